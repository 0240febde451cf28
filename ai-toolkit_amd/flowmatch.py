"""Flow-matching train schedule (host logic).

Restates toolkit/samplers/custom_flowmatch_sampler.py:
  * calculate_shift ................ lines 10-20
  * set_train_timesteps ............ lines 107-193 ('linear'/'weighted', 'sigmoid', 'shift'/'flux_shift' with dynamic shifting)
  * add_noise ...................... lines 91-102 (executed by the aitk_flow_noise_pack kernel, see trainer.py)
and the timestep-index sampling of jobs/process/BaseSDTrainProcess.py:1301-1323 (content_or_style='balanced').
Only O(1000)-element tables live here; they are produced with torch on the device the trainer uses so that the RNG
stream order equals the reference's (randn for 'sigmoid' is drawn on `device`).
"""
import math

import numpy as np
import torch


_WEIGHING = None


def default_weighing_scheme():
    global _WEIGHING
    if _WEIGHING is None:
        import json
        import os

        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "flowmatch_default_weighing_scheme.json")) as fh:
            _WEIGHING = json.load(fh)["weights"]
        assert len(_WEIGHING) == 1000
    return _WEIGHING


def calculate_shift(image_seq_len, base_seq_len=256, max_seq_len=4096, base_shift=0.5, max_shift=1.16):
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    b = base_shift - m * base_seq_len
    return image_seq_len * m + b


class FlowMatchTrainSchedule:
    def __init__(self, num_train_timesteps=1000, shift=3.0, use_dynamic_shifting=True, base_shift=0.5, max_shift=1.15,
                 base_image_seq_len=256, max_image_seq_len=4096):
        # FLUX.1-dev scheduler_config values (extensions_built_in/diffusion_models/flux_kontext/flux_kontext.py:30-38)
        self.num_train_timesteps = num_train_timesteps
        self.shift = shift
        self.use_dynamic_shifting = use_dynamic_shifting
        self.base_shift, self.max_shift = base_shift, max_shift
        self.base_image_seq_len, self.max_image_seq_len = base_image_seq_len, max_image_seq_len
        self.timesteps = None
        self.timestep_type = "linear"
        # Bell-shaped mean-normalised timestep weights and the half-bell variant (custom_flowmatch_sampler.py:29-56)
        n = 1000
        x = torch.arange(n, dtype=torch.float32)
        y = torch.exp(-2 * ((x - n / 2) / n) ** 2)
        y = y - y.min()
        self.linear_timesteps_weights = y * (n / y.sum())
        w2 = y * (n / y.sum())
        w2[n // 2:] = w2[n // 2:].max()
        self.linear_timesteps_weights2 = w2

    # few-step distillation-style schedules: a linear table, indices drawn from a fixed list (jobs/process/BaseSDTrainProcess.py:1196-1205, 1254-1272)
    N_STEP_INDICES = {"one_step": [0], "two_step": [0, 499], "four_step": [0, 250, 500, 750],
                      "eight_step": [0, 125, 250, 375, 500, 625, 750, 875]}

    def set_train_timesteps(self, num_timesteps, device, timestep_type="linear", latents=None, patch_size=1):
        self.timestep_type = timestep_type
        if timestep_type in self.N_STEP_INDICES:  # the trainer hands the scheduler 'linear' for these (BaseSDTrainProcess.py:1196-1203)
            timestep_type = "linear"
        if timestep_type in ("linear", "weighted"):
            self.timesteps = torch.linspace(1000, 1, num_timesteps, device=device)
        elif timestep_type == "sigmoid":
            t = torch.sigmoid(torch.randn((num_timesteps,), device=device))
            ts, _ = torch.sort((1 - t) * 1000, descending=True)
            self.timesteps = ts.to(device=device)
        elif timestep_type in ("flux_shift", "lumina2_shift", "shift"):
            n = self.num_train_timesteps
            # diffusers FlowMatchEulerDiscreteScheduler: sigma_max/min after the static shift
            sig = np.linspace(1, n, n, dtype=np.float32)[::-1].copy() / n
            sig = self.shift * sig / (1 + (self.shift - 1) * sig) if not self.use_dynamic_shifting else sig
            t_max, t_min = float(sig[0]) * n, float(sig[-1]) * n
            timesteps = np.linspace(t_max, t_min, num_timesteps)
            sigmas = timesteps / n
            if self.use_dynamic_shifting:
                if latents is None:
                    raise ValueError("latents is None")
                h, w = latents.shape[2], latents.shape[3]
                mu = calculate_shift(h * w // (patch_size ** 2), self.base_image_seq_len, self.max_image_seq_len,
                                     self.base_shift, self.max_shift)
                sigmas = math.exp(mu) / (math.exp(mu) + (1 / sigmas - 1) ** 1.0)
            else:
                sigmas = self.shift * sigmas / (1 + (self.shift - 1) * sigmas)
            self.timesteps = (torch.from_numpy(sigmas).to(dtype=torch.float32, device=device)) * n
        elif timestep_type == "lognorm_blend":
            # custom_flowmatch_sampler.py:194-217: 75 % log-normal(0, 0.333) samples scaled into (0, 1000], 25 % linear, sorted, int
            alpha = 0.75
            t1 = torch.distributions.LogNormal(loc=0, scale=0.333).sample((int(num_timesteps * alpha),)).to(device)
            t1 = (1 - t1 / t1.max()) * 1000
            t2 = torch.linspace(1000, 1, int(num_timesteps * (1 - alpha)), device=device)
            ts, _ = torch.sort(torch.cat((t1, t2)), descending=True)
            self.timesteps = ts.to(torch.int).to(device=device)
        else:
            raise ValueError(f"Invalid timestep type: {timestep_type}")
        return self.timesteps

    def sample_timesteps(self, batch_size, device, generator=None, min_idx=0, max_idx=None, content_or_style="balanced"):
        """Timestep indices into the table (jobs/process/BaseSDTrainProcess.py:1248-1323):
        'balanced' -> randint(min_idx, max_idx), flow-matching uses [0, num_train_timesteps - 1);
        'content' / 'style' -> cubic sampling u^3 / 1 - u^3 of u ~ U(0,1), mapped to [min_idx, max_idx] and clamped (1275-1298)."""
        if max_idx is None:
            max_idx = self.num_train_timesteps - 1
        if self.timestep_type in self.N_STEP_INDICES:
            import random

            choices = self.N_STEP_INDICES[self.timestep_type]
            if self.timestep_type == "one_step":
                idx = torch.zeros((batch_size,), device=device, dtype=torch.long)
            else:  # the reference draws with Python's global `random` (random.choices), not with torch
                idx = torch.tensor(random.choices(choices, k=batch_size), device=device).long()
            return self.timesteps[idx].float().contiguous(), idx
        if content_or_style == "balanced":
            if min_idx == max_idx:
                idx = torch.full((batch_size,), min_idx, device=device).long()
            else:
                idx = torch.randint(min_idx, max_idx, (batch_size,), device=device, generator=generator).long()
        elif content_or_style in ("content", "style"):
            u = torch.rand((batch_size,), device=device, generator=generator)
            n = self.num_train_timesteps
            t = u ** 3 * n if content_or_style == "content" else (1 - u ** 3) * n
            t = (t - 0) * (max_idx - min_idx) / ((n - 1) - 0) + min_idx  # toolkit/basic.py:7-8 value_map
            idx = t.long().clamp(min_idx, max_idx)
        else:
            raise ValueError(f"Unknown content_or_style {content_or_style}")
        return self.timesteps[idx].float().contiguous(), idx

    def get_weights_for_timesteps(self, timesteps, v2=False, timestep_type="linear"):
        """Per-sample loss weights of `linear_timesteps` / `linear_timesteps2` training (custom_flowmatch_sampler.py:59-76,
        applied at SDTrainer.py:925-944): bell-shaped weight of each timestep's INDEX in the current table.  'weighted' looks the
        index up in the reference's empirical 1000-entry default_weighing_scheme (data/flowmatch_default_weighing_scheme.json,
        imported verbatim by tools/import_reference_tables.py) and returns it in the timesteps' dtype like the reference."""
        table = self.timesteps.to(timesteps.device)
        idx = torch.stack([(table == t).nonzero()[0, 0] for t in timesteps]).cpu()
        if timestep_type == "weighted":
            return torch.tensor([default_weighing_scheme()[int(i)] for i in idx], device=timesteps.device, dtype=timesteps.dtype)
        w = (self.linear_timesteps_weights2 if v2 else self.linear_timesteps_weights)[idx]
        return w.flatten().to(timesteps.device)


def get_noise(latents, generator=None, *, noise_offset=0.0, noise_multiplier=1.0, random_noise_shift=0.0,
              random_noise_multiplier=0.0, dynamic_noise_offset=False, dtype=None, signal_correction_noise_scale=0.0,
              batch_noise_correction_scale=0.0):
    """The reference's noise for one batch, same draw order on the same device generator:
      randn(latents.shape) in fp32 on the device                         toolkit/stable_diffusion_model.py:1803-1811
      + noise_offset * randn(B, C, 1, 1)   (4-D latents only)            toolkit/train_tools.py:132-139
      cast to the training dtype                                          jobs/process/BaseSDTrainProcess.py:1020-1027
      + latents channel mean / 2           (dynamic_noise_offset)         BaseSDTrainProcess.py:1331-1335
      * noise_multiplier                                                   1344-1351
      + latents * randn(B, C, 1, 1) * signal_correction_noise_scale        1353-1361 (do_signal_correction_noise)
      + roll(latents, randint(1, B)) * randn(B, C, 1, 1) * batch_noise_correction_scale   1363-1376 (do_batch_noise_correction, B > 1)
      + randn(B, C[, F], 1, 1) * random_noise_shift                        1378-1386
      * exp(randn(B, C[, F], 1, 1) * random_noise_multiplier)              1388-1391
    Tiny per-batch tensor plumbing on [B, C, h, w]; the mixing with the latents and the 2x2 packing are aitk_flow_noise_pack."""
    dev = latents.device
    dtype = dtype or latents.dtype
    noise = torch.randn(latents.shape, device=dev, dtype=torch.float32, generator=generator)
    if noise_offset is not None and abs(noise_offset) >= 0.000001:
        if latents.dim() > 4:
            raise ValueError("Applying noise offset not supported for video models at this time.")
        noise = noise + noise_offset * torch.randn((noise.shape[0], noise.shape[1], 1, 1), device=dev, generator=generator)
    noise = noise.to(dtype)
    if dynamic_noise_offset:
        noise = noise + latents.mean(dim=(2, 3), keepdim=True).to(dtype) / 2
    s = (noise.shape[0], noise.shape[1], 1, 1) if noise.dim() == 4 else (noise.shape[0], noise.shape[1], noise.shape[2], 1, 1)
    noise = noise * noise_multiplier
    if signal_correction_noise_scale:
        scn = torch.randn(latents.shape[0], latents.shape[1], 1, 1, device=dev, dtype=dtype, generator=generator) * signal_correction_noise_scale
        noise = noise + latents.to(dtype) * scn
    if batch_noise_correction_scale and latents.shape[0] > 1:
        # another sample of the batch, never the same position (the reference draws the shift with the global CPU generator)
        shift = torch.randint(1, latents.shape[0], (1,)).item()
        bn = latents.roll(shifts=shift, dims=0).to(dtype)
        noise = noise + bn * (torch.randn(bn.shape[0], bn.shape[1], 1, 1, device=dev, dtype=dtype, generator=generator) * batch_noise_correction_scale)
    if random_noise_shift > 0.0:
        noise = noise + torch.randn(s, device=dev, dtype=dtype, generator=generator) * random_noise_shift
    if random_noise_multiplier > 0.0:
        noise = noise * torch.exp(torch.randn(s, device=dev, dtype=dtype, generator=generator) * random_noise_multiplier)
    return noise
