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

    def set_train_timesteps(self, num_timesteps, device, timestep_type="linear", latents=None, patch_size=1):
        self.timestep_type = timestep_type
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
        else:
            raise ValueError(f"Invalid timestep type: {timestep_type}")
        return self.timesteps

    def sample_timesteps(self, batch_size, device, generator=None, min_idx=0, max_idx=None):
        """'balanced': randint(min_idx, max_idx) into the table; flow-matching uses [0, num_train_timesteps - 1)
        (jobs/process/BaseSDTrainProcess.py:1301-1323)."""
        if max_idx is None:
            max_idx = self.num_train_timesteps - 1
        idx = torch.randint(min_idx, max_idx, (batch_size,), device=device, generator=generator).long()
        return self.timesteps[idx].float().contiguous(), idx
