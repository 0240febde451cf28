"""DDPM train schedule of the UNet models (SD1.5 / SDXL), host logic only (O(1000)-element tables).

Restates what the reference's eps / v-prediction path uses:
  * the scheduler ............ get_sampler('ddpm') -> DDPMScheduler.from_config(sd_config): scaled_linear betas 0.00085 -> 0.012,
                               1000 steps (toolkit/sampler.py:31-50, 136-137); set_timesteps(1000) with 'leading' spacing and
                               steps_offset 0 gives the table [999, 998, ..., 0] (jobs/process/BaseSDTrainProcess.py:1227-1229)
  * timestep indices ......... content_or_style 'balanced': randint(min + 1, max - 1) for non-flow-matching schedulers
                               (BaseSDTrainProcess.py:1301-1318), 'content' / 'style' cubic sampling (1275-1298)
  * add_noise ................ sqrt(acp[t]) x0 + sqrt(1 - acp[t]) eps with alphas_cumprod cast to the latent dtype
                               (toolkit/stable_diffusion_model.py:1854-1876 -> DDPMScheduler.add_noise) — executed by aitk_ddpm_noise_nhwc
  * loss target .............. eps (extensions_built_in/sd_trainer/SDTrainer.py:650) or the velocity (623-625)
  * min-SNR weighting ........ apply_snr_weight / get_all_snr (toolkit/train_tools.py:642-654, 720-749), applied at SDTrainer.py:1005-1011
"""
import torch


class DDPMTrainSchedule:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, prediction_type="epsilon"):
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.timesteps = None

    def set_timesteps(self, num_timesteps, device):
        """'leading' spacing, steps_offset 0: arange(0, n) * (num_train_timesteps // n), reversed."""
        ratio = self.num_train_timesteps // num_timesteps
        self.timesteps = (torch.arange(0, num_timesteps, device=device) * ratio).flip(0).long()
        return self.timesteps

    def sample_timesteps(self, batch_size, device, generator=None, min_noise_steps=0, max_noise_steps=999, content_or_style="balanced"):
        if self.timesteps is None or self.timesteps.device != torch.device(device):
            self.set_timesteps(self.num_train_timesteps, device)
        lo, hi = max(min_noise_steps, 0), min(max_noise_steps, self.num_train_timesteps - 1)
        if content_or_style == "balanced":
            if lo == hi:
                idx = torch.full((batch_size,), lo, device=device).long()
            else:
                idx = torch.randint(lo + 1, hi - 1, (batch_size,), device=device, generator=generator).long()
        elif content_or_style in ("content", "style"):
            u = torch.rand((batch_size,), device=device, generator=generator)
            n = self.num_train_timesteps
            t = u ** 3 * n if content_or_style == "content" else (1 - u ** 3) * n
            t = t * (hi - lo) / (n - 1) + lo  # toolkit/basic.py value_map(t, 0, n - 1, lo, hi)
            idx = t.long().clamp(lo, hi)
        else:
            raise ValueError(f"Unknown content_or_style {content_or_style}")
        return self.timesteps[idx], idx

    def noise_coefficients(self, timesteps, dtype):
        """(sqrt(acp[t]), sqrt(1 - acp[t])) computed in the latent dtype like DDPMScheduler.add_noise, returned as fp32 [B]."""
        ac = self.alphas_cumprod.to(device=timesteps.device, dtype=dtype)[timesteps.long()]
        return (ac ** 0.5).float().contiguous(), ((1 - ac) ** 0.5).float().contiguous()

    def snr_weights(self, timesteps, gamma, fixed=False):
        ac = self.alphas_cumprod.to(timesteps.device)
        all_snr = (torch.sqrt(ac) / torch.sqrt(1.0 - ac)) ** 2
        snr = all_snr[timesteps.long()]  # DDPM timesteps never start at 1000: offset 0 (train_tools.py:739-742)
        g = gamma / snr
        return (g if fixed else torch.minimum(g, torch.ones_like(g))).float()
