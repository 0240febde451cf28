"""ORACLE (test infrastructure): the reference's per-step sequence in plain PyTorch + autograd.

Restates, for the FLUX / flow-matching branch:
  add_noise ................ toolkit/samplers/custom_flowmatch_sampler.py:91-102 (per-sample loop collapsed; same math)
  pack / ids / unpack ...... toolkit/stable_diffusion_model.py:2157-2170, 2210-2219
  loss ..................... extensions_built_in/sd_trainer/SDTrainer.py:644-646 (target = noise - latents), 916 (mse on
                             .float()), 987-990 (mean over C,H,W), 1013 (mean over batch)
  batch_list accumulation .. SDTrainer.py:2243-2293 (zero_grad once, backward per micro-batch, losses summed, one step)
  output preservation ...... SDTrainer.py:1229-1247 (prior = adapter inactive, no_grad, preservation embeds), 2182-2220
                             (second adapter-active pass on the preservation embeds, mse(pred, prior) * multiplier added)
  clip / step / EMA ........ SDTrainer.py:2278-2293 ; toolkit/optimizer.py:78-79 (torch.optim.AdamW, eps=1e-6) ;
                             toolkit/ema.py:116-152 (s -= (1-decay)(s-p))
"""
import torch

from . import flux_ref


class RefTrainStep:
    def __init__(self, model, net, lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-6, max_grad_norm=1.0,
                 ema_decay=0.0, guidance=1.0, lr_scheduler=None):
        self.model, self.net = model, net
        self.params = [p for m in net.unet_loras for p in
                       ((m.magnitude, m.lora_up.weight, m.lora_down.weight) if hasattr(m, "magnitude") else
                        tuple(m.parameters()) if hasattr(m, "lokr_w1") else (m.lora_down.weight, m.lora_up.weight))]
        self.opt = torch.optim.AdamW(self.params, lr=lr, eps=eps, betas=betas, weight_decay=weight_decay)
        self.max_grad_norm, self.ema_decay, self.guidance = max_grad_norm, ema_decay, guidance
        self.ema = [p.detach().clone() for p in self.params] if ema_decay > 0 else None
        self.lr_scheduler = lr_scheduler(self.opt) if lr_scheduler is not None else None  # factory(optimizer) -> torch scheduler

    def step(self, latents, prompt_embeds, pooled, noise, timesteps, dtype=torch.float32, preservation=None,
             preservation_multiplier=1.0):
        return self.step_list([dict(latents=latents, prompt_embeds=prompt_embeds, pooled=pooled, noise=noise, timesteps=timesteps,
                                    preservation=preservation, preservation_multiplier=preservation_multiplier)], dtype)

    def step_list(self, batches, dtype=torch.float32):
        self.opt.zero_grad()
        total = None
        for b in batches:
            loss = self._single(dtype=dtype, **b)
            total = loss if total is None else total + loss
        if self.max_grad_norm > 0:
            torch.nn.utils.clip_grad_norm_(self.params, self.max_grad_norm)
        self.opt.step()
        if self.ema is not None:
            with torch.no_grad():
                for s, p in zip(self.ema, self.params):
                    s.sub_((1.0 - self.ema_decay) * (s - p))
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
        return total

    def _single(self, latents, prompt_embeds, pooled, noise, timesteps, dtype=torch.float32, preservation=None,
                preservation_multiplier=1.0):
        B, Cc, Hh, W = latents.shape
        lat = latents.to(dtype)
        noi = noise.to(dtype)
        t01 = (timesteps.float() / 1000).view(B, 1, 1, 1)
        noisy = (1.0 - t01) * lat + t01 * noi  # fp32 math, cast when fed to the model
        target = (noi - lat).detach()
        img_ids, txt_ids = flux_ref.make_ids(Hh, W, prompt_embeds.shape[1], latents.device)
        guidance = torch.full((B,), float(self.guidance), device=latents.device)
        packed = flux_ref.pack_latents(noisy.to(dtype))
        prior = None
        if preservation is not None:
            with torch.no_grad():  # network inactive: base-model prediction on the preservation embeds
                prior = self.model(packed, preservation[0].to(dtype), preservation[1].to(dtype), timesteps.float() / 1000,
                                   img_ids, txt_ids, guidance)
        with self.net:
            pred = self.model(packed, prompt_embeds.to(dtype), pooled.to(dtype), timesteps.float() / 1000, img_ids, txt_ids,
                              guidance)
            pred = flux_ref.unpack_latents(pred, Hh, W)
            loss = torch.nn.functional.mse_loss(pred.float(), target.float(), reduction="none").mean([1, 2, 3]).mean()
            if prior is not None:
                pres = self.model(packed, preservation[0].to(dtype), preservation[1].to(dtype), timesteps.float() / 1000,
                                  img_ids, txt_ids, guidance)
                loss = loss + torch.nn.functional.mse_loss(pres, prior) * preservation_multiplier
            loss.backward()
        return loss.detach()


class RefUNetTrainStep:
    """SD1.5 / SDXL step of the reference in plain PyTorch + autograd: DDPM add_noise in the latent dtype
    (toolkit/stable_diffusion_model.py:1854-1876), UNet call (2049-2055 / 2260-2265; SDXL time_ids 1824-1852), target = noise
    (SDTrainer.py:650) or velocity (623-625), mse(pred.float(), target.float()).mean([1,2,3]) (916, 987-990), min-SNR / fixed-SNR
    weights (1003-1011, toolkit/train_tools.py:720-749), mean (1013), clip_grad_norm_, torch.optim.AdamW(eps=1e-6)."""

    def __init__(self, model, net, lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-6, max_grad_norm=1.0, min_snr_gamma=None,
                 snr_gamma=None, prediction_type="epsilon"):
        from . import unet_ref

        self.model, self.net, self.U = model, net, unet_ref
        self.params = [p for m in net.unet_loras for p in (m.lora_down.weight, m.lora_up.weight)]
        self.opt = torch.optim.AdamW(self.params, lr=lr, eps=eps, betas=betas, weight_decay=weight_decay)
        self.max_grad_norm, self.min_snr_gamma, self.snr_gamma, self.prediction_type = max_grad_norm, min_snr_gamma, snr_gamma, prediction_type
        self.acp = unet_ref.ddpm_alphas_cumprod()
        self.is_xl = model.config["addition_embed_type"] == "text_time"

    def step(self, latents, prompt_embeds, pooled, noise, timesteps, dtype=torch.float32):
        U = self.U
        self.opt.zero_grad()
        lat, noi = latents.to(dtype), noise.to(dtype)
        noisy = U.ddpm_add_noise(lat, noi, timesteps, self.acp)
        target = U.ddpm_velocity(lat, noi, timesteps, self.acp) if self.prediction_type == "v_prediction" else noi
        added = None
        if self.is_xl:
            added = dict(text_embeds=pooled.to(dtype), time_ids=U.time_ids_from_latents(lat))
        with self.net:
            pred = self.model(noisy, timesteps.float(), prompt_embeds.to(dtype), added)
            loss = torch.nn.functional.mse_loss(pred.float(), target.float(), reduction="none").mean([1, 2, 3])
            if self.snr_gamma is not None and self.snr_gamma > 1e-6:
                loss = loss * U.min_snr_weight(timesteps, self.acp, self.snr_gamma, fixed=True)
            elif self.min_snr_gamma is not None and self.min_snr_gamma > 1e-6:
                loss = loss * U.min_snr_weight(timesteps, self.acp, self.min_snr_gamma)
            loss = loss.mean()
            loss.backward()
        if self.max_grad_norm > 0:
            torch.nn.utils.clip_grad_norm_(self.params, self.max_grad_norm)
        self.opt.step()
        return loss.detach()
