"""Model-plugin surface of the reference (`BaseModel` subclasses registered in AI_TOOLKIT_MODELS, selected by `arch`,
toolkit/util/get_model.py:20-50) for the two native transformer graphs, and the legacy `StableDiffusion` wrapper surface the reference
uses for the SD1.5 / SDXL UNets (`StableDiffusionMI355Model`, toolkit/stable_diffusion_model.py:1824-1876, 1878-2055, 2260-2301).

The classes are duck-typed mirrors: same attribute / method names, argument meaning and error behaviour as the reference's
in-tree plug-ins, without importing the reference (its `BaseModel` pulls in diffusers / transformers):

  Flux1MI355Model   <- extensions_built_in/diffusion_models/flux_kontext/flux_kontext.py:41-419 (FLUX.1 on the BaseModel API;
                       identical pack / ids / guidance / unpack to the legacy branch, toolkit/stable_diffusion_model.py:2154-2222)
  Wan21MI355Model   <- toolkit/models/wan21/wan21.py:330-342, 578-603, 717-736

A maintainer makes them real plug-ins by adding `BaseModel` to the bases and listing them in AI_TOOLKIT_MODELS (INTEGRATION.md
§2); everything below already runs on the fused graphs: `get_noise_prediction` calls the model's diffusers-signature
`forward`, whose output carries the explicit HIP backward through torch.autograd, so `accelerator.backward(loss)`
(SDTrainer.py:2238) fills the adapter gradient arena.
"""
import torch

from .convert import wan_lora_to_diffusers, wan_lora_to_original
from .flowmatch import FlowMatchTrainSchedule

# extensions_built_in/diffusion_models/flux_kontext/flux_kontext.py:28-36 (FLUX.1 training scheduler config)
FLUX_SCHEDULER_CONFIG = {"base_image_seq_len": 256, "base_shift": 0.5, "max_image_seq_len": 4096, "max_shift": 1.15,
                         "num_train_timesteps": 1000, "shift": 3.0, "use_dynamic_shifting": True}
# toolkit/models/wan21/wan21.py:80-84
WAN_SCHEDULER_CONFIG = {"num_train_timesteps": 1000, "shift": 3.0, "use_dynamic_shifting": False}


def _embeds(text_embeddings):
    """PromptEmbeds-like object (.text_embeds / .pooled_embeds, toolkit/prompt_utils.py) or a (text, pooled) tuple."""
    if hasattr(text_embeddings, "text_embeds"):
        return text_embeddings.text_embeds, getattr(text_embeddings, "pooled_embeds", None)
    if isinstance(text_embeddings, (tuple, list)):
        return text_embeddings[0], (text_embeddings[1] if len(text_embeddings) > 1 else None)
    return text_embeddings, None


class _PluginBase:
    is_flow_matching = True
    is_transformer = True
    use_old_lokr_format = False

    def __init__(self, device, model=None, vae=None, dtype=torch.bfloat16, **kwargs):
        self.device_torch = torch.device(device)
        self.torch_dtype = dtype
        self.model = model
        self.vae = vae
        self.network = None

    # the reference reads the denoiser through these aliases (toolkit/models/base_model.py:199-216)
    @property
    def unet(self):
        return self.model

    @property
    def unet_unwrapped(self):
        return self.model

    def get_model_has_grad(self):
        return False  # frozen base: only the adapter trains

    def get_te_has_grad(self):
        return False

    def get_loss_target(self, *args, **kwargs):
        noise, batch = kwargs.get("noise"), kwargs.get("batch")
        if batch is None:
            raise ValueError("Batch is not provided")
        if noise is None:
            raise ValueError("Noise is not provided")
        return (noise - batch.latents).detach()

    def convert_lora_weights_before_save(self, state_dict):
        return state_dict

    def convert_lora_weights_before_load(self, state_dict):
        return state_dict

    def encode_images(self, image_list, device=None, dtype=None):
        """[-1, 1] images -> scaled latents through the native VAE encoder (toolkit/models/base_model.py:1134-1176)."""
        if self.vae is None:
            raise RuntimeError("no VAE encoder attached (latents are expected to be cached)")
        images = torch.stack(list(image_list)) if isinstance(image_list, (list, tuple)) else image_list
        return self.vae.encode_images(images.to(self.device_torch, self.torch_dtype))


class Flux1MI355Model(_PluginBase):
    arch = "flux_mi355"
    target_lora_modules = ["FluxTransformer2DModel"]

    @staticmethod
    def get_train_scheduler():
        c = FLUX_SCHEDULER_CONFIG
        return FlowMatchTrainSchedule(num_train_timesteps=c["num_train_timesteps"], shift=c["shift"],
                                      use_dynamic_shifting=c["use_dynamic_shifting"], base_image_seq_len=c["base_image_seq_len"],
                                      max_image_seq_len=c["max_image_seq_len"], base_shift=c["base_shift"], max_shift=c["max_shift"])

    def get_bucket_divisibility(self):
        return 16

    def get_base_model_version(self):
        return "flux.1"

    def get_transformer_block_names(self):
        return ["transformer_blocks", "single_transformer_blocks"]

    def get_noise_prediction(self, latent_model_input, timestep, text_embeddings, guidance_embedding_scale=1.0,
                             bypass_guidance_embedding=False, **kwargs):
        """latent_model_input [B,16,H,W], timestep [B] on the 0..1000 scale -> prediction [B,16,H,W]
        (flux_kontext.py:243-352 without the kontext control branch)."""
        if bypass_guidance_embedding:
            raise NotImplementedError("bypass_guidance_embedding (FLUX.1-schnell training adapter) is not on the fused path")
        bs, c, h, w = latent_model_input.shape
        if c != 16:
            raise ValueError(f"expected 16 latent channels, got {c} (kontext control channels are not on the fused path)")
        dev = self.device_torch
        text, pooled = _embeds(text_embeddings)
        with torch.no_grad():
            packed = latent_model_input.reshape(bs, c, h // 2, 2, w // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(bs, (h // 2) * (w // 2), c * 4)
            img_ids = torch.zeros(h // 2, w // 2, 3)
            img_ids[..., 1] = img_ids[..., 1] + torch.arange(h // 2)[:, None]
            img_ids[..., 2] = img_ids[..., 2] + torch.arange(w // 2)[None, :]
            img_ids = img_ids.reshape(-1, 3).to(dev)
            img_ids._aitk_grid = (h // 2, w // 2, text.shape[1])
            txt_ids = torch.zeros(text.shape[1], 3, device=dev)
            if isinstance(guidance_embedding_scale, list):
                guidance = torch.tensor(guidance_embedding_scale, device=dev, dtype=torch.float32)
            else:
                guidance = torch.tensor([float(guidance_embedding_scale)], device=dev).expand(bs)
        cast = self.model.dt
        noise_pred = self.model(hidden_states=packed.to(dev, cast), timestep=timestep.to(dev) / 1000,
                                encoder_hidden_states=text.to(dev, cast), pooled_projections=pooled.to(dev, cast),
                                txt_ids=txt_ids, img_ids=img_ids, guidance=guidance, return_dict=False, **kwargs)[0]
        return noise_pred.reshape(bs, h // 2, w // 2, c, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(bs, c, h, w)


class Wan21MI355Model(_PluginBase):
    arch = "wan21_mi355"
    target_lora_modules = ["WanTransformer3DModel"]

    @staticmethod
    def get_train_scheduler():
        c = WAN_SCHEDULER_CONFIG
        return FlowMatchTrainSchedule(num_train_timesteps=c["num_train_timesteps"], shift=c["shift"],
                                      use_dynamic_shifting=c["use_dynamic_shifting"])

    def get_bucket_divisibility(self):
        return 16

    def get_base_model_version(self):
        return "wan_2.1"

    def get_transformer_block_names(self):
        return ["blocks"]

    def get_noise_prediction(self, latent_model_input, timestep, text_embeddings, **kwargs):
        """latent_model_input [B,16,F,H,W], timestep [B] on the 0..1000 scale (wan21.py:578-603)."""
        text, _ = _embeds(text_embeddings)
        return self.model(hidden_states=latent_model_input, timestep=timestep, encoder_hidden_states=text,
                          return_dict=False, **kwargs)[0]

    def encode_images(self, image_list, device=None, dtype=None):
        """list of [C,H,W] images / [T,C,H,W] clips in [-1, 1] -> normalised video latents [B,16,T',H/8,W/8] through the native
        AutoencoderKLWan encoder (toolkit/models/wan21/wan21.py:618-672)."""
        if self.vae is None:
            raise RuntimeError("no VAE encoder attached (latents are expected to be cached)")
        return self.vae.encode_images([im.to(self.device_torch) for im in image_list])

    def convert_lora_weights_before_save(self, state_dict):
        return wan_lora_to_original(state_dict)

    def convert_lora_weights_before_load(self, state_dict):
        return wan_lora_to_diffusers(state_dict)


class StableDiffusionMI355Model(_PluginBase):
    """The part of the reference's legacy `StableDiffusion` wrapper the train step calls for SD1.5 / SDXL (`is_xl`): `predict_noise`
    (toolkit/stable_diffusion_model.py:1878-1935 argument handling, SDXL branch 1968-2055, SD1.5 branch 2260-2265; training never runs
    classifier-free guidance here: the embeddings' batch size equals the latents'), `get_time_ids_from_latents` (1824-1852), `add_noise`
    through the DDPM schedule (1854-1876) and the eps / v-prediction loss target (SDTrainer.py:623-625, 650).  `model` is
    ai_toolkit_amd.unet.UNet2DConditionModel, whose diffusers-signature forward returns `.sample` and carries the explicit backward."""

    arch = "sd_mi355"
    is_flow_matching = False
    is_transformer = False
    target_lora_modules = ["Transformer2DModel"]  # toolkit/kohya_lora.py:750 (+ ResnetBlock2D / Downsample2D / Upsample2D with network.conv)

    def __init__(self, device, model=None, vae=None, dtype=torch.bfloat16, is_xl=False, prediction_type="epsilon", **kwargs):
        super().__init__(device, model=model, vae=vae, dtype=dtype, **kwargs)
        from .ddpm import DDPMTrainSchedule

        self.is_xl = bool(is_xl)
        self.prediction_type = prediction_type
        self.noise_scheduler = DDPMTrainSchedule(prediction_type=prediction_type)

    @staticmethod
    def get_train_scheduler():
        from .ddpm import DDPMTrainSchedule

        return DDPMTrainSchedule()

    def get_bucket_divisibility(self):
        return 8  # vae scale factor 8; the UNet's three (SDXL: two) stride-2 levels are covered by the reference's 64-px bucket tolerance

    def get_base_model_version(self):
        return "sdxl_1.0" if self.is_xl else "sd_1.5"

    def get_time_ids_from_latents(self, latents, requires_aesthetic_score=False):
        """(H, W, 0, 0, H, W) per sample in pixels for SDXL, None for SD1.5 (stable_diffusion_model.py:1824-1852)."""
        if not self.is_xl:
            return None
        if requires_aesthetic_score:
            raise NotImplementedError("the SDXL refiner is not on the fused path")
        bs, _, h, w = latents.shape
        ids = torch.tensor([[h * 8, w * 8, 0, 0, h * 8, w * 8]]).to(latents.device, dtype=latents.dtype)
        return torch.cat([ids for _ in range(bs)])

    def add_noise(self, original_samples, noise, timesteps):
        a, s = self.noise_scheduler.noise_coefficients(timesteps.to(original_samples.device), original_samples.dtype)
        a, s = a.to(original_samples.dtype).view(-1, 1, 1, 1), s.to(original_samples.dtype).view(-1, 1, 1, 1)
        return a * original_samples + s * noise

    def get_loss_target(self, *args, **kwargs):
        noise, batch, timesteps = kwargs.get("noise"), kwargs.get("batch"), kwargs.get("timesteps")
        if noise is None:
            raise ValueError("Noise is not provided")
        if self.prediction_type == "v_prediction":
            if batch is None or timesteps is None:
                raise ValueError("v_prediction needs the batch latents and the timesteps")
            a, s = self.noise_scheduler.noise_coefficients(timesteps.to(noise.device), noise.dtype)
            return (a.to(noise.dtype).view(-1, 1, 1, 1) * noise - s.to(noise.dtype).view(-1, 1, 1, 1) * batch.latents).detach()
        return noise.detach()

    def predict_noise(self, latents, text_embeddings=None, timestep=1, guidance_scale=7.5, guidance_rescale=0, add_time_ids=None,
                      conditional_embeddings=None, unconditional_embeddings=None, **kwargs):
        if text_embeddings is None and conditional_embeddings is None:
            raise ValueError("Either text_embeddings or conditional_embeddings must be specified")
        if unconditional_embeddings is not None:
            raise NotImplementedError("classifier-free guidance inside predict_noise (sampling) is not on the fused path")
        if text_embeddings is None:
            text_embeddings = conditional_embeddings
        text, pooled = _embeds(text_embeddings)
        if latents.shape[0] != text.shape[0]:
            raise ValueError("Batch size of latents must be the same or half the batch size of text embeddings")
        dev = self.device_torch
        timestep = torch.as_tensor(timestep).to(dev)
        if timestep.dim() == 0:
            timestep = timestep.unsqueeze(0)
        if timestep.shape[0] == 1 and latents.shape[0] > 1:
            timestep = timestep.repeat(latents.shape[0])
        cast = self.model.dt
        if self.is_xl:
            with torch.no_grad():
                if add_time_ids is None:
                    add_time_ids = self.get_time_ids_from_latents(latents)
                added = {"text_embeds": pooled.to(dev, cast), "time_ids": add_time_ids.to(dev)}
            return self.model(latents.to(dev, cast), timestep, encoder_hidden_states=text.to(dev, cast), added_cond_kwargs=added).sample
        return self.model(latents.to(dev, cast), timestep=timestep, encoder_hidden_states=text.to(dev, cast)).sample

    # the BaseModel-style name for the same call (toolkit/models/base_model.py get_noise_prediction contract)
    def get_noise_prediction(self, latent_model_input, timestep, text_embeddings, **kwargs):
        return self.predict_noise(latent_model_input, text_embeddings=text_embeddings, timestep=timestep, **kwargs)
