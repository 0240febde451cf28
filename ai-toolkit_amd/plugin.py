"""Model-plugin surface of the reference (`BaseModel` subclasses registered in AI_TOOLKIT_MODELS, selected by `arch`,
toolkit/util/get_model.py:20-50) for the two native transformer graphs, and the legacy `StableDiffusion` wrapper surface the reference
uses for the SD1.5 / SDXL UNets (`StableDiffusionMI355Model`, toolkit/stable_diffusion_model.py:1824-1876, 1878-2055, 2260-2301).

The classes are duck-typed mirrors: same attribute / method names, argument meaning and error behaviour as the reference's
in-tree plug-ins, without importing the reference (its `BaseModel` pulls in diffusers / transformers):

  Flux1MI355Model   <- extensions_built_in/diffusion_models/flux_kontext/flux_kontext.py:41-419 (FLUX.1 on the BaseModel API;
                       identical pack / ids / guidance / unpack to the legacy branch, toolkit/stable_diffusion_model.py:2154-2222)
  Wan21MI355Model   <- toolkit/models/wan21/wan21.py:330-342, 578-603, 717-736

`integration/extensions/aitk_mi355/__init__.py` turns them into REAL plug-ins — `type(name, (Mirror, BaseModel), ...)` listed in
AI_TOOLKIT_MODELS — and tests/golden/plugin_registration.json records that package executed against the reference's own classes (subclass
check, construction with the reference's ModelConfig, selection by toolkit/util/get_model.py's get_model_class); the hook set and every
signature are held to tests/golden/base_model_contract.json, introspected from toolkit/models/base_model.py (tests/test_plugin_contract_cpu.py).
`load_model` streams a diffusers checkpoint directory (sharded safetensors) into the native graph (loader.py).  Everything below runs on the fused graphs: `get_noise_prediction` calls the model's diffusers-signature
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


def _native_ops():
    """the kernel table the plug-ins hand to the native graphs they build in load_model (CPU tests inject the oracle table here)"""
    from . import ops

    return ops


class PromptEmbeds:
    """Mirror of the reference's container (toolkit/prompt_utils.py:23-100: `.text_embeds`, `.pooled_embeds`, `.attention_mask`, `to`, `detach`,
    `clone`) for code that runs without the reference importable; as a real plug-in the reference's own class is returned."""

    def __init__(self, args, attention_mask=None):
        if isinstance(args, (list, tuple)):
            self.text_embeds, self.pooled_embeds = args[0], args[1]
        else:
            self.text_embeds, self.pooled_embeds = args, None
        self.attention_mask = attention_mask

    def _map(self, fn):
        out = PromptEmbeds([fn(self.text_embeds), None if self.pooled_embeds is None else fn(self.pooled_embeds)],
                           None if self.attention_mask is None else fn(self.attention_mask))
        return out

    def to(self, *a, **k):
        n = self._map(lambda t: t.to(*a, **k))
        self.text_embeds, self.pooled_embeds, self.attention_mask = n.text_embeds, n.pooled_embeds, n.attention_mask
        return self

    def detach(self):
        return self._map(lambda t: t.detach())

    def clone(self):
        return self._map(lambda t: t.clone())


def _prompt_embeds_class():
    try:
        from toolkit.prompt_utils import PromptEmbeds as Ref  # the reference's own container when it is importable (real plug-in)

        return Ref
    except Exception:  # noqa: BLE001
        return PromptEmbeds


def encode_prompts_flux(tokenizer, text_encoder, prompts, truncate=True, max_length=None, dropout_prob=0.0, attn_mask=False):
    """toolkit/train_tools.py:510-574 restated: CLIP-L pooled output + T5 last hidden state (max_length 512) for FLUX.  `tokenizer` /
    `text_encoder` = [CLIP, T5] pairs of `transformers` objects.  LIBRARY PATH: the text encoders run through
    `transformers` on PyTorch-ROCm, once per caption when the text-embedding cache is filled — they are not part of the per-step hot path and
    are not re-implemented as HIP kernels; the fused step consumes the embeddings this returns (or the reference's `_t_e_cache` files)."""
    if max_length is None:
        max_length = 512
    if dropout_prob > 0.0:
        prompts = [p if torch.rand(1).item() > dropout_prob else "" for p in prompts]
    device, dtype = text_encoder[0].device, text_encoder[0].dtype
    ti = tokenizer[0](prompts, padding="max_length", max_length=tokenizer[0].model_max_length, truncation=True, return_overflowing_tokens=False,
                      return_length=False, return_tensors="pt")
    pooled = text_encoder[0](ti.input_ids.to(device), output_hidden_states=False).pooler_output.to(dtype=dtype, device=device)
    ti = tokenizer[1](prompts, padding="max_length", max_length=max_length, truncation=True, return_length=False, return_overflowing_tokens=False,
                      return_tensors="pt")
    embeds = text_encoder[1](ti.input_ids.to(device), output_hidden_states=False)[0]
    embeds = embeds.to(dtype=text_encoder[1].dtype, device=device)
    if attn_mask:
        m = ti["attention_mask"].unsqueeze(-1).expand(embeds.shape)
        embeds = embeds * m.to(dtype=embeds.dtype, device=embeds.device)
    return embeds, pooled


def _text_tokenize(tokenizer, prompts, truncate=True, max_length=None, max_length_multiplier=4):
    """toolkit/train_tools.py:192-230: pad / truncate to max_length; long prompts (truncate False) keep up to 4 windows of model_max_length and
    drop the windows that are nothing but padding"""
    if max_length is None:
        max_length = tokenizer.model_max_length if truncate else tokenizer.model_max_length * max_length_multiplier
    ids = tokenizer(prompts, padding="max_length", max_length=max_length, truncation=True, return_tensors="pt").input_ids
    if truncate or max_length == tokenizer.model_max_length:
        return ids
    chunks = torch.chunk(ids, chunks=ids.shape[1] // tokenizer.model_max_length, dim=1)
    return torch.cat([c for c in chunks if not c.eq(c[0, 0]).all()], dim=1)


def encode_prompts(tokenizer, text_encoder, prompts, truncate=True, max_length=None, dropout_prob=0.0):
    """SD1.x (toolkit/train_tools.py:379-422): CLIP last hidden state, long prompts window by window.  LIBRARY PATH like encode_prompts_flux."""
    if max_length is None:
        max_length = tokenizer.model_max_length
    if dropout_prob > 0.0:
        prompts = [p if torch.rand(1).item() > dropout_prob else "" for p in prompts]
    tokens = _text_tokenize(tokenizer, prompts, truncate=truncate, max_length=max_length).to(text_encoder.device)
    if truncate:
        return text_encoder(tokens)[0]
    return torch.cat([text_encoder(tokens[:, i:i + max_length])[0] for i in range(0, tokens.shape[-1], max_length)], dim=1)


def encode_prompts_xl(tokenizers, text_encoders, prompts, prompts2=None, num_images_per_prompt=1, use_text_encoder_1=True, use_text_encoder_2=True,
                      truncate=True, max_length=None, dropout_prob=0.0):
    """SDXL (toolkit/train_tools.py:234-323): penultimate hidden states of both CLIP encoders concatenated on the feature axis, pooled output of
    the second (the first window's for long prompts); a disabled encoder sees empty prompts.  LIBRARY PATH like encode_prompts_flux."""
    embeds, pooled = [], None
    prompts2 = prompts if prompts2 is None else prompts2
    for idx, (tok, te) in enumerate(zip(tokenizers, text_encoders)):
        use = prompts if idx == 0 else prompts2
        if (idx == 0 and not use_text_encoder_1) or (idx == 1 and not use_text_encoder_2):
            use = ["" for _ in prompts]
        if dropout_prob > 0.0:
            use = [p if torch.rand(1).item() > dropout_prob else "" for p in use]
        ids = _text_tokenize(tok, use, truncate=truncate, max_length=max_length)
        if idx == 0:
            max_length = ids.shape[-1]
        ids = ids.to(te.device)
        if truncate:
            o = te(ids, output_hidden_states=True)
            pooled, e = o[0], o.hidden_states[-2]
        else:
            parts, pooled = [], None
            for i in range(0, ids.shape[-1], tok.model_max_length):
                o = te(ids[:, i:i + tok.model_max_length], output_hidden_states=True)
                pooled = o[0] if pooled is None else pooled
                parts.append(o.hidden_states[-2])
            e = torch.cat(parts, dim=1)
        bs, seq, _ = e.shape
        embeds.append(e.repeat(1, num_images_per_prompt, 1).view(bs * num_images_per_prompt, seq, -1))
    bs = pooled.shape[0]
    pooled = pooled.repeat(1, num_images_per_prompt).view(bs * num_images_per_prompt, -1)
    return torch.concat(embeds, dim=-1), pooled


def encode_prompts_wan(tokenizer, text_encoder, prompts, max_sequence_length=512, dtype=None):
    """Wan2.1 (`Wan21.get_prompt_embeds`, toolkit/models/wan21/wan21.py:605-615, calls diffusers' `WanPipeline.encode_prompt(prompt,
    do_classifier_free_guidance=False, max_sequence_length=512)`).  PARITY UNPINNED: the pipeline class is not vendored by the reference; this is the
    published algorithm of its `_get_t5_prompt_embeds` — clean the text (html-unescape twice + whitespace collapse; `ftfy.fix_text` first when ftfy is
    installed), tokenize to max_sequence_length with the attention mask, UMT5 last hidden state, every sequence cut at its own length and
    zero-padded back to max_sequence_length.  LIBRARY PATH like encode_prompts_flux."""
    import html
    import re

    def clean(t):
        try:
            import ftfy

            fixed = ftfy.fix_text(t)
            t = fixed if isinstance(fixed, str) else t
        except ImportError:
            pass
        return re.sub(r"\s+", " ", html.unescape(html.unescape(t)).strip()).strip()

    ti = tokenizer([clean(p) for p in prompts], padding="max_length", max_length=max_sequence_length, truncation=True, add_special_tokens=True,
                   return_attention_mask=True, return_tensors="pt")
    ids, mask = ti.input_ids, ti.attention_mask
    lens = mask.gt(0).sum(dim=1).long()
    e = text_encoder(ids.to(text_encoder.device), mask.to(text_encoder.device)).last_hidden_state
    e = e.to(dtype=dtype or text_encoder.dtype)
    return torch.stack([torch.cat([u[:v], u.new_zeros(max_sequence_length - int(v), u.size(1))]) for u, v in zip(e, lens)], dim=0)


class FakeTextEncoder(torch.nn.Module):
    """Stand-in for a text encoder that is not loaded (toolkit/unloader.py:10-33 does the same after caching the embeddings): the reference's
    trainer calls `requires_grad_`, `eval`, `to`, `.device`, `.dtype` on whatever `sd.text_encoder` holds (BaseSDTrainProcess.py:1892-1898)."""

    def __init__(self, device, dtype):
        super().__init__()
        self.dummy_param = torch.nn.Parameter(torch.zeros(1), requires_grad=False)
        self._device, self._dtype = device, dtype

    def forward(self, *args, **kwargs):
        raise NotImplementedError("no text encoder is loaded: train with cached text embeddings (datasets: cache_text_embeddings: true) or point "
                                  "model.name_or_path at a pipeline directory that has the text_encoder / tokenizer sub-folders")

    @property
    def device(self):
        return self._device

    @property
    def dtype(self):
        return self._dtype

    def to(self, *args, **kwargs):
        return self


class FakeVAE(torch.nn.Module):
    """Stand-in for a VAE that is not loaded (latents cached on disk): takes the trainer's `vae.to(...)`, `requires_grad_`, `eval` and says what
    is missing when something tries to encode with it."""

    def __init__(self, device, dtype):
        super().__init__()
        self.dummy_param = torch.nn.Parameter(torch.zeros(1), requires_grad=False)
        self._device, self._dtype = device, dtype
        self.config = {}

    device = property(lambda self: self._device)
    dtype = property(lambda self: self._dtype)

    def to(self, *args, **kwargs):
        return self

    def encode(self, *a, **k):
        raise RuntimeError("no VAE encoder is loaded (the checkpoint directory has no 'vae' sub-folder): latents must be cached (cache_latents_to_disk)")

    encode_images = encode


def _embeds(text_embeddings):
    """PromptEmbeds-like object (.text_embeds / .pooled_embeds, toolkit/prompt_utils.py) or a (text, pooled) tuple."""
    if hasattr(text_embeddings, "text_embeds"):
        return text_embeddings.text_embeds, getattr(text_embeddings, "pooled_embeds", None)
    if isinstance(text_embeddings, (tuple, list)):
        return text_embeddings[0], (text_embeddings[1] if len(text_embeddings) > 1 else None)
    return text_embeddings, None


_DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp16": torch.float16, "float16": torch.float16, "fp32": torch.float32,
           "float32": torch.float32, "float": torch.float32}


def _torch_dtype(dtype):
    """the reference passes dtype names ('bf16', 'fp16', 'fp32': toolkit/train_tools.py get_torch_dtype); torch dtypes pass through"""
    if isinstance(dtype, torch.dtype):
        return dtype
    try:
        return _DTYPES[str(dtype).lower()]
    except KeyError:
        raise ValueError(f"unknown dtype {dtype!r}") from None


class _PluginBase:
    """Constructor and hook set of `toolkit/models/base_model.py:98-360` (pinned by tests/golden/base_model_contract.json, generated from the
    reference class): `(device, model_config, dtype, custom_pipeline, noise_scheduler, **kwargs)` like every in-tree plug-in; `model=` / `vae=`
    hand over already-built native graphs (tests, bench), otherwise `load_model()` builds them from `model_config.name_or_path`."""
    is_flow_matching = True
    is_transformer = True
    use_old_lokr_format = False
    arch = None

    def __init__(self, device, model_config=None, dtype="bf16", custom_pipeline=None, noise_scheduler=None, *, model=None, vae=None, **kwargs):
        self.device = device
        self.device_torch = torch.device(device)
        self.vae_device_torch = self.te_device_torch = self.device_torch
        self.dtype = dtype
        self.torch_dtype = _torch_dtype(dtype)
        self.model_config = model_config
        self.custom_pipeline = custom_pipeline
        self.noise_scheduler = noise_scheduler
        self.model = model
        self.vae = vae
        self.text_encoder = None   # text embeddings are cached in every example config of the reference
        self.tokenizer = None
        self.pipeline = None
        self.network = None
        self.is_loaded = model is not None
        # As a real plug-in (integration/extensions/aitk_mi355) BaseModel.__init__ has run before this constructor and left INSTANCE attributes
        # is_flow_matching = is_transformer = False, use_old_lokr_format = True (toolkit/models/base_model.py:158-185) that shadow the class
        # attributes above.  The trainer reads them off the instance: is_transformer decides PEFT format and the `transformer` key prefix of the
        # LoRA network (BaseSDTrainProcess.py:1976, toolkit/lora_special.py:423, 466), is_flow_matching the timestep / noise branches
        # (BaseSDTrainProcess.py:1699, 1730), use_old_lokr_format the LoKr file layout (lora_special.py:412-416).
        cls = type(self)
        self.is_flow_matching = bool(cls.is_flow_matching)
        self.is_transformer = bool(cls.is_transformer)
        self.use_old_lokr_format = bool(cls.use_old_lokr_format)

    # ---- "must be implemented in child classes" hooks (base_model.py:306-360)
    _component = None            # diffusers sub-folder of the denoiser ('transformer')

    def _build_native(self, config=None):     # -> un-initialised native graph of the architecture the checkpoint's config.json describes
        raise NotImplementedError

    @staticmethod
    def _read_config(component_dir):
        """<component>/config.json of a diffusers checkpoint ({} when absent): the reference's from_pretrained builds the module from it, so
        FLUX.1-schnell (guidance_embeds false), Wan2.1 14B (40 layers / 40 heads / ffn 13824) or a non-default UNet load as what they are."""
        import json
        import os

        f = os.path.join(component_dir, "config.json")
        if not os.path.exists(f):
            return {}
        with open(f) as fh:
            return json.load(fh)

    @staticmethod
    def _pick(config, keys):
        return {k: (tuple(config[k]) if isinstance(config[k], list) else config[k]) for k in keys if k in config and config[k] is not None}

    def _make_scheduler(self):
        return type(self).get_train_scheduler()

    def get_transformer_block_names(self):
        return None  # base_model.py:1636-1638 default: no block filter on the adapter discovery (the DiT families override)

    def load_model(self):
        """`model_config.name_or_path` = a diffusers pipeline directory (or the component directory itself, flux_kontext.py:84-92): the
        denoiser's sharded safetensors stream into the native graph (loader.load_component), `model_config.quantize` selects the
        weight-only fp8 base (toolkit/util/quantize.py:43-75), the VAE encoder is loaded when the directory has one.  Text encoders are
        not loaded: prompts must be cached (`get_prompt_embeds` says so)."""
        from . import loader

        cfg = self.model_config
        path = getattr(cfg, "name_or_path", None)
        if not path:
            raise ValueError("model_config.name_or_path is required")
        cdir = loader.resolve_component_dir(path, self._component)
        self.model = self._build_native(self._read_config(cdir))
        loader.load_component(self.model, cdir)
        for p in self.model.parameters():
            p.requires_grad_(False)
        if getattr(cfg, "quantize", False):
            self.model.quantize_base_fp8(release_bf16=True)
        self.model.prepare()
        base = getattr(cfg, "extras_name_or_path", None) or path
        try:
            vdir = loader.resolve_component_dir(base, "vae")
        except FileNotFoundError:
            vdir = None
        if vdir is None:
            self.vae = FakeVAE(self.vae_device_torch, self.torch_dtype)
        else:
            self.vae = self._build_vae(self._read_config(vdir))
            # only the encoder half is built: decoder.* / post_quant_conv.* of the file are skipped; every encoder tensor must be there
            missing, _ = loader.load_component(self.vae, vdir, strict=False,
                                               rename=lambda k: k if k.startswith(("encoder.", "quant_conv.")) else None)
            if missing:
                raise KeyError(f"VAE checkpoint {vdir} lacks {len(missing)} encoder tensor(s) of the configured architecture, e.g. {missing[:4]}: "
                               "encode_images would run on uninitialised weights")
            self.vae.prepare()
        self._load_text_side(base)
        if self.noise_scheduler is None:  # the trainer hands its sampler (ModelClass.get_train_scheduler()) to the constructor: keep it
            self.noise_scheduler = self._make_scheduler()
        self.is_loaded = True

    def _load_text_side(self, path):
        """text encoders / tokenizers after load_model: stand-ins, so that the reference's trainer can freeze / move / unload "them"
        (BaseSDTrainProcess.py:1892-1898, toolkit/unloader.py); the FLUX plug-in loads the real CLIP-L + T5 when the pipeline directory has them."""
        n = getattr(self, "_n_text_encoders", 1)
        self.text_encoder = [FakeTextEncoder(self.te_device_torch, self.torch_dtype) for _ in range(n)]
        self.tokenizer = [None] * n
        if n == 1:
            self.text_encoder, self.tokenizer = self.text_encoder[0], None

    def _build_vae(self, config=None):
        """AutoencoderKL encoder from vae/config.json (latent_channels, scaling / shift factor, use_quant_conv, widths); without a config
        the FLUX.1 defaults of the class."""
        from . import vae as nvae

        ops = _native_ops()
        kw = self._pick(config or {}, ("latent_channels", "block_out_channels", "layers_per_block", "scaling_factor", "shift_factor", "use_quant_conv"))
        if config and "norm_num_groups" in config:
            kw["groups"] = config["norm_num_groups"]
        if config and config.get("shift_factor", 0.0) is None:
            kw["shift_factor"] = 0.0
        return nvae.AutoencoderKLEncoder(dtype=self.torch_dtype, device=self.device_torch, ops=ops, **kw)

    def get_generation_pipeline(self):
        raise NotImplementedError("sampling / preview generation is outside the accelerated path (SURVEY.md section 8: out of scope); "
                                  "run previews through the reference's own pipeline on the saved LoRA")

    def generate_single_image(self, pipeline, gen_config, conditional_embeds, unconditional_embeds, generator, extra):
        raise NotImplementedError("sampling / preview generation is outside the accelerated path (SURVEY.md section 8: out of scope)")

    def get_prompt_embeds(self, prompt, control_images=None):
        raise NotImplementedError("text encoders are not part of the accelerated path: train with cached text embeddings "
                                  "(datasets: cache_text_embeddings: true; ai_toolkit_amd.batches reads the reference's _t_e_cache files)")

    def save_model(self, output_path, meta, save_dtype):
        """BaseModel.save_model (base_model.py:350-360): the denoiser in diffusers layout under <output_path>/<component> + aitk_meta.yaml."""
        import os

        import yaml

        from . import loader

        loader.save_component(self.model, os.path.join(output_path, self._component), dtype=_torch_dtype(save_dtype))
        with open(os.path.join(output_path, "aitk_meta.yaml"), "w") as f:
            yaml.dump(dict(meta), f)

    def encode_audio(self, audio_data_list):
        raise NotImplementedError("Audio encoding not implemented for this model.")  # base_model.py:1178-1180

    def get_model_to_train(self):
        return self.model

    # the reference reads AND assigns the denoiser through these aliases (toolkit/models/base_model.py:199-224; `self.sd.unet =
    # self.accelerator.prepare(self.sd.unet)`, jobs/process/BaseSDTrainProcess.py:751)
    @property
    def transformer(self):
        return self.model

    @transformer.setter
    def transformer(self, value):
        self.model = value

    @property
    def unet(self):
        return self.model

    @unet.setter
    def unet(self, value):
        self.model = value

    @property
    def model_unwrapped(self):
        return getattr(self.model, "module", self.model) if self.model.__class__.__name__ == "DistributedDataParallel" else self.model

    @property
    def unet_unwrapped(self):
        return self.model_unwrapped

    # `self.sd.network = self.network` (BaseSDTrainProcess.py:1985): a network the reference built itself is checked here for the options the
    # fused graph cannot honour (network level; each module is checked when apply_to attaches it, adopt.register_foreign_adapter), so a
    # config outside the accelerated path fails at set-up, not as a base-only model in the first step
    @property
    def network(self):
        return self.__dict__.get("_network")

    @network.setter
    def network(self, value):
        if value is not None:
            from .adopt import check_foreign_network

            check_foreign_network(value)
        self.__dict__["_network"] = value

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
    _component = "transformer"
    _n_text_encoders = 2  # CLIP-L + T5 (flux_kontext.py:120-170)

    def _load_text_side(self, path):
        import os

        if all(os.path.isdir(os.path.join(str(path), sub)) for sub in ("tokenizer", "tokenizer_2", "text_encoder", "text_encoder_2")):
            self.load_text_encoders(path)  # real encoders: prompts can be encoded (and cached) by the trainer
        else:
            super()._load_text_side(path)

    def _build_native(self, config=None):
        from .flux import FluxTransformer2DModel

        ops = _native_ops()
        kw = self._pick(config or {}, ("in_channels", "num_layers", "num_single_layers", "attention_head_dim", "num_attention_heads",
                                       "joint_attention_dim", "pooled_projection_dim", "guidance_embeds", "axes_dims_rope"))
        return FluxTransformer2DModel(dtype=self.torch_dtype, device=self.device_torch, ops=ops, **kw)

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

    def load_text_encoders(self, path=None):
        """CLIP-L (`text_encoder` / `tokenizer`) and T5 (`text_encoder_2` / `tokenizer_2`) of a diffusers FLUX pipeline directory through
        `transformers` (flux_kontext.py:120-170 loads the same four sub-folders) — only when prompts must be encoded here (no text-embedding
        cache); frozen, model dtype, on the plug-in's device.  Library path, see encode_prompts_flux."""
        import os

        from transformers import CLIPTextModel, CLIPTokenizer, T5EncoderModel, T5TokenizerFast

        cfg = self.model_config
        path = path or getattr(cfg, "extras_name_or_path", None) or getattr(cfg, "name_or_path", None)
        for sub in ("tokenizer", "tokenizer_2", "text_encoder", "text_encoder_2"):
            if not os.path.isdir(os.path.join(str(path), sub)):
                raise FileNotFoundError(f"{path!r} has no '{sub}' sub-folder: prompts cannot be encoded here — train with cached text embeddings "
                                        "(datasets: cache_text_embeddings: true)")
        kw = dict(local_files_only=True)
        self.tokenizer = [CLIPTokenizer.from_pretrained(path, subfolder="tokenizer", **kw), T5TokenizerFast.from_pretrained(path, subfolder="tokenizer_2", **kw)]
        te = [CLIPTextModel.from_pretrained(path, subfolder="text_encoder", torch_dtype=self.torch_dtype, **kw),
              T5EncoderModel.from_pretrained(path, subfolder="text_encoder_2", torch_dtype=self.torch_dtype, **kw)]
        for m in te:
            m.to(self.te_device_torch).requires_grad_(False).eval()
        self.text_encoder = te
        return te

    def get_prompt_embeds(self, prompt, control_images=None):
        """flux_kontext.py:354-367: encode_prompts_flux(tokenizer, text_encoder, prompt, max_length=512) -> PromptEmbeds(text) + pooled_embeds.
        Encoders are loaded on first use (load_text_encoders) or taken from `self.text_encoder` / `self.tokenizer` if the caller set them."""
        te = self.text_encoder
        if not te or not self.tokenizer or any(isinstance(t, FakeTextEncoder) or t.__class__.__name__ == "FakeTextEncoder" for t in (te if isinstance(te, (list, tuple)) else [te])):
            self.load_text_encoders()
        if not isinstance(prompt, (list, tuple)):
            prompt = [prompt]
        with torch.no_grad():
            embeds, pooled = encode_prompts_flux(self.tokenizer, self.text_encoder, list(prompt), max_length=512)
        pe = _prompt_embeds_class()(embeds)
        pe.pooled_embeds = pooled
        return pe

    def get_noise_prediction(self, latent_model_input, timestep, text_embeddings, guidance_embedding_scale=1.0,
                             bypass_guidance_embedding=False, **kwargs):
        """latent_model_input [B,16,H,W], timestep [B] on the 0..1000 scale -> prediction [B,16,H,W]
        (flux_kontext.py:243-352 without the kontext control branch)."""
        bs, c, h, w = latent_model_input.shape
        if c != 16:
            raise ValueError(f"expected 16 latent channels, got {c} (kontext control channels are not on the fused path)")
        dev = self.device_torch
        text, pooled = _embeds(text_embeddings)
        with torch.no_grad():
            packed = latent_model_input.reshape(bs, c, h // 2, 2, w // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(bs, (h // 2) * (w // 2), c * 4)
            from .trainer import make_ids

            img_ids, txt_ids = make_ids(h, w, text.shape[1], dev)  # (0, row, col) ids of the 2x2-packed grid, cached per bucket (no host-to-device copy per step)
            if bypass_guidance_embedding:  # toolkit/models/flux.py:9-35: the guidance embedder is skipped for this call
                guidance = None
            elif isinstance(guidance_embedding_scale, list):
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
    _component = "transformer"

    def _build_native(self, config=None):
        from .wan import WanTransformer3DModel

        ops = _native_ops()
        kw = self._pick(config or {}, ("patch_size", "num_attention_heads", "attention_head_dim", "in_channels", "out_channels", "text_dim",
                                       "freq_dim", "ffn_dim", "num_layers", "eps"))
        if (config or {}).get("image_dim") or (config or {}).get("added_kv_proj_dim"):
            raise NotImplementedError("Wan2.1 image-to-video checkpoints (image_dim / added_kv_proj_dim) are not on the fused path: text-to-video only")
        return WanTransformer3DModel(dtype=self.torch_dtype, device=self.device_torch, ops=ops, **kw)

    def _build_vae(self, config=None):
        from . import wan_vae

        ops = _native_ops()
        return wan_vae.AutoencoderKLWanEncoder(dtype=self.torch_dtype, device=self.device_torch, ops=ops)

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

    def load_text_encoders(self, path=None):
        """UMT5 encoder + tokenizer of a diffusers Wan2.1 pipeline directory through `transformers` (library path)."""
        import os

        from transformers import AutoTokenizer, UMT5EncoderModel

        cfg = self.model_config
        path = path or getattr(cfg, "extras_name_or_path", None) or getattr(cfg, "name_or_path", None)
        for sub in ("tokenizer", "text_encoder"):
            if not os.path.isdir(os.path.join(str(path), sub)):
                raise FileNotFoundError(f"{path!r} has no '{sub}' sub-folder: prompts cannot be encoded here — train with cached text embeddings "
                                        "(datasets: cache_text_embeddings: true)")
        self.tokenizer = AutoTokenizer.from_pretrained(path, subfolder="tokenizer", local_files_only=True)
        self.text_encoder = UMT5EncoderModel.from_pretrained(path, subfolder="text_encoder", torch_dtype=self.torch_dtype, local_files_only=True)
        self.text_encoder.to(self.te_device_torch).requires_grad_(False).eval()
        return self.text_encoder

    def _load_text_side(self, path):
        import os

        if all(os.path.isdir(os.path.join(str(path), sub)) for sub in ("tokenizer", "text_encoder")):
            self.load_text_encoders(path)
        else:
            super()._load_text_side(path)

    def get_prompt_embeds(self, prompt, control_images=None):
        te = self.text_encoder
        if te is None or te.__class__.__name__ == "FakeTextEncoder":
            self.load_text_encoders()
        prompt = list(prompt) if isinstance(prompt, (list, tuple)) else [prompt]
        with torch.no_grad():
            return _prompt_embeds_class()(encode_prompts_wan(self.tokenizer, self.text_encoder, prompt, 512, self.torch_dtype))

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

    _component = "unet"

    # BaseModel exposes is_xl as a read-only property of model_config (base_model.py:233); here it is also a constructor argument
    @property
    def is_xl(self):
        return self._is_xl

    @is_xl.setter
    def is_xl(self, v):
        self._is_xl = bool(v)

    def __init__(self, device, model_config=None, dtype="bf16", custom_pipeline=None, noise_scheduler=None, *, model=None, vae=None,
                 is_xl=None, prediction_type=None, **kwargs):
        super().__init__(device, model_config, dtype, custom_pipeline, noise_scheduler, model=model, vae=vae, **kwargs)
        from .ddpm import DDPMTrainSchedule

        if is_xl is None:  # toolkit/config_modules.py ModelConfig.is_xl / arch
            is_xl = bool(getattr(model_config, "is_xl", False)) or getattr(model_config, "arch", None) == "sdxl"
        if prediction_type is None:  # base_model.py:127
            prediction_type = "v_prediction" if getattr(model_config, "is_v_pred", False) else "epsilon"

        self.is_xl = bool(is_xl)
        self.prediction_type = prediction_type
        # the trainer hands the sampler it built (ModelClass.get_train_scheduler(), BaseSDTrainProcess.py:1767-1770, 1794-1801) to the constructor
        # and later drives THAT object (set_timesteps / timesteps / config): keep it when it is a working DDPM table (diffusers DDPMScheduler:
        # `alphas_cumprod`), the native schedule otherwise
        if not torch.is_tensor(getattr(noise_scheduler, "alphas_cumprod", None)):
            self.noise_scheduler = DDPMTrainSchedule(prediction_type=prediction_type)

    @staticmethod
    def get_train_scheduler():
        from .ddpm import DDPMTrainSchedule

        return DDPMTrainSchedule()

    def _build_native(self, config=None):
        from .unet import SD15_CONFIG, SDXL_CONFIG, UNet2DConditionModel

        ops = _native_ops()
        base = dict(SDXL_CONFIG if self.is_xl else SD15_CONFIG)
        over = self._pick(config or {}, tuple(base) + ("norm_num_groups",))
        if config and (config.get("addition_embed_type") == "text_time") != self.is_xl:
            raise ValueError(f"unet/config.json says addition_embed_type={config.get('addition_embed_type')!r} but the model was configured as "
                             f"{'SDXL' if self.is_xl else 'SD1.x'} (model.arch / is_xl)")
        base.update(over)
        return UNet2DConditionModel(**base, dtype=self.torch_dtype, device=self.device_torch, ops=ops)

    def _build_vae(self, config=None):
        """SD1.x / SDXL AutoencoderKL: 4 latent channels, `quant_conv` on the moments, scaling 0.18215 / 0.13025, no shift — read from
        vae/config.json when present (the FLUX defaults of the encoder class do not apply here)."""
        cfg = dict(latent_channels=4, use_quant_conv=True, scaling_factor=0.13025 if self.is_xl else 0.18215, shift_factor=0.0)
        cfg.update({k: v for k, v in (config or {}).items() if v is not None})
        return super()._build_vae(cfg)

    def _make_scheduler(self):
        from .ddpm import DDPMTrainSchedule

        return DDPMTrainSchedule(prediction_type=self.prediction_type)  # is_v_pred survives load_model

    def get_bucket_divisibility(self):
        return 8  # vae scale factor 8; the UNet's three (SDXL: two) stride-2 levels are covered by the reference's 64-px bucket tolerance

    @property
    def _n_text_encoders(self):
        return 2 if self.is_xl else 1

    def load_text_encoders(self, path=None):
        """CLIP text encoder(s) + tokenizer(s) of a diffusers SD1.x / SDXL pipeline directory through `transformers` (library path)."""
        import os

        from transformers import CLIPTextModel, CLIPTextModelWithProjection, CLIPTokenizer

        cfg = self.model_config
        path = path or getattr(cfg, "extras_name_or_path", None) or getattr(cfg, "name_or_path", None)
        subs = (("tokenizer", "text_encoder", CLIPTextModel),) + ((("tokenizer_2", "text_encoder_2", CLIPTextModelWithProjection),) if self.is_xl else ())
        for tk, te, _ in subs:
            for sub in (tk, te):
                if not os.path.isdir(os.path.join(str(path), sub)):
                    raise FileNotFoundError(f"{path!r} has no '{sub}' sub-folder: prompts cannot be encoded here — train with cached text embeddings "
                                            "(datasets: cache_text_embeddings: true)")
        toks = [CLIPTokenizer.from_pretrained(path, subfolder=tk, local_files_only=True) for tk, _, _ in subs]
        tes = [cls.from_pretrained(path, subfolder=te, torch_dtype=self.torch_dtype, local_files_only=True).to(self.te_device_torch).requires_grad_(False).eval()
               for _, te, cls in subs]
        self.tokenizer, self.text_encoder = (toks, tes) if self.is_xl else (toks[0], tes[0])
        return self.text_encoder

    def _load_text_side(self, path):
        import os

        if all(os.path.isdir(os.path.join(str(path), sub)) for sub in (("tokenizer", "text_encoder") + (("tokenizer_2", "text_encoder_2") if self.is_xl else ()))):
            self.load_text_encoders(path)
        else:
            super()._load_text_side(path)

    def get_prompt_embeds(self, prompt, control_images=None, prompt2=None, long_prompts=False, max_length=None, dropout_prob=0.0):
        """toolkit/stable_diffusion_model.py:2400-2440 (the sd1 / sdxl branches of encode_prompt)."""
        te = self.text_encoder
        tes = te if isinstance(te, (list, tuple)) else [te]
        if not te or any(t.__class__.__name__ == "FakeTextEncoder" for t in tes):
            self.load_text_encoders()
        prompt = list(prompt) if isinstance(prompt, (list, tuple)) else [prompt]
        with torch.no_grad():
            if self.is_xl:
                e, pooled = encode_prompts_xl(self.tokenizer, self.text_encoder, prompt, prompt2, truncate=not long_prompts, max_length=max_length, dropout_prob=dropout_prob)
                return _prompt_embeds_class()([e, pooled])
            return _prompt_embeds_class()(encode_prompts(self.tokenizer, self.text_encoder, prompt, truncate=not long_prompts, max_length=max_length, dropout_prob=dropout_prob))

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

    def _noise_coefficients(self, timesteps, dtype):
        """(sqrt(alphas_cumprod[t]), sqrt(1 - alphas_cumprod[t])) from whichever schedule object this model holds: the native one, or the
        trainer's diffusers DDPMScheduler (its add_noise / get_velocity read the same table the same way)."""
        sch = self.noise_scheduler
        if hasattr(sch, "noise_coefficients"):
            return sch.noise_coefficients(timesteps, dtype)
        acp = sch.alphas_cumprod.to(device=timesteps.device, dtype=dtype)
        t = timesteps.long().reshape(-1)
        return acp[t] ** 0.5, (1 - acp[t]) ** 0.5

    def add_noise(self, original_samples, noise, timesteps, **kwargs):
        timesteps = timesteps.to(original_samples.device).reshape(-1)
        if timesteps.numel() == 1 and original_samples.shape[0] > 1:  # stable_diffusion_model.py:1865-1866
            timesteps = timesteps.expand(original_samples.shape[0])
        a, s = self._noise_coefficients(timesteps, original_samples.dtype)
        a, s = a.to(original_samples.dtype).view(-1, 1, 1, 1), s.to(original_samples.dtype).view(-1, 1, 1, 1)
        return a * original_samples + s * noise

    def get_loss_target(self, *args, **kwargs):
        noise, batch, timesteps = kwargs.get("noise"), kwargs.get("batch"), kwargs.get("timesteps")
        if noise is None:
            raise ValueError("Noise is not provided")
        if self.prediction_type == "v_prediction":
            if batch is None or timesteps is None:
                raise ValueError("v_prediction needs the batch latents and the timesteps")
            a, s = self._noise_coefficients(timesteps.to(noise.device), noise.dtype)
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
