"""LoRA network surface for the fused MI355X path.

Mirrors the Python-visible contract of the reference's adapter layer so trainer code, optimizers, EMA and saved files
keep working:

  * LoRAModule            <- toolkit/lora_special.py:46-135 (attributes, init order, `alpha` buffer, fp32 params)
                             toolkit/network_mixins.py:170-195 (`scale` float + non-persistent `_runtime_scale` buffer)
  * FusedLoRANetwork      <- toolkit/lora_special.py:276-775 (module discovery / naming), toolkit/network_mixins.py:491-932
                             (multiplier -> torch_multiplier, context-manager activation, state-dict I/O, PEFT renaming),
                             toolkit/kohya_lora.py:1030-1074 (prepare_optimizer_params)

What is different underneath: all lora_down / lora_up weights live in ONE flat fp32 arena (plus flat grad / Adam / EMA
arenas and a bf16 shadow arena holding every matrix in both orientations).  The nn.Parameters are views into that
arena, so a fused optimizer kernel and a single RCCL all-reduce can treat the adapter as one tensor while
`state_dict()`, `torch.optim.*` and `clip_grad_norm_` still see ordinary per-layer parameters.
"""
import json
import math
import os
import weakref
from collections import OrderedDict
from typing import List

import torch
import torch.nn as nn

LINEAR_MODULES = ["Linear", "LoRACompatibleLinear", "QLinear", "OstrisLinear"]  # toolkit/lora_special.py:29-35


class LoRAModule(nn.Module):
    """State holder for one wrapped Linear.  The arithmetic is NOT here: the fused GEMM reads the bf16 shadows."""

    is_lokr = False

    def __init__(self, lora_name, org_module, multiplier=1.0, lora_dim=4, alpha=1, dropout=None, rank_dropout=None,
                 module_dropout=None, network=None, use_bias=False, **kwargs):
        super().__init__()
        self.can_merge_in = True
        self.network_ref = weakref.ref(network) if network is not None else (lambda: None)
        self.is_checkpointing = False
        self._multiplier = None
        self.lora_name = lora_name
        self.orig_module_ref = weakref.ref(org_module)
        if getattr(org_module, "bias", None) is None:
            use_bias = False
        if use_bias:
            raise NotImplementedError("use_bias LoRA is not on the fused path")
        # 3x3 Conv2d (network.conv; toolkit/lora_special.py:95-104): lora_down = Conv2d(in, r, 3, stride, padding), lora_up = Conv2d(r, out, 1).
        # Held as the flattened matrices [r, in*9] (columns cin*9 + ky*3 + kx = the Conv2d weight's own memory order) and [out, r]: the
        # same number of kaiming draws with the same fan-in as the reference's Conv2d constructors, i.e. the same RNG consumption.
        self.is_conv3x3 = bool(getattr(org_module, "is_conv3x3", False))
        if self.is_conv3x3:
            assert org_module.cin_pad == org_module.in_channels and org_module.cout_pad == org_module.out_channels
            self.conv_cin, self.conv_stride = org_module.in_channels, org_module.stride
            in_dim, out_dim = org_module.in_channels * 9, org_module.out_channels
        else:
            in_dim, out_dim = org_module.in_features, org_module.out_features
        self.lora_dim = lora_dim
        self.full_rank = False
        # same construction (hence same RNG consumption) as toolkit/lora_special.py:105-122
        self.lora_down = nn.Linear(in_dim, lora_dim, bias=False)
        self.lora_up = nn.Linear(lora_dim, out_dim, bias=False)
        if isinstance(alpha, torch.Tensor):
            alpha = float(alpha.detach().float().item())
        alpha = lora_dim if alpha is None or alpha == 0 else alpha
        self._set_runtime_scale(float(alpha) / lora_dim)
        self.register_buffer("alpha", torch.tensor(alpha))
        nn.init.kaiming_uniform_(self.lora_down.weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_up.weight)
        self.magnitude = None  # DoRAModule sets it
        self.multiplier = multiplier
        self.org_module = [org_module]  # list keeps it out of state_dict (reference lines 125-126)
        # dropout on lx = lora_down(x), rank_dropout (per sample, per rank, rescaled by 1 / (1 - p)) and module_dropout
        # (toolkit/network_mixins.py:198-228): executed as a multiplier on the rank-space activation inside aitk_lora_down
        self.dropout, self.rank_dropout, self.module_dropout = dropout, rank_dropout, module_dropout
        # arena bookkeeping (filled by FusedLoRANetwork._build_arena)
        self.off_down = self.off_up = -1
        self.sh_down = self.sh_down_lo = self.sh_downT3 = self.sh_up3 = self.sh_upT = self.sh_upT_lo = None  # build_arena
        self.sh_down_stack = self.sh_down_dgrad = None  # 3x3-conv adapters: [A_hi ; A_lo] tap-major, rotated dgrad filter over the dT slab
        self.g_down = self.g_up = None

    def _set_runtime_scale(self, value):
        self.scale = float(value)
        rs = getattr(self, "_runtime_scale", None)
        if rs is None:
            self.register_buffer("_runtime_scale", torch.tensor(self.scale, dtype=torch.float32), persistent=False)
        else:
            with torch.no_grad():
                rs.fill_(self.scale)

    def apply_to(self):
        """The reference swaps org_module.forward; the fused path instead tags the base layer with its adapter."""
        object.__setattr__(self.org_module[0], "lora", self)  # plain attribute: keeps the adapter out of the base state_dict

    @property
    def in_features(self):
        return self.lora_down.weight.shape[1]

    @property
    def out_features(self):
        return self.lora_up.weight.shape[0]


class DoRAModule(LoRAModule):
    """toolkit/models/DoRA.py:36-106: same surface as LoRAModule plus `magnitude` [out] (initialised to the row norm of the
    base weight), lora_up created first and zeroed, lora_down ~ N(0, 1/r) — same construction order, hence the same RNG
    consumption as the reference.  The fused GEMM applies c = magnitude / ||W + s*up@down||_row as a column scale."""

    def __init__(self, lora_name, org_module, multiplier=1.0, lora_dim=4, alpha=1, network=None, **kwargs):
        nn.Module.__init__(self)
        self.can_merge_in = False  # network_mixins.py:894-897: merge_in is a no-op for DoRA
        self.network_ref = weakref.ref(network) if network is not None else (lambda: None)
        self.is_checkpointing = False
        self._multiplier = None
        self.lora_name = lora_name
        self.orig_module_ref = weakref.ref(org_module)
        in_dim, out_dim = org_module.in_features, org_module.out_features
        self.lora_dim = lora_dim
        self.full_rank = False
        if isinstance(alpha, torch.Tensor):
            alpha = float(alpha.detach().float().item())
        alpha = lora_dim if alpha is None or alpha == 0 else alpha
        self._set_runtime_scale(float(alpha) / lora_dim)
        self.lora_up = nn.Linear(lora_dim, out_dim, bias=False)
        self.lora_up.weight.data = torch.zeros_like(self.lora_up.weight.data)
        self.lora_down = nn.Linear(in_dim, lora_dim, bias=False)
        self.lora_down.weight.data = torch.randn_like(self.lora_down.weight.data) * (1 / torch.sqrt(torch.tensor(lora_dim).float()))
        if getattr(org_module, "qweight", None) is not None:  # quantised base: the norm of the DEQUANTISED weight (DoRA.py:105-109, get_orig_weight)
            w = (org_module.qweight.view(torch.float8_e4m3fn).float() * org_module.wscale[:, None].float()).to(org_module.weight.dtype).float().cpu()
        else:
            w = org_module.weight.data.detach().float().cpu()
        self.magnitude = nn.Parameter(torch.linalg.norm(w, dim=1).clone())  # lora_up = 0 at init: ||W + up@down|| = ||W||
        self.multiplier = multiplier
        self.org_module = [org_module]
        self.dropout = self.rank_dropout = self.module_dropout = None
        self.off_down = self.off_up = self.off_mag = -1
        self.sh_down = self.sh_down_lo = self.sh_downT3 = self.sh_up3 = self.sh_upT = self.sh_upT_lo = None
        self.g_down = self.g_up = self.g_mag = None
        self.c = None       # fp32 [out]: magnitude / ||W + s*up@down||, refreshed after every optimizer step
        self.w2 = None      # fp32 [out]: ||W_j||^2 of the frozen base weight
        self.y_lin = None   # this step's linear output (c*z + b), kept for d magnitude


def add_model_hash_to_meta(state_dict, meta):
    """sshs_model_hash / sshs_legacy_hash of sd-webui-additional-networks, as the reference stamps every saved adapter
    (toolkit/metadata.py:32-48, toolkit/train_tools.py:162-185): serialise the tensors with only the `ss_*` metadata, sha256 of
    everything after the safetensors header, and the first 8 hex digits of sha256 over bytes [0x100000, 0x110000)."""
    import hashlib

    from safetensors.torch import save as st_save

    blob = st_save(state_dict, {k: v for k, v in meta.items() if k.startswith("ss_")})
    n = int.from_bytes(blob[:8], "little")
    meta["sshs_model_hash"] = hashlib.sha256(blob[n + 8:]).hexdigest()
    meta["sshs_legacy_hash"] = hashlib.sha256(blob[0x100000:0x110000]).hexdigest()[0:8]
    return meta


def factorization(dimension: int, factor: int = -1):
    """Split `dimension` into (m, n), m * n == dimension, m <= n.  Behaviour of the LyCORIS rule the reference uses
    (toolkit/models/lokr.py:22-59): an exact divisor `factor` is taken as is; otherwise walk the divisor pairs upwards from (1, dim)
    and keep the last pair whose sum did not grow and whose smaller member does not exceed `factor` (-1: no limit)."""
    if factor > 0 and dimension % factor == 0:
        return factor, dimension // factor
    limit = dimension if factor == -1 else factor
    best = (1, dimension)
    for cand in range(2, dimension + 1):
        if best[0] >= best[1]:
            break
        if dimension % cand:
            continue
        other = dimension // cand
        if cand + other > best[0] + best[1] or cand > limit:
            break
        best = (cand, other)
    lo, hi = min(best), max(best)
    return lo, hi


class _ParamProxy:
    """`.weight` view of a module-level nn.Parameter, so the arena builder can treat lokr_w2 / lokr_w1 like lora_down / lora_up."""

    def __init__(self, module, name):
        self._m, self._n = module, name

    @property
    def weight(self):
        return getattr(self._m, self._n)

    @weight.setter
    def weight(self, value):
        setattr(self._m, self._n, value)


def kron_lds_bytes(a_in, b_in, a_out, b_out):
    """LDS footprint of aitk_kron_apply with both factors resident (csrc/kron.hip kron_layout): the per-token Kronecker kernel keeps lokr_w1
    [a_out, a_in], W2 [b_out, b_in] and one token's X / intermediate / output tiles in the CU's 160-KB LDS."""
    c16, c32, c8 = (lambda v: -(-v // 16) * 16), (lambda v: -(-v // 32) * 32), (lambda v: -(-v // 8) * 8)
    xs, as_ = c32(b_in) + 8, c32(a_in) + 8
    return 2 * (c16(a_in) * xs + c16(b_out) * xs + c16(a_out) * as_ + c16(b_out) * as_ + c8(a_out * b_out))


def kron_mix_lds_bytes(a_in, b, a_out):
    """LDS footprint of aitk_kron_apply with ONLY the small factor resident (B = identity: out_m = A X_m, X_m [a_in, b])."""
    c16, c32, c8 = (lambda v: -(-v // 16) * 16), (lambda v: -(-v // 32) * 32), (lambda v: -(-v // 8) * 8)
    as_ = c32(a_in) + 8
    return 2 * (c16(a_out) * as_ + c16(b) * as_ + c8(a_out * b))


def check_kron_fits(name, in_m, in_n, out_l, out_k):
    """How a LoKr layer runs: "token" — both factors resident in the per-token Kronecker kernel (every layer size of FLUX / Wan / the UNets under the
    reference's default factorisation) — or "two_stage": an explicit small `network.lokr_factor` on a wide layer makes W2 [out / f, in / f] a
    GEMM-sized matrix (factor 4 on a 3072 x 3072 Linear: 768 x 768) that does not fit the kernel's LDS; the product is then split into a plain GEMM
    with W2 over (token x factor-index) rows and the per-token mix with the small factor lokr_w1 on the NARROWER side (graph._kron_any).
    Raises, where the adapter is attached and with the numbers, when even the small-factor mix does not fit."""
    need = max(kron_lds_bytes(in_m, in_n, out_l, out_k), kron_lds_bytes(out_l, out_k, in_m, in_n))  # forward and data gradient (transposed factors)
    if need <= 160 * 1024:
        return "token"
    b = min(in_n, out_k)
    mix = max(kron_mix_lds_bytes(in_m, b, out_l), kron_mix_lds_bytes(out_l, b, in_m))
    if mix <= 160 * 1024:
        return "two_stage"
    raise NotImplementedError(
        f"{name}: LoKr factors lokr_w1 {out_l}x{in_m}, W2 {out_k}x{in_n} need {need // 1024} KiB of LDS in the per-token Kronecker kernel and "
        f"{mix // 1024} KiB in the two-stage form (160 KiB per CU): use the default factorisation (network.lokr_factor: -1) or a larger factor")


class LoKrModule(LoRAModule):
    """toolkit/models/lokr.py:76-242 for a Linear: delta W = kron(lokr_w1 [out_l, in_m], W2 [out_k, in_n]) * scale,
    (out_l, out_k) = factorization(out), (in_m, in_n) = factorization(in).
      * lora_dim >= max(out_k, in_n) / 2 (the reference's default `lokr_full_rank: true`, toolkit/config_modules.py:204-209):
        W2 = lokr_w2, full; lokr_w2 = 0, lokr_w1 kaiming-uniform at init (one RNG draw); alpha = lora_dim => scale 1.
      * otherwise (lokr.py:184-197): W2 = lokr_w2_a [out_k, r] @ lokr_w2_b [r, in_n]; lokr_w2_a kaiming-uniform (drawn first), lokr_w2_b = 0,
        then lokr_w1; scale = alpha / r.  The kernels read the composed W2 from the bf16 shadows (shadow kind 3 composes a @ b in fp32);
        the pair's gradients come from the gradient of the composed factor (aitk_lokr_lowrank_grad).
    The arithmetic is aitk_kron_apply (per-token A . X . B^T, csrc/kron.hip) — kron(w1, W2) is never formed.  In the arena W2 (or the
    pair a | b, back to back) takes the `down` slot, lokr_w1 the `up` slot."""

    is_lokr = True

    def __init__(self, lora_name, org_module, multiplier=1.0, lora_dim=4, alpha=1, network=None, factor=-1, **kwargs):
        nn.Module.__init__(self)
        self.can_merge_in = True
        self.network_ref = weakref.ref(network) if network is not None else (lambda: None)
        self.is_checkpointing = False
        self._multiplier = None
        self.lora_name = lora_name
        self.orig_module_ref = weakref.ref(org_module)
        self.lora_dim = lora_dim
        self.full_rank = False
        in_dim, out_dim = org_module.in_features, org_module.out_features
        self.in_m, self.in_n = factorization(in_dim, int(factor))
        self.out_l, self.out_k = factorization(out_dim, int(factor))
        if self.in_n % 8 or self.out_k % 8:
            raise NotImplementedError(f"LoKr factor {self.out_k}x{self.in_n}: the kron kernel needs multiples of 8")
        self.kron_two_stage = check_kron_fits(lora_name, self.in_m, self.in_n, self.out_l, self.out_k) == "two_stage"
        self.use_w1 = True
        self.use_w2 = lora_dim >= max(self.out_k, self.in_n) / 2
        self.lokr_w1 = nn.Parameter(torch.empty(self.out_l, self.in_m))
        if self.use_w2:
            self.lokr_w2 = nn.Parameter(torch.empty(self.out_k, self.in_n))
        else:
            self.lokr_w2_a = nn.Parameter(torch.empty(self.out_k, lora_dim))
            self.lokr_w2_b = nn.Parameter(torch.empty(lora_dim, self.in_n))
        if isinstance(alpha, torch.Tensor):
            alpha = float(alpha.detach().float().item())
        alpha = lora_dim if alpha is None or alpha == 0 else alpha
        if self.use_w2:
            alpha = lora_dim  # both factors full: scale 1 (lokr.py:203-206)
        self._set_runtime_scale(float(alpha) / lora_dim)
        self.register_buffer("alpha", torch.tensor(alpha))
        if self.use_w2:
            nn.init.constant_(self.lokr_w2, 0)
        else:
            nn.init.kaiming_uniform_(self.lokr_w2_a, a=math.sqrt(5))
            nn.init.constant_(self.lokr_w2_b, 0)
        nn.init.kaiming_uniform_(self.lokr_w1, a=math.sqrt(5))
        self.magnitude = None
        self.multiplier = multiplier
        self.org_module = [org_module]
        self.dropout = self.rank_dropout = self.module_dropout = None
        self._in, self._out = in_dim, out_dim
        self.off_down = self.off_up = -1
        self.sh_down = self.sh_downT = self.sh_up = self.sh_upT = None
        self.g_down = self.g_up = None
        self.g_w2a = self.g_w2b = None

    def factor_params(self):
        """(key, parameter) in the reference module's named_parameters() order."""
        if self.use_w2:
            return [("lokr_w1", self.lokr_w1), ("lokr_w2", self.lokr_w2)]
        return [("lokr_w1", self.lokr_w1), ("lokr_w2_a", self.lokr_w2_a), ("lokr_w2_b", self.lokr_w2_b)]

    def composed_w2(self):
        """fp32 W2 (load-time utility for merge_in; the step reads the composed bf16 shadow instead)."""
        return self.lokr_w2.data if self.use_w2 else (self.lokr_w2_a.data @ self.lokr_w2_b.data)

    @property
    def lora_down(self):
        if not self.use_w2:
            raise AttributeError("low-rank LoKr has no single `down` matrix: use factor_params()")
        return _ParamProxy(self, "lokr_w2")

    @property
    def lora_up(self):
        return _ParamProxy(self, "lokr_w1")

    @property
    def in_features(self):
        return self._in

    @property
    def out_features(self):
        return self._out


def _default_mask_provider(name, kind, shape, device):
    """the reference's draws: torch.rand on the global generators (toolkit/network_mixins.py:200, 220) — the module_dropout coin on the host, the
    element / rank masks on the adapter's device (graph-safe there: the CUDA generator's offset is advanced by every replay of a captured graph)"""
    return torch.rand(shape, device="cpu" if kind == "module" else device)


class FusedLoRANetwork(nn.Module):
    """Drop-in for LoRASpecialNetwork: transformer models in PEFT format (FLUX, Wan: `transformer.<path>.lora_A/B.weight`, alpha
    forced to the rank) and UNet models (SD1.5 / SDXL) in kohya format (`lora_unet_<path_with_underscores>.lora_down/up.weight` +
    `.alpha`; Linear and 1x1-Conv2d children of every Transformer2DModel, toolkit/kohya_lora.py:750, lora_special.py:463-502)."""

    def __init__(self, unet, lora_dim=4, alpha=1.0, multiplier=1.0, target_lin_modules=("FluxTransformer2DModel",),
                 transformer_only=True, transformer_block_names=None, ignore_if_contains=None, only_if_contains=None,
                 is_transformer=True, peft_format=True, network_type="lora", base_model_version="flux1", lokr_factor=-1,
                 base_model=None, conv_lora_dim=None, conv_alpha=None, dropout=None, rank_dropout=None, module_dropout=None,
                 target_conv_modules=("ResnetBlock2D", "Downsample2D", "Upsample2D")):
        super().__init__()
        self.dropout, self.rank_dropout, self.module_dropout = dropout, rank_dropout, module_dropout
        if (dropout or rank_dropout or module_dropout) and network_type.lower() != "lora":
            raise NotImplementedError("dropout variants: plain LoRA modules only on the fused path")
        # draws the random numbers of the dropout masks: (lora_name, kind, shape, device) -> uniform [0, 1) tensor; the default is
        # torch.rand on the adapter device like the reference (network_mixins.py:200, 220); tests inject a keyed provider
        self.mask_provider = _default_mask_provider
        # the reference holds a weak reference to the model plug-in for the save / load key-conversion hooks (lora_special.py:373-375)
        self.base_model_ref = weakref.ref(base_model) if base_model is not None else None
        assert network_type.lower() in ("lora", "dora", "lokr"), "locon / lorm / full-rank adapters are not on the fused path"
        # network.conv (toolkit/lora_special.py:381-382, 585-590, 678-681; toolkit/kohya_lora.py:751): the Linear / Conv2d children of
        # ResnetBlock2D, Downsample2D and Upsample2D are wrapped too — 3x3 convolutions at conv_lora_dim / conv_alpha, the rest at the
        # linear rank.  UNet (kohya-format) plain-LoRA networks only.
        self.conv_lora_dim, self.conv_alpha = (conv_lora_dim or None), conv_alpha
        if self.conv_lora_dim is not None:
            if network_type.lower() != "lora" or is_transformer or peft_format:
                raise NotImplementedError("network.conv: plain LoRA on UNet (kohya-format) networks only on the fused path")
            if self.conv_lora_dim > 64:
                raise NotImplementedError("network.conv ranks above 64 are not on the fused path (the split-slab epilogue of the implicit-GEMM "
                                          "lora_down covers one 128-column tile = 64 ranks hi + lo)")
            target_lin_modules = tuple(target_lin_modules) + tuple(target_conv_modules)
        # toolkit/lora_special.py:403-408
        module_class = {"lora": LoRAModule, "dora": DoRAModule, "lokr": LoKrModule}[network_type.lower()]
        module_kwargs = {"factor": lokr_factor} if network_type.lower() == "lokr" else {}  # lora_special.py:601-602
        self.lora_dim = lora_dim
        self.network_type = network_type
        # transformer models are always PEFT format (lora_special.py:418-422); UNet models keep the kohya format
        self.peft_format = bool(peft_format or is_transformer)
        self.is_transformer = bool(is_transformer)
        if not self.peft_format and network_type.lower() != "lora":
            raise NotImplementedError("kohya-format (UNet) networks: plain LoRA only on the fused path")
        self.is_lorm = False
        self.is_active = False
        self.is_merged_in = False
        self.base_model_version = base_model_version
        # PEFT format: alpha forced to rank => scale 1 (toolkit/lora_special.py:428-433); kohya format: alpha as configured
        self.alpha = lora_dim if self.peft_format else alpha
        self._multiplier = 1.0
        self.torch_multiplier = None
        self.unet_loras: List[LoRAModule] = []
        self.text_encoder_loras: List[LoRAModule] = []
        ignore_if_contains = ignore_if_contains or []
        if self.is_transformer:
            prefix = "transformer" if self.peft_format else "lora_transformer"
        else:
            prefix = "unet" if self.peft_format else "lora_unet"  # lora_special.py:286-288, 463-469
        names = set()
        for name, module in unet.named_modules():
            if module.__class__.__name__ not in target_lin_modules:
                continue
            for child_name, child in module.named_modules():
                is_linear = child.__class__.__name__ in LINEAR_MODULES
                is_conv1x1 = bool(getattr(child, "is_conv1x1", False))  # Conv2d with kernel (1, 1): lora_special.py:487-488, 585-587
                is_conv3x3 = bool(getattr(child, "is_conv3x3", False)) and self.conv_lora_dim is not None
                if not (is_linear or is_conv1x1 or is_conv3x3):
                    continue
                clean = ".".join([x for x in (prefix, name, child_name) if x])
                lora_name = clean.replace(".", "$$") if self.peft_format else clean.replace(".", "_")
                if any(w in clean for w in ignore_if_contains):
                    continue
                if transformer_only and self.is_transformer:
                    blocks = transformer_block_names
                    if blocks is not None:
                        if not any(b in clean for b in blocks):
                            continue
                    elif "transformer_blocks" not in lora_name:
                        continue
                if only_if_contains is not None and not any(w in clean for w in only_if_contains) and not any(w in lora_name for w in only_if_contains):
                    continue
                if lora_name in names:
                    continue
                names.add(lora_name)
                if network_type.lower() == "lora":
                    module_kwargs = dict(dropout=dropout, rank_dropout=rank_dropout, module_dropout=module_dropout)
                dim, al = (self.conv_lora_dim, self.conv_alpha) if is_conv3x3 else (lora_dim, self.alpha)  # lora_special.py:585-590
                lora = module_class(lora_name, child, multiplier, dim, al, network=self, **module_kwargs)
                lora.is_conv1x1 = is_conv1x1  # saved / loaded as Conv2d weights [r, in, 1, 1] / [out, r, 1, 1] like the reference's
                self.unet_loras.append(lora)
        for lora in self.unet_loras:
            self.add_module(lora.lora_name, lora)
        self._arena_built = False
        self.multiplier = multiplier

    # ------------------------------------------------------------------ arena
    def get_all_modules(self):
        return self.text_encoder_loras + self.unet_loras

    def apply_to(self, *args, **kwargs):
        for lora in self.get_all_modules():
            lora.apply_to()

    # False: every adapter matrix becomes a NEW nn.Parameter viewing the arena (a network this class built itself).  True (adopt.AdoptedNetwork):
    # the existing Parameter objects are kept and only their storage is re-pointed (`.data = view`), because the reference's trainer already
    # holds them (optimizer param groups, EMA, accelerate) when the arena is built; Conv2d-shaped weights ([r, in, k, k] / [out, r, 1, 1]) view
    # the same [rank_pad, in*k*k] / [out, rank_pad] blocks.
    _repoint = False

    def _bind(self, holder, attr, view, gview):
        if not self._repoint:
            par = nn.Parameter(view, requires_grad=True)
            setattr(holder, attr, par)
            par.grad = gview
            return par
        par = getattr(holder, attr)
        if par.dim() == 4:
            if par.shape[2:] == (1, 1):
                view, gview = view[:, :, None, None], gview[:, :, None, None]
            else:
                view, gview = view.view(par.shape), gview.view(par.shape)
        par.data = view
        par.grad = gview
        return par

    def build_arena(self, device, ema: bool = False, groups=None, shadow_dtype=None):
        """Move every adapter matrix into flat fp32 arenas on `device` (reference: network.force_to(device, fp32),
        jobs/process/BaseSDTrainProcess.py:1982-1983) and create grad / Adam / EMA / bf16-shadow arenas.

        groups: lists of LoRAModules whose base layers read the SAME activation (q/k/v[/proj_mlp]); their lora_down
        matrices are laid out back to back so one skinny kernel launch produces T for the whole group and one wgrad
        launch produces all their lora_down gradients (the activation is streamed once instead of 3-4 times)."""
        mods = self.get_all_modules()
        # the skinny kernels handle up to 64 ranks per launch: larger groups (e.g. q,k,v at rank 32) stay ungrouped
        # Ranks are padded to a multiple of 16 inside the arenas (the MFMA K-slab and the skinny kernels work on 16-wide rank
        # blocks): lora_down lives in the first r rows of a [rank_pad, in] block, lora_up in the first r columns of an
        # [out, rank_pad] block.  The padding is zero at creation and stays exactly zero: its gradients are products with zero
        # rows / columns and AdamW maps (p, g, m, v) = 0 to 0.  The nn.Parameters are the logical [r, in] / [out, r] views.
        for m in mods:
            m.rank_pad = (1 << 30) if m.is_lokr else (m.lora_dim + 15) // 16 * 16  # LoKr: no rank slab (never grouped)

        def block_shape(m, which):
            if m.is_lokr and which == "down" and not m.use_w2:  # the pair a [out_k, r] | b [r, in_n], back to back
                return (1, m.out_k * m.lora_dim + m.lora_dim * m.in_n)
            w = (m.lora_down if which == "down" else m.lora_up).weight
            if m.is_lokr:
                return tuple(w.shape)
            return (m.rank_pad, w.shape[1:].numel()) if which == "down" else (w.shape[0], m.rank_pad)

        groups = [g for g in (groups or []) if sum(x.rank_pad for x in g) <= 64]
        n = sum(block_shape(m, "down")[0] * block_shape(m, "down")[1] + block_shape(m, "up")[0] * block_shape(m, "up")[1] for m in mods)
        n_mat = n  # [0, n_mat): the matrices (shadowed in bf16); [n_mat, n): DoRA magnitude vectors (fp32 only)
        n += sum(m.magnitude.numel() for m in mods if m.magnitude is not None)
        self.arena_p = torch.zeros(n, dtype=torch.float32, device=device)
        self.arena_g = torch.zeros(n, dtype=torch.float32, device=device)
        self.arena_m = torch.zeros(n, dtype=torch.float32, device=device)
        self.arena_v = torch.zeros(n, dtype=torch.float32, device=device)
        self.arena_ema = None
        dt = shadow_dtype or (torch.bfloat16 if torch.device(device).type == "cuda" else torch.float32)
        self.shadow_dtype = dt
        self.n_mat = n_mat
        group_of = {}
        for gi, grp in enumerate(groups or []):
            for m in grp:
                group_of[id(m)] = gi
        order = []  # (module, which)
        done_groups = set()
        for m in mods:
            gi = group_of.get(id(m))
            if gi is None:
                order += [(m, "down"), (m, "up")]
            elif gi not in done_groups:
                done_groups.add(gi)
                order += [(x, "down") for x in groups[gi]] + [(x, "up") for x in groups[gi]]
        # bf16 shadow arena in the layouts the kernels read (AitkShadowDesc, include/aitk_mi355.h).  The reference keeps the
        # adapter in fp32 (network_mixins.py:309, BaseSDTrainProcess.py:1982-1983); bf16 MFMA reaches that precision through the
        # split hi = bf16(w), lo = bf16(w - hi):
        #   lora_down A [rp, in]:  hi / lo [rp, in]   (P, P_lo of the forward lora_down; a same-input group's matrices adjacent)
        #                          [in, 3rp] = [A^T_hi | A^T_hi | A^T_lo]   (B2 of the data-gradient K-slab)
        #   lora_up   B [out, rp]: [out, 3rp] = [B_hi | B_hi | B_lo]          (B2 of the forward K-slab)
        #                          hi / lo transposed [rp, out]               (P, P_lo of the backward lora_down)
        #   LoKr factors: plain bf16, direct + transposed (the Kronecker kernel's operands).
        sizes = {"hi": 0, "lo": 0, "t3": 0, "u3": 0, "uth": 0, "utl": 0, "cs": 0}
        for m, which in order:
            rows, cols = block_shape(m, which)
            cnt = rows * cols
            if m.is_lokr:
                scnt = m.out_k * m.in_n if which == "down" else cnt  # the shadow of `down` is always the (composed) W2
                sizes["hi"] += scnt
                sizes["t3"] += scnt
            elif which == "down" and getattr(m, "is_conv3x3", False):
                sizes["cs"] += 2 * cnt  # [A_hi ; A_lo], tap-major
                sizes["t3"] += 3 * cnt  # rotated data-gradient filter [Cin, 9 * 3 rp]
            elif which == "down":
                sizes["hi"] += cnt
                sizes["lo"] += cnt
                sizes["t3"] += 3 * cnt
            else:
                sizes["u3"] += 3 * cnt
                sizes["uth"] += cnt
                sizes["utl"] += cnt
        cur, tot = {}, 0
        for k in ("hi", "lo", "t3", "u3", "uth", "utl", "cs"):
            cur[k] = tot
            tot += (sizes[k] + 63) // 64 * 64  # regions start 128-byte aligned
        self.arena_shadow = torch.zeros(max(tot, 1), dtype=dt, device=device)

        def take(region, cnt, shape):
            o = cur[region]
            cur[region] += cnt
            return o, self.arena_shadow[o:o + cnt].view(*shape)

        entries = []
        off = 0
        g_t3 = {}  # group index -> (shadow offset, [in, 3 R] view) of the group's data-gradient slab matrix
        for m, which in order:
            rows, cols = block_shape(m, which)
            cnt = rows * cols
            if m.is_lokr and which == "down" and not m.use_w2:
                r, na = m.lora_dim, m.out_k * m.lora_dim
                for name, o, shape in (("lokr_w2_a", off, (m.out_k, r)), ("lokr_w2_b", off + na, (r, m.in_n))):
                    view = self.arena_p[o:o + shape[0] * shape[1]].view(*shape)
                    view.copy_(getattr(m, name).data)
                    self._bind(m, name, view, self.arena_g[o:o + shape[0] * shape[1]].view(*shape))
                m.g_w2a, m.g_w2b = m.lokr_w2_a.grad, m.lokr_w2_b.grad
                scnt = m.out_k * m.in_n
                d0, sh = take("hi", scnt, (m.out_k, m.in_n))
                d1, shT = take("t3", scnt, (m.in_n, m.out_k))
                entries.append((off, m.out_k, m.in_n, 3, d0, d1, 0, r))
                # gradient of the composed factor: scratch outside the arenas (never seen by the optimizer)
                m.g_down = torch.zeros(m.out_k, m.in_n, dtype=torch.float32, device=device)
                m.off_down, m.sh_down, m.sh_downT, m.blk_down = off, sh, shT, (1, cnt)
                off += cnt
                continue
            lin = m.lora_down if which == "down" else m.lora_up
            w = lin.weight.data
            w = w.reshape(w.shape[0], -1)  # Conv2d-shaped adapter weights of an adopted network: the [r, in*k*k] / [out, r] matrix
            block = self.arena_p[off:off + cnt].view(rows, cols)
            view = block[: w.shape[0], : w.shape[1]]  # logical matrix: leading rows (down) / leading columns (up) of the block
            view.copy_(w)
            gblock = self.arena_g[off:off + cnt].view(rows, cols)
            self._bind(lin, "weight", view, gblock[: w.shape[0], : w.shape[1]])
            if m.is_lokr:
                d0, sh = take("hi", cnt, (rows, cols))
                d1, shT = take("t3", cnt, (cols, rows))
                entries.append((off, rows, cols, 0, d0, d1, 0))
                if which == "down":
                    m.off_down, m.g_down, m.sh_down, m.sh_downT, m.blk_down = off, gblock, sh, shT, (rows, cols)
                else:
                    m.off_up, m.g_up, m.sh_up, m.sh_upT, m.blk_up = off, gblock, sh, shT, (rows, cols)
            elif which == "down" and getattr(m, "is_conv3x3", False):
                d0, stack = take("cs", 2 * cnt, (2 * rows, cols))
                d1, dg = take("t3", 3 * cnt, (m.conv_cin, 27 * rows))
                entries.append((off, rows, cols, 4, d0, d1, 0, m.conv_cin))
                m.off_down, m.g_down, m.blk_down = off, gblock, (rows, cols)
                m.sh_down_stack, m.sh_down_dgrad, m._sh_down_off = stack, dg, (d0, d1)
            elif which == "down":
                d0, hi = take("hi", cnt, (rows, cols))
                d1, lo = take("lo", cnt, (rows, cols))
                gi = group_of.get(id(m))
                if gi is None:
                    d2, t3 = take("t3", 3 * cnt, (cols, 3 * rows))
                    entries.append((off, rows, cols, 1, d0, d1, d2))
                else:
                    # same-input group: the [in, 3 rp] data-gradient blocks of its adapters are column windows of ONE [in, 3 R] matrix, so the
                    # group's data gradient dx = [dy_q | dy_k | ...] [W_q^T | W_k^T | ...]^T + [dT_q | dT_k | ...] [that matrix]^T is one GEMM
                    grp_m = groups[gi]
                    R3 = 3 * sum(x.rank_pad for x in grp_m)
                    if m is grp_m[0]:
                        g_t3[gi] = take("t3", cols * R3, (cols, R3))
                    gb, gmat = g_t3[gi]
                    c3 = 3 * sum(x.rank_pad for x in grp_m[:[id(x) for x in grp_m].index(id(m))])
                    d2, t3 = gb + c3, gmat[:, c3:c3 + 3 * rows]
                    entries.append((off, rows, cols, 1, d0, d1, d2, R3))
                m.off_down, m.g_down, m.blk_down = off, gblock, (rows, cols)
                m.sh_down, m.sh_down_lo, m.sh_downT3, m._sh_down_off = hi, lo, t3, (d0, d1)
            else:
                d0, u3 = take("u3", 3 * cnt, (rows, 3 * cols))
                d1, th = take("uth", cnt, (cols, rows))
                d2, tl = take("utl", cnt, (cols, rows))
                entries.append((off, rows, cols, 2, d0, d1, d2))
                m.off_up, m.g_up, m.blk_up = off, gblock, (rows, cols)
                m.sh_up3, m.sh_upT, m.sh_upT_lo = u3, th, tl
            off += cnt
        assert off == n_mat
        for m in mods:
            if m.magnitude is None:
                continue
            cnt = m.magnitude.numel()
            view = self.arena_p[off:off + cnt]
            view.copy_(m.magnitude.data)
            m.g_mag = self.arena_g[off:off + cnt]
            self._bind(m, "magnitude", view, m.g_mag)
            m.off_mag = off
            m.c = torch.ones(cnt, dtype=torch.float32, device=device)
            w = m.org_module[0].weight.data
            # (a layer quantised to e4m3 has released this copy: refresh_dora takes ||W_j||^2 from the dequantised codes instead)
            m.w2 = w.float().pow(2).sum(1).to(device) if w.numel() else torch.zeros(cnt, dtype=torch.float32, device=device)
            off += cnt
        for m in mods:
            if hasattr(m, "alpha"):
                m.alpha = m.alpha.to(device)
            if getattr(m, "_runtime_scale", None) is not None:
                m._runtime_scale = m._runtime_scale.to(device)
            m.group = None
        self.groups = []
        for grp in groups or []:
            first = grp[0]
            rtot = sum(x.rank_pad for x in grp)
            cin = first.in_features
            assert all(x.in_features == cin and x.scale == first.scale for x in grp)
            o0 = first.off_down
            assert [x.off_down for x in grp] == [o0 + sum(y.rank_pad for y in grp[:i]) * cin for i in range(len(grp))]
            assert all(x.rank_pad == first.rank_pad for x in grp), "a same-input group shares one rank"
            h0, l0 = first._sh_down_off
            assert [x._sh_down_off[0] for x in grp] == [h0 + i * first.rank_pad * cin for i in range(len(grp))]
            g = {"mods": grp, "R": rtot, "rp": first.rank_pad, "sh_down": self.arena_shadow[h0:h0 + rtot * cin].view(rtot, cin),
                 "sh_down_lo": self.arena_shadow[l0:l0 + rtot * cin].view(rtot, cin),
                 "g_down": self.arena_g[o0:o0 + rtot * cin].view(rtot, cin), "scale": first.scale,
                 "sh_downT3": g_t3[groups.index(grp)][1],
                 "col": {id(x): sum(y.rank_pad for y in grp[:i]) for i, x in enumerate(grp)}}
            for x in grp:
                x.group = g
            self.groups.append(g)
        self._shadow_entries = entries
        self._shadow_table = None
        if ema:
            self.arena_ema = self.arena_p.clone()
        self._arena_built = True
        self._update_torch_multiplier()
        return self

    def arena_view(self, arena, m, which, padded=False):
        """The [rows, cols] matrix of module m inside a flat arena (arena_p / _g / _m / _v / _ema): the padded block, or the
        logical (unpadded) view of it."""
        if which in ("w2_a", "w2_b"):  # low-rank LoKr pair inside the `down` slot
            na = m.out_k * m.lora_dim
            if which == "w2_a":
                return arena[m.off_down:m.off_down + na].view(m.out_k, m.lora_dim)
            return arena[m.off_down + na:m.off_down + na + m.lora_dim * m.in_n].view(m.lora_dim, m.in_n)
        off, (rows, cols) = (m.off_down, m.blk_down) if which == "down" else (m.off_up, m.blk_up)
        block = arena[off:off + rows * cols].view(rows, cols)
        if padded:
            return block
        w = (m.lora_down if which == "down" else m.lora_up).weight
        return block[: w.shape[0], : w.shape[1:].numel()]

    @staticmethod
    def _shaped_like(par, view2d):
        """a 2-D arena view in the shape of the Parameter it backs (Conv2d-shaped adapter weights of an adopted network)"""
        if par.dim() != 4:
            return view2d
        return view2d[:, :, None, None] if par.shape[2:] == (1, 1) else view2d.view(par.shape)

    def refresh_shadows(self, ops):
        """bf16 shadows (split hi + lo, every layout the kernels read) of every adapter matrix; call after each optimizer step /
        weight load."""
        self._ops = ops
        if self._shadow_table is None:
            self._shadow_table = ops.make_shadow_table(self._shadow_entries, self.arena_p.device)
        # never part of an autograd graph (the explicit backward is the only history); under no_grad also because the shadow views may have
        # been created inside the trainer's no_grad prior prediction (first forward of a preservation run, SDTrainer.py:1244) — torch forbids
        # writing a base in grad mode whose views were made without it (torch-backed kernel table; the HIP table writes through raw pointers)
        with torch.no_grad():
            ops.refresh_shadows(self.arena_p, self.arena_shadow, self._shadow_table)
            self.refresh_dora(ops)

    def refresh_dora(self, ops):
        """c = magnitude / ||W + s*up@down||_row for every DoRA module, from the current adapter state:
        ||.||^2 = ||W||^2 + 2 s B.(W A^T) + s^2 B (A A^T) B^T — one skinny pass over each base weight (aitk_lora_down with the
        weight as the streamed operand), an r x r Gram matrix (aitk_lora_wgrad) and a row kernel.  The norm is detached in the
        reference (DoRA.py:139-147), so c is a constant of the step."""
        mods = [m for m in self.get_all_modules() if m.magnitude is not None]
        if not mods:
            return
        # per-sample multipliers: the reference scales the DoRA weight by multiplier.mean() (toolkit/network_mixins.py:333-336)
        mbar = self.multiplier_mean()
        self._dora_mbar = mbar
        for m in mods:
            lin = m.org_module[0]
            dev, r = self.arena_p.device, m.rank_pad
            wsrc = lin.weight.data
            q = getattr(lin, "qweight", None)
            if q is not None:
                # weight-only fp8 base: the reference's DoRAModule takes the norm over the DEQUANTISED weight (DoRA.py:105-109: get_orig_weight ->
                # weight.dequantize()); the bf16 copy of a quantised layer is released, so the e4m3 codes are expanded for the skinny pass (the same
                # expansion the base GEMM multiplies with), and ||W_j||^2 follows the codes it was taken from
                wsrc = torch.empty(q.shape[0], q.shape[1], dtype=self.shadow_dtype, device=dev)
                ops.dequant_fp8(q, lin.wscale, 1, wsrc)
                if getattr(m, "_w2_of", None) is not q:
                    m.w2 = wsrc.float().pow(2).sum(1)
                    m._w2_of = q
            tw = torch.empty(lin.out_features, r, dtype=self.shadow_dtype, device=dev)
            ops.lora_down(wsrc, m.sh_down, tw, scale=1.0, M=lin.out_features, p_lo=m.sh_down_lo)
            gram = torch.zeros(r, r, dtype=torch.float32, device=dev)
            at = m.sh_downT3[:, :r]  # A^T_hi [in, r] (row stride 3r)
            ops.lora_wgrad(at, at, gram, M=m.in_features)
            ops.dora_colscale(m.w2, tw, self.arena_view(self.arena_p, m, "up", padded=True), gram, m.magnitude.data, m.scale * mbar, m.c)

    def multiplier_mean(self):
        mult = self._multiplier
        vals = [float(x) for x in mult] if isinstance(mult, (list, tuple)) else [float(mult)]
        return sum(vals) / len(vals)

    def multiplier_is_per_sample(self):
        mult = self._multiplier
        vals = [float(x) for x in mult] if isinstance(mult, (list, tuple)) else [float(mult)]
        return max(vals) != min(vals)

    @property
    def has_dropout(self):
        return bool(self.dropout or self.rank_dropout or self.module_dropout)

    def dropout_is_capturable(self):
        """hipGraph replay draws fresh masks only if every draw is a device-side torch.rand on the default generator: no module_dropout (its coin is
        a HOST decision that changes the launch list) and the default mask provider."""
        if any(getattr(m, "module_dropout", None) for m in self.get_all_modules()) or self.module_dropout:
            return False
        return self.mask_provider is _default_mask_provider

    def dropout_plan(self, m, *, M, rows_per_batch, B):
        """Dropout decisions of one adapter for one forward, in the reference's order (toolkit/network_mixins.py:198-228; only in
        training mode): returns None (no dropout), "skip" (module_dropout fired: the adapter contributes nothing this call) or
        (tmask fp32 [rows, rank_pad], tmask_rows_per_batch) — the multiplier aitk_lora_down applies to lx (and to its gradient):
        neuron dropout keep / (1 - p) per element, rank dropout keep / (1 - p) per (sample, rank)."""
        if not (self.training and self.has_dropout):
            return None
        dev = self.arena_p.device
        if m.module_dropout and float(self.mask_provider(m.lora_name, "module", (1,), dev)) < m.module_dropout:
            return "skip"
        mask, rpb = None, 0
        if m.dropout:
            keep = (self.mask_provider(m.lora_name, "dropout", (M, m.lora_dim), dev) >= m.dropout).float() / (1.0 - m.dropout)
            mask = keep
        if m.rank_dropout and m.rank_dropout > 0:
            keep = (self.mask_provider(m.lora_name, "rank", (B, m.lora_dim), dev) > m.rank_dropout).float() / (1.0 - m.rank_dropout)
            if mask is None:
                mask, rpb = keep, rows_per_batch
            else:
                mask = mask * keep.repeat_interleave(rows_per_batch, 0)[:M]
        if mask is None:
            return None
        full = torch.zeros(mask.shape[0], m.rank_pad, dtype=torch.float32, device=dev)
        full[:, : m.lora_dim] = mask.to(dev)
        return full.contiguous(), rpb

    def attach_grad_views(self):
        """Every Parameter's .grad is a view of the flat gradient arena; optimizer.zero_grad(set_to_none=True) — what the reference's
        trainer calls after each step (SDTrainer.py:2288) — drops them, so they are re-attached before gradients are written."""
        for par, view, _ in self._grad_views():
            if par.grad is None:
                par.grad = view
        for m in self.get_all_modules():
            if m.is_lokr and not m.use_w2:
                m.g_w2a, m.g_w2b = m.lokr_w2_a.grad, m.lokr_w2_b.grad

    def _grad_views(self):
        """[(Parameter, its view of arena_g, the view's address)] — built once per gradient arena: the reference's trainer drops every .grad
        after every step (zero_grad(set_to_none=True)) and re-attaching 988 views must not cost 988 x (slice, view, slice) of host time on a
        GPU that the trainer's mid-step `isfinite(loss)` sync has just drained (tools/gpu_trainer_path.py: 10 ms of a 225-ms step at B = 1)."""
        cache = self.__dict__.get("_grad_view_cache")
        if cache is not None and cache[0] is self.arena_g:
            return cache[1]
        out = []
        for m in self.get_all_modules():
            if m.is_lokr and not m.use_w2:  # low-rank LoKr: the pair a | b shares the `down` slot
                for par, which in ((m.lokr_w2_a, "w2_a"), (m.lokr_w2_b, "w2_b"), (m.lokr_w1, "up")):
                    out.append((par, self.arena_view(self.arena_g, m, which)))
                continue
            out.append((m.lora_down.weight, self._shaped_like(m.lora_down.weight, self.arena_view(self.arena_g, m, "down"))))
            out.append((m.lora_up.weight, self._shaped_like(m.lora_up.weight, self.arena_view(self.arena_g, m, "up"))))
            if m.magnitude is not None:
                out.append((m.magnitude, m.g_mag))
        out = [(par, v, v.data_ptr()) for par, v in out]
        self.__dict__["_grad_view_cache"] = (self.arena_g, out)
        return out

    def grads_dropped(self):
        """True when optimizer.zero_grad(set_to_none=True) removed the .grad views (any adapter type: asks the first trainable
        Parameter, not lora_down — low-rank LoKr has no such matrix)."""
        for par in self.parameters():
            if par.requires_grad:
                return par.grad is None
        return False

    def zero_grad_arena(self):
        self.arena_g.zero_()
        self._arena_dirty = False
        self.attach_grad_views()

    def zero_grad(self, set_to_none: bool = True):
        """nn.Module.zero_grad of the adapter: the arena is zeroed in one memset; the .grad views stay attached."""
        if getattr(self, "_arena_built", False):
            self.zero_grad_arena()
        else:
            super().zero_grad(set_to_none)

    # ------------------------------------------------------------------ multiplier / activation (network_mixins.py:791-853)
    @property
    def multiplier(self):
        return self._multiplier

    @multiplier.setter
    def multiplier(self, value):
        if isinstance(value, torch.Tensor):
            value = value.detach().cpu().tolist()
        self._multiplier = value
        self._update_torch_multiplier()

    def _update_torch_multiplier(self):
        if not getattr(self, "_arena_built", False):
            return
        v = self._multiplier
        vals = [float(x) for x in v] if isinstance(v, (list, tuple)) else [float(v)]
        self.torch_multiplier = torch.tensor(vals, dtype=torch.float32, device=self.arena_p.device)

    def __enter__(self):
        self.is_active = True

    def __exit__(self, exc_type, exc_value, tb):
        self.is_active = False

    def force_to(self, device, dtype):
        assert dtype == torch.float32, "adapter weights are fp32 in the reference (BaseSDTrainProcess.py:1983)"
        if not self._arena_built:
            self.build_arena(device)
        return self

    def prepare_grad_etc(self, *args, **kwargs):
        self.requires_grad_(True)

    def enable_gradient_checkpointing(self):
        pass  # the fused step keeps activations resident in HBM (288 GB); nothing to recompute

    def disable_gradient_checkpointing(self):
        pass  # toolkit/network_mixins.py:885-888

    def prepare_optimizer_params(self, text_encoder_lr=None, unet_lr=None, default_lr=None):
        """One group with every adapter weight (toolkit/kohya_lora.py:1030-1074, unet branch)."""
        params = []
        for m in self.unet_loras:  # named_parameters() order of the reference modules
            if m.magnitude is not None:
                params.extend([m.magnitude, m.lora_up.weight, m.lora_down.weight])
            elif getattr(m, "is_lokr", False):
                params.extend(p for _, p in m.factor_params())
            else:
                params.extend([m.lora_down.weight, m.lora_up.weight])
        group = {"params": params}
        lr = unet_lr if unet_lr is not None else default_lr
        if lr is not None:
            group["lr"] = lr
        return [group]

    # ------------------------------------------------------------------ state-dict I/O (network_mixins.py:581-789)
    def _base_model(self):
        ref = getattr(self, "base_model_ref", None)
        return ref() if ref is not None else None

    def get_state_dict(self, extra_state_dict=None, dtype=torch.float16, use_ema=False):
        src = self.arena_ema if (use_ema and self.arena_ema is not None) else None
        sd = OrderedDict()
        if not self.peft_format:
            # kohya format = the network's own state_dict (toolkit/network_mixins.py:590-598): per module `.alpha` (buffer), then
            # `.lora_down.weight`, `.lora_up.weight`; 1x1-conv adapters keep their Conv2d shapes
            for m in self.get_all_modules():
                sd[f"{m.lora_name}.alpha"] = m.alpha.detach().clone().to("cpu").to(dtype)
                for key, lin, which in (("lora_down", m.lora_down, "down"), ("lora_up", m.lora_up, "up")):
                    w = lin.weight.detach() if src is None else self.arena_view(src, m, which)
                    w = w.clone().contiguous()
                    if getattr(m, "is_conv1x1", False):
                        w = w[:, :, None, None]
                    elif getattr(m, "is_conv3x3", False):  # Conv2d(in, r, 3) / Conv2d(r, out, 1) shapes (lora_special.py:95-104)
                        w = w.view(w.shape[0], m.conv_cin, 3, 3) if which == "down" else w[:, :, None, None]
                    sd[f"{m.lora_name}.{key}.weight"] = w.to("cpu").to(dtype)
            if extra_state_dict is not None:
                for k, v in extra_state_dict.items():
                    sd[k] = v.detach().clone().to("cpu").to(dtype)
            base_model = self._base_model()
            if base_model is not None:
                sd = base_model.convert_lora_weights_before_save(sd)
            return sd
        for m in self.get_all_modules():
            base = m.lora_name.replace("$$", ".")
            if getattr(m, "is_lokr", False):  # <name>.lokr_w1 / .lokr_w2 / .alpha — LoKr keeps alpha (network_mixins.py:613-616)
                # the reference's state_dict order: parameters (lokr_w1, lokr_w2 | lokr_w2_a, lokr_w2_b), then the alpha buffer
                where = {"lokr_w1": "up", "lokr_w2": "down", "lokr_w2_a": "w2_a", "lokr_w2_b": "w2_b"}
                for key, par in m.factor_params():
                    w = par.detach() if src is None else self.arena_view(src, m, where[key])
                    sd[f"{base}.{key}"] = w.clone().contiguous().to("cpu").to(dtype)
                sd[f"{base}.alpha"] = m.alpha.detach().clone().to("cpu").to(dtype)
                continue
            order = (("lora_A", m.lora_down, "down"), ("lora_B", m.lora_up, "up"))
            if m.magnitude is not None:
                # DoRAModule.state_dict() order: its own Parameter first, then the sub-modules in creation order (lora_up before lora_down,
                # toolkit/models/DoRA.py:84-98)
                w = m.magnitude.detach()
                if src is not None:
                    w = src[m.off_mag:m.off_mag + w.numel()]
                sd[f"{base}.magnitude"] = w.clone().to("cpu").to(dtype)
                order = order[::-1]
            for key, lin, which in order:
                w = lin.weight.detach()
                if src is not None:
                    w = self.arena_view(src, m, which)
                sd[f"{base}.{key}.weight"] = w.clone().contiguous().to("cpu").to(dtype)  # alpha dropped in PEFT format (607-624)
        if extra_state_dict is not None:
            for k, v in extra_state_dict.items():
                sd[k] = v.detach().clone().to("cpu").to(dtype)
        base_model = self._base_model()
        if base_model is not None:  # model-specific key names, e.g. Wan: diffusion_model.blocks.N.self_attn.q... (network_mixins.py:637-638)
            sd = base_model.convert_lora_weights_before_save(sd)
        return sd

    def save_weights(self, file, dtype=torch.float16, metadata=None, extra_state_dict=None, use_ema=False):
        from safetensors.torch import save_file

        sd = self.get_state_dict(extra_state_dict, dtype, use_ema)
        meta = OrderedDict()
        for k, v in (metadata or {}).items():  # every value JSON-stringified (toolkit/metadata.py:13-29)
            meta[k] = v if isinstance(v, str) else json.dumps(v)
        meta = add_model_hash_to_meta(sd, meta)  # toolkit/network_mixins.py:655-663
        base_model = self._base_model()
        if base_model is not None and hasattr(base_model, "save_lora"):  # network_mixins.py:657-660
            base_model.save_lora(sd, file, meta)
            return
        meta.setdefault("format", "pt")
        os.makedirs(os.path.dirname(os.path.abspath(file)), exist_ok=True)
        save_file(sd, file, meta)

    def load_weights(self, file):
        """Accepts the PEFT keys this class writes (after the base model's convert_lora_weights_before_load hook, like the
        reference: network_mixins.py:674-688); rank grow/shrink by zero-pad / truncate (737-775).  Returns the keys that matched
        no adapter (the reference's `extra_dict`), or None; a file that matches NO adapter at all raises, because training would
        otherwise silently continue from the fresh initialisation."""
        from safetensors.torch import load_file

        base_model = self._base_model()
        if isinstance(file, dict):
            sd = file
        elif base_model is not None and hasattr(base_model, "load_lora"):
            sd = base_model.load_lora(file)
        else:
            sd = load_file(file)
        if base_model is not None:
            sd = base_model.convert_lora_weights_before_load(sd)
        extra = OrderedDict()
        by_name = {m.lora_name.replace("$$", "."): m for m in self.get_all_modules()}
        n_hit = 0
        suffixes = ((".lora_A.weight", "lora_down"), (".lora_B.weight", "lora_up"), (".lokr_w1", "lokr_w1"), (".lokr_w2", "lokr_w2"),
                    (".lokr_w2_a", "lokr_w2_a"), (".lokr_w2_b", "lokr_w2_b"))
        if not self.peft_format:  # kohya keys
            suffixes = ((".lora_down.weight", "lora_down"), (".lora_up.weight", "lora_up"))
        with torch.no_grad():
            for k, v in sd.items():
                hit = False
                if v.dim() == 4 and v.shape[2:] == (1, 1):
                    v = v[:, :, 0, 0]  # 1x1-conv adapter weights
                elif v.dim() == 4:
                    v = v.reshape(v.shape[0], -1)  # 3x3-conv lora_down [r, in, 3, 3] -> [r, in*9] (the Conv2d weight's memory order)
                for which, attr in suffixes:
                    if k.endswith(which) and k[: -len(which)] in by_name:
                        mod = by_name[k[: -len(which)]]
                        if which.startswith(".lokr") != bool(getattr(mod, "is_lokr", False)):
                            continue
                        if which.startswith(".lokr"):
                            w = getattr(mod, attr, None)  # a full-rank file does not load into a low-rank module and vice versa
                            if w is None:
                                continue
                        else:
                            w = getattr(mod, attr).weight
                        v = v.to(w.device, torch.float32)
                        if v.shape != w.shape:
                            new = torch.zeros_like(w)
                            r0, r1 = min(v.shape[0], w.shape[0]), min(v.shape[1], w.shape[1])
                            new[:r0, :r1] = v[:r0, :r1]
                            v = new
                        w.copy_(v)
                        hit = True
                        n_hit += 1
                mk = k[: -len(".magnitude")] if k.endswith(".magnitude") else None
                if mk is not None and mk in by_name and by_name[mk].magnitude is not None:
                    mag = by_name[mk].magnitude
                    mag.copy_(v.to(mag.device, torch.float32))
                    hit = True
                    n_hit += 1
                if k.endswith(".alpha") and k[: -len(".alpha")] in by_name:
                    hit = True  # constant buffer (LoKr / kohya files carry it; the scale was fixed at construction like the reference's)
                    if not self.peft_format:
                        by_name[k[: -len(".alpha")]].alpha.copy_(v.to(by_name[k[: -len(".alpha")]].alpha))
                if not hit:
                    extra[k] = v
        if n_hit == 0 and len(sd) > 0:
            raise ValueError(f"load_weights: none of the {len(sd)} keys matches an adapter of this network "
                             f"(first key: {next(iter(sd))!r}); wrong key format or missing base_model conversion hook")
        if getattr(self, "_arena_built", False) and getattr(self, "_ops", None) is not None:
            self.refresh_shadows(self._ops)  # the kernels read the shadows (and DoRA's column scale), not the fp32 arena
        return extra if len(extra) else None

    def set_multiplier(self, multiplier):
        self.multiplier = multiplier

    # ------------------------------------------------------------------ merge (network_mixins.py:356-462, 894-906)
    @torch.no_grad()
    def merge_in(self, merge_weight=1.0, ops=None):
        """W <- W + merge_weight * scale * (lora_up @ lora_down) for every wrapped Linear (and its transposed copy),
        as a rank-r GEMM with the accumulate epilogue: C += (c*B) A  — the MFMA form of ToolkitModuleMixin.merge_in.  The product
        uses the split shadows (B_hi A_hi + B_hi A_lo + B_lo A_hi), i.e. the fp32 product of the reference to 2^-17."""
        if self.network_type.lower() == "dora":
            return  # toolkit/network_mixins.py:894-897
        ops = ops or self._ops
        if self.network_type.lower() == "lokr":  # toolkit/models/lokr.py:261-309: W += kron(w1, w2) * scale * merge_weight
            for m in self.get_all_modules():
                lin = m.org_module[0]
                a = float(merge_weight) * m.scale
                w2 = m.composed_w2()
                if getattr(lin, "qweight", None) is not None:
                    # weight-only fp8 base: dequantise, add kron(w1, w2) * scale * merge_weight, re-quantise with a fresh per-channel scale —
                    # what the reference does for any quantised org_module (toolkit/models/lokr.py:261-309 ends in the same
                    # load_state_dict + requantise as LoRA's merge, toolkit/network_mixins.py:452-459); the model stays quantised
                    from .graph import quantize_linear_fp8

                    w = (lin.qweight.view(torch.float8_e4m3fn).float() * lin.wscale[:, None]).to(self.shadow_dtype).contiguous()
                    ops.kron_merge(w, m.lokr_w1.data.contiguous(), w2.contiguous(), a)
                    quantize_linear_fp8(lin, w)
                    continue
                ops.kron_merge(lin.weight.data, m.lokr_w1.data.contiguous(), w2.contiguous(), a)
                if getattr(lin, "weight_t", None) is not None:
                    ops.kron_merge(lin.weight_t, m.lokr_w1.data.t().contiguous(), w2.t().contiguous(), a)
            self.is_merged_in = merge_weight > 0
            return
        self.refresh_shadows(ops)
        for m in self.get_all_modules():
            lin = m.org_module[0]
            if not m.can_merge_in:
                continue
            if not bool(m.lora_up.weight.any()) or not bool(m.lora_down.weight.any()):
                continue  # a zero delta merges to identity: skipped (matters on quantised bases; network_mixins.py:381-389)
            if getattr(m, "is_conv3x3", False):
                # 3x3-conv adapter (toolkit/network_mixins.py:424-433): W [out, in, 3, 3] += c * (up [out, r] @ down [r, in*9]).  A set-up time
                # operation on the diffusers-layout weight in fp32 (not part of the step), followed by the kernel-layout rebuild.
                delta = float(merge_weight) * m.scale * (m.lora_up.weight.data.float() @ m.lora_down.weight.data.float())
                lin.weight.data.copy_((lin.weight.data.float() + delta.view_as(lin.weight.data)).to(lin.weight.dtype))
                lin.prepare()
                continue
            rp = m.rank_pad
            bs = torch.empty_like(m.sh_up3)  # (c B) as [B_hi | B_hi | B_lo]
            ops.ew(3, m.sh_up3, bs, alpha=float(merge_weight) * m.scale)
            t3 = m.sh_downT3
            at = torch.cat((t3[:, :rp], t3[:, 2 * rp:], t3[:, rp:2 * rp]), dim=1).contiguous()  # [A^T_hi | A^T_lo | A^T_hi]
            if getattr(lin, "qweight", None) is not None:
                # weight-only fp8 base: dequantise, add the delta, re-quantise (toolkit/network_mixins.py:452-459) — the model
                # stays quantised across merge / reset cycles; the scale per output channel is recomputed from the merged row
                from .graph import quantize_linear_fp8

                w = (lin.qweight.view(torch.float8_e4m3fn).float() * lin.wscale[:, None]).to(bs.dtype)
                ops.gemm_nt(bs, at, w, flags=2)
                quantize_linear_fp8(lin, w)
                continue
            ops.gemm_nt(bs, at, lin.weight.data, flags=2)          # [out,in] += (cB)[out,3r] . A^T[in,3r]^T
            if getattr(lin, "weight_t", None) is not None:
                ops.gemm_nt(at, bs, lin.weight_t, flags=2)         # [in,out] += A^T[in,3r] . (cB)[out,3r]^T
        self.is_merged_in = merge_weight > 0

    @torch.no_grad()
    def merge_out(self, merge_weight=1.0, ops=None):
        if not self.is_merged_in:
            return  # toolkit/network_mixins.py:900-902: nothing was merged, nothing to subtract
        self.merge_in(-abs(merge_weight), ops=ops)
        self.is_merged_in = False

    def reset_weights(self):
        """The reference zeroes every `lora_up` and leaves lora_down as trained (toolkit/network_mixins.py:464-471, 890-892), e.g.
        after a merge-and-reset cycle; LoKr modules have no lora_up key and are left untouched by that loop."""
        with torch.no_grad():
            for m in self.get_all_modules():
                if getattr(m, "is_lokr", False):
                    continue
                nn.init.zeros_(m.lora_up.weight)
        if getattr(self, "_arena_built", False) and getattr(self, "_ops", None) is not None:
            self.refresh_shadows(self._ops)

    # ------------------------------------------------------------------ optimizer state checkpoint (BaseSDTrainProcess.py:701-714)
    def _opt_slices(self, arena):
        """Views of a flat arena (arena_m / arena_v / ...) for every trainable tensor, in prepare_optimizer_params() order —
        i.e. the reference modules' named_parameters() order: LoRA (lora_down, lora_up); DoRA (magnitude, lora_up, lora_down);
        LoKr (lokr_w1, lokr_w2)."""
        out = []
        for m in self.unet_loras:
            if m.magnitude is not None:
                out.append(arena[m.off_mag:m.off_mag + m.magnitude.numel()])
                out += [self.arena_view(arena, m, "up"), self.arena_view(arena, m, "down")]
            elif getattr(m, "is_lokr", False):  # lokr_w1, lokr_w2 | lokr_w2_a, lokr_w2_b
                out += [self.arena_view(arena, m, w) for w in (("up", "down") if m.use_w2 else ("up", "w2_a", "w2_b"))]
            else:
                out += [self.arena_view(arena, m, "down"), self.arena_view(arena, m, "up")]
        return out

    def _opt_ref_shapes(self):
        """Shapes of the reference's parameters in _opt_slices order: conv adapters are Conv2d weights there ([r, in, 1, 1] / [r, in, 3, 3]
        down, [out, r, 1, 1] up; toolkit/lora_special.py:95-104), so the exported Adam moments take those shapes."""
        out = []
        for m in self.unet_loras:
            if m.magnitude is not None:
                out += [tuple(m.magnitude.shape), tuple(m.lora_up.weight.shape), tuple(m.lora_down.weight.shape)]
            elif getattr(m, "is_lokr", False):
                out += [tuple(p.shape) for _, p in m.factor_params()]
            else:
                d, u = tuple(m.lora_down.weight.shape), tuple(m.lora_up.weight.shape)
                if getattr(m, "is_conv3x3", False):
                    d, u = (d[0], m.conv_cin, 3, 3), u + (1, 1)
                elif getattr(m, "is_conv1x1", False):
                    d, u = d + (1, 1), u + (1, 1)
                out += [d, u]
        return out

    def optimizer_state_dict(self, step, lr, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01):
        """The fused AdamW state exported in torch.optim.AdamW.state_dict() layout (params in prepare_optimizer_params
        order), so `optimizer.pt` written here can be loaded by the reference's torch optimizer and vice versa."""
        state = {}
        shapes = self._opt_ref_shapes()
        for i, (mv, vv) in enumerate(zip(self._opt_slices(self.arena_m), self._opt_slices(self.arena_v))):
            state[i] = {"step": torch.tensor(float(step)), "exp_avg": mv.clone().contiguous().cpu().reshape(shapes[i]),
                        "exp_avg_sq": vv.clone().contiguous().cpu().reshape(shapes[i])}
        group = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(state)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        """Inverse of optimizer_state_dict; returns the step count."""
        step = 0
        ms, vs = self._opt_slices(self.arena_m), self._opt_slices(self.arena_v)
        if len(sd["state"]) != len(ms):
            raise ValueError(f"optimizer state has {len(sd['state'])} tensors, the network has {len(ms)} trainable tensors")
        for i, (mv, vv) in enumerate(zip(ms, vs)):
            st = sd["state"][i]
            if st["exp_avg"].numel() != mv.numel() or tuple(st["exp_avg"].shape[:1]) != tuple(mv.shape[:1]):
                raise ValueError(f"optimizer state {i}: shape {tuple(st['exp_avg'].shape)} does not match parameter {tuple(mv.shape)}")
            mv.copy_(st["exp_avg"].reshape(mv.shape))  # conv adapters arrive as 4-D Conv2d-shaped moments
            vv.copy_(st["exp_avg_sq"].reshape(vv.shape))
            step = int(float(st["step"]))
        return step
