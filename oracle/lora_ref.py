"""ORACLE (test infrastructure): restatement of the reference LoRA layer on plain nn.Linear modules.

Follows toolkit/lora_special.py:46-135 (construction / init) and toolkit/network_mixins.py:197-239, 274-348 (forward):
    org = org_forward(x)
    lx  = lora_up(lora_down(x.to(fp32))) * scale
    out = org + (lx * multiplier[b]).to(org.dtype)
Pinned against the reference classes themselves by tests/golden/make_golden.py.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn


class RefLoRAModule(nn.Module):
    def __init__(self, lora_name, org_module, lora_dim, alpha, network):
        super().__init__()
        self.lora_name = lora_name
        self.lora_dim = lora_dim
        if isinstance(org_module, nn.Conv2d) or org_module.__class__.__name__ == "Conv2d":  # toolkit/lora_special.py:80-104 goes by class name (UNet: 1x1 proj_in / proj_out of SD1.5, 3x3 with network.conv)
            self.lora_down = nn.Conv2d(org_module.in_channels, lora_dim, org_module.kernel_size, org_module.stride, org_module.padding, bias=False)
            self.lora_up = nn.Conv2d(lora_dim, org_module.out_channels, (1, 1), (1, 1), bias=False)
        else:
            self.lora_down = nn.Linear(org_module.in_features, lora_dim, bias=False)
            self.lora_up = nn.Linear(lora_dim, org_module.out_features, bias=False)
        alpha = lora_dim if alpha is None or alpha == 0 else alpha
        self.scale = float(alpha) / lora_dim
        self.register_buffer("alpha", torch.tensor(alpha))
        nn.init.kaiming_uniform_(self.lora_down.weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_up.weight)
        self.org_module = [org_module]
        self.network = [network]

    def apply_to(self):
        self.org_forward = self.org_module[0].forward
        self.org_module[0].forward = self.forward

    def forward(self, x, *args, **kwargs):
        net = self.network[0]
        if not net.is_active or net.multiplier_is_zero():
            return self.org_forward(x, *args, **kwargs)
        org = self.org_forward(x, *args, **kwargs)
        # dropout variants of ToolkitModuleMixin._call_forward (toolkit/network_mixins.py:198-228), training mode only; the uniform
        # draws come from network.mask_provider(lora_name, kind, shape, device) (default torch.rand, like the reference)
        prov = getattr(net, "mask_provider", None) if (self.training and getattr(net, "dropout_cfg", None)) else None
        cfg = getattr(net, "dropout_cfg", None) or {}
        if prov is not None and cfg.get("module_dropout") and float(prov(self.lora_name, "module", (1,), x.device)) < cfg["module_dropout"]:
            return org
        lx = self.lora_down(x.to(self.lora_down.weight.dtype))
        scale = self.scale
        if prov is not None and cfg.get("dropout"):
            # F.dropout draws one number per element; the keyed provider hands them out as [positions, rank] with positions in (b, y, x)
            # order for a Conv2d's [B, r, H, W] activation
            chan_last = lx.permute(0, 2, 3, 1) if lx.dim() == 4 else lx
            flat = chan_last.reshape(-1, self.lora_dim)
            keep = (prov(self.lora_name, "dropout", tuple(flat.shape), x.device) >= cfg["dropout"]).to(lx.dtype) / (1.0 - cfg["dropout"])
            flat = (flat * keep).reshape(chan_last.shape)
            lx = flat.permute(0, 3, 1, 2) if lx.dim() == 4 else flat
        if prov is not None and cfg.get("rank_dropout"):
            mask = (prov(self.lora_name, "rank", (lx.size(0), self.lora_dim), x.device) > cfg["rank_dropout"]).to(lx.dtype)
            if lx.dim() == 3:
                mask = mask.unsqueeze(1)
            elif lx.dim() == 4:  # Conv2d (toolkit/network_mixins.py:223-224)
                mask = mask.unsqueeze(-1).unsqueeze(-1)
            lx = lx * mask
            scale = scale * (1.0 / (1.0 - cfg["rank_dropout"]))
        lx = self.lora_up(lx) * scale
        m = net.torch_multiplier
        if lx.size(0) != m.size(0):
            m = m.repeat_interleave(lx.size(0) // m.size(0))
        lx = lx * m.view(-1, *([1] * (lx.dim() - 1)))
        return org + lx.to(org.dtype)


class RefDoRAModule(nn.Module):
    """toolkit/models/DoRA.py:36-148 + the DoRA branch of toolkit/network_mixins.py:323-339:
        out = org(x) + scale * m_b * up(down(x32)) + (magnitude / ||W + s*up@down||_row - 1) * F.linear(x32, W + s*up@down)
    with s = multiplier.mean(), the row norm detached (DoRA paper §4.3), lora_down ~ N(0, 1/r), lora_up = 0,
    magnitude initialised to the row norm of the base weight.  Parameter registration order (magnitude is created last but
    nn.Module lists Parameters before sub-modules): magnitude, lora_up.weight, lora_down.weight."""

    def __init__(self, lora_name, org_module, lora_dim, alpha, network):
        super().__init__()
        self.lora_name = lora_name
        self.lora_dim = lora_dim
        alpha = lora_dim if alpha is None or alpha == 0 else alpha
        self.scale = float(alpha) / lora_dim
        self.lora_up = nn.Linear(lora_dim, org_module.out_features, bias=False)
        self.lora_up.weight.data = torch.zeros_like(self.lora_up.weight.data)
        self.lora_down = nn.Linear(org_module.in_features, lora_dim, bias=False)
        self.lora_down.weight.data = torch.randn_like(self.lora_down.weight.data) * (1 / torch.sqrt(torch.tensor(lora_dim).float()))
        self.org_module = [org_module]
        self.network = [network]
        w = org_module.weight.data.detach().float().to(self.lora_up.weight.device)
        self.magnitude = nn.Parameter(torch.linalg.norm(w + self.lora_up.weight @ self.lora_down.weight, dim=1).detach().clone())

    def apply_to(self):
        self.org_forward = self.org_module[0].forward
        self.org_module[0].forward = self.forward

    def forward(self, x, *args, **kwargs):
        net = self.network[0]
        if not net.is_active or net.multiplier_is_zero():
            return self.org_forward(x, *args, **kwargs)
        org = self.org_forward(x, *args, **kwargs)
        x32 = x.to(self.lora_down.weight.dtype)
        lx = self.lora_up(self.lora_down(x32)) * self.scale
        m = net.torch_multiplier
        if lx.size(0) != m.size(0):
            m = m.repeat_interleave(lx.size(0) // m.size(0))
        lx = (lx * m.view(-1, *([1] * (lx.dim() - 1)))).to(org.dtype)
        slw = (self.lora_up.weight @ self.lora_down.weight) * m.mean()
        w = self.org_module[0].weight.data.detach().to(slw.dtype)
        norm = torch.linalg.norm(w + slw, dim=1).detach()
        dora = (self.magnitude / norm - 1).view(1, -1) * torch.nn.functional.linear(x32, w + slw)
        return org + lx + dora.to(org.dtype)


def factorization(dimension, factor=-1):
    """ORACLE restatement of the LyCORIS split used at toolkit/models/lokr.py:22-59, written over the explicit divisor list: among the
    divisor pairs (a, dim/a), a ascending from 1, advance while a < dim/a, the pair sum does not grow and a stays <= factor."""
    if factor > 0 and dimension % factor == 0:
        return factor, dimension // factor
    cap = dimension if factor == -1 else factor
    divisors = [a for a in range(1, dimension + 1) if dimension % a == 0]
    pick = 0
    while divisors[pick] < dimension // divisors[pick] and pick + 1 < len(divisors):
        nxt = divisors[pick + 1]
        if nxt + dimension // nxt > divisors[pick] + dimension // divisors[pick] or nxt > cap:
            break
        pick += 1
    a, b = divisors[pick], dimension // divisors[pick]
    return (a, b) if a <= b else (b, a)


class RefLokrModule(nn.Module):
    """toolkit/models/lokr.py:76-242 (Linear) + the factorised forward 331-399:
        X = x.unflatten(-1, (in_m, in_n)); tmp = einsum('...qs,os->...qo', X, w2); delta = einsum('...qo,pq->...po', tmp, w1*scale)
        out = org(x) + delta.flatten(-2) * mean(multiplier), computed in the base output's dtype.
    lora_dim >= max(out_k, in_n) / 2: full lokr_w2 (= 0 at init), alpha forced to lora_dim (scale 1).  Otherwise (184-197) the
    low-rank pair lokr_w2_a [out_k, r] (kaiming-uniform, drawn BEFORE w1) @ lokr_w2_b [r, in_n] (= 0), scale = alpha / r, and the
    forward folds the rank through first: tmp = (X w2_b^T) w2_a^T."""

    def __init__(self, lora_name, org_module, lora_dim, alpha, network, factor=-1):
        super().__init__()
        self.lora_name = lora_name
        self.lora_dim = lora_dim
        self.in_m, self.in_n = factorization(org_module.in_features, factor)
        self.out_l, self.out_k = factorization(org_module.out_features, factor)
        self.use_w2 = lora_dim >= max(self.out_k, self.in_n) / 2
        self.lokr_w1 = nn.Parameter(torch.empty(self.out_l, self.in_m))
        if self.use_w2:
            self.lokr_w2 = nn.Parameter(torch.empty(self.out_k, self.in_n))
            alpha = lora_dim  # both factors full: scale 1 (lokr.py:203-206)
            nn.init.constant_(self.lokr_w2, 0)
        else:
            self.lokr_w2_a = nn.Parameter(torch.empty(self.out_k, lora_dim))
            self.lokr_w2_b = nn.Parameter(torch.empty(lora_dim, self.in_n))
            nn.init.kaiming_uniform_(self.lokr_w2_a, a=math.sqrt(5))
            nn.init.constant_(self.lokr_w2_b, 0)
        self.scale = float(alpha) / lora_dim
        self.register_buffer("alpha", torch.tensor(alpha))
        nn.init.kaiming_uniform_(self.lokr_w1, a=math.sqrt(5))
        self.org_module = [org_module]
        self.network = [network]

    def apply_to(self):
        self.org_forward = self.org_module[0].forward
        self.org_module[0].forward = self.forward

    def forward(self, x, *args, **kwargs):
        net = self.network[0]
        if not net.is_active or net.multiplier_is_zero():
            return self.org_forward(x, *args, **kwargs)
        org = self.org_forward(x, *args, **kwargs)
        dt = org.dtype
        X = x.to(dt).unflatten(-1, (self.in_m, self.in_n))
        if self.use_w2:
            tmp = torch.einsum("...qs,os->...qo", X, self.lokr_w2.to(dt))
        else:
            tmp = torch.einsum("...qs,rs->...qr", X, self.lokr_w2_b.to(dt))
            tmp = torch.einsum("...qr,or->...qo", tmp, self.lokr_w2_a.to(dt))
        delta = torch.einsum("...qo,pq->...po", tmp, self.lokr_w1.to(dt) * self.scale).flatten(-2, -1)
        return (org + delta * net.torch_multiplier.mean().to(dt)).to(x.dtype)


class RefLoRANetwork(nn.Module):
    """PEFT-format transformer network: module discovery + naming of toolkit/lora_special.py:457-647 (flux branch)."""

    def __init__(self, unet, lora_dim, multiplier=1.0, target=("FluxTransformer2DModel",), block_names=("transformer_blocks",),
                 network_type="lora", lokr_factor=-1, kohya_unet=False, alpha=None, conv_lora_dim=None, conv_alpha=None):
        """kohya_unet: the UNet branch of toolkit/lora_special.py:457-647 — prefix lora_unet, dots -> underscores, Linear and 1x1
        Conv2d children of every `target` module (Transformer2DModel), no block filter, alpha as configured (scale = alpha / rank).
        conv_lora_dim (network.conv; lora_special.py:381-382, 585-590, toolkit/kohya_lora.py:750-751): the ResnetBlock2D / Downsample2D /
        Upsample2D modules are targets too, and 3x3 Conv2d children are wrapped at conv_lora_dim / conv_alpha."""
        super().__init__()
        self.is_active = False
        self.torch_multiplier = torch.tensor([float(multiplier)])
        self.unet_loras = []
        if kohya_unet and conv_lora_dim:
            target = tuple(target) + ("ResnetBlock2D", "Downsample2D", "Upsample2D")
        for name, module in unet.named_modules():
            if module.__class__.__name__ not in target:
                continue
            for child_name, child in module.named_modules():
                if kohya_unet:
                    is_conv = child.__class__.__name__ == "Conv2d"
                    if not (child.__class__.__name__ == "Linear" or (is_conv and (child.kernel_size == (1, 1) or conv_lora_dim))):
                        continue
                    lora_name = ".".join([x for x in ("lora_unet", name, child_name) if x]).replace(".", "_")
                    dim, al = lora_dim, (lora_dim if alpha is None else alpha)
                    if is_conv and child.kernel_size != (1, 1):
                        dim, al = conv_lora_dim, (conv_lora_dim if conv_alpha is None else conv_alpha)
                    self.unet_loras.append(RefLoRAModule(lora_name, child, dim, al, self))
                    continue
                if child.__class__.__name__ != "Linear":
                    continue
                clean = ".".join([x for x in ("transformer", name, child_name) if x])
                lora_name = clean.replace(".", "$$")
                if not any(b in clean for b in block_names):
                    continue
                if network_type == "lokr":
                    # PEFT-format transformer networks carry no alpha: alpha = rank (lora_special.py:428-433)
                    self.unet_loras.append(RefLokrModule(lora_name, child, lora_dim, lora_dim, self, factor=lokr_factor))
                    continue
                cls = RefDoRAModule if network_type == "dora" else RefLoRAModule
                self.unet_loras.append(cls(lora_name, child, lora_dim, lora_dim, self))
        for lo in self.unet_loras:
            self.add_module(lo.lora_name, lo)

    def multiplier_is_zero(self):
        return bool((self.torch_multiplier == 0).all())

    def apply_to(self, text_encoder=None, unet=None, apply_text_encoder=True, apply_unet=True):
        """toolkit/kohya_lora.py:952-965 (every module's apply_to swaps its layer's forward)"""
        for lo in self.unet_loras:
            lo.apply_to()

    # ---- the rest of the surface the reference's trainer touches (toolkit/network_mixins.py:791-883, toolkit/kohya_lora.py:1030-1074),
    # restated so that tests can walk the trainer's sequence (BaseSDTrainProcess.py:1949-2039) without importing the reference
    is_merged_in = False
    is_lorm = False
    text_encoder_loras = ()

    @property
    def multiplier(self):
        m = self.torch_multiplier.tolist()
        return m[0] if len(m) == 1 else m

    @multiplier.setter
    def multiplier(self, value):
        vals = [float(v) for v in value] if isinstance(value, (list, tuple)) else (value.flatten().tolist() if isinstance(value, torch.Tensor) else [float(value)])
        self.torch_multiplier = torch.tensor(vals, dtype=torch.float32, device=self.torch_multiplier.device)

    def _update_torch_multiplier(self):
        dev = next(self.unet_loras[0].parameters()).device
        self.torch_multiplier = self.torch_multiplier.to(dev, torch.float32)

    def get_all_modules(self):
        return list(self.unet_loras)

    def force_to(self, device, dtype):
        self.to(device, dtype)
        for lo in self.unet_loras:
            lo.to(device, dtype)

    def prepare_grad_etc(self, *a, **k):
        self.requires_grad_(True)

    def prepare_optimizer_params(self, text_encoder_lr=None, unet_lr=None, default_lr=None):
        params = []
        for lo in self.unet_loras:
            params.extend(lo.parameters())
        group = {"params": params}
        if unet_lr is not None:
            group["lr"] = unet_lr
        return [group]

    def __enter__(self):
        self.is_active = True

    def __exit__(self, *a):
        self.is_active = False

    def peft_state_dict(self, dtype=torch.float16):
        """PEFT renaming of toolkit/network_mixins.py:607-624."""
        sd = OrderedDict()
        lokr = any(isinstance(m, RefLokrModule) for m in self.unet_loras)
        for k, v in self.state_dict().items():
            if k.endswith(".alpha") and not lokr:  # LoKr files keep alpha (network_mixins.py:613-616)
                continue
            k = k.replace("lora_down", "lora_A").replace("lora_up", "lora_B").replace("$$", ".")
            sd[k] = v.detach().clone().to("cpu").to(dtype)
        return sd
