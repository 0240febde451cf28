"""AutoencoderKL encoder on the gfx950 kernels (host graph) + the reference's on-disk latent cache format.

Replaces `StableDiffusion.encode_images` (toolkit/stable_diffusion_model.py:2533-2575): `vae.encode(images)
.latent_dist.sample()` then `scaling_factor * (latents - shift_factor)`.  Module / parameter names are diffusers'
(`encoder.down_blocks.N.resnets.M.conv1.weight` ...) so a diffusers VAE checkpoint loads by key; `prepare()` re-lays every
3x3 weight out as [Cout, 9*Cin] (tap-major, channels-last) for the implicit-GEMM convolution and pads conv_in to 8 input
channels.  Activations are NHWC bf16 [B*H*W, C]; convolutions are `conv3x3` (gemm_nt_kernel<CONV>, MFMA), norms are the
two-pass GroupNorm(+SiLU) kernel, the single-head mid-block attention materialises its scores with the GEMM.

Latent cache (toolkit/dataloader_mixins.py:1779-1842, 2019-2081): `<dir>/_latent_cache/<stem>_<b64(md5(json(info)))>.safetensors`
holding tensor `latent` [C,h,w] in the model dtype.
"""
import base64
import hashlib
import json
import os
from collections import OrderedDict

import torch
import torch.nn as nn


class _Conv(nn.Module):
    def __init__(self, cin, cout, k, dtype, device):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k, dtype=dtype, device=device), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout, dtype=dtype, device=device), requires_grad=False)
        self.wk = None  # kernel layout, built by prepare()


class _Norm(nn.Module):
    def __init__(self, c, dtype, device):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c, dtype=dtype, device=device), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(c, dtype=dtype, device=device), requires_grad=False)


class _Lin(nn.Module):
    def __init__(self, cin, cout, dtype, device):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, dtype=dtype, device=device), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout, dtype=dtype, device=device), requires_grad=False)


class _Resnet(nn.Module):
    def __init__(self, cin, cout, dtype, device):
        super().__init__()
        self.norm1 = _Norm(cin, dtype, device)
        self.conv1 = _Conv(cin, cout, 3, dtype, device)
        self.norm2 = _Norm(cout, dtype, device)
        self.conv2 = _Conv(cout, cout, 3, dtype, device)
        self.conv_shortcut = _Conv(cin, cout, 1, dtype, device) if cin != cout else None


class _Holder(nn.Module):
    pass


class AutoencoderKLEncoder(nn.Module):
    def _apply(self, fn, recurse=True):
        """`.to()` / `.cpu()` / `.float()` are no-ops: the reference's trainer parks the VAE on the CPU and re-casts it (jobs/process/
        BaseSDTrainProcess.py:1902); the native encoder stays on its device in its dtype (kernel-layout buffers are not nn.Parameters)."""
        return self

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return self.dt

    def __init__(self, latent_channels=16, block_out_channels=(128, 256, 512, 512), layers_per_block=2, groups=32,
                 scaling_factor=0.3611, shift_factor=0.1159, use_quant_conv=False, dtype=torch.bfloat16, device=None, ops=None):
        """Defaults = FLUX.1's VAE.  SD1.5 / SDXL: latent_channels=4, scaling_factor=0.18215 / 0.13025, shift_factor=0.0,
        use_quant_conv=True (the 1x1 `quant_conv` on the moments that those AutoencoderKL configs carry)."""
        super().__init__()
        self.ops, self.dt, self.groups = ops, dtype, groups
        self.scaling_factor, self.shift_factor, self.latent_channels = scaling_factor, shift_factor, latent_channels
        enc = _Holder()
        enc.conv_in = _Conv(3, block_out_channels[0], 3, dtype, device)
        blocks = []
        c = block_out_channels[0]
        for i, co in enumerate(block_out_channels):
            blk = _Holder()
            blk.resnets = nn.ModuleList([_Resnet(c if j == 0 else co, co, dtype, device) for j in range(layers_per_block)])
            if i != len(block_out_channels) - 1:
                ds = _Holder()
                ds.conv = _Conv(co, co, 3, dtype, device)
                blk.downsamplers = nn.ModuleList([ds])
            else:
                blk.downsamplers = None
            blocks.append(blk)
            c = co
        enc.down_blocks = nn.ModuleList(blocks)
        mid = _Holder()
        att = _Holder()
        att.group_norm = _Norm(c, dtype, device)
        att.to_q, att.to_k, att.to_v = _Lin(c, c, dtype, device), _Lin(c, c, dtype, device), _Lin(c, c, dtype, device)
        att.to_out = nn.ModuleList([_Lin(c, c, dtype, device), nn.Identity()])
        mid.attentions = nn.ModuleList([att])
        mid.resnets = nn.ModuleList([_Resnet(c, c, dtype, device), _Resnet(c, c, dtype, device)])
        enc.mid_block = mid
        enc.conv_norm_out = _Norm(c, dtype, device)
        enc.conv_out = _Conv(c, 2 * latent_channels, 3, dtype, device)
        self.encoder = enc
        self.quant_conv = _Conv(2 * latent_channels, 2 * latent_channels, 1, dtype, device) if use_quant_conv else None
        self._prepared = False

    def prepare(self):
        for m in self.modules():
            if isinstance(m, _Conv) and m.weight.shape[-1] == 3:
                w = m.weight.data
                cout, cin = w.shape[0], w.shape[1]
                cpad = (cin + 7) // 8 * 8
                wk = torch.zeros(cout, 3, 3, cpad, dtype=w.dtype, device=w.device)
                wk[..., :cin] = w.permute(0, 2, 3, 1)
                m.wk = wk.reshape(cout, 9 * cpad).contiguous()
            elif isinstance(m, _Conv):
                m.wk = m.weight.data.reshape(m.weight.shape[0], -1).contiguous()
        self._prepared = True
        return self

    def _new(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.dt, device=self.encoder.conv_in.weight.device)

    def _resnet(self, r, x, B, H, W):
        ops = self.ops
        M, cout = B * H * W, r.conv1.weight.shape[0]
        h = self._new(M, x.shape[1])
        ops.groupnorm(x, r.norm1.weight, r.norm1.bias, h, B=B, HW=H * W, G=self.groups, silu=True)
        h1 = self._new(M, cout)
        ops.conv3x3(h, r.conv1.wk, h1, B=B, H=H, W=W, bias=r.conv1.bias)
        ops.groupnorm(h1, r.norm2.weight, r.norm2.bias, h1, B=B, HW=H * W, G=self.groups, silu=True)
        res = x
        if r.conv_shortcut is not None:
            res = self._new(M, cout)
            ops.gemm_nt(x, r.conv_shortcut.wk, res, bias=r.conv_shortcut.bias)
        out = self._new(M, cout)
        ops.conv3x3(h1, r.conv2.wk, out, B=B, H=H, W=W, bias=r.conv2.bias, flags=ops.EPI_ADD_AUX, aux_in=res)
        return out

    def _attention(self, a, x, B, HW):
        ops = self.ops
        Cc = x.shape[1]
        gn = self._new(B * HW, Cc)
        ops.groupnorm(x, a.group_norm.weight, a.group_norm.bias, gn, B=B, HW=HW, G=self.groups, silu=False)
        q, k = self._new(B * HW, Cc), self._new(B * HW, Cc)
        ops.gemm_nt(gn, a.to_q.weight, q, bias=a.to_q.bias)
        ops.gemm_nt(gn, a.to_k.weight, k, bias=a.to_k.bias)
        o = self._new(B * HW, Cc)
        scores = self._new(HW, HW)
        vT = self._new(Cc, HW)
        for b in range(B):
            sl = slice(b * HW, (b + 1) * HW)
            ops.gemm_nt(a.to_v.weight, gn[sl], vT, bias=a.to_v.bias, flags=ops.EPI_BIAS_ROW)  # V^T = W_v x^T + b_v (per row)
            ops.gemm_nt(q[sl], k[sl], scores)
            ops.softmax_rows(scores, Cc ** -0.5)
            ops.gemm_nt(scores, vT, o[sl])
        out = self._new(B * HW, Cc)
        ops.gemm_nt(o, a.to_out[0].weight, out, bias=a.to_out[0].bias, flags=ops.EPI_ADD_AUX, aux_in=x)
        return out

    def moments(self, images):
        """images [B,3,H,W] fp32 in [-1,1] -> NHWC moments [B*h*w, 2*latent_channels] (mean | logvar), (h, w)."""
        if not self._prepared:
            self.prepare()
        ops, enc = self.ops, self.encoder
        B, _, H, W = images.shape
        x8 = self._new(B * H * W, 8)
        ops.image_to_nhwc8(images.float().contiguous(), x8)
        x = self._new(B * H * W, enc.conv_in.weight.shape[0])
        ops.conv3x3(x8, enc.conv_in.wk, x, B=B, H=H, W=W, bias=enc.conv_in.bias)
        for blk in enc.down_blocks:
            for r in blk.resnets:
                x = self._resnet(r, x, B, H, W)
            if blk.downsamplers is not None:
                Ho, Wo = H // 2, W // 2
                y = self._new(B * Ho * Wo, x.shape[1])
                ds = blk.downsamplers[0].conv
                ops.conv3x3(x, ds.wk, y, B=B, H=H, W=W, stride=2, pad_t=0, pad_l=0, Ho=Ho, Wo=Wo, bias=ds.bias)  # F.pad (0,1,0,1)
                x, H, W = y, Ho, Wo
        x = self._resnet(enc.mid_block.resnets[0], x, B, H, W)
        x = self._attention(enc.mid_block.attentions[0], x, B, H * W)
        x = self._resnet(enc.mid_block.resnets[1], x, B, H, W)
        ops.groupnorm(x, enc.conv_norm_out.weight, enc.conv_norm_out.bias, x, B=B, HW=H * W, G=self.groups, silu=True)
        mom = self._new(B * H * W, enc.conv_out.weight.shape[0])
        ops.conv3x3(x, enc.conv_out.wk, mom, B=B, H=H, W=W, bias=enc.conv_out.bias)
        if self.quant_conv is not None:  # AutoencoderKL.encode: moments = quant_conv(encoder(x))
            mq = self._new(B * H * W, mom.shape[1])
            ops.gemm_nt(mom, self.quant_conv.wk, mq, bias=self.quant_conv.bias)
            mom = mq
        return mom, (H, W)

    @torch.no_grad()
    def encode_images(self, images, eps=None, generator=None):
        """-> scaled latents [B, latent_channels, H/8, W/8] (model dtype), as StableDiffusion.encode_images returns."""
        mom, (h, w) = self.moments(images)
        B, L = images.shape[0], self.latent_channels
        if eps is None:
            eps = torch.randn(B, L, h, w, device=mom.device, dtype=torch.float32, generator=generator)
        out = self._new(B, L, h, w)
        self.ops.latent_sample(mom, eps.float().contiguous(), out, scale=self.scaling_factor, shift=self.shift_factor)
        return out


# ---------------------------------------------------------------------------------------------------------- latent cache
def latent_cache_path(image_path, info: OrderedDict):
    """toolkit/dataloader_mixins.py:1827-1842: md5 of the JSON info dict, urlsafe-base64 without padding."""
    img_dir = os.path.dirname(image_path)
    stem = os.path.splitext(os.path.basename(image_path))[0]
    hash_input = json.dumps(info, sort_keys=True).encode("utf-8")
    hash_str = base64.urlsafe_b64encode(hashlib.md5(hash_input).digest()).decode("ascii").replace("=", "")
    return os.path.join(img_dir, "_latent_cache", f"{stem}_{hash_str}.safetensors")


def save_latent_cache(path, latent):
    from safetensors.torch import save_file

    os.makedirs(os.path.dirname(path), exist_ok=True)
    save_file(OrderedDict([("latent", latent.detach().to("cpu").contiguous())]), path)


def load_latent_cache(path, device=None, dtype=None):
    from safetensors.torch import load_file

    t = load_file(path)["latent"]
    return t.to(device=device, dtype=dtype) if (device is not None or dtype is not None) else t


def latent_info_dict(filename, plan, latent_space_version, latent_version=1, flip_x=False, flip_y=False):
    """Image-only subset of FileItemDTO.get_latent_info_dict (toolkit/dataloader_mixins.py:1779-1826); `plan` is a
    buckets.CropPlan."""
    item = OrderedDict([
        ("filename", os.path.basename(filename)),
        ("scale_to_width", plan.scale_to_width), ("scale_to_height", plan.scale_to_height),
        ("crop_x", plan.crop_x), ("crop_y", plan.crop_y),
        ("crop_width", plan.crop_width), ("crop_height", plan.crop_height),
        ("latent_space_version", latent_space_version), ("latent_version", latent_version),
    ])
    if flip_x:
        item["flip_x"] = True
    if flip_y:
        item["flip_y"] = True
    return item
