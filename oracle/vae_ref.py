"""ORACLE (test infrastructure): AutoencoderKL *encoder* + latent sampling/scaling, plain PyTorch.

The reference calls `self.vae.encode(images).latent_dist.sample()` then `scaling_factor * (latents - shift_factor)`
(toolkit/stable_diffusion_model.py:2533-2575); the encoder arithmetic itself is diffusers' `AutoencoderKL`
(un-vendored, `requirements_base.txt:3`); this file restates the published architecture with diffusers' module / parameter
names (so a diffusers VAE checkpoint loads by key).  The encoder ARITHMETIC is pinned on the reference's own in-tree LDM-style
Encoder of the same architecture (extensions_built_in/diffusion_models/flux2/src/autoencoder.py:36-233), executed by
tests/golden/make_golden.py (golden_vae_encoder) on this file's weights mapped to its names; tests/test_vae_cpu.py compares the
moments.  Unpinned: the DiagonalGaussian sample / clamp and the FLUX.1 scaling constants (diffusers config values).

  encoder.conv_in -> down_blocks[i].resnets[j] (GroupNorm32 -> SiLU -> conv3x3 -> GroupNorm32 -> SiLU -> conv3x3,
  1x1 conv_shortcut when channels change) -> downsamplers[0].conv (3x3 stride 2 on F.pad(x,(0,1,0,1))) ->
  mid_block (resnet, single-head attention over h*w tokens with GroupNorm + residual, resnet) ->
  conv_norm_out -> SiLU -> conv_out (2*latent_channels) -> DiagonalGaussian(mean, clamp(logvar,-30,20)).sample().

FLUX.1 VAE config: block_out_channels (128,256,512,512), layers_per_block 2, latent_channels 16, norm_num_groups 32,
scaling_factor 0.3611, shift_factor 0.1159, no quant_conv.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, groups=32, eps=1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))


class DownEncoderBlock2D(nn.Module):
    def __init__(self, cin, cout, layers, add_down, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class VaeAttention(nn.Module):
    def __init__(self, c, groups, eps=1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=eps)
        self.to_q = nn.Linear(c, c)
        self.to_k = nn.Linear(c, c)
        self.to_v = nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def forward(self, x):
        B, C, H, W = x.shape
        res = x
        h = self.group_norm(x).view(B, C, H * W).transpose(1, 2)
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
        o = self.to_out[0](o)
        return o.transpose(1, 2).reshape(B, C, H, W) + res


class MidBlock(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.attentions = nn.ModuleList([VaeAttention(c, groups)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, groups), ResnetBlock2D(c, c, groups)])

    def forward(self, x):
        x = self.resnets[0](x)
        x = self.attentions[0](x)
        return self.resnets[1](x)


class Encoder(nn.Module):
    def __init__(self, in_channels=3, latent_channels=16, block_out_channels=(128, 256, 512, 512), layers_per_block=2, groups=32):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[0], 3, padding=1)
        blocks = []
        c = block_out_channels[0]
        for i, co in enumerate(block_out_channels):
            blocks.append(DownEncoderBlock2D(c, co, layers_per_block, i != len(block_out_channels) - 1, groups))
            c = co
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = MidBlock(c, groups)
        self.conv_norm_out = nn.GroupNorm(groups, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, 2 * latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKLEncoder(nn.Module):
    """`vae.encode(x).latent_dist` with FLUX.1's config by default (scaling 0.3611, shift 0.1159, no quant_conv)."""

    def __init__(self, latent_channels=16, block_out_channels=(128, 256, 512, 512), layers_per_block=2, groups=32,
                 scaling_factor=0.3611, shift_factor=0.1159, use_quant_conv=False):
        super().__init__()
        self.encoder = Encoder(3, latent_channels, block_out_channels, layers_per_block, groups)
        # SD1.5 / SDXL AutoencoderKL: moments = quant_conv(encoder(x)), a 1x1 convolution (FLUX.1's VAE has none)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1) if use_quant_conv else None
        self.scaling_factor, self.shift_factor = scaling_factor, shift_factor
        self.latent_channels = latent_channels

    def moments(self, images):
        m = self.encoder(images)
        return m if self.quant_conv is None else self.quant_conv(m)

    def encode_images(self, images, eps):
        """toolkit/stable_diffusion_model.py:2567-2573 with the Gaussian sample made explicit (eps ~ N(0,1), NCHW)."""
        m = self.moments(images)
        mean, logvar = torch.chunk(m, 2, dim=1)
        logvar = torch.clamp(logvar, -30.0, 20.0)
        z = mean + torch.exp(0.5 * logvar) * eps.to(mean.dtype)
        return self.scaling_factor * (z - self.shift_factor)


def init_synthetic_(model, seed=4321):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                fan_in = m.weight[0].numel()
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (1.0 / fan_in) ** 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.02)
            elif isinstance(m, nn.GroupNorm):
                m.weight.copy_(1 + 0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.05 * torch.randn(m.bias.shape, generator=g))
    return model
