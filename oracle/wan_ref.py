"""ORACLE (test infrastructure): Wan2.1 T2V video DiT (`WanTransformer3DModel`), plain PyTorch.

PARITY UNPINNED for the block arithmetic: the reference calls diffusers' `WanTransformer3DModel`
(toolkit/models/wan21/wan21.py:343-420 loads it, :578-603 calls it) and diffusers is not in this image, so this file
restates the published architecture with diffusers' module / parameter names (checkpoints and LoRA keys match by name).
The one piece that IS reference code is the attention processor, toolkit/models/wan21/wan_attn.py:12-103: q/k/v
projections -> norm_q / norm_k (RMSNorm over the whole projection, "rms_norm_across_heads") -> split heads -> rotary
embedding as a complex multiply on (2i, 2i+1) pairs in float64 (self-attention only) -> SDPA -> to_out; `Attention`
below follows it line by line for the T2V case (no image-conditioning branch) and is PINNED to it: make_golden.py executes the
reference processor on this module and tests/test_wan_cpu.py compares (self-attention with RoPE, text cross-attention).

Model:
  patch_embedding Conv3d(16, D, k=s=(1,2,2)) -> tokens (f, h/2, w/2)
  condition_embedder: sinusoidal(256, flip_sin_to_cos, shift 0) -> Linear-SiLU-Linear = temb [B,D];
      time_proj(SiLU(temb)) -> [B,6,D]; text_embedder Linear-GELU(tanh)-Linear on the UMT5 states
  block: (scale_shift_table[1,6,D] + time_proj).chunk(6) in fp32;
      x += gate_msa * attn1(LN(x)(1+scale_msa)+shift_msa);  x += attn2(LN_affine(x), text);
      x += c_gate * ffn(LN(x)(1+c_scale)+c_shift)     (FP32LayerNorm, eps 1e-6; ffn = Linear-GELU(tanh)-Linear)
  head: LN(x)(1+scale)+shift with (scale_shift_table[1,2,D] + temb) -> proj_out Linear(D, 64) -> unpatchify.
Wan2.1-T2V-1.3B: D=1536 (12 heads x 128), ffn 8960, 30 blocks, text_dim 4096, freq_dim 256.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .flux_ref import TimestepEmbedding, get_timestep_embedding


class FP32LayerNorm(nn.LayerNorm):
    def forward(self, x):
        return F.layer_norm(x.float(), self.normalized_shape, self.weight.float() if self.weight is not None else None,
                            self.bias.float() if self.bias is not None else None, self.eps).to(x.dtype)


class RMSNormAcrossHeads(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        dt = x.dtype
        var = x.float().pow(2).mean(-1, keepdim=True)
        x = x.float() * torch.rsqrt(var + self.eps)
        if self.weight.dtype in (torch.float16, torch.bfloat16):
            x = x.to(self.weight.dtype)
        return (x * self.weight).to(dt)


def wan_rope_freqs(F_, H_, W_, head_dim=128, theta=10000.0):
    """WanRotaryPosEmbed: per-axis complex frequencies, dims (d-4*(d//6), 2*(d//6), 2*(d//6)); returns angles [S, d/2] f64."""
    h_dim = w_dim = 2 * (head_dim // 6)
    t_dim = head_dim - h_dim - w_dim
    outs = []
    for dim, n, shape in ((t_dim, F_, (F_, 1, 1)), (h_dim, H_, (1, H_, 1)), (w_dim, W_, (1, 1, W_))):
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float64)[: dim // 2] / dim))
        ang = torch.outer(torch.arange(n, dtype=torch.float64), freqs)  # [n, dim/2]
        outs.append(ang.view(*shape, dim // 2).expand(F_, H_, W_, dim // 2))
    return torch.cat(outs, dim=-1).reshape(F_ * H_ * W_, head_dim // 2)


def apply_rotary_emb(x, ang):
    """x [B,H,S,D]; complex multiply in float64 (wan_attn.py:48-54)."""
    xr = torch.view_as_complex(x.to(torch.float64).unflatten(3, (-1, 2)))
    fr = torch.polar(torch.ones_like(ang), ang)[None, None].to(x.device)
    return torch.view_as_real(xr * fr).flatten(3, 4).type_as(x)


class Attention(nn.Module):
    def __init__(self, dim, heads, dim_head, eps=1e-6):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = nn.Linear(dim, inner)
        self.to_k = nn.Linear(dim, inner)
        self.to_v = nn.Linear(dim, inner)
        self.to_out = nn.ModuleList([nn.Linear(inner, dim), nn.Dropout(0.0)])
        self.norm_q = RMSNormAcrossHeads(inner, eps)
        self.norm_k = RMSNormAcrossHeads(inner, eps)

    def forward(self, hidden_states, encoder_hidden_states=None, rotary_emb=None):
        enc = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        q = self.norm_q(self.to_q(hidden_states))
        k = self.norm_k(self.to_k(enc))
        v = self.to_v(enc)
        q = q.unflatten(2, (self.heads, -1)).transpose(1, 2)
        k = k.unflatten(2, (self.heads, -1)).transpose(1, 2)
        v = v.unflatten(2, (self.heads, -1)).transpose(1, 2)
        if rotary_emb is not None:
            q = apply_rotary_emb(q, rotary_emb)
            k = apply_rotary_emb(k, rotary_emb)
        o = F.scaled_dot_product_attention(q, k, v)
        o = o.transpose(1, 2).flatten(2, 3).type_as(q)
        return self.to_out[0](o)


class _GELUProj(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.proj = nn.Linear(cin, cout)

    def forward(self, x):
        return F.gelu(self.proj(x), approximate="tanh")


class FeedForward(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.net = nn.ModuleList([_GELUProj(dim, inner), nn.Dropout(0.0), nn.Linear(inner, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class WanTransformerBlock(nn.Module):
    def __init__(self, dim, ffn_dim, heads, eps=1e-6):
        super().__init__()
        self.norm1 = FP32LayerNorm(dim, eps, elementwise_affine=False)
        self.attn1 = Attention(dim, heads, dim // heads, eps)
        self.attn2 = Attention(dim, heads, dim // heads, eps)
        self.norm2 = FP32LayerNorm(dim, eps, elementwise_affine=True)
        self.ffn = FeedForward(dim, ffn_dim)
        self.norm3 = FP32LayerNorm(dim, eps, elementwise_affine=False)
        self.scale_shift_table = nn.Parameter(torch.randn(1, 6, dim) / dim ** 0.5)

    def forward(self, hidden_states, encoder_hidden_states, temb, rotary_emb):
        shift_msa, scale_msa, gate_msa, c_shift, c_scale, c_gate = (self.scale_shift_table + temb.float()).chunk(6, dim=1)
        n = (self.norm1(hidden_states.float()) * (1 + scale_msa) + shift_msa).type_as(hidden_states)
        a = self.attn1(n, rotary_emb=rotary_emb)
        hidden_states = (hidden_states.float() + a * gate_msa).type_as(hidden_states)
        n = self.norm2(hidden_states.float()).type_as(hidden_states)
        hidden_states = hidden_states + self.attn2(n, encoder_hidden_states=encoder_hidden_states)
        n = (self.norm3(hidden_states.float()) * (1 + c_scale) + c_shift).type_as(hidden_states)
        f = self.ffn(n)
        return (hidden_states.float() + f.float() * c_gate).type_as(hidden_states)


class _TextProjection(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.gelu(self.linear_1(x), approximate="tanh"))


class WanTimeTextImageEmbedding(nn.Module):
    def __init__(self, dim, freq_dim, text_dim):
        super().__init__()
        self.freq_dim = freq_dim
        self.time_embedder = TimestepEmbedding(freq_dim, dim)
        self.time_proj = nn.Linear(dim, 6 * dim)
        self.text_embedder = _TextProjection(text_dim, dim)

    def forward(self, timestep, encoder_hidden_states):
        proj = get_timestep_embedding(timestep, self.freq_dim)
        temb = self.time_embedder(proj.to(encoder_hidden_states.dtype)).type_as(encoder_hidden_states)
        timestep_proj = self.time_proj(F.silu(temb))
        return temb, timestep_proj, self.text_embedder(encoder_hidden_states)


class WanTransformer3DModel(nn.Module):
    def __init__(self, patch_size=(1, 2, 2), num_attention_heads=12, attention_head_dim=128, in_channels=16, out_channels=16,
                 text_dim=4096, freq_dim=256, ffn_dim=8960, num_layers=30, eps=1e-6):
        super().__init__()
        d = num_attention_heads * attention_head_dim
        self.patch_size, self.out_channels, self.head_dim = tuple(patch_size), out_channels, attention_head_dim
        self.patch_embedding = nn.Conv3d(in_channels, d, kernel_size=patch_size, stride=patch_size)
        self.condition_embedder = WanTimeTextImageEmbedding(d, freq_dim, text_dim)
        self.blocks = nn.ModuleList([WanTransformerBlock(d, ffn_dim, num_attention_heads, eps) for _ in range(num_layers)])
        self.norm_out = FP32LayerNorm(d, eps, elementwise_affine=False)
        self.proj_out = nn.Linear(d, out_channels * math.prod(patch_size))
        self.scale_shift_table = nn.Parameter(torch.randn(1, 2, d) / d ** 0.5)

    def forward(self, hidden_states, timestep, encoder_hidden_states):
        B, C, Fr, Hh, W = hidden_states.shape
        pt, ph, pw = self.patch_size
        f2, h2, w2 = Fr // pt, Hh // ph, W // pw
        rotary = wan_rope_freqs(f2, h2, w2, self.head_dim)
        x = self.patch_embedding(hidden_states).flatten(2).transpose(1, 2)
        temb, timestep_proj, enc = self.condition_embedder(timestep, encoder_hidden_states)
        timestep_proj = timestep_proj.unflatten(1, (6, -1))
        for blk in self.blocks:
            x = blk(x, enc, timestep_proj, rotary)
        shift, scale = (self.scale_shift_table + temb.unsqueeze(1)).chunk(2, dim=1)
        x = (self.norm_out(x.float()) * (1 + scale) + shift).type_as(x)
        x = self.proj_out(x)
        x = x.reshape(B, f2, h2, w2, pt, ph, pw, -1).permute(0, 7, 1, 4, 2, 5, 3, 6)
        return x.flatten(6, 7).flatten(4, 5).flatten(2, 3)


def init_synthetic_(model, seed=2468, std=0.02):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "norm_q" in name or "norm_k" in name or (name.endswith("norm2.weight")):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith("norm2.bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            elif "scale_shift_table" in name:
                p.copy_(torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5)
            elif p.ndim >= 2:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * min(std * 3, (1.0 / fan_in) ** 0.5))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * std)
    return model


def pack_video_latents(x, patch=(1, 2, 2)):
    """[B,C,F,H,W] -> [B, F*(H/2)*(W/2), C*4] with the Conv3d weight's (c, pt, ph, pw) feature order."""
    B, C, Fr, Hh, W = x.shape
    pt, ph, pw = patch
    x = x.view(B, C, Fr // pt, pt, Hh // ph, ph, W // pw, pw).permute(0, 2, 4, 6, 1, 3, 5, 7)
    return x.reshape(B, (Fr // pt) * (Hh // ph) * (W // pw), C * pt * ph * pw)


def pack_video_output(x, patch=(1, 2, 2)):
    """[B,C,F,H,W] -> tokens with proj_out's (pt, ph, pw, c) feature order (inverse of the model's unpatchify)."""
    B, C, Fr, Hh, W = x.shape
    pt, ph, pw = patch
    x = x.view(B, C, Fr // pt, pt, Hh // ph, ph, W // pw, pw).permute(0, 2, 4, 6, 3, 5, 7, 1)
    return x.reshape(B, (Fr // pt) * (Hh // ph) * (W // pw), pt * ph * pw * C)
