"""ORACLE (test infrastructure, never shipped / never measured as the product).

CPU restatement, in plain PyTorch, of the FLUX.1 DiT that the reference trains LoRAs on.  The arithmetic is NOT in
/root/reference: it lives in `diffusers @ c943837899b16cbae2f619b8dd4f7bb6f07dd81a` (requirements_base.txt:3),
class `FluxTransformer2DModel`, which is absent from this image.  So this file restates the published algorithm and
anchors it on the reference's own call sites / in-tree restatements:

  * call site + I/O packing .......... toolkit/stable_diffusion_model.py:2154-2222
  * module inventory / key names ..... scripts/convert_diffusers_to_comfy.py:67-285 (norm_out = [scale, shift], 287-290)
  * attention order (norm->cat->rope) . toolkit/models/flux_sage_attn.py:26-93
  * RoPE ............................. extensions_built_in/diffusion_models/chroma/src/math.py:34-51
  * double / single block, last layer . extensions_built_in/diffusion_models/chroma/src/layers.py:417-455,471-607,610-681,684-720
  * timestep sinusoid ................ extensions_built_in/diffusion_models/chroma/src/layers.py:30-53
  * embedder composition ............. toolkit/models/flux.py:10-16

PARITY: the reference holds no golden vector / known-answer test at this boundary and diffusers cannot be
imported here, but the block ARITHMETIC is pinned on the reference's own executable restatement of the FLUX blocks: the Chroma
model's DoubleStreamBlock / SingleStreamBlock / LastLayer / EmbedND / timestep_embedding / MLPEmbedder are run by
tests/golden/make_golden.py (golden_flux_blocks) on this file's weights mapped through the reference's diffusers<->BFL key map,
and tests/test_flux_blocks_golden.py holds this file to those outputs (RoPE table and application, QK-RMSNorm, joint attention
order, gate / residual / MLP arithmetic, single-block split and concat order, last layer with the [scale, shift] swap, timestep
sinusoid and its MLP, and the timestep + pooled-text embedder sum through the reference's guidance_embed_bypass_forward,
toolkit/models/flux.py:8-14).  Not executable-pinned (diffusers-only glue): the chunk order of the adaLN projections — (shift, scale,
gate) x 2, the order in which the reference's distribute_modulations (chroma/src/layers.py:92-176) fills `img_mod.lin` /
`txt_mod.lin` / `modulation.lin`, which the key map sends to norm1.linear / norm1_context.linear / norm.linear unchanged —
the guidance-embedder term (same form as the pinned timestep embedder) and the plain x_embedder / context_embedder Linears.
Class and attribute names follow diffusers exactly so the reference's LoRASpecialNetwork attaches to it and produces the
reference's state-dict keys (pinned as well, see tests/golden/make_golden.py).
"""
import math
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


def get_timestep_embedding(timesteps: torch.Tensor, dim: int = 256, max_period: int = 10000) -> torch.Tensor:
    """diffusers `Timesteps(num_channels=256, flip_sin_to_cos=True, downscale_freq_shift=0)` -> [cos | sin]."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device) / half
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels: int, time_embed_dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, x):
        return self.linear_2(self.act(self.linear_1(x)))


class PixArtAlphaTextProjection(nn.Module):
    def __init__(self, in_features: int, hidden_size: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_features, hidden_size)
        self.act_1 = nn.SiLU()
        self.linear_2 = nn.Linear(hidden_size, hidden_size)

    def forward(self, x):
        return self.linear_2(self.act_1(self.linear_1(x)))


class CombinedTimestepGuidanceTextProjEmbeddings(nn.Module):
    """toolkit/models/flux.py:10-16 shows the composition (timestep_embedder(time_proj(t)) + text_embedder(pooled));
    the guidance-distilled model adds guidance_embedder(time_proj(g))."""

    def __init__(self, embedding_dim: int, pooled_projection_dim: int):
        super().__init__()
        self.timestep_embedder = TimestepEmbedding(256, embedding_dim)
        self.guidance_embedder = TimestepEmbedding(256, embedding_dim)
        self.text_embedder = PixArtAlphaTextProjection(pooled_projection_dim, embedding_dim)

    @staticmethod
    def time_proj(t):
        """diffusers `Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0)` — the attribute the reference's
        guidance_embed_bypass_forward calls (toolkit/models/flux.py:9)."""
        return get_timestep_embedding(t, 256)

    def forward(self, timestep, guidance, pooled_projection):
        t_emb = self.timestep_embedder(get_timestep_embedding(timestep).to(pooled_projection.dtype))
        if guidance is None:  # guidance_embed_bypass_forward (toolkit/models/flux.py:9-15): timestep + pooled text only
            return t_emb + self.text_embedder(pooled_projection)
        g_emb = self.guidance_embedder(get_timestep_embedding(guidance).to(pooled_projection.dtype))
        return t_emb + g_emb + self.text_embedder(pooled_projection)


class RMSNorm(nn.Module):
    """diffusers RMSNorm(dim_head, eps=1e-6) with learned weight: variance in fp32, cast to weight dtype, scale."""

    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        variance = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        x = x * torch.rsqrt(variance + self.eps)
        if self.weight.dtype in (torch.float16, torch.bfloat16):
            x = x.to(self.weight.dtype)
        return x * self.weight


def rope_freqs(ids: torch.Tensor, axes_dim=(16, 56, 56), theta: float = 10000.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """FluxPosEmbed: per axis, freqs = pos * theta^(-2i/d) in float64; cos/sin repeat-interleaved x2; axes concatenated.
    Same angles as chroma/src/math.py:34-44 (`rope`), laid out as (cos, sin) [S, sum(axes_dim)] fp32."""
    cos_out, sin_out = [], []
    pos = ids.to(torch.float64)
    for i, d in enumerate(axes_dim):
        freqs = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64, device=ids.device)[: d // 2] / d))
        ang = torch.outer(pos[:, i], freqs)
        cos_out.append(ang.cos().repeat_interleave(2, dim=1).float())
        sin_out.append(ang.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos_out, dim=-1), torch.cat(sin_out, dim=-1)


def apply_rotary_emb(x: torch.Tensor, freqs: Tuple[torch.Tensor, torch.Tensor]) -> torch.Tensor:
    """x [B,H,S,D]; out[2i] = x[2i] cos - x[2i+1] sin ; out[2i+1] = x[2i+1] cos + x[2i] sin   (chroma math.py:47-51)."""
    cos, sin = freqs
    cos, sin = cos[None, None], sin[None, None]
    x_real, x_imag = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    x_rot = torch.stack([-x_imag, x_real], dim=-1).flatten(3)
    return (x.float() * cos + x_rot.float() * sin).to(x.dtype)


class Attention(nn.Module):
    def __init__(self, dim: int, heads: int, dim_head: int, added_kv: bool, pre_only: bool = False):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.norm_q = RMSNorm(dim_head)
        self.norm_k = RMSNorm(dim_head)
        self.to_q = nn.Linear(dim, inner)
        self.to_k = nn.Linear(dim, inner)
        self.to_v = nn.Linear(dim, inner)
        if added_kv:
            self.add_k_proj = nn.Linear(dim, inner)
            self.add_v_proj = nn.Linear(dim, inner)
            self.add_q_proj = nn.Linear(dim, inner)
            self.norm_added_q = RMSNorm(dim_head)
            self.norm_added_k = RMSNorm(dim_head)
        if not pre_only:
            self.to_out = nn.ModuleList([nn.Linear(inner, dim), nn.Dropout(0.0)])
        if added_kv:
            self.to_add_out = nn.Linear(inner, dim)
        self.added_kv = added_kv
        self.pre_only = pre_only

    def forward(self, hidden_states, encoder_hidden_states=None, image_rotary_emb=None):
        # order restated from toolkit/models/flux_sage_attn.py:26-93
        B = hidden_states.shape[0]
        H = self.heads

        def split(t):
            return t.view(B, -1, H, t.shape[-1] // H).transpose(1, 2)

        q = self.norm_q(split(self.to_q(hidden_states)))
        k = self.norm_k(split(self.to_k(hidden_states)))
        v = split(self.to_v(hidden_states))
        if encoder_hidden_states is not None:
            eq = self.norm_added_q(split(self.add_q_proj(encoder_hidden_states)))
            ek = self.norm_added_k(split(self.add_k_proj(encoder_hidden_states)))
            ev = split(self.add_v_proj(encoder_hidden_states))
            q = torch.cat([eq, q], dim=2)
            k = torch.cat([ek, k], dim=2)
            v = torch.cat([ev, v], dim=2)
        if image_rotary_emb is not None:
            q = apply_rotary_emb(q, image_rotary_emb)
            k = apply_rotary_emb(k, image_rotary_emb)
        o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(B, -1, H * o.shape[-1]).to(q.dtype)
        if encoder_hidden_states is not None:
            n_txt = encoder_hidden_states.shape[1]
            eo, o = o[:, :n_txt], o[:, n_txt:]
            o = self.to_out[1](self.to_out[0](o))
            eo = self.to_add_out(eo)
            return o, eo
        return o


class AdaLayerNormZero(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(dim, 6 * dim)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, emb):
        emb = self.linear(self.silu(emb))
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = emb.chunk(6, dim=1)
        x = self.norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
        return x, gate_msa, shift_mlp, scale_mlp, gate_mlp


class AdaLayerNormZeroSingle(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(dim, 3 * dim)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, emb):
        emb = self.linear(self.silu(emb))
        shift_msa, scale_msa, gate_msa = emb.chunk(3, dim=1)
        x = self.norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
        return x, gate_msa


class AdaLayerNormContinuous(nn.Module):
    """norm_out: emb -> [scale, shift] (order per scripts/convert_diffusers_to_comfy.py:287-290 swap_scale_shift)."""

    def __init__(self, dim: int, cond_dim: int):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(cond_dim, 2 * dim)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, cond):
        emb = self.linear(self.silu(cond).to(x.dtype))
        scale, shift = torch.chunk(emb, 2, dim=1)
        return self.norm(x) * (1 + scale)[:, None, :] + shift[:, None, :]


class GELU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)

    def forward(self, x):
        return F.gelu(self.proj(x), approximate="tanh")


class FeedForward(nn.Module):
    def __init__(self, dim: int, mult: int = 4):
        super().__init__()
        self.net = nn.ModuleList([GELU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class FluxTransformerBlock(nn.Module):
    def __init__(self, dim: int, heads: int, dim_head: int):
        super().__init__()
        self.norm1 = AdaLayerNormZero(dim)
        self.norm1_context = AdaLayerNormZero(dim)
        self.attn = Attention(dim, heads, dim_head, added_kv=True)
        self.norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff = FeedForward(dim)
        self.norm2_context = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff_context = FeedForward(dim)

    def forward(self, hidden_states, encoder_hidden_states, temb, image_rotary_emb):
        n_h, gate_msa, shift_mlp, scale_mlp, gate_mlp = self.norm1(hidden_states, temb)
        n_e, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = self.norm1_context(encoder_hidden_states, temb)
        attn_out, ctx_attn_out = self.attn(n_h, n_e, image_rotary_emb)
        hidden_states = hidden_states + gate_msa.unsqueeze(1) * attn_out
        n_h = self.norm2(hidden_states) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
        hidden_states = hidden_states + gate_mlp.unsqueeze(1) * self.ff(n_h)
        encoder_hidden_states = encoder_hidden_states + c_gate_msa.unsqueeze(1) * ctx_attn_out
        n_e = self.norm2_context(encoder_hidden_states) * (1 + c_scale_mlp[:, None]) + c_shift_mlp[:, None]
        encoder_hidden_states = encoder_hidden_states + c_gate_mlp.unsqueeze(1) * self.ff_context(n_e)
        return encoder_hidden_states, hidden_states


class FluxSingleTransformerBlock(nn.Module):
    def __init__(self, dim: int, heads: int, dim_head: int, mlp_ratio: float = 4.0):
        super().__init__()
        self.mlp_hidden_dim = int(dim * mlp_ratio)
        self.norm = AdaLayerNormZeroSingle(dim)
        self.proj_mlp = nn.Linear(dim, self.mlp_hidden_dim)
        self.act_mlp = nn.GELU(approximate="tanh")
        self.proj_out = nn.Linear(dim + self.mlp_hidden_dim, dim)
        self.attn = Attention(dim, heads, dim_head, added_kv=False, pre_only=True)

    def forward(self, hidden_states, temb, image_rotary_emb):
        residual = hidden_states
        n_h, gate = self.norm(hidden_states, temb)
        mlp = self.act_mlp(self.proj_mlp(n_h))
        attn_out = self.attn(n_h, None, image_rotary_emb)
        hidden_states = torch.cat([attn_out, mlp], dim=2)
        return residual + gate.unsqueeze(1) * self.proj_out(hidden_states)


class FluxTransformer2DModel(nn.Module):
    """Same constructor keys as the diffusers config.  Real FLUX.1-dev: defaults below."""

    def __init__(self, in_channels=64, num_layers=19, num_single_layers=38, attention_head_dim=128,
                 num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768,
                 guidance_embeds=True, axes_dims_rope=(16, 56, 56)):
        super().__init__()
        self.inner_dim = num_attention_heads * attention_head_dim
        self.axes_dims_rope = tuple(axes_dims_rope)
        assert sum(self.axes_dims_rope) == attention_head_dim
        assert guidance_embeds, "FLUX.1-dev is guidance-distilled"
        self.config = dict(in_channels=in_channels, num_layers=num_layers, num_single_layers=num_single_layers,
                           attention_head_dim=attention_head_dim, num_attention_heads=num_attention_heads,
                           joint_attention_dim=joint_attention_dim, pooled_projection_dim=pooled_projection_dim,
                           guidance_embeds=guidance_embeds, axes_dims_rope=self.axes_dims_rope)
        d = self.inner_dim
        self.time_text_embed = CombinedTimestepGuidanceTextProjEmbeddings(d, pooled_projection_dim)
        self.context_embedder = nn.Linear(joint_attention_dim, d)
        self.x_embedder = nn.Linear(in_channels, d)
        self.transformer_blocks = nn.ModuleList(
            [FluxTransformerBlock(d, num_attention_heads, attention_head_dim) for _ in range(num_layers)])
        self.single_transformer_blocks = nn.ModuleList(
            [FluxSingleTransformerBlock(d, num_attention_heads, attention_head_dim) for _ in range(num_single_layers)])
        self.norm_out = AdaLayerNormContinuous(d, d)
        self.proj_out = nn.Linear(d, in_channels)

    def forward(self, hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids, guidance):
        """hidden_states [B, N_img, 64]; encoder_hidden_states [B, N_txt, 4096]; timestep, guidance in [0,1] / raw
        (the model multiplies both by 1000, as the pinned diffusers does; the trainer passes timestep/1000,
        toolkit/stable_diffusion_model.py:2192-2205)."""
        hidden_states = self.x_embedder(hidden_states)
        timestep = timestep.to(hidden_states.dtype) * 1000
        guidance = None if guidance is None else guidance.to(hidden_states.dtype) * 1000
        temb = self.time_text_embed(timestep, guidance, pooled_projections)
        encoder_hidden_states = self.context_embedder(encoder_hidden_states)
        ids = torch.cat((txt_ids, img_ids), dim=0)
        rot = rope_freqs(ids, self.axes_dims_rope)
        for blk in self.transformer_blocks:
            encoder_hidden_states, hidden_states = blk(hidden_states, encoder_hidden_states, temb, rot)
        n_txt = encoder_hidden_states.shape[1]
        hidden_states = torch.cat([encoder_hidden_states, hidden_states], dim=1)
        for blk in self.single_transformer_blocks:
            hidden_states = blk(hidden_states, temb, rot)
        hidden_states = hidden_states[:, n_txt:]
        hidden_states = self.norm_out(hidden_states, temb)
        return self.proj_out(hidden_states)


def init_synthetic_(model: nn.Module, seed: int = 1234, std: float = 0.02):
    """BASELINE.md §2 synthetic weights: every Linear W ~ N(0, std^2), biases 0, RMSNorm scales 1."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, nn.Linear):
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * std)
                if m.bias is not None:
                    m.bias.zero_()
            elif isinstance(m, RMSNorm):
                m.weight.fill_(1.0)
    return model


def pack_latents(x: torch.Tensor) -> torch.Tensor:
    """b c (h 2) (w 2) -> b (h w) (c 2 2)   (toolkit/stable_diffusion_model.py:2157-2163)."""
    B, Cc, H, W = x.shape
    x = x.view(B, Cc, H // 2, 2, W // 2, 2)
    return x.permute(0, 2, 4, 1, 3, 5).reshape(B, (H // 2) * (W // 2), Cc * 4)


def unpack_latents(x: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """b (h w) (c 2 2) -> b c (h 2) (w 2)   (toolkit/stable_diffusion_model.py:2210-2219)."""
    B, _, Cp = x.shape
    Cc = Cp // 4
    x = x.view(B, H // 2, W // 2, Cc, 2, 2)
    return x.permute(0, 3, 1, 4, 2, 5).reshape(B, Cc, H, W)


def make_ids(H: int, W: int, n_txt: int, device="cpu"):
    """img_ids [h/2*w/2, 3] with (0, row, col); txt_ids zeros [n_txt, 3]  (stable_diffusion_model.py:2165-2170, 2187-2190)."""
    h2, w2 = H // 2, W // 2
    img_ids = torch.zeros(h2, w2, 3, device=device)
    img_ids[..., 1] = img_ids[..., 1] + torch.arange(h2, device=device)[:, None]
    img_ids[..., 2] = img_ids[..., 2] + torch.arange(w2, device=device)[None, :]
    return img_ids.reshape(h2 * w2, 3), torch.zeros(n_txt, 3, device=device)
