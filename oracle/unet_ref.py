"""ORACLE (test infrastructure): plain-PyTorch restatement of the diffusers `UNet2DConditionModel` subset the reference trains
for SD1.5 (BASELINE config 1) and SDXL (config 2) — never imported by the product path.

The class lives in un-vendored diffusers@c9438378 (requirements_base.txt:3) and the reference holds no in-tree restatement or
golden vector of it => block arithmetic is **parity unpinned**; what IS pinned:
  * the architecture (channel plan, layer counts, module names) by the published parameter counts of the two checkpoints
    (SD1.5 UNet 859,520,964; SDXL UNet 2,567,463,684 — tests/test_unet_cpu.py) and by the reference's own LoRA discovery run on
    this module tree (tests/golden/make_golden.py: 192 / 722 adapters, kohya key names);
  * ResnetBlock arithmetic (GroupNorm -> SiLU -> conv) on the reference's in-tree LDM-style block
    (extensions_built_in/diffusion_models/flux2/src/autoencoder.py), shared with oracle/vae_ref.py.
Call sites in the reference: toolkit/stable_diffusion_model.py:2049-2055 (SDXL, added_cond_kwargs text_embeds + time_ids from
1824-1852), 2260-2265 (SD1.5); DDPM schedule toolkit/sampler.py:31-50, 136-137; loss target extensions_built_in/sd_trainer/
SDTrainer.py:650 (eps), 623-625 (v-prediction), min-SNR 1005-1011 + toolkit/train_tools.py:642-654, 720-749.
Module / parameter names are diffusers' so checkpoints load by key and LoRASpecialNetwork produces the reference's names.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

SD15 = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
            down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
            up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
            cross_attention_dim=768, attention_head_dim=8, transformer_layers_per_block=1, use_linear_projection=False,
            addition_embed_type=None)
SDXL = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=2,
            down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
            up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
            cross_attention_dim=2048, attention_head_dim=(5, 10, 20), transformer_layers_per_block=(1, 2, 10),
            use_linear_projection=True, addition_embed_type="text_time", addition_time_embed_dim=256,
            projection_class_embeddings_input_dim=2816)


def timestep_embedding(t, dim, flip_sin_to_cos=True, downscale_freq_shift=0.0, max_period=10000):
    """diffusers get_timestep_embedding (Timesteps module; no parameters)."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / (half - downscale_freq_shift)
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_channels=1280, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Attention(nn.Module):
    def __init__(self, query_dim, heads, dim_head, cross_attention_dim=None):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        kv = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(kv, inner, bias=False)
        self.to_v = nn.Linear(kv, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Identity()])

    def forward(self, x, context=None):
        context = x if context is None else context
        B, S, _ = x.shape
        q = self.to_q(x).view(B, S, self.heads, -1).transpose(1, 2)
        k = self.to_k(context).view(B, context.shape[1], self.heads, -1).transpose(1, 2)
        v = self.to_v(context).view(B, context.shape[1], self.heads, -1).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v)
        return self.to_out[0](o.transpose(1, 2).reshape(B, S, -1))


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, 4 * dim), nn.Identity(), nn.Linear(4 * dim, dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, heads, dim_head, cross_attention_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, x, context):
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), context)
        return x + self.ff(self.norm3(x))


class Transformer2DModel(nn.Module):
    def __init__(self, heads, dim_head, in_channels, num_layers, cross_attention_dim, use_linear_projection, groups=32):
        super().__init__()
        inner = heads * dim_head
        self.use_linear_projection = use_linear_projection
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner) if use_linear_projection else nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim) for _ in range(num_layers)])
        self.proj_out = nn.Linear(inner, in_channels) if use_linear_projection else nn.Conv2d(inner, in_channels, 1)

    def forward(self, x, context):
        B, Cc, H, W = x.shape
        res = x
        h = self.norm(x)
        if self.use_linear_projection:
            h = self.proj_in(h.permute(0, 2, 3, 1).reshape(B, H * W, Cc))
        else:
            h = self.proj_in(h).permute(0, 2, 3, 1).reshape(B, H * W, -1)
        for blk in self.transformer_blocks:
            h = blk(h, context)
        if self.use_linear_projection:
            h = self.proj_out(h).reshape(B, H, W, Cc).permute(0, 3, 1, 2)
        else:
            h = self.proj_out(h.reshape(B, H, W, -1).permute(0, 3, 1, 2))
        return h + res


class Downsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _DownBlock(nn.Module):
    def __init__(self, cin, cout, layers, add_down, attn=None, temb=1280, groups=32):
        super().__init__()
        # registration order of diffusers' blocks (attentions, resnets, samplers): it is the order named_modules() walks, hence the
        # order the reference creates adapters in (and consumes the RNG) once `network.conv` also wraps the ResnetBlock2D children
        if attn is not None:
            self.attentions = nn.ModuleList([Transformer2DModel(in_channels=cout, **attn) for _ in range(layers)])
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb, groups) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None

    def forward(self, x, temb, context):
        outs = []
        for i, r in enumerate(self.resnets):
            x = r(x, temb)
            if hasattr(self, "attentions"):
                x = self.attentions[i](x, context)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class CrossAttnDownBlock2D(_DownBlock):
    pass


class DownBlock2D(_DownBlock):
    pass


class UNetMidBlock2DCrossAttn(nn.Module):
    def __init__(self, ch, attn, temb=1280, groups=32):
        super().__init__()
        self.attentions = nn.ModuleList([Transformer2DModel(in_channels=ch, **attn)])
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb, groups), ResnetBlock2D(ch, ch, temb, groups)])

    def forward(self, x, temb, context):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, context)
        return self.resnets[1](x, temb)


class _UpBlock(nn.Module):
    def __init__(self, cin, cout, prev, layers, add_up, attn=None, temb=1280, groups=32):
        super().__init__()
        res = []
        for i in range(layers):
            skip = cin if i == layers - 1 else cout
            rin = prev if i == 0 else cout
            res.append(ResnetBlock2D(rin + skip, cout, temb, groups))
        if attn is not None:
            self.attentions = nn.ModuleList([Transformer2DModel(in_channels=cout, **attn) for _ in range(layers)])
        self.resnets = nn.ModuleList(res)
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, x, skips, temb, context):
        for i, r in enumerate(self.resnets):
            x = r(torch.cat([x, skips.pop()], dim=1), temb)
            if hasattr(self, "attentions"):
                x = self.attentions[i](x, context)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class CrossAttnUpBlock2D(_UpBlock):
    pass


class UpBlock2D(_UpBlock):
    pass


class UNet2DConditionModel(nn.Module):
    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 down_block_types=SD15["down_block_types"], up_block_types=SD15["up_block_types"], cross_attention_dim=768,
                 attention_head_dim=8, transformer_layers_per_block=1, use_linear_projection=False, addition_embed_type=None,
                 addition_time_embed_dim=None, projection_class_embeddings_input_dim=None, norm_num_groups=32):
        super().__init__()
        n = len(block_out_channels)
        heads = (attention_head_dim,) * n if isinstance(attention_head_dim, int) else tuple(attention_head_dim)
        tl = (transformer_layers_per_block,) * n if isinstance(transformer_layers_per_block, int) else tuple(transformer_layers_per_block)
        self.config = dict(in_channels=in_channels, out_channels=out_channels, block_out_channels=tuple(block_out_channels),
                           layers_per_block=layers_per_block, down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
                           cross_attention_dim=cross_attention_dim, attention_head_dim=heads, transformer_layers_per_block=tl,
                           use_linear_projection=use_linear_projection, addition_embed_type=addition_embed_type,
                           addition_time_embed_dim=addition_time_embed_dim,
                           projection_class_embeddings_input_dim=projection_class_embeddings_input_dim, norm_num_groups=norm_num_groups)
        c0 = block_out_channels[0]
        temb = 4 * c0
        self.conv_in = nn.Conv2d(in_channels, c0, 3, padding=1)
        self.time_embedding = TimestepEmbedding(c0, temb)
        if addition_embed_type == "text_time":
            self.add_embedding = TimestepEmbedding(projection_class_embeddings_input_dim, temb)

        def attn_kw(i):
            return dict(heads=heads[i], dim_head=block_out_channels[i] // heads[i], num_layers=tl[i],
                        cross_attention_dim=cross_attention_dim, use_linear_projection=use_linear_projection, groups=norm_num_groups)

        downs = []
        out_ch = c0
        for i, t in enumerate(down_block_types):
            in_ch, out_ch = out_ch, block_out_channels[i]
            cls = CrossAttnDownBlock2D if t == "CrossAttnDownBlock2D" else DownBlock2D
            downs.append(cls(in_ch, out_ch, layers_per_block, add_down=i != n - 1, attn=attn_kw(i) if t == "CrossAttnDownBlock2D" else None,
                             temb=temb, groups=norm_num_groups))
        mid = UNetMidBlock2DCrossAttn(block_out_channels[-1], attn_kw(n - 1), temb, norm_num_groups)
        rev = list(reversed(block_out_channels))
        ups = []
        out_ch = rev[0]
        for i, t in enumerate(up_block_types):
            prev, out_ch, in_ch = out_ch, rev[i], rev[min(i + 1, n - 1)]
            cls = CrossAttnUpBlock2D if t == "CrossAttnUpBlock2D" else UpBlock2D
            ups.append(cls(in_ch, out_ch, prev, layers_per_block + 1, add_up=i != n - 1,
                           attn=attn_kw(n - 1 - i) if t == "CrossAttnUpBlock2D" else None, temb=temb,
                           groups=norm_num_groups))
        # diffusers registers down_blocks and up_blocks (empty ModuleLists) before mid_block: named_modules() order = adapter order
        self.down_blocks = nn.ModuleList(downs)
        self.up_blocks = nn.ModuleList(ups)
        self.mid_block = mid
        self.conv_norm_out = nn.GroupNorm(norm_num_groups, c0, eps=1e-5)
        self.conv_out = nn.Conv2d(c0, out_channels, 3, padding=1)

    def forward(self, sample, timestep, encoder_hidden_states, added_cond_kwargs=None):
        """sample [B,4,h,w], timestep [B] (0..999), encoder_hidden_states [B,77,D]; SDXL: added_cond_kwargs = {text_embeds
        [B,1280], time_ids [B,6]} (toolkit/stable_diffusion_model.py:1985-1990)."""
        cfg = self.config
        B = sample.shape[0]
        t_emb = timestep_embedding(timestep.expand(B), cfg["block_out_channels"][0]).to(sample.dtype)
        emb = self.time_embedding(t_emb)
        if cfg["addition_embed_type"] == "text_time":
            te = timestep_embedding(added_cond_kwargs["time_ids"].flatten(), cfg["addition_time_embed_dim"]).reshape(B, -1)
            add = torch.cat([added_cond_kwargs["text_embeds"], te.to(added_cond_kwargs["text_embeds"].dtype)], dim=-1).to(emb.dtype)
            emb = emb + self.add_embedding(add)
        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, emb, encoder_hidden_states)
            skips += outs
        x = self.mid_block(x, emb, encoder_hidden_states)
        for blk in self.up_blocks:
            x = blk(x, skips, emb, encoder_hidden_states)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


def init_synthetic_(model, seed=1234, std=0.02):
    """synthetic weights: Linear / Conv W ~ N(0, std^2) with small random biases, norm scales near 1 — non-trivial
    affine parameters so every path is exercised."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                fan_in = m.weight[0].numel()
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * min(std * 4, 1.0 / math.sqrt(fan_in)))
                if m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.01)
            elif isinstance(m, (nn.GroupNorm, nn.LayerNorm)):
                m.weight.copy_(1 + 0.05 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.02 * torch.randn(m.bias.shape, generator=g))
    return model


# ---------------------------------------------------------------------------------------------------- scheduler / loss
def ddpm_alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    """DDPMScheduler(beta_schedule='scaled_linear') from the reference's sd_config (toolkit/sampler.py:31-50, 136-137)."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def ddpm_add_noise(x0, noise, timesteps, alphas_cumprod):
    """DDPMScheduler.add_noise as reached from StableDiffusion.add_noise (toolkit/stable_diffusion_model.py:1854-1876)."""
    ac = alphas_cumprod.to(x0.device, x0.dtype)
    a = ac[timesteps.long()] ** 0.5
    s = (1 - ac[timesteps.long()]) ** 0.5
    return a.view(-1, 1, 1, 1) * x0 + s.view(-1, 1, 1, 1) * noise


def ddpm_velocity(x0, noise, timesteps, alphas_cumprod):
    """DDPMScheduler.get_velocity (v-prediction target, SDTrainer.py:623-625)."""
    ac = alphas_cumprod.to(x0.device, x0.dtype)
    a = ac[timesteps.long()] ** 0.5
    s = (1 - ac[timesteps.long()]) ** 0.5
    return a.view(-1, 1, 1, 1) * noise - s.view(-1, 1, 1, 1) * x0


def min_snr_weight(timesteps, alphas_cumprod, gamma, fixed=False):
    """apply_snr_weight (toolkit/train_tools.py:720-749) with all_snr from get_all_snr (642-654); DDPM timesteps never start at
    1000 so offset = 0."""
    ac = alphas_cumprod.float()
    all_snr = (torch.sqrt(ac) / torch.sqrt(1.0 - ac)) ** 2
    snr = all_snr.to(timesteps.device)[timesteps.long()]
    g = gamma / snr
    return g.float() if fixed else torch.minimum(g, torch.ones_like(g)).float()


def time_ids_from_latents(latents, vae_scale_factor=8):
    """StableDiffusion.get_time_ids_from_latents (toolkit/stable_diffusion_model.py:1824-1852): (H, W, 0, 0, H, W) per sample."""
    bs, _, h, w = latents.shape
    H, W = h * vae_scale_factor, w * vae_scale_factor
    ids = torch.tensor([[H, W, 0, 0, H, W]], dtype=latents.dtype, device=latents.device)
    return ids.repeat(bs, 1)
