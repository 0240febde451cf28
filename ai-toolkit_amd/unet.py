"""UNet2DConditionModel (SD1.5 / SDXL) forward / backward as an explicit op graph over the gfx950 kernels (host logic only).

Module tree, class names and parameter names are those of diffusers' UNet2DConditionModel — the class the reference loads through
StableDiffusion(XL)Pipeline and calls at toolkit/stable_diffusion_model.py:2049-2055 (SDXL: added_cond_kwargs text_embeds + time_ids,
1824-1852, 1985-1990) and 2260-2265 (SD1.5) — so diffusers checkpoints load by key and the LoRA network produces the reference's
kohya names (`lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q`, toolkit/lora_special.py:463-502;
LoRA targets = Linear and 1x1 Conv2d children of every Transformer2DModel, toolkit/kohya_lora.py:750).

Activations are NHWC: one row per pixel, [B*H*W, C] bf16, so every 1x1 convolution / Linear is the LoRA-fused token GEMM of the
transformer path (graph.py), every 3x3 convolution the implicit-GEMM MFMA kernel (aitk_gemm_nt conv mode; data gradient = the same
kernel on the rotated, in/out-swapped filter; a stride-2 convolution's data gradient = zero insertion + stride-1 conv), GroupNorm /
LayerNorm / GEGLU / resampling are row kernels, self- and cross-attention (77 text tokens) run on the flash kernels with head_dim
40 / 64 / 80 zero-padded to 128 (exact).  The UNet is a DAG (skip connections, residuals), so backward is a reverse sweep over a
tape of kernel-level closures recorded by forward; only tensors downstream of an adapter are recorded (SDXL's first down block and
conv_in never see backward).  All arithmetic is C-ABI kernel calls; `ops` = ai_toolkit_amd.ops on MI355X, oracle/ref_ops.py in the
CPU host-logic tests.
"""
import math

import torch
import torch.nn as nn

from .graph import EPI_ACCUM, EPI_ADD_AUX, FusedGraphBase, Linear, _Holder

SD15_CONFIG = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                   down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
                   up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
                   cross_attention_dim=768, attention_head_dim=8, transformer_layers_per_block=1, use_linear_projection=False,
                   addition_embed_type=None)
SDXL_CONFIG = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=2,
                   down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                   up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
                   cross_attention_dim=2048, attention_head_dim=(5, 10, 20), transformer_layers_per_block=(1, 2, 10),
                   use_linear_projection=True, addition_embed_type="text_time", addition_time_embed_dim=256,
                   projection_class_embeddings_input_dim=2816)
PAD_D = 128  # head_dim of the flash attention kernels


class Conv1x1(Linear):
    """1x1 Conv2d of the diffusers tree (Transformer2DModel.proj_in / proj_out of SD1.5, ResnetBlock2D.conv_shortcut) held as the
    [out, in] matrix the token GEMM multiplies with; state_dict keeps the diffusers [out, in, 1, 1] layout."""

    is_conv1x1 = True
    # the attributes the reference's LoRAModule reads off a Conv2d when it builds the adapter (toolkit/lora_special.py:80-104)
    kernel_size, stride, padding, dilation, groups = (1, 1), (1, 1), (0, 0), (1, 1), 1

    def __init__(self, cin, cout, dtype, device):
        super().__init__(cin, cout, True, dtype, device)
        self.in_channels, self.out_channels = cin, cout
        self._register_load_state_dict_pre_hook(self._squeeze)
        self._register_state_dict_hook(self._unsqueeze)

    @staticmethod
    def _squeeze(state_dict, prefix, *args):
        k = prefix + "weight"
        if k in state_dict and state_dict[k].dim() == 4:
            state_dict[k] = state_dict[k][:, :, 0, 0]

    @staticmethod
    def _unsqueeze(module, state_dict, prefix, local_metadata):
        k = prefix + "weight"
        if k in state_dict and state_dict[k].dim() == 2:
            state_dict[k] = state_dict[k][:, :, None, None]


class Conv3x3(nn.Module):
    """Frozen 3x3 Conv2d (diffusers parameter layout [out, in, 3, 3]); prepare() builds the implicit-GEMM operands:
    wk [out, 9*in] (k = (ky*3+kx)*in + cin) for forward and wd [in, 9*out] (rotated filter, in/out swapped) for the data gradient."""

    kernel_size, padding, dilation, groups = (3, 3), (1, 1), (1, 1), 1
    is_conv3x3 = True  # LoRA discovery: wrapped when network.conv is set (toolkit/lora_special.py:585-590)

    def __init__(self, cin, cout, stride, dtype, device, cin_pad=None, cout_pad=None):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = cin, cout, stride
        self.cin_pad = cin_pad or cin  # conv_in: 4 latent channels zero-padded to 8 (the kernel's 16-byte channel chunks)
        self.cout_pad = cout_pad or cout  # conv_out: 4 output channels zero-padded to 8 (so its data gradient reads 16-byte chunks)
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 3, dtype=dtype, device=device), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout, dtype=dtype, device=device), requires_grad=False)
        self.wk = self.wd = self.bias_k = None
        object.__setattr__(self, "lora", None)  # set by LoRAModule.apply_to when network.conv wraps 3x3 convolutions
        # load_state_dict on a prepared layer (the reference's merge_in / merge_out of a network.conv adapter writes the merged filter back
        # through org_module.load_state_dict, toolkit/network_mixins.py:452-462; a whole-model load after prepare() does the same): the
        # implicit-GEMM operands wk / wd / bias_k are rebuilt from the new weight, like graph.Linear refreshes its transposed copy
        from .graph import _note_loaded_keys

        self._register_load_state_dict_pre_hook(_note_loaded_keys, with_module=True)
        self.register_load_state_dict_post_hook(_conv3x3_weights_loaded)

    @torch.no_grad()
    def prepare(self, need_dgrad=True):
        w = self.weight.data
        if self.cin_pad != self.in_channels:
            w = torch.cat((w, torch.zeros(w.shape[0], self.cin_pad - self.in_channels, 3, 3, dtype=w.dtype, device=w.device)), 1)
        self.bias_k = self.bias.data
        if self.cout_pad != self.out_channels:
            w = torch.cat((w, torch.zeros(self.cout_pad - self.out_channels, w.shape[1], 3, 3, dtype=w.dtype, device=w.device)), 0)
            self.bias_k = torch.cat((self.bias.data, torch.zeros(self.cout_pad - self.out_channels, dtype=w.dtype, device=w.device))).contiguous()
        self.wk = w.permute(0, 2, 3, 1).reshape(w.shape[0], 9 * w.shape[1]).contiguous()
        if need_dgrad:
            self.wd = w.flip(2, 3).permute(1, 2, 3, 0).reshape(w.shape[1], 9 * w.shape[0]).contiguous()

    def forward(self, x):
        raise RuntimeError("fused path: executed inside UNet2DConditionModel.forward_native")

    def __setattr__(self, name, value):
        if name == "forward":  # the reference's LoRAModule.apply_to swaps forward: adopt its module as this layer's adapter (graph.Linear does the same)
            from .adopt import register_foreign_adapter

            register_foreign_adapter(self, value)
            return
        super().__setattr__(name, value)


def _conv3x3_weights_loaded(module, incompatible_keys):
    if not module.__dict__.pop("_sd_touched", True) or module.wk is None:
        return
    module.prepare(need_dgrad=module.wd is not None)


# The reference's adapter discovery goes by CLASS NAME: a child is wrapped when `child.__class__.__name__` is in LINEAR_MODULES / CONV_MODULES
# (toolkit/lora_special.py:29-40, 488-490), and `kernel_size == (1, 1)` separates the 1x1 projections (linear rank) from the 3x3 convolutions
# (network.conv rank).  The two holders therefore carry diffusers' class name, so that the reference's OWN LoRASpecialNetwork finds exactly the
# layers it finds on a diffusers UNet2DConditionModel (state-dict keys and shapes are diffusers' already).
Conv1x1.__name__ = Conv3x3.__name__ = "Conv2d"


class _Norm(nn.Module):
    """GroupNorm / LayerNorm parameters (frozen)."""

    def __init__(self, ch, eps, dtype, device, groups=0):
        super().__init__()
        self.eps, self.groups = eps, groups
        self.weight = nn.Parameter(torch.ones(ch, dtype=dtype, device=device), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(ch, dtype=dtype, device=device), requires_grad=False)
        self._mod = None

    def ln_mod(self):
        """[beta | gamma - 1] as one row: LayerNorm(x) * gamma + beta = LN(x) * (1 + (gamma - 1)) + beta, i.e. the adaLN kernel with a
        constant modulation (one "batch" spanning every row)."""
        if self._mod is None:
            self._mod = torch.cat((self.bias.data, (self.weight.data.float() - 1.0).to(self.bias.dtype)))[None].contiguous()
        return self._mod


class ResnetBlock2D(_Holder):
    """Class names matter: with `network.conv` set the reference's LoRA discovery also targets the Linear / Conv2d children of
    ResnetBlock2D, Downsample2D and Upsample2D (toolkit/kohya_lora.py:751, toolkit/lora_special.py:678-681)."""


class Downsample2D(_Holder):
    pass


class Upsample2D(_Holder):
    pass


def _resnet(cin, cout, temb, dtype, device, groups):
    r = ResnetBlock2D()
    r.norm1 = _Norm(cin, 1e-5, dtype, device, groups)
    r.conv1 = Conv3x3(cin, cout, 1, dtype, device)
    r.time_emb_proj = Linear(temb, cout, True, dtype, device)
    r.norm2 = _Norm(cout, 1e-5, dtype, device, groups)
    r.conv2 = Conv3x3(cout, cout, 1, dtype, device)
    if cin != cout:
        r.conv_shortcut = Conv1x1(cin, cout, dtype, device)
    return r


def _attn(dim, heads, ctx_dim, dtype, device):
    a = _Holder()
    a.heads, a.dim_head = heads, dim // heads
    kv = ctx_dim if ctx_dim is not None else dim
    a.to_q = Linear(dim, dim, False, dtype, device)
    a.to_k = Linear(kv, dim, False, dtype, device)
    a.to_v = Linear(kv, dim, False, dtype, device)
    a.to_out = nn.ModuleList([Linear(dim, dim, True, dtype, device), nn.Identity()])
    return a


class UNet2DConditionOutput(tuple):
    """`(sample,)` with the `.sample` attribute of diffusers' output class."""

    @property
    def sample(self):
        return self[0]


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, ctx_dim, dtype, device):
        super().__init__()
        self.norm1 = _Norm(dim, 1e-5, dtype, device)
        self.attn1 = _attn(dim, heads, None, dtype, device)
        self.norm2 = _Norm(dim, 1e-5, dtype, device)
        self.attn2 = _attn(dim, heads, ctx_dim, dtype, device)
        self.norm3 = _Norm(dim, 1e-5, dtype, device)
        ff, g = _Holder(), _Holder()
        g.proj = Linear(dim, 8 * dim, True, dtype, device)
        ff.net = nn.ModuleList([g, nn.Identity(), Linear(4 * dim, dim, True, dtype, device)])
        self.ff = ff


class Transformer2DModel(nn.Module):
    """Class name matters: the reference's LoRA discovery targets children of `Transformer2DModel` (toolkit/kohya_lora.py:750)."""

    def __init__(self, ch, heads, layers, ctx_dim, linear_proj, dtype, device, groups):
        super().__init__()
        self.use_linear_projection = linear_proj
        self.norm = _Norm(ch, 1e-6, dtype, device, groups)
        mk = (lambda: Linear(ch, ch, True, dtype, device)) if linear_proj else (lambda: Conv1x1(ch, ch, dtype, device))
        self.proj_in = mk()
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(ch, heads, ctx_dim, dtype, device) for _ in range(layers)])
        self.proj_out = mk()


def _block(resnets, attentions=None, sampler=None, kind="down"):
    b = _Holder()
    if attentions is not None:  # diffusers registers attentions before resnets (the order named_modules() walks = adapter creation order)
        b.attentions = nn.ModuleList(attentions)
    b.resnets = nn.ModuleList(resnets)
    if sampler is not None:
        s = Downsample2D() if kind == "down" else Upsample2D()
        s.conv = sampler
        setattr(b, "downsamplers" if kind == "down" else "upsamplers", nn.ModuleList([s]))
    return b


class _Tape:
    """Reverse-mode tape over kernel-level closures.  A tensor `needs` gradient iff an adapter sits upstream of it."""

    def __init__(self, ops, enabled):
        self.ops, self.enabled = ops, enabled
        self.steps = []
        self.need = set()
        self.grads = {}

    def needs(self, *ts):
        return self.enabled and any(t is not None and id(t) in self.need for t in ts)

    def mark(self, t):
        if self.enabled:
            self.need.add(id(t))
        return t

    def record(self, out, fn):
        self.mark(out)
        self.steps.append((out, fn))

    def acc(self, t, g, own):
        """add gradient g to tensor t's gradient; `own` = g's buffer is handed over (may be accumulated into in place)."""
        if id(t) not in self.need:
            return
        cur = self.grads.get(id(t))
        if cur is None:
            self.grads[id(t)] = (g, own, t)
            return
        e, eown, _ = cur
        if not eown:
            n = torch.empty(e.shape, dtype=e.dtype, device=e.device)
            self.ops.ew(2, e, n, a=g)
            self.grads[id(t)] = (n, True, t)
        else:
            self.ops.ew(2, e, e, a=g)

    def backward(self, out, dout):
        self.grads[id(out)] = (dout, False, out)
        for t, fn in reversed(self.steps):
            g = self.grads.pop(id(t), None)
            if g is not None:
                fn(g[0])
        self.steps, self.grads, self.need = [], {}, set()


class UNet2DConditionModel(FusedGraphBase):
    _graph_slots = ("tape", "_pred")  # what one forward leaves for its backward (FusedGraphBase._take_graph_state)

    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 down_block_types=SD15_CONFIG["down_block_types"], up_block_types=SD15_CONFIG["up_block_types"],
                 cross_attention_dim=768, attention_head_dim=8, transformer_layers_per_block=1, use_linear_projection=False,
                 addition_embed_type=None, addition_time_embed_dim=None, projection_class_embeddings_input_dim=None,
                 norm_num_groups=32, dtype=torch.bfloat16, device=None, ops=None):
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
        for i, c in enumerate(block_out_channels):
            d = c // heads[i]
            if d > 256 or d % 8:
                raise NotImplementedError(f"attention head_dim {d} (channels {c} / {heads[i]} heads): head_dim must be a multiple of 8, <= 256")
        G, c0 = norm_num_groups, block_out_channels[0]
        temb = 4 * c0
        self.temb_dim = temb
        self.conv_in = Conv3x3(in_channels, c0, 1, dtype, device, cin_pad=8)
        te = _Holder()
        te.linear_1, te.linear_2 = Linear(c0, temb, True, dtype, device), Linear(temb, temb, True, dtype, device)
        self.time_embedding = te
        if addition_embed_type == "text_time":
            ae = _Holder()
            ae.linear_1 = Linear(projection_class_embeddings_input_dim, temb, True, dtype, device)
            ae.linear_2 = Linear(temb, temb, True, dtype, device)
            self.add_embedding = ae

        def tf(i, ch):
            return Transformer2DModel(ch, heads[i], tl[i], cross_attention_dim, use_linear_projection, dtype, device, G)

        downs, out_ch = [], c0
        for i, t in enumerate(down_block_types):
            in_ch, out_ch = out_ch, block_out_channels[i]
            res = [_resnet(in_ch if j == 0 else out_ch, out_ch, temb, dtype, device, G) for j in range(layers_per_block)]
            att = [tf(i, out_ch) for _ in range(layers_per_block)] if t == "CrossAttnDownBlock2D" else None
            downs.append(_block(res, att, Conv3x3(out_ch, out_ch, 2, dtype, device) if i != n - 1 else None, "down"))
        cm = block_out_channels[-1]
        mid = _block([_resnet(cm, cm, temb, dtype, device, G), _resnet(cm, cm, temb, dtype, device, G)], [tf(n - 1, cm)])
        rev = list(reversed(block_out_channels))
        ups, out_ch = [], rev[0]
        for i, t in enumerate(up_block_types):
            prev, out_ch, in_ch = out_ch, rev[i], rev[min(i + 1, n - 1)]
            L = layers_per_block + 1
            res = [_resnet((prev if j == 0 else out_ch) + (in_ch if j == L - 1 else out_ch), out_ch, temb, dtype, device, G) for j in range(L)]
            att = [tf(n - 1 - i, out_ch) for _ in range(L)] if t == "CrossAttnUpBlock2D" else None
            ups.append(_block(res, att, Conv3x3(out_ch, out_ch, 1, dtype, device) if i != n - 1 else None, "up"))
        # registration order of diffusers' UNet2DConditionModel (down_blocks and up_blocks are created before mid_block): it is the
        # order named_modules() walks, hence the order adapters are created in and consume the RNG
        self.down_blocks = nn.ModuleList(downs)
        self.up_blocks = nn.ModuleList(ups)
        self.mid_block = mid
        self.conv_norm_out = _Norm(c0, 1e-5, dtype, device, G)
        self.conv_out = Conv3x3(c0, out_channels, 1, dtype, device, cout_pad=8)
        self._init_graph(ops, dtype)
        self.tape = None

    # ------------------------------------------------------------------ setup
    def _transformers(self):
        return [m for m in self.modules() if isinstance(m, Transformer2DModel)]

    def _token_linears(self):
        out = []
        for t in self._transformers():
            out += [t.proj_in, t.proj_out]
            for b in t.transformer_blocks:
                out += [b.attn1.to_q, b.attn1.to_k, b.attn1.to_v, b.attn1.to_out[0], b.attn2.to_q, b.attn2.to_out[0], b.ff.net[0].proj, b.ff.net[2]]
        out += [m for m in self.modules() if isinstance(m, Conv1x1) and m not in out]
        return out

    def _dgrad_groups(self):
        return [(b.attn1.to_q, b.attn1.to_k, b.attn1.to_v) for t in self._transformers() for b in t.transformer_blocks]

    def prepare(self):
        super().prepare()
        for m in self.modules():
            if isinstance(m, Conv3x3):
                m.prepare(need_dgrad=m is not self.conv_in)
        return self

    def lora_groups(self):
        """same-input adapters: (q, k, v) of self-attention, (k, v) of cross-attention (both read the text states)."""
        groups = []
        for t in self._transformers():
            for b in t.transformer_blocks:
                for lins in ((b.attn1.to_q, b.attn1.to_k, b.attn1.to_v), (b.attn2.to_k, b.attn2.to_v)):
                    mods = [l.lora for l in lins]
                    if all(m is not None for m in mods):
                        groups.append(mods)
        return groups

    def grad_split_offset(self, network):
        """DP all-reduce pieces: adapters of the up blocks get their final gradients first ('late' piece = arena tail)."""
        offs = [min(m.off_down, m.off_up) for m in network.unet_loras if "up_blocks" in m.lora_name]
        return min(offs) if offs else network.arena_p.numel()

    # ------------------------------------------------------------------ kernel-level graph helpers (each records its backward)
    def _conv_scale(self, lo):
        """fp32 [2 rank_pad] column scale of a conv adapter's lora_down launch: the runtime scale.  The network multiplier (one number or one
        per sample — slider training) is a row factor the implicit-GEMM epilogue cannot apply: aitk_slab_rescale multiplies the rows of T."""
        c = lo.scale
        cached = getattr(lo, "_cs", None)
        if cached is None or cached[0] != c or cached[1].device != self._device():
            cached = (c, torch.full((2 * lo.rank_pad,), c, dtype=torch.float32, device=self._device()))
            lo._cs = cached
        return cached[1]

    def _conv(self, x, conv, B, H, W, res=None, tape=None):
        """y = conv3x3(x) + bias (+ res) (+ LoRA when network.conv wrapped the layer); NHWC contiguous in, [B*Ho*Wo, Cout] out.

        3x3-conv adapter (toolkit/lora_special.py:95-104): lora_down = Conv2d(in, r, 3, stride, padding) is the same implicit-GEMM kernel
        with the 2 rank_pad stacked filters (16-rank blocks [A_hi ; A_lo]) whose fp32 sum leaves the epilogue as the [hi | lo | hi] slab T; lora_up (1x1) is the
        K-slab of the base convolution, exactly like a Linear's.  Backward: dT = c (dy B) and dB by the skinny kernels; dx += the 3x3
        convolution of the dT slab image with the rotated [A_hi | A_hi | A_lo] filter; dA = nine per-tap skinny contractions of the
        zero-framed dT and x grids (a flat shift per tap: wrapped pairs always meet a zero of the frame)."""
        ops = self.ops
        s = conv.stride
        Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
        Mo = B * Ho * Wo
        y = self._new(Mo, conv.cout_pad)
        lo = conv.lora if self._lora_active(conv) else None
        T, kw = None, {}
        # dropout / rank_dropout / module_dropout of the adapter (toolkit/network_mixins.py:198-229), training mode only
        plan = self.network.dropout_plan(lo, M=Mo, rows_per_batch=Ho * Wo, B=B) if lo is not None else None
        if plan == "skip":
            lo = None
        tm, tm_rpb = plan if isinstance(plan, tuple) else (None, 0)
        if lo is not None:
            T = self._new(Mo, 3 * lo.rank_pad)
            ops.conv3x3(x, lo.sh_down_stack, T, B=B, H=H, W=W, stride=s, Ho=Ho, Wo=Wo, split_slab=True, col_scale=self._conv_scale(lo))
            mult, rpb = self._mult(Ho * Wo, B)
            if mult is not None or tm is not None:  # row factors the convolution epilogue cannot apply
                ops.slab_rescale(T, lo.rank_pad, mult=mult, rows_per_batch=rpb, tmask=tm, tmask_rows_per_batch=tm_rpb, M=Mo)
            kw = dict(a2=T, b2=lo.sh_up3)
        ops.conv3x3(x, conv.wk, y, B=B, H=H, W=W, stride=s, Ho=Ho, Wo=Wo, bias=conv.bias_k, flags=EPI_ADD_AUX if res is not None else 0,
                    aux_in=res, **kw)
        if tape.needs(x, res) or T is not None:
            need_x = tape.needs(x)

            def bwd(dy, x=x, res=res, T=T, lo=lo, tm=tm, tm_rpb=tm_rpb):
                if res is not None:
                    tape.acc(res, dy, False)
                if not need_x and T is None:
                    return
                g = dy if dy.is_contiguous() else self._contig(dy)
                dT = None
                if T is not None:
                    rp = lo.rank_pad
                    dT = self._new(Mo, 3 * rp)
                    mult, rpb = self._mult(Ho * Wo, B)
                    ops.lora_down(g, lo.sh_upT, dT, scale=lo.scale, mult=mult, rows_per_batch=rpb, M=Mo, p_lo=lo.sh_upT_lo, split=rp, tmask=tm,
                                  tmask_rows_per_batch=tm_rpb)
                    ops.lora_wgrad(T, g, lo.g_up, transpose_out=True, accumulate=True, M=Mo, split=rp)
                if s == 2:  # zero insertion to the input grid (even H, W: pad 1 / stride 2 maps 2Ho x 2Wo back to H x W)
                    if need_x:
                        z = self._new(B * 4 * Ho * Wo, conv.cout_pad)
                        ops.resample2x(g, z, B=B, H=Ho, W=Wo, mode=2)
                        g = z
                    if dT is not None:
                        zt = self._new(B * 4 * Ho * Wo, dT.shape[1])
                        ops.resample2x(dT, zt, B=B, H=Ho, W=Wo, mode=2)
                        dT = zt
                if dT is not None:  # lora_down.weight.grad [r, Cin, 3, 3]: one skinny launch per tap on the framed grids
                    Wp, Np, cin = W + 2, B * (H + 2) * (W + 2), lo.conv_cin
                    dTp, xp = self._new(Np, dT.shape[1]), self._new(Np, cin)
                    ops.pad_nhwc(dT, dTp, B=B, H=H, W=W)
                    ops.pad_nhwc(x, xp, B=B, H=H, W=W)
                    gflat = lo.g_down.view(-1)
                    for tap in range(9):
                        sh = (tap // 3 - 1) * Wp + (tap % 3 - 1)
                        j0, j1 = max(0, -sh), min(Np, Np - sh)
                        ops.lora_wgrad(dTp[j0:j1], xp[j0 + sh:j1 + sh], gflat[tap:], accumulate=True, M=j1 - j0, split=lo.rank_pad,
                                       out_strides=(9 * cin, 9))
                if not need_x:
                    return
                dx = self._new(B * H * W, conv.cin_pad)
                ops.conv3x3(g, conv.wd, dx, B=B, H=H, W=W)
                if dT is not None:
                    ops.conv3x3(dT, lo.sh_down_dgrad, dx, B=B, H=H, W=W, flags=EPI_ACCUM)
                tape.acc(x, dx, True)

            tape.record(y, bwd)
        return y, Ho, Wo

    def _batch_indicator(self, B, HW, Rb):
        """[B*HW, Rb] one-hot of the sample index (bf16, cached per shape): the S operand that turns aitk_lora_wgrad into a per-sample
        column sum."""
        key = (B, HW, Rb, str(self._device()))
        cache = self.__dict__.setdefault("_ind_cache", {})
        ind = cache.get(key)
        if ind is None:
            ind = torch.zeros(B * HW, Rb, dtype=self.dt, device=self._device())
            ar = torch.arange(B, device=ind.device)
            ind.view(B, HW, Rb)[ar, :, ar] = 1
            cache[key] = ind
        return ind

    def _contig(self, t):
        c = self._new(t.shape[0], t.shape[1])
        self.ops.copy_rows(c, t)
        return c

    def _gn(self, x, norm, B, HW, silu, tape):
        ops = self.ops
        y = self._new(x.shape[0], x.shape[1])
        rec = tape.needs(x)
        stats = self._new(B * norm.groups * 2, dtype=torch.float32) if rec else None
        ops.groupnorm(x, norm.weight, norm.bias, y, B=B, HW=HW, G=norm.groups, eps=norm.eps, silu=silu, stats_out=stats)
        if rec:
            def bwd(dy, x=x):
                dx = self._new(x.shape[0], x.shape[1])
                ops.groupnorm_bwd(dy, x, norm.weight, norm.bias, stats, dx, B=B, HW=HW, G=norm.groups, silu=silu)
                tape.acc(x, dx, True)

            tape.record(y, bwd)
        return y

    def _ln(self, x, norm, tape):
        ops = self.ops
        M, Cc = x.shape
        mod = norm.ln_mod()
        y = self._new(M, Cc)
        rec = tape.needs(x)
        mean = self._new(M, dtype=torch.float32) if rec else None
        rstd = self._new(M, dtype=torch.float32) if rec else None
        ops.ln_mod_fwd(x, mod[:, :Cc], mod[:, Cc:], y, rows_per_batch=M, mean=mean, rstd=rstd, eps=norm.eps)
        if rec:
            def bwd(dy, x=x):
                dx = self._new(M, Cc)
                ops.ln_mod_bwd(dy, x, mean, rstd, mod[:, Cc:], dx, B=1, S=M)
                tape.acc(x, dx, True)

            tape.record(y, bwd)
        return y

    def _lin(self, x, lin, *, M, rpb, B, tape, res=None, T=None, need_dx=True):
        """y = x W^T + b (+ LoRA) (+ res) through the LoRA-fused GEMM; backward = adapter gradients + data gradient."""
        y = self._new(M, lin.out_features)
        T = self._lin_fwd(lin, x, y, M=M, rows_per_batch=rpb, B=B, flags=EPI_ADD_AUX if res is not None else 0, aux_in=res, T=T)
        if tape.needs(x, res) or T is not None:
            want_dx = need_dx and tape.needs(x)

            def bwd(dy, x=x, res=res, T=T):
                if res is not None:
                    tape.acc(res, dy, False)
                g = dy if dy.stride(1) == 1 else self._contig(dy)
                if want_dx:
                    dx = self._new(M, lin.in_features)
                    self._lin_bwd(lin, g, T, x, dx, M=M, rows_per_batch=rpb, B=B)
                    tape.acc(x, dx, True)
                elif T is not None:
                    self._lora_grads(lin, self._dora_dz(lin, g, M), T, x, M=M, rows_per_batch=rpb, B=B)

            tape.record(y, bwd)
        return y

    def _attention(self, xq, ctx, a, *, B, Sq, Skv, tape, res):
        """diffusers Attention (AttnProcessor2_0): to_q / to_k / to_v (+LoRA), SDPA over `heads` of dim_head, to_out[0] (+LoRA) + res.
        ctx is None for self-attention (keys / values from xq).  Heads of 40 / 80 columns are zero-padded to 128 for the flash kernels, 64-wide
        heads are read in place."""
        ops, H, d = self.ops, a.heads, a.dim_head
        Mq, Mk = B * Sq, B * Skv
        self_attn = ctx is None
        src = xq if self_attn else ctx
        if self_attn:
            Tg = self._group_down((a.to_q, a.to_k, a.to_v), xq, M=Mq, rows_per_batch=Sq, B=B)
        else:
            Tg = self._group_down((a.to_k, a.to_v), ctx, M=Mk, rows_per_batch=Skv, B=B)
        dim = H * d
        q = self._new(Mq, dim)
        k = self._new(Mk, dim)
        v = self._new(Mk, dim)
        Tq = self._lin_fwd(a.to_q, xq, q, M=Mq, rows_per_batch=Sq, B=B, T=Tg.get(id(a.to_q)))
        Tk = self._lin_fwd(a.to_k, src, k, M=Mk, rows_per_batch=Skv, B=B, T=Tg.get(id(a.to_k)))
        Tv = self._lin_fwd(a.to_v, src, v, M=Mk, rows_per_batch=Skv, B=B, T=Tg.get(id(a.to_v)))
        scale = 1.0 / math.sqrt(d)
        lse = self._new(B, H, Sq, dtype=torch.float32)
        kvn = 0 if self_attn else Skv
        small = d > PAD_D  # SD1.5's head_dim 160: generic fp32 kernels on the unpadded heads (tiny sequences)
        native = d in (64, 96)  # whole contraction steps and output blocks of an exact kernel instantiation: no padding needed
        if small:
            qp, kp, vp = q, k, v
            o = op_ = self._new(Mq, dim)
            ops.attn_small_fwd(q, k, v, o, lse, B=B, H=H, S=Sq, D=d, scale=scale, Skv=kvn)
        elif native:  # SDXL's 64-wide heads: the flash kernels read q / k / v where the projections wrote them (AitkAttnArgs.hstride)
            qp, kp, vp = q, k, v
            o = op_ = self._new(Mq, dim)
            ops.attn_fwd(q, k, v, o, lse, B=B, H=H, S=Sq, scale=scale, Skv=kvn, dv=d, hstride=d)
        else:
            if d != PAD_D:  # head_dim 40 / 80: zero-padded to the flash kernels' 128 columns (exact)
                qp, kp, vp = self._new(Mq, H * PAD_D), self._new(Mk, H * PAD_D), self._new(Mk, H * PAD_D)
                for s_, d_ in ((q, qp), (k, kp), (v, vp)):
                    ops.copy_heads(s_, d_, H=H, d_src=d, d_dst=PAD_D)
            else:
                qp, kp, vp = q, k, v
            op_ = self._new(Mq, H * PAD_D)
            ops.attn_fwd(qp, kp, vp, op_, lse, B=B, H=H, S=Sq, scale=scale, Skv=kvn, dv=d if d != PAD_D else 0)
            if d != PAD_D:
                o = self._new(Mq, dim)
                ops.copy_heads(op_, o, H=H, d_src=PAD_D, d_dst=d)
            else:
                o = op_
        y = self._new(Mq, dim)
        To = self._lin_fwd(a.to_out[0], o, y, M=Mq, rows_per_batch=Sq, B=B, flags=EPI_ADD_AUX, aux_in=res)
        need_x = tape.needs(xq)
        if not (tape.enabled and (need_x or tape.needs(res) or any(t is not None for t in (Tq, Tk, Tv, To)))):
            return y

        def bwd(dy):
            tape.acc(res, dy, False)
            g = dy if dy.stride(1) == 1 else self._contig(dy)
            do = self._new(Mq, dim)
            self._lin_bwd(a.to_out[0], g, To, o, do, M=Mq, rows_per_batch=Sq, B=B)
            if self_attn:
                dgrp = self._new(Mq, 3 * dim)  # d[q | k | v] side by side: one K-concatenated data-gradient GEMM for the group (graph._group_bwd)
                _qkv_new = lambda: (dgrp[:, :dim], dgrp[:, dim:2 * dim], dgrp[:, 2 * dim:])  # noqa: E731
            else:
                _qkv_new = lambda: (self._new(Mq, dim), self._new(Mk, dim), self._new(Mk, dim))  # noqa: E731
            if small:
                dq, dk, dv = _qkv_new()
                ops.attn_small_bwd(q, k, v, o, lse, do, dq, dk, dv, B=B, H=H, S=Sq, D=d, scale=scale, Skv=kvn)
            elif native:
                dq, dk, dv = _qkv_new()
                ops.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, B=B, H=H, S=Sq, scale=scale, Skv=kvn, dvalid=d, hstride=d)
            else:
                if d != PAD_D:
                    dop = self._new(Mq, H * PAD_D)
                    ops.copy_heads(do, dop, H=H, d_src=d, d_dst=PAD_D)
                else:
                    dop = do
                dqp, dkp, dvp = (self._new(Mq, H * PAD_D), self._new(Mk, H * PAD_D), self._new(Mk, H * PAD_D)) if d != PAD_D else _qkv_new()
                ops.attn_bwd(qp, kp, vp, op_, lse, dop, dqp, dkp, dvp, B=B, H=H, S=Sq, scale=scale, Skv=kvn, dvalid=d if d != PAD_D else 0)
                if d != PAD_D:
                    dq, dk, dv = _qkv_new()
                    for s_, d_ in ((dqp, dq), (dkp, dk), (dvp, dv)):
                        ops.copy_heads(s_, d_, H=H, d_src=PAD_D, d_dst=d)
                else:
                    dq, dk, dv = dqp, dkp, dvp
            if self_attn:
                dx = self._new(Mq, a.to_q.in_features)
                self._group_bwd((a.to_q, a.to_k, a.to_v), [dq, dk, dv], [Tq, Tk, Tv], xq, dx, M=Mq, rows_per_batch=Sq, B=B)
                tape.acc(xq, dx, True)
            else:
                if need_x:
                    dx = self._new(Mq, a.to_q.in_features)
                    self._lin_bwd(a.to_q, dq, Tq, xq, dx, M=Mq, rows_per_batch=Sq, B=B)
                    tape.acc(xq, dx, True)
                elif Tq is not None:
                    self._lora_grads(a.to_q, self._dora_dz(a.to_q, dq, Mq), Tq, xq, M=Mq, rows_per_batch=Sq, B=B)
                self._wgrad_only((a.to_k, a.to_v), [dk, dv], [Tk, Tv], ctx, Mk, Skv, B)  # text states: no data gradient

        tape.record(y, bwd)
        return y

    def _wgrad_only(self, lins, dys, Ts, x_in, M, rpb, B):
        """adapter weight gradients of same-input Linears whose input needs no data gradient (cross-attention k / v)."""
        if all(t is None for t in Ts):
            return
        grp = getattr(lins[0].lora, "group", None) if all(t is not None for t in Ts) else None
        if grp is not None and [id(m) for m in grp["mods"]] != [id(l.lora) for l in lins]:
            grp = None
        dTcat = self._new(M, 3 * grp["R"]) if grp is not None else None
        for lin, dy, T in zip(lins, dys, Ts):
            dy = self._dora_dz(lin, dy, M)
            dT_out = None
            if grp is not None:
                c0 = 3 * grp["col"][id(lin.lora)]
                dT_out = dTcat[:, c0:c0 + 3 * lin.lora.rank_pad]
            self._lora_grads(lin, dy, T, x_in, M=M, rows_per_batch=rpb, B=B, dT_out=dT_out)
        if grp is not None:
            self.ops.lora_wgrad(dTcat, x_in, grp["g_down"], accumulate=True, M=M, split=grp["rp"])

    def _geglu(self, hg, tape):
        ops = self.ops
        M, F2 = hg.shape
        y = self._new(M, F2 // 2)
        ops.geglu_fwd(hg, y)
        if tape.needs(hg):
            def bwd(dy, hg=hg):
                g = dy if dy.stride(1) == 1 else self._contig(dy)
                dhg = self._new(M, F2)
                ops.geglu_bwd(g, hg, dhg)
                tape.acc(hg, dhg, True)

            tape.record(y, bwd)
        return y

    def _resnet_fwd(self, x, r, temb_act, B, H, W, tape):
        """diffusers ResnetBlock2D: conv2(silu(norm2(conv1(silu(norm1(x))) + time_emb_proj(silu(temb))))) + shortcut(x)."""
        ops = self.ops
        HW = H * W
        h = self._gn(x, r.norm1, B, HW, True, tape)
        h, _, _ = self._conv(h, r.conv1, B, H, W, tape=tape)
        tproj = r.time_emb_proj
        if self._lora_active(tproj):  # network.conv also wraps the Linear children of ResnetBlock2D (toolkit/kohya_lora.py:751)
            tp, Tt = self._ada_fwd(tproj, temb_act, B)
        else:
            tp, Tt = self._new(B, r.conv1.out_channels), None
            ops.gemv_nt(temb_act, tproj.weight, tp, bias=tproj.bias)
        h2 = self._new(h.shape[0], h.shape[1])
        ops.ew(2, h, h2, a=tp, a_rows_per_batch=HW)
        if tape.needs(h) or Tt is not None:
            def bwd_t(dy, h=h, Tt=Tt):
                tape.acc(h, dy, False)
                if Tt is None:
                    return
                # d(time_emb_proj output)[b] = sum over the pixels of sample b of dy: one skinny contraction against the batch-indicator
                # matrix (out[b][c] = sum_m [m in sample b] dy[m][c]), then the small-batch adapter backward of the adaLN projections
                g = dy if dy.is_contiguous() else self._contig(dy)
                dtp = self._new(B, g.shape[1])
                for b0 in range(0, B, 64):  # the skinny kernel contracts against at most 64 indicator columns per launch
                    nb = min(64, B - b0)
                    Rb = (nb + 15) // 16 * 16
                    ind = self._batch_indicator(nb, HW, Rb)
                    part = torch.zeros(Rb, g.shape[1], dtype=torch.float32, device=g.device)
                    ops.lora_wgrad(ind, g[b0 * HW:(b0 + nb) * HW], part, M=nb * HW)
                    ops.colsum_finish(part, 1, nb, 1, g.shape[1], dtp[b0:b0 + nb])
                self._ada_bwd(tproj, dtp, Tt, temb_act, B)

            tape.record(h2, bwd_t)
        h2 = self._gn(h2, r.norm2, B, HW, True, tape)
        sc = x
        if hasattr(r, "conv_shortcut"):
            sc = self._lin(x, r.conv_shortcut, M=B * HW, rpb=HW, B=B, tape=tape)
        y, _, _ = self._conv(h2, r.conv2, B, H, W, res=sc, tape=tape)
        return y

    def _transformer_fwd(self, x, t, ctx, B, H, W, Skv, tape):
        """diffusers Transformer2DModel: GroupNorm -> proj_in -> BasicTransformerBlocks -> proj_out -> + input."""
        HW, M = H * W, B * H * W
        h = self._gn(x, t.norm, B, HW, False, tape)
        h = self._lin(h, t.proj_in, M=M, rpb=HW, B=B, tape=tape)
        for b in t.transformer_blocks:
            n1 = self._ln(h, b.norm1, tape)
            h = self._attention(n1, None, b.attn1, B=B, Sq=HW, Skv=HW, tape=tape, res=h)
            n2 = self._ln(h, b.norm2, tape)
            h = self._attention(n2, ctx, b.attn2, B=B, Sq=HW, Skv=Skv, tape=tape, res=h)
            n3 = self._ln(h, b.norm3, tape)
            hg = self._lin(n3, b.ff.net[0].proj, M=M, rpb=HW, B=B, tape=tape)
            ge = self._geglu(hg, tape)
            h = self._lin(ge, b.ff.net[2], M=M, rpb=HW, B=B, tape=tape, res=h)
        return self._lin(h, t.proj_out, M=M, rpb=HW, B=B, tape=tape, res=x)

    def _cat(self, a, s, tape):
        """channel concat [a | s] (NHWC: column concat)."""
        M, C1, C2 = a.shape[0], a.shape[1], s.shape[1]
        y = self._new(M, C1 + C2)
        self.ops.copy_rows(y[:, :C1], a)
        self.ops.copy_rows(y[:, C1:], s)
        if tape.needs(a, s):
            def bwd(dy, a=a, s=s):
                tape.acc(a, dy[:, :C1], False)
                tape.acc(s, dy[:, C1:], False)

            tape.record(y, bwd)
        return y

    def _upsample(self, x, conv, B, H, W, tape):
        ops = self.ops
        u = self._new(B * 4 * H * W, x.shape[1])
        ops.resample2x(x, u, B=B, H=H, W=W, mode=0)
        if tape.needs(x):
            def bwd(dy, x=x):
                g = dy if dy.is_contiguous() else self._contig(dy)
                dx = self._new(B * H * W, x.shape[1])
                ops.resample2x(g, dx, B=B, H=2 * H, W=2 * W, mode=1)
                tape.acc(x, dx, True)

            tape.record(u, bwd)
        y, _, _ = self._conv(u, conv, B, 2 * H, 2 * W, tape=tape)
        return y

    def _embed(self, proj_in, emb):
        """TimestepEmbedding: linear_2(silu(linear_1(x)))  (small-batch projections: weight-streaming GEMV, 8 rows per launch)."""
        ops = self.ops
        B = proj_in.shape[0]
        h1 = self._new(B, emb.linear_1.out_features)
        ops.gemv_nt(proj_in, emb.linear_1.weight, h1, bias=emb.linear_1.bias)
        ops.ew(0, h1, h1)
        out = self._new(B, emb.linear_2.out_features)
        ops.gemv_nt(h1, emb.linear_2.weight, out, bias=emb.linear_2.bias)
        return out

    # ------------------------------------------------------------------ forward / backward
    def forward_native(self, sample_nhwc, timestep, encoder_hidden_states, added_cond_kwargs=None, *, B, H, W, save_for_backward=True):
        """sample_nhwc [B*H*W, 8] (4 latent channels + zero padding), timestep [B] fp32 (0..999), encoder_hidden_states [B, 77, D];
        SDXL: added_cond_kwargs = {text_embeds [B,1280], time_ids [B,6]}.  Returns the prediction NHWC [B*H*W, out_channels]."""
        ops, dt, cfg = self.ops, self.dt, self.config
        if not self._prepared:
            self.prepare()
        tape = _Tape(ops, save_for_backward and self.network is not None and self.network.is_active)
        c0 = cfg["block_out_channels"][0]
        # ---- time embedding (no trainable ancestor)
        tproj = self._new(B, c0)
        ops.timestep_embed(timestep.float().contiguous(), tproj)
        emb = self._embed(tproj, self.time_embedding)
        if cfg["addition_embed_type"] == "text_time":
            tid = added_cond_kwargs["time_ids"].float().reshape(-1).contiguous()
            tp = self._new(tid.numel(), cfg["addition_time_embed_dim"])
            ops.timestep_embed(tid, tp)
            add = torch.cat((added_cond_kwargs["text_embeds"].to(dt), tp.reshape(B, -1)), dim=-1).contiguous()
            aug = self._embed(add, self.add_embedding)
            ops.ew(2, emb, emb, a=aug)
        temb_act = self._new(B, self.temb_dim)
        ops.ew(0, emb, temb_act)
        Skv = encoder_hidden_states.shape[1]
        ctx = encoder_hidden_states.to(dt).reshape(B * Skv, -1).contiguous()

        x, h, w = self._conv(sample_nhwc, self.conv_in, B, H, W, tape=tape)
        skips = [(x, h, w)]
        for blk in self.down_blocks:
            for i, r in enumerate(blk.resnets):
                x = self._resnet_fwd(x, r, temb_act, B, h, w, tape)
                if hasattr(blk, "attentions"):
                    x = self._transformer_fwd(x, blk.attentions[i], ctx, B, h, w, Skv, tape)
                skips.append((x, h, w))
            if hasattr(blk, "downsamplers"):
                x, h, w = self._conv(x, blk.downsamplers[0].conv, B, h, w, tape=tape)
                skips.append((x, h, w))
        mb = self.mid_block
        x = self._resnet_fwd(x, mb.resnets[0], temb_act, B, h, w, tape)
        x = self._transformer_fwd(x, mb.attentions[0], ctx, B, h, w, Skv, tape)
        x = self._resnet_fwd(x, mb.resnets[1], temb_act, B, h, w, tape)
        for blk in self.up_blocks:
            for i, r in enumerate(blk.resnets):
                s, sh, sw = skips.pop()
                assert (sh, sw) == (h, w)
                x = self._resnet_fwd(self._cat(x, s, tape), r, temb_act, B, h, w, tape)
                if hasattr(blk, "attentions"):
                    x = self._transformer_fwd(x, blk.attentions[i], ctx, B, h, w, Skv, tape)
            if hasattr(blk, "upsamplers"):
                x = self._upsample(x, blk.upsamplers[0].conv, B, h, w, tape)
                h, w = 2 * h, 2 * w
        x = self._gn(x, self.conv_norm_out, B, h * w, True, tape)
        pred8, _, _ = self._conv(x, self.conv_out, B, h, w, tape=tape)  # [M, 8]: the 4 prediction channels + zero padding
        Co = cfg["out_channels"]
        pred = self._new(pred8.shape[0], Co)
        ops.copy_rows(pred, pred8[:, :Co])
        self.tape = tape if tape.enabled else None
        self._pred = pred8
        return pred

    def backward_native(self, dpred):
        """dpred NHWC [B*H*W, out_channels]: accumulates every adapter gradient into network.arena_g; frees the tape."""
        tape = self.tape
        assert tape is not None, "forward_native(save_for_backward=True) inside `with network:` must run first"
        Co = self.config["out_channels"]
        d8 = torch.zeros(self._pred.shape, dtype=self.dt, device=self._pred.device)
        self.ops.copy_rows(d8[:, :Co], dpred.to(self.dt).reshape(-1, Co).contiguous())
        ops = self.ops
        wdefer = getattr(ops, "wgrad_defer_begin", None) is not None and ops.wgrad_defer_begin(d8.device)  # weight-gradient finishes eight at a time (flux.py)
        try:
            tape.backward(self._pred, d8)
        finally:
            if wdefer:
                ops.wgrad_defer_end()
        self.tape = self._pred = None
        if self.grad_ready_hook is not None:
            self.grad_ready_hook("single")
            self.grad_ready_hook("double")

    # diffusers-signature call used by the plug-in / reference-style trainers (NCHW in / out; autograd bridge like flux.py).  The reference's
    # legacy UNet call sites read `.sample` from the result (toolkit/stable_diffusion_model.py:2049-2055, 2260-2265): return_dict=True (the
    # diffusers default) returns a tuple that also carries `.sample`; return_dict=False the plain tuple.
    def forward(self, sample, timestep, encoder_hidden_states, added_cond_kwargs=None, return_dict=True, **kwargs):
        self._resolve_network()  # adopts / syncs a network the reference built itself (adopt.py)
        B, Cc, H, W = sample.shape
        x = torch.zeros(B * H * W, 8, dtype=self.dt, device=sample.device)
        x[:, :Cc] = sample.to(self.dt).permute(0, 2, 3, 1).reshape(B * H * W, Cc)
        ts = timestep.reshape(-1).float().expand(B).contiguous() if timestep.numel() == 1 else timestep.float()
        pred = self.forward_native(x, ts, encoder_hidden_states, added_cond_kwargs, B=B, H=H, W=W, save_for_backward=torch.is_grad_enabled())
        if torch.is_grad_enabled() and self.network is not None and self.network.is_active:
            from .flux import _FluxGraphFn

            pred = _FluxGraphFn.apply(pred.detach(), self, self.network.arena_p.requires_grad_(True))  # detach: the explicit graph is the only history (a torch-backed kernel table would otherwise leave autograd history of its own on pred)
        out = pred.reshape(B, H, W, -1).permute(0, 3, 1, 2)
        return UNet2DConditionOutput((out,)) if return_dict else (out,)
