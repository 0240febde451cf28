"""ORACLE (test infrastructure): plain-PyTorch restatement of every kernel in ai-toolkit_amd/ops.py, same signatures.

Used only by tests/ (and __graft_entry__.smoke / bench cpu_baseline as the checker) in two ways:
  1. per-kernel parity on the GPU: HIP kernel output vs the function of the same name here;
  2. host-logic parity on the CPU: ai_toolkit_amd.flux.FluxTransformer2DModel driven by THIS table in fp32 must
     reproduce autograd of oracle/flux_ref.py, which proves the hand-written backward graph independent of any GPU.
Each function cites the reference line it restates.  Math is done in fp32 (fp64 where noted) and cast to the output
dtype, i.e. one rounding per kernel output like the HIP kernels.
"""
import math

import torch
import torch.nn.functional as F

EPI_BIAS, EPI_ACCUM, EPI_GELU, EPI_DGELU, EPI_GATE_RES, EPI_BIAS_ROW, EPI_ADD_AUX = 1, 2, 4, 8, 16, 32, 64
EPI_SPLIT_SLAB = 256


def _seg_view(t, seg, M):
    """logical [M, C] tensor for a (seg_rows, seg_stride)-segmented 2-D view `t` (first segment)."""
    if seg is None:
        return t[:M]
    seg_rows, seg_stride = seg
    nseg = (M + seg_rows - 1) // seg_rows
    v = torch.as_strided(t, (nseg, seg_rows, t.shape[1]), (seg_stride, t.stride(0), 1))
    return v.reshape(nseg * seg_rows, t.shape[1])[:M]


def _seg_store(t, seg, M, value):
    if seg is None:
        t[:M].copy_(value.to(t.dtype))
        return
    seg_rows, seg_stride = seg
    nseg = M // seg_rows
    assert nseg * seg_rows == M
    v = torch.as_strided(t, (nseg, seg_rows, t.shape[1]), (seg_stride, t.stride(0), 1))
    v.copy_(value.view(nseg, seg_rows, -1).to(t.dtype))


def rows_per_block():
    return 16


EMIT_T_ROW_TILE = 256  # AITK_EPI_EMIT_T: whole 256-row tiles, as on the HIP kernel (the graphs ask the table); tests/test_emit_t_cpu.py lowers it to 1 so that tiny CPU models walk the route
_ws = {}


def workspace(nbytes, device, tag="ws"):
    """grow-only fp32 scratch, like ai_toolkit_amd.ops.workspace"""
    key = (tag, str(device))
    n = (nbytes + 3) // 4
    if key not in _ws or _ws[key].numel() < n:
        _ws[key] = torch.empty(n, dtype=torch.float32, device=device)
    return _ws[key]


def quant_rows_fp8(x, q, row_scale, *, col_mul=None, x_seg=None, M=None):
    """Per-token dynamic e4m3 quantisation (the A operand of the W8A8 GEMM, b_scale_mode 3): row_scale = amax / 448 (IEEE division),
    q = e4m3(x * col_mul * (1 / row_scale)), round-to-nearest-even — the standard per-token activation recipe of fp8 training stacks;
    the reference itself has no activation quantisation (toolkit/util/quantize.py:43-75 is weight-only), see DESIGN.md."""
    if M is None:
        M = x.shape[0]
    v = _seg_view(x, x_seg, M).float()
    if col_mul is not None:
        v = v * col_mul.float()[None, :]
    sc = (v.abs().amax(dim=1) / torch.full((M,), 448.0, device=v.device)).clamp_min(1.17549435e-38)
    inv = torch.ones_like(sc) / sc
    q[:M].copy_((v * inv[:, None]).to(torch.float8_e4m3fn).view(torch.uint8))
    row_scale[:M].copy_(sc)
    return q, row_scale


def gemm_nt(a, b, out, *, bias=None, a2=None, b2=None, flags=0, aux_out=None, aux_in=None, gate=None, gate_rows=0,
            a_seg=None, c_seg=None, M=None, stage_mode=None, tile_mode=None, b_scale=None, b_scale_mode=0, col_scale=None, a_scale=None, emit_t=None):
    """org Linear + LoRA up-projection + epilogue (toolkit/network_mixins.py:304-342).  b_scale: weight-only fp8 base,
    dequantised as (fp8 * scale) rounded to the activation dtype (quanto / torchao weight-only semantics)."""
    if M is None:
        M = a.shape[0]
    act_dt = torch.bfloat16 if b_scale_mode == 3 else a.dtype  # type the branch output y of GATE_RES is rounded to
    if b_scale_mode == 3:  # W8A8: exact products of e4m3 values accumulated in fp32, scaled by the operands' quantisation scales
        A = a[:M].view(torch.float8_e4m3fn).float() * a_scale[:M, None]
        Bq = b.view(torch.float8_e4m3fn).float()
        if b_scale is not None:
            Bq = Bq * b_scale[:, None]
        a, b, b_scale = A, Bq, None
    A = _seg_view(a, a_seg, M).float()
    if b_scale is not None:
        bq = b.view(torch.float8_e4m3fn).float()
        bq = bq * (b_scale[:, None] if b_scale_mode == 1 else b_scale[None, :])
        b = bq.to(a.dtype)
    v = A @ b.float().t()
    if a2 is not None:
        v = v + a2[:M].float() @ b2.float().t()
    if col_scale is not None:  # DoRA: magnitude / ||W + dW||_row applied to the un-biased product
        v = v * col_scale.float()[None, :]
    if bias is not None:
        v = v + (bias.float()[:, None] if flags & EPI_BIAS_ROW else bias.float())
    if flags & EPI_ADD_AUX:
        v = v + aux_in[:M].float()
    if flags & EPI_ACCUM:
        v = v + _seg_view(out, c_seg, M).float()
    if flags & EPI_GELU:
        aux_out[:M].copy_(v.to(aux_out.dtype))
        v = F.gelu(aux_out[:M].float(), approximate="tanh")
        if emit_t is not None:  # AITK_EPI_EMIT_T: per 256-column tile, the stored GELU values against the consumer's lora_down rows (hi + lo shadows)
            p_hi, p_lo, partial, tile0 = emit_t
            hq = v.to(out.dtype).float()
            N = b.shape[0]
            P = (p_hi.float() + p_lo.float())[:, :N]
            for t in range(N // 256):
                partial[tile0 + t, :M].copy_(hq[:, t * 256:(t + 1) * 256] @ P[:, t * 256:(t + 1) * 256].t())
    if flags & EPI_DGELU:
        u = aux_in[:M].float().requires_grad_(True)
        with torch.enable_grad():
            F.gelu(u, approximate="tanh").sum().backward()
        v = v * u.grad
    if flags & EPI_GATE_RES:
        y = v.to(act_dt)  # the branch output is rounded to the activation type; `out` (the residual stream) may be fp32 (precision="high")
        if aux_out is not None:
            aux_out[:M].copy_(y)
        g = gate.float().repeat_interleave(gate_rows, 0)[:M]
        v = aux_in[:M].float() + g * y.float()
    _seg_store(out, c_seg, M, v)
    return out


def _split_cols(R, rp):
    """column indices of (hi, lo, hi-again) for the [M, 3R] K-slab layout: rank block b at columns 3*b*rp."""
    r = torch.arange(R)
    base = (r // rp) * 3 * rp + r % rp
    return base, base + rp, base + 2 * rp


def lora_down(x, pmat, out, *, scale=1.0, mult=None, rows_per_batch=0, x_seg=None, M=None, p_lo=None, split=0, tmask=None,
              tmask_rows_per_batch=0):
    """lora_down(x.float()) * scale * multiplier (toolkit/network_mixins.py:197-239, 309-318).  p_lo: second half of a split
    (hi + lo) projection; split: write the fp32 result as the [hi | lo | hi] bf16-pair slab layout (hi = round(t), lo = round(t - hi))."""
    if M is None:
        M = x.shape[0]
    P = pmat.float() if p_lo is None else pmat.float() + p_lo.float()
    v = _seg_view(x, x_seg, M).float() @ P.t() * scale
    if mult is not None:
        v = v * mult.repeat_interleave(rows_per_batch)[:M, None]
    if tmask is not None:  # dropout / rank_dropout on lx = lora_down(x) (toolkit/network_mixins.py:212-228), pre-scaled by 1 / keep
        v = v * (tmask.repeat_interleave(tmask_rows_per_batch, 0)[:M] if tmask_rows_per_batch else tmask[:M])
    if not split:
        out[:M].copy_(v.to(out.dtype))
        return out
    hi = v.to(out.dtype)
    lo = (v - hi.float()).to(out.dtype)
    c_hi, c_lo, c_hi2 = _split_cols(pmat.shape[0], split)
    out[:M, c_hi] = hi
    out[:M, c_lo] = lo
    out[:M, c_hi2] = hi
    return out


def lora_down_raw(x, pmat, raw, *, p_lo=None, x_seg=None, M=None):
    """un-scaled fp32 x @ (pmat + p_lo)^T: one tile of a partial-sum slab (include/aitk_mi355.h aitk_lora_down_raw)."""
    if M is None:
        M = x.shape[0]
    P = pmat.float() if p_lo is None else pmat.float() + p_lo.float()
    raw[:M].copy_(_seg_view(x, x_seg, M).float() @ P.t())
    return raw


def lora_t_finish(partial, ntiles, out, *, scale=1.0, mult=None, rows_per_batch=0, split=0, tmask=None, tmask_rows_per_batch=0, M=None):
    """T from the sum of the partial tiles (tile order), written as lora_down writes it (toolkit/network_mixins.py:309-318 on the summed product)."""
    M = partial.shape[1] if M is None else M
    R = partial.shape[2]
    v = torch.zeros(M, R, dtype=torch.float32, device=partial.device)
    for t in range(ntiles):
        v = v + partial[t, :M]
    v = v * scale
    if mult is not None:
        v = v * mult.repeat_interleave(rows_per_batch)[:M, None]
    if tmask is not None:
        v = v * (tmask.repeat_interleave(tmask_rows_per_batch, 0)[:M] if tmask_rows_per_batch else tmask[:M])
    if not split:
        out[:M].copy_(v.to(out.dtype))
        return out
    hi = v.to(out.dtype)
    lo = (v - hi.float()).to(out.dtype)
    c_hi, c_lo, c_hi2 = _split_cols(R, split)
    out[:M, c_hi] = hi
    out[:M, c_lo] = lo
    out[:M, c_hi2] = hi
    return out


def slab_rescale(T, rp, *, mult=None, rows_per_batch=0, tmask=None, tmask_rows_per_batch=0, M=None):
    """LoRAModule._call_forward's dropout / rank_dropout masks and forward()'s per-sample multiplier on the rank-space activation of a
    conv adapter (toolkit/network_mixins.py:211-229, 235-239), applied to the fp32 value of the [hi | lo | hi] slab."""
    M = T.shape[0] if M is None else M
    v = T[:M, :rp].float() + T[:M, rp:2 * rp].float()
    rows = torch.arange(M, device=T.device)
    if mult is not None:
        v = v * mult.float()[rows // rows_per_batch][:, None]
    if tmask is not None:
        v = v * (tmask.float()[rows // tmask_rows_per_batch] if tmask_rows_per_batch > 0 else tmask.float()[:M])
    hi = v.to(T.dtype)
    lo = (v - hi.float()).to(T.dtype)
    T[:M, :3 * rp] = torch.cat((hi, lo, hi), dim=1)
    return T


def lora_wgrad(s, g, out, *, transpose_out=False, accumulate=False, g_seg=None, M=None, split=0, out_strides=None, g2=None, g2_act=None):
    """autograd of lora_down / lora_up weights (split: s is the slab layout, read as hi + lo); out_strides: strided destination
    (one tap of a conv adapter's [r, Cin, 3, 3] gradient).  g2: the layer input was [g | act(g2)] with act = F.gelu(approximate="tanh") of the
    pre-activation rounded to the activation dtype, exactly what the forward pass fed the layer (diffusers FeedForward / FluxSingleTransformerBlock)."""
    if M is None:
        M = s.shape[0]
    if g2 is not None:
        h = F.gelu(g2[:M].float(), approximate="tanh").to(g2.dtype) if g2_act == "gelu" else g2[:M]
        g = h if g is None else torch.cat((g[:M], h), dim=1)
        g_seg = None
    if split:
        c_hi, c_lo, _ = _split_cols(s.shape[1] // 3, split)
        s = s[:M, c_hi].float() + s[:M, c_lo].float()
    v = s[:M].float().t() @ _seg_view(g, g_seg, M).float()
    if out_strides is not None:
        dst = torch.as_strided(out, v.shape, out_strides)
        dst.add_(v) if accumulate else dst.copy_(v)
        return out
    if transpose_out:
        v = v.t()
    if accumulate:
        out.add_(v)
    else:
        out.copy_(v)
    return out


def ln_mod_fwd(x, shift, scale, out, *, rows_per_batch, mean=None, rstd=None, eps=1e-6):
    xf = x.float()
    mu = xf.mean(-1, keepdim=True)
    var = ((xf - mu) ** 2).mean(-1, keepdim=True)
    rs = torch.rsqrt(var + eps)
    M = x.shape[0]
    sc = scale.float().repeat_interleave(rows_per_batch, 0)[:M]
    sh = shift.float().repeat_interleave(rows_per_batch, 0)[:M]
    out.copy_(((xf - mu) * rs * (1 + sc) + sh).to(out.dtype))
    if mean is not None:
        mean.copy_(mu[:, 0])
        rstd.copy_(rs[:, 0])
    return out


def ln_mod_bwd(dxn, x, mean, rstd, scale, dx, *, B, S, dres=None, dshift=None, dscale=None):
    M = B * S
    xh = (x[:M].float() - mean[:M, None]) * rstd[:M, None]
    gr = dxn[:M].float()
    sc = scale.float().repeat_interleave(S, 0)
    g = gr * (1 + sc)
    c1 = g.mean(-1, keepdim=True)
    c2 = (g * xh).mean(-1, keepdim=True)
    v = rstd[:M, None] * (g - c1 - xh * c2)
    if dres is not None:
        v = v + dres[:M].float()
    if dshift is not None:
        dshift.copy_(gr.view(B, S, -1).sum(1).to(dshift.dtype))
        dscale.copy_((gr * xh).view(B, S, -1).sum(1).to(dscale.dtype))
    dx[:M].copy_(v.to(dx.dtype))
    return dx


def colsum_finish(partial, nchunk, B, V, Cc, out0, out1=None):
    """fp32 partial column sums [B][nchunk][V][C] -> out_v [B, C] (second stage of the deterministic column sums)."""
    p = partial.view(-1)[: B * nchunk * V * Cc].view(B, nchunk, V, Cc).float().sum(1)
    out0.copy_(p[:, 0].to(out0.dtype))
    if out1 is not None:
        out1.copy_(p[:, 1].to(out1.dtype))
    return out0


def gate_bwd(dx, y, gate, dy, dgate, *, B, S):
    M = B * S
    d = dx[:M].float()
    if y is not None:
        dgate.copy_((d * y[:M].float()).view(B, S, -1).sum(1).to(dgate.dtype))
    dy[:M].copy_((gate.float().repeat_interleave(S, 0) * d).to(dy.dtype))
    return dy


def _rope(t, cos, sin):
    a, b = t[..., 0::2], t[..., 1::2]
    o = torch.empty_like(t)
    o[..., 0::2] = a * cos[..., 0::2] - b * sin[..., 0::2]
    o[..., 1::2] = b * cos[..., 1::2] + a * sin[..., 1::2]
    return o


def qkv_post_fwd(jobs, cos, sin, *, B, H, S_src, S_dst, s_off, eps=1e-6):
    """diffusers RMSNorm + apply_rotary_emb (order: toolkit/models/flux_sage_attn.py:36-74)."""
    HD = H * 128
    c = cos[s_off:s_off + S_src][None, :, None, :]
    s = sin[s_off:s_off + S_src][None, :, None, :]
    for j in jobs:
        src, dst, w = j["src"], j["dst"], j.get("weight")
        x = src[: B * S_src, :HD].reshape(B, S_src, H, 128)
        dv = dst[: B * S_dst].view(B, S_dst, -1)[:, s_off:s_off + S_src, :HD]
        if w is None:
            dv.copy_(x.reshape(B, S_src, HD))
            continue
        xf = x.float()
        t = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(x.dtype)
        t = (t * w.to(x.dtype)).float()
        dv.copy_(_rope(t, c, s).reshape(B, S_src, HD).to(dst.dtype))


def qkv_post_bwd(jobs, cos, sin, *, B, H, S_src, S_dst, s_off, eps=1e-6):
    HD = H * 128
    c = cos[s_off:s_off + S_src][None, :, None, :]
    s = sin[s_off:s_off + S_src][None, :, None, :]
    for j in jobs:
        graw, gj, w = j["src"], j["dst"], j.get("weight")
        g = gj[: B * S_dst].view(B, S_dst, -1)[:, s_off:s_off + S_src, :HD]
        if w is None:
            graw[: B * S_src, :HD].copy_(g.reshape(B * S_src, HD))
            continue
        x = j["raw"][: B * S_src, :HD].reshape(B, S_src, H, 128).float().requires_grad_(True)
        with torch.enable_grad():
            t = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w.float()
            o = _rope(t, c, s)
            o.backward(g.reshape(B, S_src, H, 128).float())
        graw[: B * S_src, :HD].copy_(x.grad.reshape(B * S_src, HD).to(graw.dtype))


def ew(op, x, y, a=None, alpha=1.0, a_rows_per_batch=0):
    xf = x.float()
    if op == 0:
        v = F.silu(xf)
    elif op == 1:
        v = xf
    elif op == 2:
        af = a.float()
        if a_rows_per_batch:  # row m // rpb of `a`: ResnetBlock2D's hidden + time_emb_proj(...)[:, :, None, None]
            af = af.repeat_interleave(a_rows_per_batch, 0)[: xf.shape[0]]
        v = af + xf
    else:
        v = alpha * xf
    y.copy_(v.to(y.dtype))
    return y


def timestep_embed(t, out, tscale=1.0):
    """extensions_built_in/diffusion_models/chroma/src/layers.py:30-53 with flip_sin_to_cos (cos first)."""
    half = out.shape[1] // 2
    freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    ang = t[:, None].float() * tscale * freq[None]
    out.copy_(torch.cat([ang.cos(), ang.sin()], -1).to(out.dtype))
    return out


def copy_rows(dst, src):
    dst.copy_(src)
    return dst


def _heads(t, B, S, H, hs=128):
    return t[: B * S, : H * hs].reshape(B, S, H, hs).transpose(1, 2).float()


def attn_fwd(q, k, v, o, lse, *, B, H, S, scale, Skv=0, dv=0, hstride=0):
    """F.scaled_dot_product_attention (toolkit/models/flux_sage_attn.py:76; chroma/src/math.py:27; cross-attention:
    toolkit/models/wan21/wan_attn.py:67-75).  hstride: head width of the native [tokens, H*hstride] layout (0: 128-column heads)."""
    Skv, hs = Skv or S, hstride or 128
    qf, kf, vf = _heads(q, B, S, H, hs), _heads(k, B, Skv, H, hs), _heads(v, B, Skv, H, hs)
    sc = (qf @ kf.transpose(-1, -2)) * scale
    lse.copy_(torch.logsumexp(sc, -1) / math.log(2.0))
    out = sc.softmax(-1) @ vf
    o[: B * S, : H * hs].copy_(out.transpose(1, 2).reshape(B * S, H * hs).to(o.dtype))
    return o


def attn_bwd(q, k, v, o, lse, do, dq, dk, dv, *, B, H, S, scale, Skv=0, dvalid=0, hstride=0):
    Skv, hs = Skv or S, hstride or 128
    qf = _heads(q, B, S, H, hs).requires_grad_(True)
    kf = _heads(k, B, Skv, H, hs).requires_grad_(True)
    vf = _heads(v, B, Skv, H, hs).requires_grad_(True)
    with torch.enable_grad():
        out = ((qf @ kf.transpose(-1, -2)) * scale).softmax(-1) @ vf
        out.backward(_heads(do, B, S, H, hs))
    for dst, src, n in ((dq, qf, S), (dk, kf, Skv), (dv, vf, Skv)):
        dst[: B * n, : H * hs].copy_(src.grad.transpose(1, 2).reshape(B * n, H * hs).to(dst.dtype))


def gemv_nt(x, w, out, *, bias=None, t=None, bl=None, accumulate=False, col_scale=None):
    v = x.float() @ w.float().t()
    if col_scale is None:
        if bias is not None:
            v = v + bias.float()
        if t is not None:
            v = v + t.float() @ bl.float().t()
    else:  # DoRA: c * (x W^T + T B^T) + b
        if t is not None:
            v = v + t.float() @ bl.float().t()
        v = v * col_scale.float()[None, :]
        if bias is not None:
            v = v + bias.float()
    if accumulate:
        v = v + out.float()
    out.copy_(v.to(out.dtype))
    return out


def flow_noise_pack(latents, noise, t, noisy, target):
    """add_noise (toolkit/samplers/custom_flowmatch_sampler.py:91-102), target noise - latents
    (extensions_built_in/sd_trainer/SDTrainer.py:644-646), packing (toolkit/stable_diffusion_model.py:2157-2163)."""
    B, Cc, Hh, W = latents.shape
    t01 = (t.float() / 1000.0).view(B, 1, 1, 1)
    x0, e = latents.float(), noise.float()

    def pack(x):
        x = x.view(B, Cc, Hh // 2, 2, W // 2, 2)
        return x.permute(0, 2, 4, 1, 3, 5).reshape(B, (Hh // 2) * (W // 2), Cc * 4)

    noisy.copy_(pack((1.0 - t01) * x0 + t01 * e).to(noisy.dtype))
    target.copy_(pack(e - x0).to(target.dtype))


def mse_loss_grad(pred, target, dpred, loss_per_sample, loss, weight=None, mask=None, loss_type="mse", huber_c=0.01, guard=None, max_loss=None):
    """mse(pred.float(), target.float()) [* mask_multiplier] -> mean over (C,H,W) -> * multiplier -> mean over batch
    (SDTrainer.py:916-1013); mask [B, tokens, 4] is the reference's [B,1,h,w] mask in the packed 2x2-patch layout."""
    B = pred.shape[0]
    d = pred.float().reshape(B, -1) - target.float().reshape(B, -1)
    n = d.shape[1]
    mk = torch.ones_like(d)
    if mask is not None:
        feat = pred.shape[-1]
        mk = mask.float().reshape(B, -1, 1, 4).expand(B, n // feat, feat // 4, 4).reshape(B, n)
    if loss_type == "mse":
        el, gr = d * d, 2.0 * d
    elif loss_type == "mae":  # SDTrainer.py:907-908: l1_loss(reduction="none")
        el, gr = d.abs(), torch.sign(d)
    elif loss_type == "pseudo_huber":  # SDTrainer.py:903-906: sqrt(diff^2 + c^2) - c, c = 0.01
        r = torch.sqrt(d * d + huber_c * huber_c)
        el, gr = r - huber_c, d / r
    else:
        raise ValueError(loss_type)
    lps = (mk * el).mean(1)
    w = weight.float() if weight is not None else torch.ones(B, device=pred.device)
    loss_per_sample.copy_(lps)
    loss.copy_((lps * w).mean().reshape(1))
    dpred.copy_((gr * mk * w[:, None] / (n * B)).reshape(dpred.shape).to(dpred.dtype))
    if guard is not None:  # SDTrainer.py:2221-2224 (non-finite loss -> a fresh zero) and 1049-1050 (clamp(loss, max=max_loss): zero derivative above)
        gate = 0
        if not bool(torch.isfinite(loss).all()):  # no graph behind the replacement: no .grad anywhere -> counts toward the skipped step
            loss.zero_()
            gate = 1
            guard[1] += 1
            guard[0] += 1
        elif max_loss and float(loss) > max_loss:  # clamp: zero gradients DO arrive, the optimizer steps on them
            loss.fill_(max_loss)
            gate = 1
            guard[2] += 1
        if gate:
            dpred.zero_()
        guard[6] = gate


def adamw_ema_step(p, g, m, v, *, lr, beta1, beta2, eps, weight_decay, step, max_norm=0.0, ema=None, ema_decay=0.0,
                   grad_scale=1.0, norm_out=None, ema_feedback=0.0, param_multiplier=1.0, guard=None, n_micro=1):
    """clip_grad_norm_ + torch.optim.AdamW + EMA (SDTrainer.py:2278-2293, toolkit/optimizer.py:78-79, toolkit/ema.py:116-152).  guard: the
    step is skipped (p, m, v untouched; EMA still updated, as ema.update() runs regardless) when the gradient norm is not finite or every
    micro-batch was gated by the loss guard — torch.optim.AdamW over parameters whose .grad is None; the step count of the bias corrections is
    guard[3] (applied steps)."""
    gs = g * grad_scale
    norm = gs.double().pow(2).sum().sqrt().float()
    coef = 1.0
    if max_norm > 0:
        coef = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
    gs = gs * coef
    if norm_out is not None:
        norm_out.copy_(norm.reshape(1))
    skip = False
    if guard is not None:
        skip = (not bool(torch.isfinite(norm))) or (n_micro > 0 and int(guard[0]) >= n_micro)
        guard[0] = 0
        guard[5] = int(skip)
        guard[4 if skip else 3] += 1
        step = int(guard[3]) + (1 if skip else 0)
    if not skip:
        # the tensor ops of torch/optim/adam.py::_single_tensor_adam (decoupled weight decay), in its order — so that a trainer whose
        # optimizer.step() is served by this entry (ai_toolkit_amd/adopt.py) lands on the bits torch.optim.AdamW produces on the CPU
        p.mul_(1 - lr * weight_decay)
        m.lerp_(gs, 1 - beta1)
        v.mul_(beta2).addcmul_(gs, gs, value=1 - beta2)
        bc1 = 1 - beta1 ** step
        bc2s = (1 - beta2 ** step) ** 0.5
        p.addcdiv_(m, (v.sqrt() / bc2s).add_(eps), value=-(lr / bc1))
    if ema is not None:  # toolkit/ema.py:135-143
        tmp = (1 - ema_decay) * (ema - p)
        ema.sub_(tmp)
        if ema_feedback:
            p.add_(tmp * ema_feedback)
        if param_multiplier != 1.0:
            p.mul_(param_multiplier)


def ema_update(p, ema, *, decay, ema_feedback=0.0, param_multiplier=1.0):
    """toolkit/ema.py:126-152, the tensor ops of its loop body in their order, over the whole arena"""
    tmp = ema - p
    tmp.mul_(1.0 - decay)
    ema.sub_(tmp)
    if ema_feedback:
        p.add_(tmp * ema_feedback)
    if param_multiplier != 1.0:
        p.mul_(param_multiplier)


def make_shadow_table(entries, device):
    return entries, len(entries)


class recording:
    """ops.recording() of the product defers kernel launches so that two independent launch lists can be merged; this table
    executes eagerly, so there is nothing to defer or replay."""

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def replay_paired(a, b):
    pass


def lokr_lowrank_grad(dw, a, b, ga, gb, *, accumulate=True):
    da, db = dw @ b.t(), a.t() @ dw
    if accumulate:
        ga.add_(da)
        gb.add_(db)
    else:
        ga.copy_(da)
        gb.copy_(db)


def grad_compress_bf16(g, out):
    """transport format of the bf16 DP all-reduce: one round-to-nearest-even per element"""
    out[: g.numel()].copy_(g.to(torch.bfloat16))
    return out


def grad_expand_bf16(src, g):
    g.copy_(src[: g.numel()].float())
    return g


def refresh_shadows(arena, shadow, table):
    """hi = round(w), lo = round(w - hi) in the layouts of AitkShadowDesc (include/aitk_mi355.h)."""
    entries, _ = table
    for so, r, c, kind, d0, d1, d2, *aux in entries:
        if kind == 3:  # low-rank LoKr factor composed in fp32
            k = aux[0]
            w = arena[so:so + r * k].view(r, k) @ arena[so + r * k:so + r * k + k * c].view(k, c)
            shadow[d0:d0 + r * c].view(r, c).copy_(w.to(shadow.dtype))
            shadow[d1:d1 + r * c].view(c, r).copy_(w.to(shadow.dtype).t())
            continue
        w = arena[so:so + r * c].view(r, c)
        hi = w.to(shadow.dtype)
        if kind == 4:  # lora_down of a 3x3-conv adapter: [r, Cin, 3, 3] -> [A_hi ; A_lo] tap-major, and the rotated dgrad filter over the slab
            cin = aux[0]
            lo = (w - hi.float()).to(shadow.dtype)
            tm = lambda t: t.view(r, cin, 9).permute(0, 2, 1).reshape(r, 9 * cin)  # noqa: E731  columns tap*Cin + cin
            # rows in 16-rank blocks [A_hi(16) ; A_lo(16)]: the slab epilogue adds columns n and n + 16 of every 32-column block
            st = torch.stack((tm(hi).view(r // 16, 16, c), tm(lo).view(r // 16, 16, c)), dim=1)
            shadow[d0:d0 + 2 * r * c].view(2 * r, c).copy_(st.reshape(2 * r, c))
            h3, l3 = hi.view(r, cin, 9).flip(2), lo.view(r, cin, 9).flip(2)  # [r, cin, tap'] with tap' = 8 - tap
            blk = torch.cat((h3, h3, l3), dim=0)  # [3r, cin, tap']: slab channel j -> (A_hi | A_hi | A_lo)
            shadow[d1:d1 + 3 * r * c].view(cin, 27 * r).copy_(blk.permute(1, 2, 0).reshape(cin, 27 * r))
            continue
        if kind == 0:
            shadow[d0:d0 + r * c].view(r, c).copy_(hi)
            shadow[d1:d1 + r * c].view(c, r).copy_(hi.t())
            continue
        lo = (w - hi.float()).to(shadow.dtype)
        if kind == 1:  # A [rank, in]
            shadow[d0:d0 + r * c].view(r, c).copy_(hi)
            shadow[d1:d1 + r * c].view(r, c).copy_(lo)
            ld3 = aux[0] if aux and aux[0] > 0 else 3 * r  # aux: row stride of the [in, 3 rank] block (column window of a group's [in, 3 R])
            torch.as_strided(shadow, (c, 3 * r), (ld3, 1), d2).copy_(torch.cat((hi.t(), hi.t(), lo.t()), dim=1))
        else:  # B [out, rank]
            shadow[d0:d0 + 3 * r * c].view(r, 3 * c).copy_(torch.cat((hi, hi, lo), dim=1))
            shadow[d1:d1 + r * c].view(c, r).copy_(hi.t())
            shadow[d2:d2 + r * c].view(c, r).copy_(lo.t())


# ---------------------------------------------------------------------------------------------------------- VAE encoder
def conv3x3(x, w, out, *, B, H, W, stride=1, pad_t=1, pad_l=1, Ho=None, Wo=None, bias=None, flags=0, aux_in=None, a2=None, b2=None,
            split_slab=False, col_scale=None):
    """nn.Conv2d 3x3 on NHWC rows (diffusers AutoencoderKL, reached from toolkit/stable_diffusion_model.py:2567); with a2 / b2 the
    lora_up K-slab of a conv adapter is added; split_slab: the conv adapter's lora_down (w = 16-rank blocks [A_hi ; A_lo]) written as [hi | lo | hi]
    (toolkit/lora_special.py:95-104, toolkit/network_mixins.py:304-342)."""
    Cin = x.shape[1]
    Ho = H if Ho is None else Ho
    Wo = W if Wo is None else Wo
    xi = x.float().view(B, H, W, Cin).permute(0, 3, 1, 2)
    wk = w.float().view(w.shape[0], 3, 3, Cin).permute(0, 3, 1, 2)
    pad_b = (Ho - 1) * stride + 3 - H - pad_t
    pad_r = (Wo - 1) * stride + 3 - W - pad_l
    xi = F.pad(xi, (pad_l, max(pad_r, 0), pad_t, max(pad_b, 0)))
    y = F.conv2d(xi, wk, None, stride=stride)[:, :, :Ho, :Wo]
    v = y.permute(0, 2, 3, 1).reshape(B * Ho * Wo, -1)
    if split_slab:
        v4 = v.view(v.shape[0], -1, 2, 16)  # 16-rank blocks [hi | lo] of the stacked filter
        t = (v4[:, :, 0] + v4[:, :, 1]).reshape(v.shape[0], -1)
        if col_scale is not None:
            t = t * col_scale.float()[: t.shape[1]]
        hi = t.to(out.dtype)
        lo = (t - hi.float()).to(out.dtype)
        out.copy_(torch.cat((hi, lo, hi), dim=1))
        return out
    if a2 is not None:
        v = v + a2[:v.shape[0]].float() @ b2.float().t()
    if col_scale is not None:
        v = v * col_scale.float()[None, :]
    if bias is not None:
        v = v + bias.float()
    if flags & EPI_ADD_AUX:
        v = v + aux_in.float()
    if flags & EPI_ACCUM:
        v = v + out.float()
    out.copy_(v.to(out.dtype))
    return out


def pad_nhwc(src, dst, *, B, H, W):
    Cc = src.shape[1]
    y = torch.zeros(B, H + 2, W + 2, Cc, dtype=dst.dtype, device=dst.device)
    y[:, 1:H + 1, 1:W + 1] = src.view(B, H, W, Cc).to(dst.dtype)
    dst.copy_(y.view(dst.shape))
    return dst


def conv3d(x, w, out, *, T, H, W, kt=3, ks=3, tstride=1, stride=1, pad_t=1, pad_l=1, Ho=None, Wo=None, bias=None, flags=0, aux_in=None):
    """nn.Conv3d on frames of NHWC rows (diffusers AutoencoderKLWan's WanCausalConv3d / time_conv, reached from
    toolkit/models/wan21/wan21.py:659): the caller's buffer carries the causal front padding, frame t*tstride + dt feeds tap dt."""
    Cin = x.shape[1]
    Ho = H if Ho is None else Ho
    Wo = W if Wo is None else Wo
    Tin = (T - 1) * tstride + kt
    xi = x[:Tin * H * W].float().view(1, Tin, H, W, Cin).permute(0, 4, 1, 2, 3)
    wk = w.float().view(w.shape[0], kt, ks, ks, Cin).permute(0, 4, 1, 2, 3)
    pad_b = max((Ho - 1) * stride + ks - H - pad_t, 0)
    pad_r = max((Wo - 1) * stride + ks - W - pad_l, 0)
    xi = F.pad(xi, (pad_l, pad_r, pad_t, pad_b, 0, 0))
    y = F.conv3d(xi, wk, None, stride=(tstride, stride, stride))[:, :, :T, :Ho, :Wo]
    v = y[0].permute(1, 2, 3, 0).reshape(T * Ho * Wo, -1)
    if bias is not None:
        v = v + bias.float()
    if flags & EPI_ADD_AUX:
        v = v + aux_in.float()
    out.copy_(v.to(out.dtype))
    return out


def rmsnorm_rows(x, gamma, out, *, eps=1e-12, silu=False):
    """WanRMS_norm: F.normalize(x, dim=channel) * sqrt(C) * gamma (+ SiLU)."""
    v = F.normalize(x.float(), dim=1, eps=eps) * (x.shape[1] ** 0.5) * gamma.float()
    out.copy_((F.silu(v) if silu else v).to(out.dtype))
    return out


def latent_sample_affine(moments, eps, out, *, ch_shift, ch_scale):
    """DiagonalGaussianDistribution.sample + (z - latents_mean) * (1 / latents_std) (toolkit/models/wan21/wan21.py:659-670)."""
    B, L = out.shape[0], out.shape[1]
    hw = out[0, 0].numel()
    m = moments.float().view(B, hw, -1)
    mean, logvar = m[..., :L], m[..., L:2 * L].clamp(-30.0, 20.0)
    z = mean + torch.exp(0.5 * logvar) * eps.reshape(B, L, hw).transpose(1, 2)
    z = (z - ch_shift.float()) * ch_scale.float()
    out.copy_(z.transpose(1, 2).reshape(out.shape).to(out.dtype))
    return out


def groupnorm(x, gamma, beta, out, *, B, HW, G=32, eps=1e-6, silu=False, stats_out=None):
    Cc = x.shape[1]
    xi = x.float().view(B, HW, Cc).transpose(1, 2)
    y = F.group_norm(xi, G, gamma.float(), beta.float(), eps)
    if silu:
        y = F.silu(y)
    out.copy_(y.transpose(1, 2).reshape(B * HW, Cc).to(out.dtype))
    if stats_out is not None:
        xg = xi.reshape(B, G, -1).double()
        mean = xg.mean(-1)
        var = (xg * xg).mean(-1) - mean * mean
        stats_out.view(B, G, 2).copy_(torch.stack((mean, 1.0 / torch.sqrt(var.clamp_min(0) + eps)), -1).float())
    return out


# ---------------------------------------------------------------------------------------------------------- UNet (SD1.5 / SDXL)
def groupnorm_bwd(dy, x, gamma, beta, stats, dx, *, B, HW, G=32, silu=False, dres=None):
    """autograd of act(nn.GroupNorm(x)) wrt x with the forward's (mean, rstd) (diffusers ResnetBlock2D / Transformer2DModel norms)."""
    Cc = x.shape[1]
    st = stats.view(B, G, 2)
    xg = x.float().view(B, HW, G, Cc // G)
    xh = (xg - st[:, None, :, 0:1]) * st[:, None, :, 1:2]
    ga, be = gamma.float().view(1, 1, G, -1), beta.float().view(1, 1, G, -1)
    dz = dy.float().view(B, HW, G, Cc // G)
    if silu:
        z = xh * ga + be
        sg = torch.sigmoid(z)
        dz = dz * sg * (1 + z * (1 - sg))
    dxh = dz * ga
    m1 = dxh.mean(dim=(1, 3), keepdim=True)
    m2 = (dxh * xh).mean(dim=(1, 3), keepdim=True)
    v = st[:, None, :, 1:2] * (dxh - m1 - xh * m2)
    v = v.reshape(B * HW, Cc)
    if dres is not None:
        v = v + dres.float()
    dx.copy_(v.to(dx.dtype))
    return dx


def geglu_fwd(hg, out):
    """diffusers GEGLU: hidden * F.gelu(gate) (exact erf GELU)."""
    Fd = hg.shape[1] // 2
    out.copy_((hg[:, :Fd].float() * F.gelu(hg[:, Fd:].float())).to(out.dtype))
    return out


def geglu_bwd(dy, hg, dhg):
    Fd = hg.shape[1] // 2
    h, g = hg[:, :Fd].float(), hg[:, Fd:].float()
    d = dy.float()
    cdf = 0.5 * (1 + torch.erf(g * 0.7071067811865476))
    pdf = torch.exp(-0.5 * g * g) * 0.3989422804014327
    dhg[:, :Fd].copy_((d * g * cdf).to(dhg.dtype))
    dhg[:, Fd:].copy_((d * h * (cdf + g * pdf)).to(dhg.dtype))
    return dhg


def resample2x(src, dst, *, B, H, W, mode):
    """mode 0: F.interpolate(scale_factor=2, mode='nearest') (Upsample2D); 1: its adjoint (2x2 sum); 2: zero insertion."""
    Cc = src.shape[1]
    x = src.float().view(B, H, W, Cc)
    if mode == 0:
        y = x.repeat_interleave(2, 1).repeat_interleave(2, 2)
    elif mode == 1:
        y = x.view(B, H // 2, 2, W // 2, 2, Cc).sum(dim=(2, 4))
    else:
        y = torch.zeros(B, 2 * H, 2 * W, Cc, dtype=torch.float32, device=src.device)
        y[:, ::2, ::2] = x
    dst.copy_(y.reshape(dst.shape).to(dst.dtype))
    return dst


def copy_heads(src, dst, *, H, d_src, d_dst):
    M = src.shape[0]
    d = min(d_src, d_dst)
    out = torch.zeros(M, H, d_dst, dtype=dst.dtype, device=dst.device)
    out[:, :, :d] = src[:, : H * d_src].reshape(M, H, d_src)[:, :, :d]
    dst[:, : H * d_dst].copy_(out.reshape(M, H * d_dst))
    return dst


def ddpm_noise_nhwc(latents, noise, a, s, noisy, target, *, v_prediction=False):
    """DDPMScheduler.add_noise / get_velocity in the latent dtype (products and the sum each rounded), written NHWC."""
    B, Cc, h, w = latents.shape
    dt = latents.dtype
    av, sv = a.to(dt).view(B, 1, 1, 1), s.to(dt).view(B, 1, 1, 1)
    nz = av * latents + sv * noise
    tg = (av * noise - sv * latents) if v_prediction else noise
    noisy.zero_()
    noisy[:, :Cc].copy_(nz.permute(0, 2, 3, 1).reshape(B * h * w, Cc))
    target.copy_(tg.permute(0, 2, 3, 1).reshape(B * h * w, Cc))


def softmax_rows(x, scale):
    x.copy_(torch.softmax(x.float() * scale, dim=-1).to(x.dtype))
    return x


def image_to_nhwc8(img, out):
    B, Cc, H, W = img.shape
    out.zero_()
    out[:, :3].copy_(img.permute(0, 2, 3, 1).reshape(B * H * W, 3).to(out.dtype))
    return out


def image_resize_to_nhwc8(img, out, *, Hd, Wd):
    """toolkit/models/wan21/wan21.py:652-657: F.interpolate(images.to(vae dtype), size, mode='bilinear', align_corners=False)."""
    B = img.shape[0]
    r = F.interpolate(img.to(out.dtype), size=(Hd, Wd), mode="bilinear", align_corners=False)
    out.zero_()
    out.view(B, Hd, Wd, 8)[..., :3].copy_(r.permute(0, 2, 3, 1))
    return out


def latent_sample(moments, eps, out, *, scale, shift):
    """DiagonalGaussianDistribution.sample + scaling (toolkit/stable_diffusion_model.py:2567-2573)."""
    B, L, h, w = out.shape
    m = moments.float().view(B, h * w, -1)
    mean, logvar = m[..., :L], m[..., L:2 * L].clamp(-30.0, 20.0)
    z = mean + torch.exp(0.5 * logvar) * eps.view(B, L, h * w).transpose(1, 2)
    out.copy_((scale * (z - shift)).transpose(1, 2).reshape(B, L, h, w).to(out.dtype))
    return out


# ---------------------------------------------------------------------------------------------------------- Wan2.1
def _rms_full(x, weight, S, cos, sin, eps, round_dtype):
    xf = x.float() if not x.requires_grad else x
    t = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    if round_dtype is not None:
        t = (t.to(round_dtype) * weight.to(round_dtype)).float()
    else:
        t = t * weight.float()
    if cos is not None:
        M, Cc = t.shape
        H = Cc // 128
        pos = torch.arange(M, device=t.device) % S
        t = _rope(t.view(M, H, 128), cos[pos][:, None, :], sin[pos][:, None, :]).reshape(M, Cc)
    return t


def rms_full_fwd(x, weight, y, *, S, cos=None, sin=None, eps=1e-6):
    """norm_q / norm_k (RMSNorm across heads) + rotary embedding (toolkit/models/wan21/wan_attn.py:36-54)."""
    y.copy_(_rms_full(x, weight, S, cos, sin, eps, x.dtype).to(y.dtype))
    return y


def rms_full_bwd(g, x, weight, dx, *, S, cos=None, sin=None, eps=1e-6):
    xx = x.float().detach().requires_grad_(True)
    with torch.enable_grad():
        _rms_full(xx, weight, S, cos, sin, eps, None).backward(g.float())
    dx.copy_(xx.grad.to(dx.dtype))
    return dx


# ---------------------------------------------------------------------------------------------------------- DoRA
def dora_colscale(w2, tw, up, gram, mag, s, c_out):
    """c_j = magnitude_j / ||W_j + s * B_j A||  (toolkit/models/DoRA.py:126-148), from precomputed pieces:
    ||.||^2 = ||W_j||^2 + 2 s B_j . (W A^T)_j + s^2 B_j (A A^T) B_j^T."""
    upf = up.float()
    n2 = w2.float() + 2.0 * s * (upf * tw.float()).sum(1) + (s * s) * ((upf @ gram.float()) * upf).sum(1)
    c_out.copy_(mag.float() / torch.sqrt(n2))
    return c_out


def dora_bwd(dy, y, c, bias, mag, dz, dmag, *, M):
    """y = c * z + b  (z = x W^T + T B^T):  dz = c * dy ;  d magnitude_j += sum_m dy_mj z_mj / ||.||_j = (sum dy*y - b_j sum dy) / mag_j."""
    d = dy[:M].float()
    yf = y[:M].float()
    dz[:M].copy_((d * c.float()[None, :]).to(dz.dtype))
    s1 = (d * yf).sum(0)
    s0 = d.sum(0)
    b = bias.float() if bias is not None else torch.zeros_like(s0)
    dmag += (s1 - b * s0) / mag.float()
    return dz


def dequant_fp8(q, scale, mode, out):
    f = q.view(torch.float8_e4m3fn).float()
    out.copy_((f * (scale[:, None] if mode == 1 else scale[None, :])).to(out.dtype))
    return out


def kron_apply(x, A, Bm, out, *, a_in, b_in, a_out, b_out, scale=1.0, transpose_out=False, accumulate=False, col0=0, ncols=0,
               x_seg=None, out_seg=None, M=None):
    """LoKr per-token product (toolkit/models/lokr.py:331-399): X = x.unflatten(-1, (a_in, b_in));
    tmp = einsum('mqs,os->mqo', X, Bm) rounded to the storage dtype; out = scale * einsum('mqo,pq->mpo', tmp, A)."""
    if M is None:
        M = x.shape[0]
    X = _seg_view(x, x_seg, M).float().reshape(M, a_in, b_in)
    tmp = X if Bm is None else torch.einsum("mqs,os->mqo", X, Bm.float())
    if A is not None and Bm is not None:
        tmp = tmp.to(out.dtype).float()  # the kernel keeps the intermediate in the storage dtype (as the reference does)
    res = (tmp if A is None else torch.einsum("mqo,pq->mpo", tmp, A.float())) * scale
    if transpose_out:
        res = res.transpose(1, 2)
    res = res.reshape(M, a_out * b_out)
    if ncols:
        res = res[:, col0:col0 + ncols]
    if accumulate:
        res = res + _seg_view(out, out_seg, M).float()
    _seg_store(out, out_seg, M, res)
    return out


def kron_merge(W, A, Bm, alpha):
    """LokrModule.merge_in (toolkit/models/lokr.py:261-309): weight + kron(w1, w2) * scale * merge_weight, cast back."""
    W.copy_((W.float() + alpha * torch.kron(A.float(), Bm.float())).to(W.dtype))
    return W



def _heads_d(t, B, S, H, D):
    return t[: B * S, : H * D].reshape(B, S, H, D).transpose(1, 2).float()


def attn_small_fwd(q, k, v, o, lse, *, B, H, S, D, scale, Skv=0):
    """F.scaled_dot_product_attention for any head_dim (diffusers AttnProcessor2_0 of the UNet attention blocks)."""
    Skv = Skv or S
    qf, kf, vf = _heads_d(q, B, S, H, D), _heads_d(k, B, Skv, H, D), _heads_d(v, B, Skv, H, D)
    sc = (qf @ kf.transpose(-1, -2)) * scale
    lse.copy_(torch.logsumexp(sc, -1))
    o[: B * S, : H * D].copy_((sc.softmax(-1) @ vf).transpose(1, 2).reshape(B * S, H * D).to(o.dtype))
    return o


def attn_small_bwd(q, k, v, o, lse, do, dq, dk, dv, *, B, H, S, D, scale, Skv=0):
    Skv = Skv or S
    qf = _heads_d(q, B, S, H, D).requires_grad_(True)
    kf = _heads_d(k, B, Skv, H, D).requires_grad_(True)
    vf = _heads_d(v, B, Skv, H, D).requires_grad_(True)
    with torch.enable_grad():
        out = ((qf @ kf.transpose(-1, -2)) * scale).softmax(-1) @ vf
        out.backward(_heads_d(do, B, S, H, D))
    for dst, src, n in ((dq, qf, S), (dk, kf, Skv), (dv, vf, Skv)):
        dst[: B * n, : H * D].copy_(src.grad.transpose(1, 2).reshape(B * n, H * D).to(dst.dtype))


def host_call(fn):
    """ops.host_call: deferred torch-side work of a recorded launch list; the oracle table launches nothing and runs it at once."""
    fn()
