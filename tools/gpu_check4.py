"""Per-kernel parity of the UNet-side kernels (csrc/unet_kernels.hip, conv mode of aitk_gemm_nt) against the oracle's functions of the
same name (oracle/ref_ops.py) on the MI355X.  Used by tests/test_gpu_unet.py; `python tools/gpu_check4.py` prints every record."""
import json
import math
import sys

import torch

sys.path.insert(0, ".")
from ai_toolkit_amd import ops  # noqa: E402
from oracle import ref_ops  # noqa: E402

bf = torch.bfloat16
dev = "cuda"


def rel(x, ref):
    x, ref = x.float(), ref.float()
    return ((x - ref).norm() / (ref.norm() + 1e-30)).item()


def R(*shape, s=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(*shape, generator=g) * s


def t_groupnorm(B=2, HW=4096, C=320, G=32, silu=True):
    x = (R(B * HW, C, seed=1) * 1.5 + 0.3).to(bf).to(dev)
    dy = R(B * HW, C, seed=2).to(bf).to(dev)
    ga = (1 + 0.1 * R(C, seed=3)).to(bf).to(dev)
    be = (0.1 * R(C, seed=4)).to(bf).to(dev)
    y, y2 = torch.empty_like(x), torch.empty_like(x)
    st, st2 = torch.empty(B * G * 2, device=dev), torch.empty(B * G * 2, device=dev)
    ops.groupnorm(x, ga, be, y, B=B, HW=HW, G=G, eps=1e-5, silu=silu, stats_out=st)
    ref_ops.groupnorm(x, ga, be, y2, B=B, HW=HW, G=G, eps=1e-5, silu=silu, stats_out=st2)
    dres = R(B * HW, C, seed=5).to(bf).to(dev)
    dx, dx2 = torch.empty_like(x), torch.empty_like(x)
    ops.groupnorm_bwd(dy, x, ga, be, st, dx, B=B, HW=HW, G=G, silu=silu, dres=dres)
    ref_ops.groupnorm_bwd(dy, x, ga, be, st2, dx2, B=B, HW=HW, G=G, silu=silu, dres=dres)
    # independent check: autograd of torch's group_norm in fp32
    xf = x.float().view(B, HW, C).transpose(1, 2).requires_grad_(True)
    yy = torch.nn.functional.group_norm(xf, G, ga.float(), be.float(), 1e-5)
    if silu:
        yy = torch.nn.functional.silu(yy)
    yy.backward(dy.float().view(B, HW, C).transpose(1, 2))
    dx3 = xf.grad.transpose(1, 2).reshape(B * HW, C) + dres.float()
    torch.cuda.synchronize()
    r = {"fwd": rel(y, y2), "stats": rel(st, st2), "bwd": rel(dx, dx2), "bwd_vs_autograd": rel(dx, dx3)}
    r["ok"] = r["fwd"] < 3e-3 and r["stats"] < 1e-5 and r["bwd"] < 3e-3 and r["bwd_vs_autograd"] < 4e-3
    return r


def t_geglu(M=3000, Fd=1280):
    hg = R(M, 2 * Fd, seed=1).to(bf).to(dev)
    dy = R(M, Fd, seed=2).to(bf).to(dev)
    o, o2 = torch.empty(M, Fd, dtype=bf, device=dev), torch.empty(M, Fd, dtype=bf, device=dev)
    ops.geglu_fwd(hg, o)
    ref_ops.geglu_fwd(hg, o2)
    d, d2 = torch.empty_like(hg), torch.empty_like(hg)
    ops.geglu_bwd(dy, hg, d)
    ref_ops.geglu_bwd(dy, hg, d2)
    torch.cuda.synchronize()
    r = {"fwd": rel(o, o2), "bwd": rel(d, d2)}
    r["ok"] = r["fwd"] < 2e-3 and r["bwd"] < 2e-3
    return r


def t_resample(B=2, H=12, W=20, C=64):
    x = R(B * H * W, C, seed=1).to(bf).to(dev)
    out = {}
    for mode, shape in ((0, (B * 4 * H * W, C)), (2, (B * 4 * H * W, C)), (1, (B * (H // 2) * (W // 2), C))):
        a, b = torch.empty(shape, dtype=bf, device=dev), torch.empty(shape, dtype=bf, device=dev)
        ops.resample2x(x, a, B=B, H=H, W=W, mode=mode)
        ref_ops.resample2x(x, b, B=B, H=H, W=W, mode=mode)
        torch.cuda.synchronize()
        out[f"mode{mode}"] = bool(torch.equal(a, b)) if mode != 1 else rel(a, b) < 2e-3
    out["ok"] = all(out.values())
    return out


def t_copy_heads(M=777, H=10, d=64):
    x = R(M, H * d, seed=1).to(bf).to(dev)
    p, p2 = torch.full((M, H * 128), 7.0, dtype=bf, device=dev), torch.empty(M, H * 128, dtype=bf, device=dev)
    ops.copy_heads(x, p, H=H, d_src=d, d_dst=128)
    ref_ops.copy_heads(x, p2, H=H, d_src=d, d_dst=128)
    back = torch.empty(M, H * d, dtype=bf, device=dev)
    ops.copy_heads(p, back, H=H, d_src=128, d_dst=d)
    torch.cuda.synchronize()
    return {"pad": bool(torch.equal(p, p2)), "unpad": bool(torch.equal(back, x)), "ok": bool(torch.equal(p, p2) and torch.equal(back, x))}


def t_ddpm(B=3, C=4, h=24, w=16, v=False):
    lat = R(B, C, h, w, seed=1).to(bf).to(dev)
    noi = R(B, C, h, w, seed=2).to(bf).to(dev)
    a = torch.tensor([0.9961, 0.5, 0.0684], device=dev).to(bf).float()
    s = torch.tensor([0.0889, 0.8672, 0.9961], device=dev).to(bf).float()
    nz, tg = torch.full((B * h * w, 8), 3.0, dtype=bf, device=dev), torch.empty(B * h * w, C, dtype=bf, device=dev)
    nz2, tg2 = torch.empty_like(nz), torch.empty_like(tg)
    ops.ddpm_noise_nhwc(lat, noi, a, s, nz, tg, v_prediction=v)
    ref_ops.ddpm_noise_nhwc(lat, noi, a, s, nz2, tg2, v_prediction=v)
    torch.cuda.synchronize()
    return {"noisy": bool(torch.equal(nz, nz2)), "target": bool(torch.equal(tg, tg2)), "ok": bool(torch.equal(nz, nz2) and torch.equal(tg, tg2))}


def t_ew_broadcast(B=3, HW=100, C=320):
    x = R(B * HW, C, seed=1).to(bf).to(dev)
    a = R(B, C, seed=2).to(bf).to(dev)
    y, y2 = torch.empty_like(x), torch.empty_like(x)
    ops.ew(2, x, y, a=a, a_rows_per_batch=HW)
    ref_ops.ew(2, x, y2, a=a, a_rows_per_batch=HW)
    torch.cuda.synchronize()
    return {"ok": bool(torch.equal(y, y2))}


def t_attn_small(B=2, H=8, S=256, D=160, Skv=0):
    kv = Skv or S
    q = R(B * S, H * D, s=0.5, seed=1).to(bf).to(dev)
    k = R(B * kv, H * D, s=0.5, seed=2).to(bf).to(dev)
    v = R(B * kv, H * D, seed=3).to(bf).to(dev)
    do = R(B * S, H * D, seed=4).to(bf).to(dev)
    sc = 1.0 / math.sqrt(D)
    o, o2 = torch.empty_like(q), torch.empty_like(q)
    l, l2 = torch.empty(B, H, S, device=dev), torch.empty(B, H, S, device=dev)
    ops.attn_small_fwd(q, k, v, o, l, B=B, H=H, S=S, D=D, scale=sc, Skv=Skv)
    ref_ops.attn_small_fwd(q, k, v, o2, l2, B=B, H=H, S=S, D=D, scale=sc, Skv=Skv)
    g = [torch.empty_like(t) for t in (q, k, v)]
    g2 = [torch.empty_like(t) for t in (q, k, v)]
    ops.attn_small_bwd(q, k, v, o, l, do, *g, B=B, H=H, S=S, D=D, scale=sc, Skv=Skv)
    ref_ops.attn_small_bwd(q, k, v, o2, l2, do, *g2, B=B, H=H, S=S, D=D, scale=sc, Skv=Skv)
    torch.cuda.synchronize()
    r = {"o": rel(o, o2), "lse": rel(l, l2), "dq": rel(g[0], g2[0]), "dk": rel(g[1], g2[1]), "dv": rel(g[2], g2[2])}
    r["ok"] = r["o"] < 3e-3 and r["lse"] < 1e-5 and max(r["dq"], r["dk"], r["dv"]) < 4e-3
    return r


def t_attn_padded(B=2, H=10, S=1000, D=64, Skv=77):
    """head_dim 64 / 40 / 80 through the head_dim-128 flash kernels by zero padding == attention at the true head_dim; run twice: with the
    full 128-wide kernels and with the `dv` variants that skip the all-zero contraction steps / output blocks — the two must agree bit for
    bit on the valid columns (same MFMA sequence on the non-zero data) and both leave exact zeros in the padded columns."""
    kv = Skv or S
    q = R(B * S, H * D, s=0.5, seed=1).to(bf).to(dev)
    k = R(B * kv, H * D, s=0.5, seed=2).to(bf).to(dev)
    v = R(B * kv, H * D, seed=3).to(bf).to(dev)
    do = R(B * S, H * D, seed=4).to(bf).to(dev)
    sc = 1.0 / math.sqrt(D)
    pads = [torch.empty(t.shape[0], H * 128, dtype=bf, device=dev) for t in (q, k, v, do)]
    for s_, d_ in zip((q, k, v, do), pads):
        ops.copy_heads(s_, d_, H=H, d_src=D, d_dst=128)
    runs = {}
    for dvv in (0, D):
        op_ = torch.full((B * S, H * 128), float("nan"), dtype=bf, device=dev)
        lse = torch.empty(B, H, S, device=dev)
        ops.attn_fwd(pads[0], pads[1], pads[2], op_, lse, B=B, H=H, S=S, scale=sc, Skv=Skv, dv=dvv)
        gp = [torch.full_like(pads[0], float("nan")), torch.full_like(pads[1], float("nan")), torch.full_like(pads[2], float("nan"))]
        ops.attn_bwd(pads[0], pads[1], pads[2], op_, lse, pads[3], *gp, B=B, H=H, S=S, scale=sc, Skv=Skv, dvalid=dvv)
        runs[dvv] = (op_, lse, gp)
    torch.cuda.synchronize()
    op_, lse, gp = runs[D]
    same = all(torch.equal(a, b) for a, b in zip([runs[0][0], runs[0][1]] + runs[0][2], [op_, lse] + gp))
    o = torch.empty_like(q)
    ops.copy_heads(op_, o, H=H, d_src=128, d_dst=D)
    g = [torch.empty_like(t) for t in (q, k, v)]
    for s_, d_ in zip(gp, g):
        ops.copy_heads(s_, d_, H=H, d_src=128, d_dst=D)
    o2, l2 = torch.empty_like(q), torch.empty(B, H, S, device=dev)
    ref_ops.attn_small_fwd(q, k, v, o2, l2, B=B, H=H, S=S, D=D, scale=sc, Skv=Skv)
    g2 = [torch.empty_like(t) for t in (q, k, v)]
    ref_ops.attn_small_bwd(q, k, v, o2, l2, do, *g2, B=B, H=H, S=S, D=D, scale=sc, Skv=Skv)
    torch.cuda.synchronize()
    padz = max(float(t.view(t.shape[0], H, 128)[:, :, D:].float().abs().max()) for t in [op_] + gp)
    r = {"o": rel(o, o2), "dq": rel(g[0], g2[0]), "dk": rel(g[1], g2[1]), "dv": rel(g[2], g2[2]), "pad_cols_max": padz,
         "dv_variant_bitwise_equal_to_full": same}
    r["ok"] = r["o"] < 4e-3 and max(r["dq"], r["dk"], r["dv"]) < 1.5e-2 and padz == 0.0 and same
    return r


def t_conv(B=2, H=32, W=24, Cin=64, Cout=96, stride=1, res=False, dgrad=False):
    """forward implicit-GEMM conv vs F.conv2d; dgrad = the same kernel on the rotated, in/out-swapped filter (+ zero insertion for
    stride 2) vs autograd of F.conv2d."""
    import torch.nn.functional as F

    from ai_toolkit_amd.unet import Conv3x3

    conv = Conv3x3(Cin, Cout, stride, bf, dev, cout_pad=max(Cout, 8))
    with torch.no_grad():
        conv.weight.copy_((R(Cout, Cin, 3, 3, seed=1) / math.sqrt(9 * Cin)).to(bf))
        conv.bias.copy_((R(Cout, seed=2) * 0.1).to(bf))
    conv.prepare()
    x = R(B * H * W, Cin, seed=3).to(bf).to(dev)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    Cp = conv.cout_pad
    y = torch.empty(B * Ho * Wo, Cp, dtype=bf, device=dev)
    aux = R(B * Ho * Wo, Cout, seed=4).to(bf).to(dev) if res else None
    ops.conv3x3(x, conv.wk, y, B=B, H=H, W=W, stride=stride, Ho=Ho, Wo=Wo, bias=conv.bias_k, flags=ops.EPI_ADD_AUX if res else 0, aux_in=aux)
    pad_zero = Cp == Cout or float(y[:, Cout:].abs().max()) == 0.0
    y = y[:, :Cout]
    xi = x.float().view(B, H, W, Cin).permute(0, 3, 1, 2).requires_grad_(True)
    yr = F.conv2d(xi, conv.weight.float(), conv.bias.float(), stride=stride, padding=1)
    yref = yr.permute(0, 2, 3, 1).reshape(B * Ho * Wo, Cout) + (aux.float() if res else 0)
    r = {"fwd": rel(y, yref)}
    if dgrad:
        dy = R(B * Ho * Wo, Cout, seed=5).to(bf).to(dev)
        g = dy
        if Cp != Cout:  # conv_out: the data gradient reads the zero-padded [M, 8] gradient
            g = torch.zeros(B * Ho * Wo, Cp, dtype=bf, device=dev)
            g[:, :Cout] = dy
        if stride == 2:
            g = torch.empty(B * 4 * Ho * Wo, Cout, dtype=bf, device=dev)
            ops.resample2x(dy, g, B=B, H=Ho, W=Wo, mode=2)
        dx = torch.empty(B * H * W, Cin, dtype=bf, device=dev)
        ops.conv3x3(g, conv.wd, dx, B=B, H=H, W=W)
        yr.backward(dy.float().view(B, Ho, Wo, Cout).permute(0, 3, 1, 2))
        r["dgrad"] = rel(dx, xi.grad.permute(0, 2, 3, 1).reshape(B * H * W, Cin))
    torch.cuda.synchronize()
    r["ok"] = r["fwd"] < 3e-3 and r.get("dgrad", 0.0) < 3e-3 and pad_zero
    return r


def main():
    out = {}
    for name, fn in (("groupnorm_320", lambda: t_groupnorm()), ("groupnorm_1920_nosilu", lambda: t_groupnorm(1, 1024, 1920, 32, False)),
                     ("geglu", t_geglu), ("resample", t_resample), ("copy_heads", t_copy_heads), ("ddpm_eps", t_ddpm), ("ddpm_v", lambda: t_ddpm(v=True)),
                     ("ew_broadcast", t_ew_broadcast), ("attn_small_self", t_attn_small), ("attn_small_cross77", lambda: t_attn_small(Skv=77)),
                     ("attn_pad64_cross77", t_attn_padded), ("attn_pad40_self", lambda: t_attn_padded(1, 8, 1024, 40, 0)),
                     ("conv_s1", lambda: t_conv(dgrad=True)), ("conv_s2", lambda: t_conv(stride=2, dgrad=True)), ("conv_res", lambda: t_conv(res=True)),
                     ("conv_in_8_to_320", lambda: t_conv(2, 64, 64, 8, 320)), ("conv_out_320_to_4", lambda: t_conv(2, 64, 64, 320, 4, dgrad=True)),
                     ("conv_1920_640", lambda: t_conv(1, 32, 32, 1920, 640, dgrad=True))):
        try:
            out[name] = fn()
        except Exception as ex:
            out[name] = {"ok": False, "error": f"{type(ex).__name__}: {ex}"[:300]}
        print(name, json.dumps(out[name]), flush=True)
    print("ALL_OK", all(v.get("ok") for v in out.values()))


if __name__ == "__main__":
    main()
