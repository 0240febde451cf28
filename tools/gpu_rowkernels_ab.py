"""Row kernels of the DiT block at the FLUX shape (B = 7, 4096 + 512 tokens, C = 3072, 24 heads): LN + modulate backward in its two forms (mode 0 = the two-phase
kernel, AITK_LN_BWD_TWO_PHASE=1; mode 1 = the single-pass kernel, two waves per 16-row chunk) — comparison against mode 0 and time per launch — and the per-head
RMSNorm + RoPE kernels (time per launch, algorithmic bytes per second).  Prints JSON lines.  (profiles/r04_rowkernels_ab.log was taken while a third form, one wave per chunk,
still existed: its "mode1" is that form, its "mode2" the kernel kept.)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import ops  # noqa: E402

bf = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def ln_bwd():
    g = torch.Generator(device="cuda").manual_seed(1)
    for (B, S, C) in ((7, 4608, 3072), (2, 201, 3072), (3, 77, 1536)):
        M = B * S
        x = torch.randn(M, C, device="cuda", generator=g).to(bf)
        dxn = (torch.randn(M, C, device="cuda", generator=g) * 0.1).to(bf)
        dres = (torch.randn(M, C, device="cuda", generator=g) * 0.1).to(bf)
        scale = (torch.randn(B, C, device="cuda", generator=g) * 0.2).to(bf)
        mean = x.float().mean(1)
        rstd = torch.rsqrt(x.float().var(1, unbiased=False) + 1e-6)
        res = {}
        for mode in (0, 1):
            if mode == 0:
                os.environ["AITK_LN_BWD_TWO_PHASE"] = "1"
            else:
                os.environ.pop("AITK_LN_BWD_TWO_PHASE", None)
            dx = torch.full((M, C), float("nan"), dtype=bf, device="cuda")
            dsh, dsc = torch.empty(B, C, dtype=bf, device="cuda"), torch.empty(B, C, dtype=bf, device="cuda")
            ops.ln_mod_bwd(dxn, x, mean, rstd, scale, dx, B=B, S=S, dres=dres, dshift=dsh, dscale=dsc)
            torch.cuda.synchronize()
            ms = timeit(lambda: ops.ln_mod_bwd(dxn, x, mean, rstd, scale, dx, B=B, S=S, dres=dres, dshift=dsh, dscale=dsc))
            res[mode] = (dx.clone(), dsh.clone(), dsc.clone(), ms)
        row = {"ln_mod_bwd": [B, S, C]}
        for mode in (0, 1):
            dx, dsh, dsc, ms = res[mode]
            row[f"mode{mode}_us"] = round(ms * 1e3, 1)
            row[f"mode{mode}_TBps_alg"] = round(4 * M * C * 2 / ms / 1e9, 2)
            if mode:
                row[f"mode{mode}_dx_equal"] = bool(torch.equal(dx, res[0][0]))
                row[f"mode{mode}_dx_max_rel"] = float(((dx.float() - res[0][0].float()).abs().max() / res[0][0].float().abs().max()).item())
                row[f"mode{mode}_colsums_equal"] = bool(torch.equal(dsh, res[0][1]) and torch.equal(dsc, res[0][2]))
        print(json.dumps(row), flush=True)
    os.environ.pop("AITK_LN_BWD_TWO_PHASE", None)


def qkv():
    from tools import gpu_check2 as g2

    print(json.dumps({"qkv_post_vs_oracle": g2.t_qkv_post(2, 24, 100, 4), "ragged_heads": g2.t_qkv_post(1, 7, 33, 3)}), flush=True)
    B, H, Si, St = 7, 24, 4096, 512
    S = Si + St
    cos, sin = g2.rope_tables(S)
    ld = 3 * H * 128
    gen = torch.Generator(device="cuda").manual_seed(2)
    raw = torch.randn(B * Si, ld, device="cuda", generator=gen).to(bf)
    joint = torch.empty(B * S, ld, dtype=bf, device="cuda")
    graw = torch.empty_like(raw)
    w = torch.ones(128, dtype=bf, device="cuda")
    HD = H * 128
    jobs = [dict(src=raw[:, :HD], dst=joint[:, :HD], weight=w), dict(src=raw[:, HD:2 * HD], dst=joint[:, HD:2 * HD], weight=w),
            dict(src=raw[:, 2 * HD:], dst=joint[:, 2 * HD:], weight=None)]
    ms = timeit(lambda: ops.qkv_post_fwd(jobs, cos, sin, B=B, H=H, S_src=Si, S_dst=S, s_off=St))
    jb = [dict(src=graw[:, :HD], dst=joint[:, :HD], weight=w, raw=raw[:, :HD]), dict(src=graw[:, HD:2 * HD], dst=joint[:, HD:2 * HD], weight=w, raw=raw[:, HD:2 * HD]),
          dict(src=graw[:, 2 * HD:], dst=joint[:, 2 * HD:], weight=None)]
    mb = timeit(lambda: ops.qkv_post_bwd(jb, cos, sin, B=B, H=H, S_src=Si, S_dst=S, s_off=St))
    nbytes = B * Si * ld * 2
    print(json.dumps({"qkv_post image stream B=7": {"fwd_us": round(ms * 1e3, 1), "fwd_TBps_alg": round(2 * nbytes / ms / 1e9, 2), "bwd_us": round(mb * 1e3, 1),
                                                     "bwd_TBps_alg": round((2 * nbytes + nbytes * 2 / 3) / mb / 1e9, 2)}}), flush=True)


if __name__ == "__main__":
    ln_bwd()
    qkv()
