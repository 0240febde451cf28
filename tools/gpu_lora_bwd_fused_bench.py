"""aitk_lora_bwd_fused against aitk_lora_down + aitk_lora_wgrad on the headline step's backward shapes (M = 7 * 4608 image rows / 7 * 512 text
rows, rank 16): microseconds per layer and the dY bytes per second both ways, cold (a 1-GB write in between, like the in-step producer) and warm."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ai_toolkit_amd import ops  # noqa: E402

bf = torch.bfloat16


def timed(fn, cold, iters=12):
    junk = torch.empty(512 * 2 ** 20, dtype=bf, device="cuda") if cold else None
    ts = []
    for _ in range(iters):
        if junk is not None:
            junk.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    out = []
    for M, L in ((32256, 3072), (32256, 12288), (32256, 15360), (3584, 3072), (32256, 9216), (4608, 3072), (4608, 12288)):
        R = 16
        g = torch.Generator(device="cuda").manual_seed(1)
        dy = (torch.randn(M, L, device="cuda", generator=g) * 0.5).to(bf)
        p_hi = (torch.randn(R, L, device="cuda", generator=g) * 0.05).to(bf)
        p_lo = (torch.randn(R, L, device="cuda", generator=g) * 0.0002).to(bf)
        T = (torch.randn(M, 3 * R, device="cuda", generator=g) * 0.3).to(bf)
        dT, gu = torch.empty(M, 3 * R, dtype=bf, device="cuda"), torch.zeros(L, R, device="cuda")

        def sep():
            ops.lora_down(dy, p_hi, dT, scale=1.0, M=M, p_lo=p_lo, split=R)
            ops.lora_wgrad(T, dy, gu, transpose_out=True, accumulate=True, M=M, split=R)

        def fused():
            ops.lora_bwd_fused(dy, T, p_hi, p_lo, dT, gu, scale=1.0, M=M, split=R)

        for _ in range(3):
            sep()
            fused()
        row = {"M": M, "L": L, "dY_MB": M * L * 2 / 1e6}
        for cold in (True, False):
            a = timed(sep, cold)
            r = {"separate_us": round(a, 1), "separate_TBps_dY": round(2 * M * L * 2 / a / 1e6, 2)}
            for ct in ("1", "2", "4", "0"):  # column tiles per workgroup: forced forms, then the launcher's own choice ("0")
                os.environ["AITK_LORA_BWD_CT"] = ct
                b = timed(fused, cold)
                r[f"fused_ct{ct if ct != '0' else '_auto'}_us"] = round(b, 1)
            os.environ.pop("AITK_LORA_BWD_CT", None)
            row["cold" if cold else "warm"] = r
        out.append(row)
        print(json.dumps(row), flush=True)
    with open("gpurun_out/r05_lora_bwd_fused_bench_ct.json", "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
