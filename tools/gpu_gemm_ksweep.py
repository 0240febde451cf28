"""Per-output-tile fixed cost of the persistent 8-phase GEMM: time(K) at fixed M x N for the bf16 and the W8A8 kernels (affine fit:
slope = time per K-tile, intercept = prologue + epilogue + tile switch), epilogue variants at K = 3072.  Prints JSON."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import _capi, ops  # noqa: E402

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


def main():
    M, N = 32256, 3072
    g = torch.Generator(device="cuda").manual_seed(1)
    res = {"M": M, "N": N, "rounds": (M // 256) * (N // 256) / 256.0, "bf16": {}, "f8": {}}
    a2 = (torch.randn(M, 48, device="cuda", generator=g) * 0.1).to(bf)
    b2 = (torch.randn(N, 48, device="cuda", generator=g) * 0.1).to(bf)
    o = torch.empty(M, N, dtype=bf, device="cuda")
    for K in (1024, 2048, 3072, 4096, 6144, 12288):
        x = torch.randn(M, K, device="cuda", generator=g).to(bf)
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(bf)
        xq = (torch.randint(0, 120, (M, K), device="cuda", generator=g, dtype=torch.int32)).to(torch.uint8)
        wq = (torch.randint(0, 120, (N, K), device="cuda", generator=g, dtype=torch.int32)).to(torch.uint8)
        xs, ws = torch.ones(M, device="cuda"), torch.ones(N, device="cuda")
        for slab in (True, False):
            kw = dict(a2=a2, b2=b2) if slab else {}
            key = f"K{K}" + ("" if slab else "_noslab")
            res["bf16"][key] = round(timeit(lambda: ops.gemm_nt(x, w, o, **kw)), 4)
            res["f8"][key] = round(timeit(lambda: ops.gemm_nt(xq, wq, o, a_scale=xs, b_scale=ws, b_scale_mode=3, **kw)), 4)
        del x, w, xq, wq
    print(json.dumps(res))
    for kind in ("bf16", "f8"):
        ks = [1024, 2048, 3072, 4096, 6144, 12288]
        ts = [res[kind][f"K{k}"] for k in ks]
        n = len(ks)
        mx, my = sum(ks) / n, sum(ts) / n
        slope = sum((k - mx) * (t - my) for k, t in zip(ks, ts)) / sum((k - mx) ** 2 for k in ks)
        icpt = my - slope * mx
        ktile = 64 if kind == "bf16" else 128
        print(kind, "ms per launch = %.4f + %.6f * K  -> per tile round (%.2f rounds): fixed %.1f us, per K-tile %.3f us" %
              (icpt, slope, res["rounds"], 1e3 * icpt / 6, 1e3 * slope * ktile / 6))


if __name__ == "__main__":
    main()
