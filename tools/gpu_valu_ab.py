"""Same-box A/B of the round-5 VALU variants (tools/build_ab_round5.py), one subprocess per library through AITK_LIB_PATH, alternating, two rounds:
the GELU / dGELU GEMM launches of the FLUX step (M = 32256, 12288 x 3072 + slab) + a bias launch as the control, and attention forward / backward at
B = 4, H = 24, S = 4608.  Prints one JSON line per (library, round) and a summary; gpurun_out/valu_ab.json.

    python tools/gpu_valu_ab.py            # every libaitk_abl_{gelu,attn_fwd}*.so present + the product
"""
import glob
import json
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch

    import ai_toolkit_amd  # noqa: F401
    from ai_toolkit_amd import _capi, ops

    bf = torch.bfloat16
    what = sys.argv[2]

    def timeit(fn, n=12):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / n)
        return sorted(ts)[1]

    out = {}
    if what == "gemm":
        g = torch.Generator(device="cuda").manual_seed(1)
        M, N, K = 32256, 12288, 3072
        x = torch.randn(M, K, device="cuda", generator=g).to(bf)
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(bf)
        a2 = (torch.randn(M, 48, device="cuda", generator=g) * 0.1).to(bf)
        b2 = (torch.randn(N, 48, device="cuda", generator=g) * 0.1).to(bf)
        bias = torch.randn(N, device="cuda", generator=g).to(bf)
        aux = torch.randn(M, N, device="cuda", generator=g).to(bf)
        o, u = torch.empty(M, N, dtype=bf, device="cuda"), torch.empty(M, N, dtype=bf, device="cuda")
        fl = 2.0 * M * N * (K + 48)
        for name, kw in (("bias", dict(bias=bias, a2=a2, b2=b2)), ("gelu", dict(bias=bias, flags=_capi.EPI_GELU, aux_out=u, a2=a2, b2=b2)),
                         ("dgelu", dict(flags=_capi.EPI_DGELU, aux_in=aux, a2=a2, b2=b2))):
            ms = timeit(lambda: ops.gemm_nt(x, w, o, **kw))
            out[name + "_ms"] = round(ms, 4)
            out[name + "_tflops"] = round(fl / ms / 1e9, 1)
            out[name + "_chk"] = float(o.view(torch.int16).to(torch.int64).sum().item())
    else:
        B, H, S = 4, 24, 4608
        d = H * 128
        torch.manual_seed(0)
        q, k, v, do = [torch.randn(B * S, d, device="cuda").to(bf) for _ in range(4)]
        o = torch.empty_like(q)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        lse = torch.empty(B, H, S, dtype=torch.float32, device="cuda")
        sc = 1 / math.sqrt(128)
        fw = timeit(lambda: ops.attn_fwd(q, k, v, o, lse, B=B, H=H, S=S, scale=sc))
        bw = timeit(lambda: ops.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, B=B, H=H, S=S, scale=sc), n=6)
        fl = 4.0 * S * S * 128 * B * H
        out = {"fwd_ms": round(fw, 4), "bwd_ms": round(bw, 4), "fwd_tflops": round(fl / fw / 1e9, 1), "bwd_tflops_alg": round(2.5 * fl / bw / 1e9, 1),
               "chk": [float(t_.view(torch.int16).to(torch.int64).sum().item()) for t_ in (o, dq, dk, dv)],
               "lse_sum": float(lse.double().sum().item())}
    print("RESULT", json.dumps(out))
else:
    lib = lambda n: os.path.join(ROOT, "ai-toolkit_amd", n)  # noqa: E731
    gemm_v = [("product", {})] + [(os.path.basename(p), {"AITK_LIB_PATH": p}) for p in sorted(glob.glob(lib("libaitk_abl_gelu*.so")))]
    attn_v = [("product", {})] + [(os.path.basename(p), {"AITK_LIB_PATH": p}) for p in sorted(glob.glob(lib("libaitk_abl_attn_fwd*.so")))]
    res = {}
    for rep in range(2):
        for what, variants in (("gemm", gemm_v), ("attn", attn_v)):
            for name, extra in variants:
                r = subprocess.run([sys.executable, __file__, "child", what], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=300)
                line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
                key = f"{what}/{name}#{rep}"
                res[key] = json.loads(line[0][7:]) if line else {"error": r.stderr[-400:]}
                print(key, json.dumps(res[key]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "valu_ab.json"), "w"), indent=1)
