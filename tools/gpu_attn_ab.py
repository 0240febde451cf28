"""Attention fwd / bwd timing, one subprocess per library (ai-toolkit_amd/libaitk_abl_*.so) on the same box."""
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
    from ai_toolkit_amd import ops

    B, H, S = int(os.environ.get("AITK_AB_B", "4")), 24, 4608
    d = H * 128
    torch.manual_seed(0)
    q, k, v, do = [torch.randn(B * S, d, device="cuda").to(torch.bfloat16) for _ in range(4)]
    o = torch.empty_like(q)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    lse = torch.empty(B, H, S, dtype=torch.float32, device="cuda")
    sc = 1 / math.sqrt(128)

    def t(fn, n=5):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts)[n // 2]

    fw = t(lambda: ops.attn_fwd(q, k, v, o, lse, B=B, H=H, S=S, scale=sc))
    ops.attn_fwd(q, k, v, o, lse, B=B, H=H, S=S, scale=sc)
    lse.clamp_(-50.0, 50.0)  # ablation libraries produce meaningless statistics: keep the backward's exp2 arguments finite
    bw = t(lambda: ops.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, B=B, H=H, S=S, scale=sc))
    fl = 4.0 * S * S * 128 * B * H
    # bit-level fingerprint of the gradients (identical accumulation order in every valid build => identical bits)
    chk = [float(t_.view(torch.int16).to(torch.int64).sum().item()) for t_ in (o, dq, dk, dv)]
    print("RESULT", json.dumps({"fwd_ms": round(fw, 3), "bwd_ms": round(bw, 3), "fwd_tflops": round(fl / fw / 1e9, 1),
                                "bwd_tflops_alg": round(2.5 * fl / bw / 1e9, 1), "chk": chk}))
else:
    res = {}
    variants = [("product", {}), ("dkdv_wave_specialised", {"AITK_ATTN_DKDV_WS": "1"}), ("dkdv_pipelined", {"AITK_ATTN_DKDV_WS": "0"})]
    variants += [(os.path.basename(l), {"AITK_LIB_PATH": l}) for l in sorted(glob.glob(os.path.join(ROOT, "ai-toolkit_amd", "libaitk_abl_attn*.so")))]
    for rep in range(2):
        for name, extra in variants:
            env = dict(os.environ, **extra)
            r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True, timeout=200)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
            res[f"{name}#{rep}"] = json.loads(line[0][7:]) if line else r.stderr[-300:]
            print(name, rep, res[f"{name}#{rep}"], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "attn_ab.json"), "w"), indent=1)
