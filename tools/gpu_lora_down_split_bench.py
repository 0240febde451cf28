"""aitk_lora_down in split precision (P = hi + lo, T written as the [hi | lo | hi] slab) on the row counts of the B = 7 FLUX step:
bytes of X streamed / time.  AITK_LORA_DOWN_U=8 selects the 3-waves-per-SIMD variant of the rank-16 kernel."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import ops  # noqa: E402

dev, bf = "cuda", torch.bfloat16


def t(fn, n=9):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[n // 2]


out = {"AITK_LORA_DOWN_U": os.environ.get("AITK_LORA_DOWN_U", "default")}
for (M, K, R) in ((32256, 3072, 16), (28672, 3072, 16), (32256, 12288, 16), (32256, 15360, 16), (3584, 3072, 16), (32256, 3072, 64), (28672, 3072, 48)):
    x = torch.randn(M, K, device=dev).to(bf)
    w = torch.randn(R, K, device=dev) * 0.05
    hi = w.to(bf)
    lo = (w - hi.float()).to(bf)
    rp = 16
    T = torch.empty(M, 3 * R, dtype=bf, device=dev)
    us = t(lambda: ops.lora_down(x, hi, T, scale=0.5, M=M, p_lo=lo, split=rp))
    ref = 0.5 * (x[:256].float() @ w.t())
    got = torch.cat([T[:256, 3 * b * rp:3 * b * rp + rp].float() + T[:256, 3 * b * rp + rp:3 * b * rp + 2 * rp].float() for b in range(R // rp)], 1)
    err = ((got - ref).norm() / ref.norm()).item()
    key = f"down_split_{M}x{K}_r{R}"
    out[key] = {"us": round(us, 1), "TBps": round(M * K * 2 / us / 1e6, 2), "rel_err_vs_fp32": err}
    print(key, out[key], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/lora_down_split_{out['AITK_LORA_DOWN_U']}.json", "w"), indent=1)
