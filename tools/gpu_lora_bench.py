"""lora_down / lora_wgrad bandwidth on the step's shapes (bytes streamed / time)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import ops  # noqa: E402
from oracle import ref_ops  # noqa: E402

dev = "cuda"
bf = torch.bfloat16


def t(fn, n=7):
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


out = {}
for (M, K, R) in ((18432, 3072, 16), (18432, 3072, 64), (18432, 12288, 16), (18432, 15360, 16), (16384, 3072, 48), (2048, 3072, 16), (4, 3072, 16)):
    x = torch.randn(M, K, device=dev).to(bf)
    P = (torch.randn(R, K, device=dev) * 0.05).to(bf)
    T = torch.empty(M, R, dtype=bf, device=dev)
    Tr = torch.empty(M, R, dtype=bf, device=dev)
    us = t(lambda: ops.lora_down(x, P, T, scale=0.5, M=M))
    ref_ops.lora_down(x, P, Tr, scale=0.5, M=M)
    err = ((T.float() - Tr.float()).norm() / Tr.float().norm()).item()
    out[f"down_{M}x{K}_r{R}"] = {"us": round(us, 1), "TBps": round(M * K * 2 / us / 1e6, 2), "rel_err": err}
    print(f"down_{M}x{K}_r{R}", out[f"down_{M}x{K}_r{R}"], flush=True)
    # wgrad: dA[R,K] += dT^T x  (S = T, G = x)
    g = torch.zeros(R, K, dtype=torch.float32, device=dev)
    us = t(lambda: ops.lora_wgrad(T, x, g, accumulate=True, M=M))
    out[f"wgrad_{M}x{K}_r{R}"] = {"us": round(us, 1), "TBps": round(M * K * 2 / us / 1e6, 2)}
    print(f"wgrad_{M}x{K}_r{R}", out[f"wgrad_{M}x{K}_r{R}"], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/lora_bench.json", "w"), indent=1)
