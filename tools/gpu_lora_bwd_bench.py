"""Micro-benchmark of the skinny backward launches over one FLUX layer's dY / X (operands rotated so every launch streams from HBM):
aitk_lora_bwd_fused (dT + lora_up gradient from one read of dY) and aitk_lora_wgrad (lora_down gradient) at the widths the step has.
python tools/gpu_lora_bwd_bench.py [M]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ai_toolkit_amd  # noqa: F401,E402
from ai_toolkit_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32256
dev, bf = "cuda", torch.bfloat16
g = torch.Generator().manual_seed(0)
res = {"M": M}


def timed(fn, n=40):
    for i in range(4):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for L in (3072, 12288):
    nrot = 6 if L == 3072 else 3
    dys = [torch.randn(M, L, generator=g).to(bf).to(dev) for _ in range(nrot)]
    for R in (16, 64):
        T = torch.randn(M, 3 * R, generator=g).to(bf).to(dev)
        if R == 16:
            p32 = (torch.randn(R, L, generator=g) * 0.05).to(dev)
            hi = p32.to(bf)
            lo = (p32 - hi.float()).to(bf)
            dT = torch.empty(M, 3 * R, dtype=bf, device=dev)
            g_up = torch.zeros(L, R, device=dev)
            us = timed(lambda i: ops.lora_bwd_fused(dys[i % nrot], T, hi, lo, dT, g_up, scale=0.5, M=M, split=16))
            res[f"bwd_fused_L{L}_R16"] = {"us": round(us, 1), "TBps": round(M * L * 2 / us / 1e6, 2)}
        g_dn = torch.zeros(R, L, device=dev)
        us = timed(lambda i: ops.lora_wgrad(T, dys[i % nrot], g_dn, accumulate=True, M=M, split=16))
        res[f"wgrad_L{L}_R{R}"] = {"us": round(us, 1), "TBps": round(M * L * 2 / us / 1e6, 2)}
    del dys
print(json.dumps(res))
