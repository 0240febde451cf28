"""Micro-benchmark of aitk_lora_down over one FLUX layer input (M x K bf16) for the rank widths the graphs launch (16: single adapter, 48: q,k,v group, 64: q,k,v,proj_mlp
group), with and without the lo half of the split-precision projection: what does a launch cost beside the time its X stream takes?  python tools/gpu_lora_down_bench.py [M] [K]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ai_toolkit_amd  # noqa: F401,E402
from ai_toolkit_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
dev, bf = "cuda", torch.bfloat16
g = torch.Generator().manual_seed(0)
xs = [torch.randn(M, K, generator=g).to(bf).to(dev) for _ in range(6)]  # rotate operands: every launch streams X from HBM
res = {"M": M, "K": K}
for R in (16, 32, 48, 64):
    p32 = (torch.randn(R, K, generator=g) * 0.05).to(dev)
    hi = p32.to(bf)
    lo = (p32 - hi.float()).to(bf)
    out = torch.empty(M, 3 * R, dtype=bf, device=dev)
    for with_lo in (True, False):
        kw = dict(p_lo=lo) if with_lo else {}
        for i in range(6):
            ops.lora_down(xs[i % 6], hi, out, scale=0.5, split=16, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 60
        e0.record()
        for i in range(n):
            ops.lora_down(xs[i % 6], hi, out, scale=0.5, split=16, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        res[f"R{R}_{'hi+lo' if with_lo else 'hi'}"] = {"us": round(us, 1), "x_TBps": round(M * K * 2 / us / 1e6, 2)}
print(json.dumps(res))
