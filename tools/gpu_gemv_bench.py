"""Timing of aitk_gemv_nt on the adaLN projection shapes (weight streaming: N x 3072 bf16)."""
import sys

import torch

sys.path.insert(0, ".")
import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import ops  # noqa: E402

for Bm in (4, 7):
    for N in (18432, 9216, 3072):
        K = 3072
        x = torch.randn(Bm, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
        b = torch.randn(N, device="cuda").to(torch.bfloat16)
        o = torch.empty(Bm, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(3):
            ops.gemv_nt(x, w, o, bias=b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemv_nt(x, w, o, bias=b)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        print(f"Bm={Bm} N={N:6d}: {us:7.1f} us  {N * K * 2 / 1e9 / (us * 1e-6):6.0f} GB/s")
