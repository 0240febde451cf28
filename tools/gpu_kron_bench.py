"""Timing of aitk_kron_apply on the FLUX LoKr shapes (M = 4 x 4608 tokens): microseconds and effective HBM GB/s
(bytes = input row + output row per token, bf16)."""
import sys

import torch

sys.path.insert(0, ".")
import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import ops  # noqa: E402

BF = torch.bfloat16
M = 18432
CASES = [  # name, a_in, b_in, a_out, b_out, hasA, hasB, transpose
    ("fwd 3072->3072", 48, 64, 48, 64, 1, 1, 0),
    ("fwd 3072->12288", 48, 64, 96, 128, 1, 1, 0),
    ("fwd 12288->3072", 96, 128, 48, 64, 1, 1, 0),
    ("fwd 15360->3072", 120, 128, 48, 64, 1, 1, 0),
    ("tmpT (I,B) 3072", 48, 64, 48, 64, 0, 1, 1),
    ("U (A,I) 3072", 48, 64, 48, 64, 1, 0, 0),
    ("dyT (I,I) 3072", 48, 64, 48, 64, 0, 0, 1),
    ("dyT (I,I) 12288", 96, 128, 96, 128, 0, 0, 1),
    ("U (A,I) 12288->in 3072", 96, 128, 48, 128, 1, 0, 0),
]
for name, ai, bi, ao, bo, hA, hB, tr in CASES:
    x = torch.randn(M, ai * bi, device="cuda").to(BF)
    A = (torch.randn(ao, ai, device="cuda") / ai ** 0.5).to(BF) if hA else None
    Bm = (torch.randn(bo, bi, device="cuda") / bi ** 0.5).to(BF) if hB else None
    out = torch.empty(M, ao * bo, device="cuda", dtype=BF)
    kw = dict(a_in=ai, b_in=bi, a_out=ao, b_out=bo, transpose_out=bool(tr))
    for _ in range(3):
        ops.kron_apply(x, A, Bm, out, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.kron_apply(x, A, Bm, out, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    gb = M * (ai * bi + ao * bo) * 2 / 1e9
    print(f"{name:28s} {us:8.1f} us  {gb / (us * 1e-6):7.0f} GB/s")
# factor-gradient reductions (aitk_lora_wgrad on M * factor rows)
for name, rows, R, L in (("dw1 3072 (rows M*64, 48x48)", M * 64, 48, 48), ("dw2 3072 (rows M*48, 64x64)", M * 48, 64, 64),
                         ("dw2 12288->3072 (rows M*96, 64x128)", M * 96, 64, 128)):
    s = torch.randn(rows, R, device="cuda").to(BF)
    g = torch.randn(rows, L, device="cuda").to(BF)
    o = torch.zeros(R, L, device="cuda")
    for _ in range(2):
        ops.lora_wgrad(s, g, o, accumulate=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.lora_wgrad(s, g, o, accumulate=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"{name:40s} {us:8.1f} us  {rows * (R + L) * 2 / 1e9 / (us * 1e-6):7.0f} GB/s")
