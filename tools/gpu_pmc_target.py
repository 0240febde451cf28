"""Small fixed workload for rocprofv3 --pmc passes: 4 launches each of the dominant GEMM shape and attention fwd/bwd."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import ops  # noqa: E402

dev = "cuda"
bf = torch.bfloat16
M, N, K = int(os.environ.get("AITK_PMC_M", "18432")), 3072, 3072  # 18432 = batch 4, 32256 = batch 7
a = torch.randn(M, K, device=dev).to(bf)
b = (torch.randn(N, K, device=dev) * 0.02).to(bf)
R2 = int(os.environ.get("AITK_PMC_K2", "48"))  # LoRA K-slab: 48 = rank 16 in the split [hi | lo | hi] layout (round 2), 16 = round 1
a2 = torch.randn(M, R2, device=dev).to(bf)
b2 = torch.randn(N, R2, device=dev).to(bf)
bias = torch.randn(N, device=dev).to(bf)
out = torch.empty(M, N, dtype=bf, device=dev)
for _ in range(4):
    ops.gemm_nt(a, b, out, bias=bias, a2=a2, b2=b2)
if os.environ.get("AITK_PMC_F8", "1") != "0":  # W8A8 (MX-scaled fp8 MFMA) GEMM of the same shape + the per-token quantisation that feeds it
    ws = (b.float().abs().amax(1) / 448.0).contiguous()
    wq = (b.float() / ws[:, None]).to(torch.float8_e4m3fn).view(torch.uint8).contiguous()
    xq, xs = torch.empty(M, K, dtype=torch.uint8, device=dev), torch.empty(M, device=dev)
    for _ in range(4):
        ops.quant_rows_fp8(a, xq, xs)
        ops.gemm_nt(xq, wq, out, bias=bias, a2=a2, b2=b2, a_scale=xs, b_scale=ws, b_scale_mode=3)
B, H, S = 1, 24, 4608
HD = H * 128
qkv = torch.randn(B * S, 3 * HD, device=dev).to(bf)
q, k, v = qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:]
o = torch.empty(B * S, HD, dtype=bf, device=dev)
do = torch.randn(B * S, HD, device=dev).to(bf)
dqkv = torch.empty_like(qkv)
lse = torch.empty(B, H, S, device=dev)
sc = 1 / math.sqrt(128)
for _ in range(4):
    ops.attn_fwd(q, k, v, o, lse, B=B, H=H, S=S, scale=sc)
    ops.attn_bwd(q, k, v, o, lse, do, dqkv[:, :HD], dqkv[:, HD:2 * HD], dqkv[:, 2 * HD:], B=B, H=H, S=S, scale=sc)
torch.cuda.synchronize()
print("done")
