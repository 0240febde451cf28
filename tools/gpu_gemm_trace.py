"""Cycle trace of the 8-phase GEMM loop (debug library built with -DAITK_GEMM_TRACE, loaded through AITK_LIB_PATH):
lane 0 of waves 0 (group 0) and 4 (group 1) of one workgroup stamps s_memtime at every segment boundary of K-tiles 8..11."""
import ctypes as C
import json
import os
import sys

os.environ["AITK_LIB_PATH"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ai-toolkit_amd", "libaitk_trace.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import _capi, ops  # noqa: E402

dev = "cuda"
M, N, K, r = 18432, 3072, 3072, 16
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
b = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
a2 = torch.randn(M, r, device=dev).to(torch.bfloat16)
b2 = torch.randn(N, r, device=dev).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
trace = torch.zeros(512, dtype=torch.float32, device=dev)  # 256 x u64
lib = _capi.lib()
real = lib.aitk_gemm_nt
BLOCK = int(sys.argv[1]) if len(sys.argv) > 1 else 300


class Hook:
    def __call__(self, argref, stream):
        g = argref._obj
        g.b_scale = C.c_void_p(trace.data_ptr())
        g._pad4 = BLOCK + 1
        return real(argref, stream)


for _ in range(3):
    ops.gemm_nt(a, b, out, a2=a2, b2=b2, tile_mode=2, stage_mode=4)
torch.cuda.synchronize()
lib.aitk_gemm_nt = Hook()
ops.gemm_nt(a, b, out, a2=a2, b2=b2, tile_mode=2, stage_mode=4)
torch.cuda.synchronize()
lib.aitk_gemm_nt = real
t = trace.view(torch.int64).cpu().tolist()
names = ["L0", "a0", "M0", "b0", "L1", "a1", "M1", "b1", "L2", "a2", "M2", "b2", "L3", "a3", "M3", "b3"]
res = {}
for grp in (0, 1):
    st = t[grp * 128: grp * 128 + 64]
    base = st[0]
    rel = [x - base for x in st]
    res[f"group{grp}_abs0"] = base
    res[f"group{grp}"] = rel
    print("group", grp, "start", base)
    for tile in range(4):
        row = rel[tile * 16:(tile + 1) * 16]
        # segment durations: L (reads+stage issue), wait at barrier a, lgkm wait+MFMA issue, barrier b
        segs = [row[i + 1] - row[i] for i in range(15)]
        print(" tile", 8 + tile, " ".join(f"{n}:{d}" for n, d in zip(names[:-1], segs)), "| tile total", (rel[(tile + 1) * 16] - row[0]) if tile < 3 else "")
print("group1 - group0 start offset:", res["group1_abs0"] - res["group0_abs0"])
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/gemm_trace.json", "w"))
