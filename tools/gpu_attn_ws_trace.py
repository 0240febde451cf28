"""Barrier timeline of the wave-specialised dK/dV kernel (AITK_ATTN_DKDV_WS=2: the TRACE instantiation stamps s_memtime before and after every
workgroup barrier of query tiles 8-11 in workgroup 0, producer wave 0 and consumer wave 4).  Prints, per tile and barrier, how long each role
worked before arriving and how long it then waited — i.e. which half-step of which role is the long one."""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import _capi, ops  # noqa: E402

B, H, S = 7, 24, 4608
d = H * 128
q, k, v, do = [torch.randn(B * S, d, device="cuda").to(torch.bfloat16) for _ in range(4)]
o = torch.empty_like(q)
dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
lse = torch.empty(B, H, S, dtype=torch.float32, device="cuda")
sc = 1 / math.sqrt(128)
ops.attn_fwd(q, k, v, o, lse, B=B, H=H, S=S, scale=sc)
os.environ["AITK_ATTN_DKDV_WS"] = "2"
for _ in range(2):
    ops.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, B=B, H=H, S=S, scale=sc)
torch.cuda.synchronize()
buf = (C.c_uint64 * 64)()
L = _capi.lib()
L.aitk_probe_attn_ws_trace.argtypes = [C.c_void_p]
assert L.aitk_probe_attn_ws_trace(buf) == 0
v = list(buf)
names = ["H1 of even step", "H2 of even step", "H1 of odd step", "H2 of odd step"]
for role, rn in ((0, "producer"), (1, "consumer")):
    prev_after = None
    for t in range(4):
        for bidx in range(4):
            before, after = v[(role * 4 + t) * 8 + 2 * bidx], v[(role * 4 + t) * 8 + 2 * bidx + 1]
            work = None if prev_after is None else before - prev_after
            print(f"{rn} tile {8 + t} {names[bidx]:16s}: worked {work} ticks, waited at the barrier {after - before} ticks")
            prev_after = after
print("(s_memtime ticks: 100 MHz constant clock on gfx9 -> 10 ns per tick)")
