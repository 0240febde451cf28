"""Does a consumer find freshly WRITTEN data in the Infinity Cache?  lora_down reading X [32256 x 3072] (198 MB, fits the 256-MB cache):
(a) right after another READ of X, (b) right after a kernel that WROTE X (torch copy; our ln_mod_fwd), (c) after an unrelated 1-GB fill.
In the training step every HBM-bound kernel is case (b)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import ops  # noqa: E402

dev = "cuda"
M, K = 32256, 3072
src = torch.randn(M, K, device=dev).to(torch.bfloat16)
x = torch.empty_like(src)
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
P = (torch.randn(16, K, device=dev) * 0.05).to(torch.bfloat16)
T = torch.empty(M, 16, dtype=torch.bfloat16, device=dev)
shift = torch.zeros(7, K, device=dev, dtype=torch.bfloat16)
scale = torch.zeros(7, K, device=dev, dtype=torch.bfloat16)


def timed(pre, n=9):
    ts = []
    for _ in range(n):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.lora_down(x, P, T, scale=0.5, M=M)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[n // 2]


out = {}
x.copy_(src)
cases = {
    "after a read of X (lora_down twice)": lambda: ops.lora_down(x, P, T, scale=0.5, M=M),
    "after torch wrote X (copy_)": lambda: x.copy_(src),
    "after ln_mod_fwd wrote X": lambda: ops.ln_mod_fwd(src, shift, scale, x, rows_per_batch=4608),
    "after an unrelated 1-GB fill": lambda: big.fill_(1),
    "after torch wrote X, then a 64-MB unrelated read": lambda: (x.copy_(src), big[: 64 << 20].sum()),
}
for name, pre in cases.items():
    us = timed(pre)
    out[name] = {"us": round(us, 1), "TBps": round(M * K * 2 / us / 1e6, 2)}
    print(f"{name:52s} {us:7.1f} us  {M * K * 2 / us / 1e6:5.2f} TB/s", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r03_producer_consumer.json", "w"), indent=1)
