"""Shader clock and power while one kernel type runs back to back: does the attention forward run at a lower clock than a GEMM?
Polls rocm-smi from a thread while the main thread keeps the GPU busy for ~4 s per workload."""
import math
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import ops  # noqa: E402

samples = []
stop = False


def poll():
    while not stop:
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "-d", "0"], capture_output=True, text=True, timeout=5).stdout
            sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", r)
            pw = re.search(r"Power \(W\): ([\d.]+)", r)
            samples.append((time.time(), int(sclk.group(1)) if sclk else -1, float(pw.group(1)) if pw else -1.0))
        except Exception as e:  # noqa: BLE001
            samples.append((time.time(), -2, -2.0))
        time.sleep(0.05)


def run(name, fn, secs=4.0):
    global samples
    fn()
    torch.cuda.synchronize()
    samples = []
    t0 = time.time()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    mid = [s for s in samples if t0 + 1.0 < s[0] < t0 + secs]
    clk = sorted(s[1] for s in mid)
    pw = sorted(s[2] for s in mid)
    med = lambda v: v[len(v) // 2] if v else -1  # noqa: E731
    line = f"{name:34s} {ms:8.3f} ms/launch   sclk median {med(clk)} MHz (min {clk[0] if clk else -1}, max {clk[-1] if clk else -1}, {len(clk)} samples)   power median {med(pw)} W"
    print(line, flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/r03_clock_power_by_kernel.txt", "a") as fh:
        fh.write(line + "\n")


th = threading.Thread(target=poll, daemon=True)
th.start()
B, H, S = 4, 24, 4608
d = H * 128
torch.manual_seed(0)
q, k, v, do = [torch.randn(B * S, d, device="cuda").to(torch.bfloat16) for _ in range(4)]
o = torch.empty_like(q)
dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
lse = torch.empty(B, H, S, dtype=torch.float32, device="cuda")
sc = 1 / math.sqrt(128)
run("attention forward (B=4)", lambda: ops.attn_fwd(q, k, v, o, lse, B=B, H=H, S=S, scale=sc))
run("attention backward (B=4)", lambda: ops.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, B=B, H=H, S=S, scale=sc))
M = 32256
x = torch.randn(M, 3072, device="cuda").to(torch.bfloat16)
w = (torch.randn(3072, 3072, device="cuda") * 0.02).to(torch.bfloat16)
y = torch.empty(M, 3072, dtype=torch.bfloat16, device="cuda")
run("GEMM 32256x3072x3072", lambda: ops.gemm_nt(x, w, y))
w2 = (torch.randn(3072, 12288, device="cuda") * 0.02).to(torch.bfloat16)
x2 = torch.randn(M, 12288, device="cuda").to(torch.bfloat16)
run("GEMM 32256x3072x12288", lambda: ops.gemm_nt(x2, w2, y))
import ai_toolkit_amd.graph  # noqa: E402,F401
xq = torch.empty(M, 3072, dtype=torch.uint8, device="cuda")
xs = torch.empty(M, dtype=torch.float32, device="cuda")
ops.quant_rows_fp8(x, xq, xs, M=M)
wq = torch.empty(3072, 3072, dtype=torch.uint8, device="cuda")
ws = torch.empty(3072, dtype=torch.float32, device="cuda")
ops.quant_rows_fp8(w, wq, ws, M=3072)
run("W8A8 GEMM 32256x3072x3072 (fp8 MFMA)", lambda: ops.gemm_nt(xq, wq, y, a_scale=xs, b_scale=ws, b_scale_mode=3))
P = (torch.randn(16, 3072, device="cuda") * 0.05).to(torch.bfloat16)
T = torch.empty(M, 16, dtype=torch.bfloat16, device="cuda")
run("lora_down 32256x3072 (HBM-bound)", lambda: ops.lora_down(x, P, T, scale=0.5, M=M))
stop = True
