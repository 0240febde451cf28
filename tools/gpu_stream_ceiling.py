"""What does a plain streaming kernel reach on this box?  Reference points for the HBM-bound kernels of the step (lora_down, lora_wgrad,
qkv_post, ln_mod, gate_bwd all sit at 2.7-4.6 TB/s in-step): torch's copy (read + write), fill (write only), and a read-only reduction
on step-sized buffers, cold (a 1-GB buffer is touched in between so nothing is left in the 256-MB Infinity Cache) and warm."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

dev = "cuda"
out = {}


def timed(fn, flush=None, n=9):
    ts = []
    for _ in range(n):
        if flush is not None:
            flush()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[n // 2]


big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)


def flush():
    big.fill_(1)


for rows, cols in ((32256, 3072), (32256, 12288)):
    x = torch.randn(rows, cols, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x)
    nbytes = x.numel() * 2
    for tag, fl in (("cold", flush), ("warm", None)):
        us = timed(lambda: y.copy_(x), fl)
        out[f"copy_{rows}x{cols}_{tag}"] = {"us": round(us, 1), "TBps_rw": round(2 * nbytes / us / 1e6, 2)}
        us = timed(lambda: y.fill_(0.5), fl)
        out[f"fill_{rows}x{cols}_{tag}"] = {"us": round(us, 1), "TBps_w": round(nbytes / us / 1e6, 2)}
        us = timed(lambda: x.view(torch.int16).sum(dtype=torch.int32) if False else torch.amax(x), fl)
        out[f"amax_{rows}x{cols}_{tag}"] = {"us": round(us, 1), "TBps_r": round(nbytes / us / 1e6, 2)}
    # ours on the same buffer: lora_down (read-only stream + tiny write), cold and warm
    import ai_toolkit_amd  # noqa: F401
    from ai_toolkit_amd import ops

    P = (torch.randn(16, cols, device=dev) * 0.05).to(torch.bfloat16)
    T = torch.empty(rows, 16, dtype=torch.bfloat16, device=dev)
    for tag, fl in (("cold", flush), ("warm", None)):
        us = timed(lambda: ops.lora_down(x, P, T, scale=0.5, M=rows), fl)
        out[f"lora_down_{rows}x{cols}_{tag}"] = {"us": round(us, 1), "TBps_r": round(nbytes / us / 1e6, 2)}
        g = torch.zeros(16, cols, dtype=torch.float32, device=dev)
        us = timed(lambda: ops.lora_wgrad(T, x, g, accumulate=True, M=rows), fl)
        out[f"lora_wgrad_{rows}x{cols}_{tag}"] = {"us": round(us, 1), "TBps_r": round(nbytes / us / 1e6, 2)}
    del x, y
for k, v in out.items():
    print(k, v, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r03_stream_ceiling.json", "w"), indent=1)
