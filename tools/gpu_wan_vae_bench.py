"""Full-size Wan2.1 VAE encoder (AutoencoderKLWan 96/192/384/384; BASELINE config 4's clip: 49 frames of 512x512 -> 16x13x64x64 latents)
timing on MI355X.  FLOPs are counted from the convolution / GEMM shapes the graph launches (2*M*N*K each)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import ops  # noqa: E402
from ai_toolkit_amd import wan_vae as nwv  # noqa: E402

dev = "cuda"
enc = nwv.AutoencoderKLWanEncoder(dtype=torch.bfloat16, device=dev, ops=ops)
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    for name, p in enc.named_parameters():
        if name.endswith("weight"):
            p.copy_((torch.randn(p.shape, device=dev, generator=g) * p[0].numel() ** -0.5).to(p.dtype))
enc.prepare()

flops = [0.0]
_conv3d, _conv3x3, _gemm = ops.conv3d, ops.conv3x3, ops.gemm_nt


def c3d(x, w, out, **kw):
    flops[0] += 2.0 * out.shape[0] * w.shape[0] * w.shape[1]
    return _conv3d(x, w, out, **kw)


def c2d(x, w, out, **kw):
    flops[0] += 2.0 * out.shape[0] * w.shape[0] * w.shape[1]
    return _conv3x3(x, w, out, **kw)


def gm(a, b, out, **kw):
    flops[0] += 2.0 * a.shape[0] * b.shape[0] * a.shape[1]
    return _gemm(a, b, out, **kw)


class CountingOps:
    def __getattr__(self, k):
        return {"conv3d": c3d, "conv3x3": c2d, "gemm_nt": gm}.get(k) or getattr(ops, k)


res = {}
T, H, W = (int(v) for v in os.environ.get("AITK_WANVAE_SHAPE", "49,512,512").split(","))
clip = torch.rand(T, 3, H, W, device=dev, generator=g) * 2 - 1
enc.ops = CountingOps()
lat = enc.encode_images([clip], generator=g)
torch.cuda.synchronize()
fl = flops[0]
enc.ops = ops
for _ in range(2):
    lat = enc.encode_images([clip], generator=g)
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n):
    lat = enc.encode_images([clip], generator=g)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
res = {"clip": [T, H, W], "ms": 1e3 * dt, "clips_per_s": 1 / dt, "frames_per_s": T / dt, "finite": bool(torch.isfinite(lat.float()).all()),
       "shape": list(lat.shape), "tflop_per_clip": fl / 1e12, "tflops": fl / dt / 1e12, "peak_mem_GiB": torch.cuda.max_memory_allocated() / 2 ** 30}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/r02_wan_vae_bench.json", "w"), indent=1)
print(json.dumps(res))
