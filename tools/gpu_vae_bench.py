"""Full-size FLUX.1 VAE encoder (AutoencoderKL 128/256/512/512, 1024x1024 -> 16x128x128 latents) timing on MI355X."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import ops  # noqa: E402
from ai_toolkit_amd import vae as nvae  # noqa: E402

dev = "cuda"
enc = nvae.AutoencoderKLEncoder(dtype=torch.bfloat16, device=dev, ops=ops)
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    for name, p in enc.named_parameters():
        if name.endswith("weight") and p.dim() > 1:
            fan = p[0].numel()
            p.copy_((torch.randn(p.shape, device=dev, generator=g) * fan ** -0.5).to(p.dtype))
enc.prepare()
out = {}
for B in (1, 2, 4):
    img = torch.rand(B, 3, 1024, 1024, device=dev, generator=g) * 2 - 1
    for _ in range(2):
        lat = enc.encode_images(img, generator=g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        lat = enc.encode_images(img, generator=g)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    out[f"B{B}"] = {"ms": 1e3 * dt, "images_per_s": B / dt, "finite": bool(torch.isfinite(lat.float()).all()), "shape": list(lat.shape),
                    "tflops_est": 4.6e12 * B / dt / 1e12}
    print(B, out[f"B{B}"], flush=True)
out["peak_mem_GiB"] = torch.cuda.max_memory_allocated() / 2 ** 30
os.makedirs("gpurun_out", exist_ok=True)
out["conv8"] = os.environ.get("AITK_CONV8", "1") != "0"
json.dump(out, open(os.environ.get("AITK_VAE_BENCH_OUT", "gpurun_out/vae_bench.json"), "w"), indent=1)
print(json.dumps(out))
