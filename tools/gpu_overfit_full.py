"""Full-size FLUX.1-dev architecture (19+38 blocks, random init), LoRA r16, B = 1, one fixed 1024^2 batch with a fixed noise / timestep draw:
N AdamW steps through the fused HIP step; prints the loss curve (it must fall: the adapters fit the fixed target)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ai_toolkit_amd.trainer import FluxLoRATrainStep  # noqa: E402

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
model, net, ops = bench.build_flux(dev, rank=16)
step = FluxLoRATrainStep(model, net, ops, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0, ema_decay=0.99, timestep_type="linear", seed=3)
lat, emb, pooled = bench.make_batch(dev, 1, seed=5)
noise = torch.randn(lat.shape, generator=torch.Generator().manual_seed(9)).to(lat.dtype).to(dev)
ts = torch.tensor([600.0], device=dev)
losses = []
t0 = time.time()
for i in range(n):
    losses.append(float(step.step(lat, emb, pooled, noise=noise, timesteps=ts).item()))
torch.cuda.synchronize()
out = {"steps": n, "seconds": round(time.time() - t0, 1), "loss_first": losses[0], "loss_last": losses[-1], "loss_min": min(losses),
       "curve_every_5": [round(v, 5) for v in losses[::5]], "finite": all(v == v and abs(v) < 1e30 for v in losses)}
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r03_overfit_full.json", "w"), indent=1)
