"""Reference-equivalent PyTorch-ROCm EAGER step on the same GPU (comparator, not the product): the oracle's plain-PyTorch
FLUX.1-dev (bf16 base, F.scaled_dot_product_attention) + the oracle restatement of the reference LoRA modules (fp32 adapter, fp32
activation copy per wrapped Linear, toolkit/network_mixins.py:304-342) + autograd + clip_grad_norm_ + torch.optim.AdamW.
No gradient checkpointing (the reference's default recomputes every block, which would make it slower still)."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from oracle import flux_ref, lora_ref  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = "cuda"
torch.manual_seed(0)
torch.set_default_dtype(torch.bfloat16)
with torch.device(dev):
    model = flux_ref.FluxTransformer2DModel()
torch.set_default_dtype(torch.float32)
with torch.no_grad():
    for m in model.modules():
        if isinstance(m, torch.nn.Linear):
            m.weight.normal_(0, 0.02)
            if m.bias is not None:
                m.bias.zero_()
for p in model.parameters():
    p.requires_grad_(False)
net = lora_ref.RefLoRANetwork(model, 16).to(dev)
net.torch_multiplier = net.torch_multiplier.to(dev)
with torch.no_grad():
    for m in net.unet_loras:
        m.lora_up.weight.normal_(0, 1e-3)
net.apply_to()
params = [p for m in net.unet_loras for p in (m.lora_down.weight, m.lora_up.weight)]
opt = torch.optim.AdamW(params, lr=1e-4, eps=1e-6, weight_decay=0.01)
g = torch.Generator(device=dev).manual_seed(1)
lat = torch.randn(B, 16, 128, 128, device=dev, generator=g).to(torch.bfloat16)
emb = (torch.randn(B, 512, 4096, device=dev, generator=g) * 0.1).to(torch.bfloat16)
pooled = (torch.randn(B, 768, device=dev, generator=g) * 0.1).to(torch.bfloat16)
img_ids, txt_ids = flux_ref.make_ids(128, 128, 512, dev)
guid = torch.ones(B, device=dev)


def step():
    noise = torch.randn_like(lat)
    t = torch.rand(B, device=dev)
    noisy = ((1 - t.view(B, 1, 1, 1)) * lat.float() + t.view(B, 1, 1, 1) * noise.float()).to(torch.bfloat16)
    opt.zero_grad(set_to_none=True)
    with net:
        pred = flux_ref.unpack_latents(model(flux_ref.pack_latents(noisy), emb, pooled, t, img_ids, txt_ids, guid), 128, 128)
        loss = torch.nn.functional.mse_loss(pred.float(), (noise.float() - lat.float()))
        loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 1.0)
    opt.step()
    return loss


step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 3
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("EAGER", json.dumps({"per_gpu_batch": B, "ms_per_step": round(dt * 1e3, 1), "images_per_s": round(B / dt, 3), "loss": float(loss.detach()),
                           "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))
