"""Wan2.1-T2V-1.3B LoRA r16 step at BASELINE config 4's shape (49 frames x 512^2 -> latents [B,16,13,64,64], 13 312 tokens,
UMT5 embeds [B,512,4096]) on one MI355X: synthetic weights / data, bf16, full step (noise mix .. AdamW).  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # the repo root, wherever the profiler starts us
import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import ops  # noqa: E402
from ai_toolkit_amd.lora import FusedLoRANetwork  # noqa: E402
from ai_toolkit_amd.trainer import WanLoRATrainStep  # noqa: E402
from ai_toolkit_amd.wan import WanTransformer3DModel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=13)
    ap.add_argument("--layers", type=int, default=30)
    a = ap.parse_args()
    dev = "cuda"
    torch.manual_seed(0)
    model = WanTransformer3DModel(num_layers=a.layers, dtype=torch.bfloat16, device=dev, ops=ops)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.ndim >= 2 and "scale_shift_table" not in n:
                p.normal_(0, 0.02)
            elif "scale_shift_table" in n:
                p.normal_(0, 1.0 / 1536 ** 0.5)
            elif n.endswith("bias"):
                p.normal_(0, 0.01)
    net = FusedLoRANetwork(model, lora_dim=16, target_lin_modules=("WanTransformer3DModel",), transformer_block_names=["blocks"],
                           base_model_version="wan_2.1")
    net.apply_to()
    net.build_arena(dev, groups=model.lora_groups())
    with torch.no_grad():
        for m in net.unet_loras:
            m.lora_up.weight.normal_(0, 0.01)
    net.refresh_shadows(ops)
    model.attach_network(net)
    model.prepare()
    step = WanLoRATrainStep(model, net, ops, lr=1e-4, seed=1)
    B = a.batch
    lat = torch.randn(B, 16, a.frames, 64, 64, device=dev).to(torch.bfloat16)
    txt = (torch.randn(B, 512, 4096, device=dev) * 0.3).to(torch.bfloat16)
    for _ in range(a.warmup):
        step.step(lat, txt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step.step(lat, txt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    S = a.frames * 32 * 32
    d, f, L = 1536, 8960, a.layers
    # matmul flops per sample: fwd 2*S*params_tok + attention; x3 for fwd+bwd
    lin = 2 * S * (8 * d * d + 2 * d * f) - 2 * S * 2 * d * d + 2 * 512 * 2 * d * d  # attn2 k/v run on 512 text tokens
    attn = 4 * S * S * d + 4 * S * 512 * d
    flops = 3 * L * (lin + attn)
    print(json.dumps({"workload": f"Wan2.1-T2V-1.3B LoRA r16 {a.frames} latent frames x 64x64, B={B}", "s_per_step": dt,
                      "samples_per_s": B / dt, "loss": float(loss), "adapters": len(net.unet_loras),
                      "lora_params_M": net.arena_p.numel() / 1e6, "tflops_per_sample": flops / 1e12,
                      "mfma_frac_of_2.5PF": flops * B / dt / 2.5e15,
                      "peak_mem_GiB": torch.cuda.max_memory_allocated() / 2 ** 30}))


if __name__ == "__main__":
    main()
