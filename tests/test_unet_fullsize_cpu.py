"""BASELINE config 1 at FULL size on the CPU (the plumbing configuration: SD1.5 UNet, LoRA r4, 512x512, batch 1, float32): the
859.5 M-parameter native graph (ai_toolkit_amd.unet, oracle kernel table) against autograd of the oracle UNet + oracle LoRA layer —
one complete train step (DDPM add_noise, eps target, MSE, clip, AdamW) with 192 adapters, and the kohya file it saves.
~1 minute and ~21 GB of host memory on 8 cores."""
import hashlib
import json
import os

import torch
from safetensors import safe_open
from safetensors.torch import load_file

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd.trainer import UNetLoRATrainStep
from oracle import ref_ops, train_ref, unet_ref
from tests.test_unet_cpu import build_pair


def test_sd15_full_size_train_step_and_kohya_file(tmp_path):
    ref, ref_net, nat, net = build_pair(unet_ref.SD15, rank=4, alpha=4.0)
    assert len(net.unet_loras) == 192 and sum(p.numel() for p in nat.parameters()) == 859_520_964
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(1, 4, 64, 64, generator=g)
    ctx = torch.randn(1, 77, 768, generator=g) * 0.5
    noise = torch.randn(1, 4, 64, 64, generator=g)
    ts = torch.tensor([500])
    kw = dict(lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    l_ref = train_ref.RefUNetTrainStep(ref, ref_net, **kw).step(lat, ctx, None, noise, ts).item()
    l = UNetLoRATrainStep(nat, net, ref_ops, **kw).step(lat, ctx, None, noise=noise, timesteps=ts).item()
    assert abs(l - l_ref) <= 1e-5 * abs(l_ref), (l, l_ref)
    worst = 0.0
    for a, b in zip(net.unet_loras, ref_net.unet_loras):  # weights after clip + AdamW (first step ~ lr * sign(g): atol = 1% of a step)
        for pa, pb in ((a.lora_down.weight, b.lora_down.weight), (a.lora_up.weight, b.lora_up.weight)):
            pb = pb.reshape(pa.shape)
            worst = max(worst, (pa - pb).abs().max().item())
    assert worst <= 2e-3 * 1e-3 * 20 + 1e-5, worst
    # gradients of the step (before the optimizer touched them) agree to fp32 summation order
    for a, b in list(zip(net.unet_loras, ref_net.unet_loras))[::24]:
        ga, gb = a.lora_up.weight.grad, b.lora_up.weight.grad.reshape(a.lora_up.weight.shape)
        assert ((ga - gb).norm() / (gb.norm() + 1e-12)).item() < 5e-4, a.lora_name
    # the kohya file: key set = the reference network's own (hash pinned by tests/golden/make_golden.py), values = the trained weights
    f = tmp_path / "sd15_full.safetensors"
    net.save_weights(str(f), dtype=torch.float16, metadata={"name": "cfg1"})
    sd = load_file(str(f))
    with safe_open(os.path.join(os.path.dirname(__file__), "golden", "unet_lora_tiny.safetensors"), "pt") as fh:
        meta = json.loads(fh.metadata()["meta"])["sd15_full"]
    names = [k[: -len(".alpha")] for k in net.get_state_dict(dtype=torch.float16) if k.endswith(".alpha")]
    assert len(sd) == 3 * 192 and hashlib.sha256("\n".join(names).encode()).hexdigest() == meta["names_sha256"]
    m0 = net.unet_loras[0]
    assert torch.equal(sd[f"{m0.lora_name}.lora_up.weight"][:, :, 0, 0], m0.lora_up.weight.detach().to(torch.float16))
    with safe_open(str(f), "pt") as fh:
        md = fh.metadata()
    assert md["format"] == "pt" and "sshs_model_hash" in md and "sshs_legacy_hash" in md
