"""Experiment: does carrying the residual stream (hidden_states / encoder_hidden_states and their gradients) in
fp32 across the blocks move the bf16 path's adapter-gradient error towards north_star's 1e-3?

Runs on CPU or GPU with the ORACLE kernel table (plain torch, one rounding per op output = the HIP kernels' rounding points), so the
answer is about the arithmetic, not about a kernel:  fp32 truth (oracle autograd)  vs  rm16 (our graph, bf16 storage)  vs
rm16 + fp32 residual stream (`set_precision("high")`)  vs  ref16 (the reference's arithmetic: bf16 modules + fp32 adapter).
Test infrastructure: imports oracle/.
"""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bf = torch.bfloat16


def rel_lists(a, b):
    num = sum(((x.float() - y.float()) ** 2).sum().item() for x, y in zip(a, b))
    den = sum((y.float() ** 2).sum().item() for y in b)
    return math.sqrt(num / max(den, 1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--double", type=int, default=4)
    ap.add_argument("--single", type=int, default=8)
    ap.add_argument("--heads", type=int, default=3)
    ap.add_argument("--hl", type=int, default=16)
    ap.add_argument("--wl", type=int, default=12)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--device", default="cpu")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import ai_toolkit_amd  # noqa: F401
    from ai_toolkit_amd.flux import FluxTransformer2DModel
    from ai_toolkit_amd.lora import FusedLoRANetwork
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from oracle import flux_ref, lora_ref, ref_ops, train_ref

    dev = a.device
    cfg = dict(in_channels=64, num_layers=a.double, num_single_layers=a.single, attention_head_dim=128, num_attention_heads=a.heads,
               joint_attention_dim=256, pooled_projection_dim=64)
    torch.manual_seed(0)
    ref = flux_ref.FluxTransformer2DModel(**cfg)
    flux_ref.init_synthetic_(ref, seed=1234, std=0.03)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if n.endswith("bias"):
                p.copy_(torch.randn_like(p) * 0.02)
            if "norm_" in n and n.endswith("weight"):
                p.copy_(1 + 0.1 * torch.randn_like(p))
            p.copy_(p.to(bf).float())
    ref = ref.to(dev)
    ref_net = lora_ref.RefLoRANetwork(ref, 16).to(dev)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for b in ref_net.unet_loras:
            b.lora_up.weight.copy_(torch.randn(b.lora_up.weight.shape, generator=g) * 0.02)
    ref_net.torch_multiplier = ref_net.torch_multiplier.to(dev)
    ref_net.apply_to()

    def build(precision):
        nat = FluxTransformer2DModel(**cfg, dtype=bf, device=dev, ops=ref_ops)
        nat.load_state_dict({k: v.to(bf) for k, v in ref.state_dict().items()}, strict=True)
        net = FusedLoRANetwork(nat, lora_dim=16)
        with torch.no_grad():
            for x, y in zip(net.unet_loras, ref_net.unet_loras):
                x.lora_down.weight.copy_(y.lora_down.weight.detach().cpu())
                x.lora_up.weight.copy_(y.lora_up.weight.detach().cpu())
        net.apply_to()
        net.build_arena(dev, groups=nat.lora_groups(), shadow_dtype=torch.float32)
        net.refresh_shadows(ref_ops)
        nat.attach_network(net)
        nat.prepare()
        nat.set_precision(precision)
        return nat, net

    gen = torch.Generator().manual_seed(5)
    B = a.batch
    lat = torch.randn(B, 16, a.hl, a.wl, generator=gen).to(bf).to(dev)
    emb = (torch.randn(B, 40, 256, generator=gen) * 0.5).to(bf).to(dev)
    pooled = (torch.randn(B, 64, generator=gen) * 0.5).to(bf).to(dev)
    noise = torch.randn(B, 16, a.hl, a.wl, generator=gen).to(bf).to(dev)
    ts = torch.tensor([700.0, 250.0, 999.0, 31.0][:B]).to(dev)
    kw = dict(lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    oracle = train_ref.RefTrainStep(ref, ref_net, **kw)
    l32 = oracle.step(lat.float(), emb.float(), pooled.float(), noise.float(), ts).item()
    g32 = [p.grad.clone() for p in oracle.params]
    ref.to(bf)
    l16 = oracle.step(lat, emb, pooled, noise, ts, dtype=bf).item()
    g16 = [p.grad.clone() for p in oracle.params]
    ref.float()
    res = {"config": vars(a), "loss_fp32": l32, "loss_ref16": l16, "ref16_vs_fp32": rel_lists(g16, g32)}
    for prec in ("default", "high"):
        nat, net = build(prec)
        l = FluxLoRATrainStep(nat, net, ref_ops, **kw).step(lat, emb, pooled, noise=noise, timesteps=ts).item()
        gr = []
        for m in net.unet_loras:
            gr += [m.lora_down.weight.grad.detach().clone(), m.lora_up.weight.grad.detach().clone()]
        res[f"loss_rm16_{prec}"] = l
        res[f"rm16_{prec}_vs_fp32"] = rel_lists(gr, g32)
        res[f"rm16_{prec}_loss_rel"] = abs(l - l32) / abs(l32)
    print(json.dumps(res, indent=1))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
