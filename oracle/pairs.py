"""ORACLE-SIDE TEST INFRASTRUCTURE (not product code): the small FLUX pair every parity check starts from — the oracle model + oracle LoRA
network (plain torch, oracle/flux_ref.py / lora_ref.py) and the native graph + FusedLoRANetwork on the SAME bf16-representable base weights and
the same adapter state — and the synthetic batch that goes with it.  Imported by tests/, bench.py's `parity` leg and __graft_entry__.smoke()
only (the checker, never the thing measured)."""
import torch

CFG = dict(in_channels=64, num_layers=2, num_single_layers=3, attention_head_dim=128, num_attention_heads=3,
           joint_attention_dim=256, pooled_projection_dim=64)


def build(rank=16, dev="cuda", attach=True, dropout_cfg=None, mask_provider=None, only=None):
    """attach=False: the native model is returned WITHOUT an adapter network (4th value None) — for the adoption tests, where the
    reference-side network is built over it afterwards.  only = substrings (network_kwargs.only_if_contains): adapters on the matching Linears only"""
    import ai_toolkit_amd  # noqa: F401
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.flux import FluxTransformer2DModel
    from ai_toolkit_amd.lora import FusedLoRANetwork
    from oracle import flux_ref, lora_ref

    torch.manual_seed(0)
    ref = flux_ref.FluxTransformer2DModel(**CFG)
    flux_ref.init_synthetic_(ref, seed=1234, std=0.03)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if n.endswith("bias"):
                p.copy_(torch.randn_like(p) * 0.02)
            if "norm_" in n and n.endswith("weight"):
                p.copy_(1 + 0.1 * torch.randn_like(p))
            p.copy_(p.to(torch.bfloat16).float())  # bf16-representable base so every path sees identical weights
    ref = ref.to(dev)
    nat = FluxTransformer2DModel(**CFG, dtype=torch.bfloat16, device=dev, ops=ops)
    nat.load_state_dict({k: v.to(torch.bfloat16) for k, v in ref.state_dict().items()}, strict=True)
    ref_net = lora_ref.RefLoRANetwork(ref, rank)
    if only is not None:  # prune the oracle network to the same subset (before the forward swap)
        keep = [m for m in ref_net.unet_loras if any(w in m.lora_name.replace("$$", ".") for w in only)]
        for m in ref_net.unet_loras:
            if m not in keep:
                delattr(ref_net, m.lora_name)
        ref_net.unet_loras = keep
    ref_net = ref_net.to(dev)
    if not attach:
        g = torch.Generator().manual_seed(7)
        with torch.no_grad():
            for b in ref_net.unet_loras:
                b.lora_up.weight.copy_(torch.randn(b.lora_up.weight.shape, generator=g) * 0.02)
        ref_net.torch_multiplier = ref_net.torch_multiplier.to(dev)
        ref_net.apply_to()
        nat.prepare()
        return ref, ref_net, nat, None
    net = FusedLoRANetwork(nat, lora_dim=rank, **(dropout_cfg or {}), **({"only_if_contains": list(only)} if only is not None else {}))
    assert [m.lora_name for m in net.unet_loras] == [m.lora_name for m in ref_net.unet_loras]
    if dropout_cfg:  # LoRA dropout / rank_dropout / module_dropout: both sides draw their uniforms from the same keyed provider
        ref_net.dropout_cfg, ref_net.mask_provider = dict(dropout_cfg), mask_provider
        net.mask_provider = mask_provider
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            up = torch.randn(b.lora_up.weight.shape, generator=g) * 0.02
            b.lora_up.weight.copy_(up)
            a.lora_down.weight.copy_(b.lora_down.weight.cpu())
            a.lora_up.weight.copy_(up)
    ref_net.torch_multiplier = ref_net.torch_multiplier.to(dev)
    ref_net.apply_to()
    net.apply_to()
    net.build_arena(dev, groups=nat.lora_groups())
    net.refresh_shadows(ops)
    nat.attach_network(net)
    nat.prepare()
    return ref, ref_net, nat, net


def batch(B, Hl=16, Wl=12, n_txt=40, dev="cuda", seed=5):
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn(B, 16, Hl, Wl, generator=g).to(torch.bfloat16)
    emb = (torch.randn(B, n_txt, CFG["joint_attention_dim"], generator=g) * 0.5).to(torch.bfloat16)
    pooled = (torch.randn(B, CFG["pooled_projection_dim"], generator=g) * 0.5).to(torch.bfloat16)
    noise = torch.randn(B, 16, Hl, Wl, generator=g).to(torch.bfloat16)
    ts = torch.tensor([700.0, 250.0, 999.0, 31.0][:B])
    return [t.to(dev) for t in (lat, emb, pooled, noise, ts)]


def cpu_twin(ref, ref_net, rank):
    """A second copy of the oracle pair on the CPU (same base weights, same adapter state, forward swap applied): the reference arithmetic on a
    second backend for the `ref16_self` statistic (oracle bf16 on the GPU through rocBLAS vs the same code on the host's CPU kernels)."""
    from oracle import flux_ref, lora_ref

    cfg = {k: ref.config[k] for k in ("in_channels", "num_layers", "num_single_layers", "attention_head_dim", "num_attention_heads",
                                      "joint_attention_dim", "pooled_projection_dim")}
    twin = flux_ref.FluxTransformer2DModel(**cfg)
    twin.load_state_dict({k: v.detach().float().cpu() for k, v in ref.state_dict().items()}, strict=True)
    for p in twin.parameters():
        p.requires_grad_(False)
    twin_net = lora_ref.RefLoRANetwork(twin, rank)
    with torch.no_grad():
        for a, b in zip(ref_net.unet_loras, twin_net.unet_loras):
            assert a.lora_name == b.lora_name
            b.lora_down.weight.copy_(a.lora_down.weight.detach().cpu())
            b.lora_up.weight.copy_(a.lora_up.weight.detach().cpu())
    twin_net.apply_to()
    return twin, twin_net
