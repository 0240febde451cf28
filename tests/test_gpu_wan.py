"""GPU parity of the Wan2.1 path through the C ABI: the RMSNorm-across-heads + RoPE kernel, cross-attention (Sq != Skv)
and one whole LoRA training step against the oracle (oracle/wan_ref.py) on identical inputs.

Tolerances as in test_gpu_e2e.py: loss within 1e-3 relative of the fp32 oracle; adapter gradients within 1.5x of the
error the oracle itself makes when run in bf16 (the reference-equivalent PyTorch bf16 path), or 1e-2."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = dict(num_attention_heads=3, attention_head_dim=128, in_channels=16, out_channels=16, text_dim=256, freq_dim=256,
           ffn_dim=896, num_layers=3)


def _rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


@pytest.mark.parametrize("rope", [True, False])
@pytest.mark.parametrize("M,S,C", [(96, 48, 384), (200, 100, 1536), (37, 37, 4096)])
def test_rms_full_fwd_bwd(M, S, C, rope):
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    g = torch.Generator().manual_seed(M + C)
    x = (torch.randn(M, C + 64, generator=g) * 1.7).to(torch.bfloat16).cuda()[:, :C]  # strided rows
    w = (1 + 0.2 * torch.randn(C, generator=g)).to(torch.bfloat16).cuda()
    gy = torch.randn(M, C, generator=g).to(torch.bfloat16).cuda()
    cos = sin = None
    if rope:
        ang = torch.rand(S, 64, generator=g, dtype=torch.float64) * 6.28
        cos = ang.cos().repeat_interleave(2, 1).float().cuda().contiguous()
        sin = ang.sin().repeat_interleave(2, 1).float().cuda().contiguous()
    y, y_ref = torch.empty(M, C, dtype=torch.bfloat16, device="cuda"), torch.empty(M, C, dtype=torch.bfloat16, device="cuda")
    ops.rms_full_fwd(x, w, y, S=S, cos=cos, sin=sin)
    ref_ops.rms_full_fwd(x, w, y_ref, S=S, cos=cos, sin=sin)
    assert (y.float() - y_ref.float()).abs().max().item() <= 2 ** -7 * y_ref.float().abs().max().item()
    assert _rel(y, y_ref) < 3e-3
    dx, dx_ref = torch.empty_like(y), torch.empty_like(y)
    ops.rms_full_bwd(gy, x, w, dx, S=S, cos=cos, sin=sin)
    ref_ops.rms_full_bwd(gy, x, w, dx_ref, S=S, cos=cos, sin=sin)
    assert _rel(dx, dx_ref) < 6e-3, _rel(dx, dx_ref)


@pytest.mark.parametrize("B,H,S,Skv", [(2, 3, 192, 64), (1, 2, 520, 512), (2, 1, 100, 77)])
def test_cross_attention_fwd_bwd(B, H, S, Skv):
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    g = torch.Generator().manual_seed(S + Skv)
    d = H * 128
    q = torch.randn(B * S, d, generator=g).to(torch.bfloat16).cuda()
    k = torch.randn(B * Skv, d, generator=g).to(torch.bfloat16).cuda()
    v = torch.randn(B * Skv, d, generator=g).to(torch.bfloat16).cuda()
    do = torch.randn(B * S, d, generator=g).to(torch.bfloat16).cuda()
    sc = 1 / math.sqrt(128)
    outs = []
    for o_ in (ops, ref_ops):
        o = torch.empty_like(q)
        lse = torch.empty(B, H, S, dtype=torch.float32, device="cuda")
        o_.attn_fwd(q, k, v, o, lse, B=B, H=H, S=S, scale=sc, Skv=Skv)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        o_.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, B=B, H=H, S=S, scale=sc, Skv=Skv)
        outs.append((o, dq, dk, dv))
    for a, b, nm in zip(outs[0], outs[1], ("o", "dq", "dk", "dv")):
        assert _rel(a, b) < 8e-3, (nm, _rel(a, b))


def _build(rank=16, dev="cuda"):
    import ai_toolkit_amd  # noqa: F401
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.lora import FusedLoRANetwork
    from ai_toolkit_amd.wan import WanTransformer3DModel
    from oracle import lora_ref, wan_ref

    torch.manual_seed(0)
    ref = wan_ref.WanTransformer3DModel(**CFG)
    wan_ref.init_synthetic_(ref, seed=99, std=0.03)
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    ref = ref.to(dev)
    nat = WanTransformer3DModel(**CFG, dtype=torch.bfloat16, device=dev, ops=ops)
    nat.load_state_dict({k: v.to(torch.bfloat16) for k, v in ref.state_dict().items()}, strict=True)
    ref_net = lora_ref.RefLoRANetwork(ref, rank, target=("WanTransformer3DModel",), block_names=("blocks",)).to(dev)
    net = FusedLoRANetwork(nat, lora_dim=rank, target_lin_modules=("WanTransformer3DModel",), transformer_block_names=["blocks"],
                           base_model_version="wan_2.1")
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


def _oracle_step(ref, ref_net, lat, txt, noise, ts, dtype):
    params = [p for m in ref_net.unet_loras for p in (m.lora_down.weight, m.lora_up.weight)]
    for p in params:
        p.grad = None
    tt = (ts / 1000).view(-1, 1, 1, 1, 1)
    noisy = ((1 - tt) * lat.float() + tt * noise.float()).to(dtype)
    with ref_net:
        pred = ref(noisy, ts, txt.to(dtype))
        loss = (pred.float() - (noise.float() - lat.float())).pow(2).mean()
        loss.backward()
    return loss.item(), [p.grad.clone() for p in params]


def test_wan_step_gradients_and_loss_vs_oracle():
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.trainer import WanLoRATrainStep

    ref, ref_net, nat, net = _build()
    g = torch.Generator().manual_seed(5)
    B, Fr, Hl, Wl, n_txt = 2, 3, 16, 12, 40
    lat = torch.randn(B, 16, Fr, Hl, Wl, generator=g).to(torch.bfloat16).cuda()
    noise = torch.randn(B, 16, Fr, Hl, Wl, generator=g).to(torch.bfloat16).cuda()
    txt = (torch.randn(B, n_txt, CFG["text_dim"], generator=g) * 0.5).to(torch.bfloat16).cuda()
    ts = torch.tensor([700.0, 250.0]).cuda()
    loss32, g32 = _oracle_step(ref, ref_net, lat, txt, noise, ts, torch.float32)
    ref.to(torch.bfloat16)
    loss16, g16 = _oracle_step(ref, ref_net, lat, txt, noise, ts, torch.bfloat16)
    ref.float()
    ours = WanLoRATrainStep(nat, net, ops, lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    loss = ours.step(lat, txt, noise=noise, timesteps=ts).item()
    assert math.isfinite(loss)
    mine = []
    for m in net.unet_loras:
        mine += [m.lora_down.weight.grad, m.lora_up.weight.grad]
    num_o = sum(((a - b) ** 2).sum().item() for a, b in zip(mine, g32))
    num_r = sum(((a - b) ** 2).sum().item() for a, b in zip(g16, g32))
    den = sum((b ** 2).sum().item() for b in g32)
    e_ours, e_ref16 = math.sqrt(num_o / den), math.sqrt(num_r / den)
    print(f"wan loss ours {loss:.6f} fp32 {loss32:.6f} bf16-oracle {loss16:.6f}; grad rel err ours {e_ours:.4e} bf16-oracle {e_ref16:.4e}")
    assert abs(loss - loss32) <= 1e-3 * abs(loss32), (loss, loss32, loss16)
    assert e_ours <= max(1.5 * e_ref16, 1e-2), (e_ours, e_ref16)
    assert max(_rel(a, b) for a, b in zip(mine, g32)) < 0.1


def test_wan_step_is_bitwise_reproducible_and_zero_adapter_is_base():
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.trainer import WanLoRATrainStep

    ref, ref_net, nat, net = _build()
    g = torch.Generator().manual_seed(6)
    lat = torch.randn(1, 16, 2, 16, 16, generator=g).to(torch.bfloat16).cuda()
    noise = torch.randn(1, 16, 2, 16, 16, generator=g).to(torch.bfloat16).cuda()
    txt = torch.randn(1, 24, CFG["text_dim"], generator=g).to(torch.bfloat16).cuda()
    ts = torch.tensor([500.0]).cuda()
    p0 = net.arena_p.clone()
    ours = WanLoRATrainStep(nat, net, ops, lr=1e-3)
    l1 = ours.step(lat, txt, noise=noise, timesteps=ts).clone()
    g1, p1 = net.arena_g.clone(), net.arena_p.clone()
    net.arena_p.copy_(p0)
    net.arena_m.zero_()
    net.arena_v.zero_()
    net.refresh_shadows(ops)
    ours.set_step_count(0)  # host copy + the device-resident count of applied steps
    l2 = ours.step(lat, txt, noise=noise, timesteps=ts)
    assert torch.equal(l1, l2) and torch.equal(g1, net.arena_g) and torch.equal(p1, net.arena_p)
    # adapter with lora_up = 0 (the reference's init) predicts exactly like the base model
    tok = nat.pack_tokens(lat).contiguous()
    base = nat.forward_native(tok, ts, txt, (2, 8, 8), save_for_backward=False).clone()
    for m in net.unet_loras:
        m.lora_up.weight.data.zero_()
    net.refresh_shadows(ops)
    with net:
        with_zero = nat.forward_native(tok, ts, txt, (2, 8, 8), save_for_backward=False)
    assert torch.equal(base, with_zero)
