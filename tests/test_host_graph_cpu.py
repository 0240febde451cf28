"""Host-logic parity on CPU: the hand-written forward/backward graph of ai_toolkit_amd.flux, driven by the oracle's
plain-torch kernel table in fp32, must reproduce autograd of the oracle model + oracle LoRA layer."""
import torch

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd.flux import FluxTransformer2DModel
from ai_toolkit_amd.lora import FusedLoRANetwork
from oracle import flux_ref, lora_ref, ref_ops

CFG = dict(in_channels=64, num_layers=2, num_single_layers=2, attention_head_dim=128, num_attention_heads=2,
           joint_attention_dim=64, pooled_projection_dim=32)


def build_pair(rank=8, multiplier=1.0, seed=0, grouped=True, network_type="lora"):
    torch.manual_seed(seed)
    ref = flux_ref.FluxTransformer2DModel(**CFG)
    flux_ref.init_synthetic_(ref, seed=1234, std=0.05)
    with torch.no_grad():  # non-trivial norms / biases so every path is exercised
        for n, p in ref.named_parameters():
            if n.endswith("bias"):
                p.copy_(torch.randn_like(p) * 0.02)
            if "norm_" in n and n.endswith("weight"):
                p.copy_(1 + 0.1 * torch.randn_like(p))
    nat = FluxTransformer2DModel(**CFG, dtype=torch.float32, device="cpu", ops=ref_ops)
    missing, unexpected = nat.load_state_dict(ref.state_dict(), strict=True)
    ref_net = lora_ref.RefLoRANetwork(ref, rank, multiplier, network_type=network_type)
    net = FusedLoRANetwork(nat, lora_dim=rank, multiplier=multiplier, network_type=network_type)
    assert [m.lora_name for m in net.unet_loras] == [m.lora_name for m in ref_net.unet_loras]
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            b.lora_up.weight.copy_(torch.randn(b.lora_up.weight.shape, generator=g) * 0.05)
            a.lora_down.weight.copy_(b.lora_down.weight)
            a.lora_up.weight.copy_(b.lora_up.weight)
    ref_net.apply_to()
    net.apply_to()
    net.build_arena("cpu", groups=nat.lora_groups() if grouped else None)
    net.refresh_shadows(ref_ops)
    nat.attach_network(net)
    nat.prepare()
    return ref, ref_net, nat, net


def inputs(B=2, Hl=8, Wl=4, n_txt=6, seed=3):
    g = torch.Generator().manual_seed(seed)
    n_img = (Hl // 2) * (Wl // 2)
    hidden = torch.randn(B, n_img, 64, generator=g)
    enc = torch.randn(B, n_txt, CFG["joint_attention_dim"], generator=g)
    pooled = torch.randn(B, CFG["pooled_projection_dim"], generator=g)
    timestep = torch.tensor([0.3, 0.8][:B])
    guidance = torch.ones(B)
    img_ids, txt_ids = flux_ref.make_ids(Hl, Wl, n_txt)
    return hidden, enc, pooled, timestep, img_ids, txt_ids, guidance


def test_forward_and_lora_grads_match_oracle_autograd():
    ref, ref_net, nat, net = build_pair()
    hidden, enc, pooled, t, img_ids, txt_ids, guid = inputs()
    with ref_net:
        pred_ref = ref(hidden, enc, pooled, t, img_ids, txt_ids, guid)
        w = torch.randn(pred_ref.shape, generator=torch.Generator().manual_seed(11))
        (pred_ref * w).sum().backward()
    with net:
        pred = nat.forward_native(hidden, enc, pooled, t, img_ids, txt_ids, guid)
        assert torch.allclose(pred, pred_ref, rtol=1e-4, atol=1e-5), (pred - pred_ref).abs().max()
        net.zero_grad_arena()
        nat.backward_native(w)
    worst = 0.0
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        for x, y, nm in ((a.lora_down.weight.grad, b.lora_down.weight.grad, "down"), (a.lora_up.weight.grad, b.lora_up.weight.grad, "up")):
            err = (x - y).norm() / (y.norm() + 1e-12)
            worst = max(worst, err.item())
            assert err < 2e-4, (a.lora_name, nm, err.item())
    assert worst > 0  # grads are non-trivial


def test_per_sample_multiplier_and_inactive_network():
    ref, ref_net, nat, net = build_pair(multiplier=1.0, grouped=False)  # ungrouped arena layout path
    hidden, enc, pooled, t, img_ids, txt_ids, guid = inputs()
    # inactive network == base model
    p0 = nat.forward_native(hidden, enc, pooled, t, img_ids, txt_ids, guid, save_for_backward=False)
    p0_ref = ref(hidden, enc, pooled, t, img_ids, txt_ids, guid)
    assert torch.allclose(p0, p0_ref, rtol=1e-4, atol=1e-5)
    net.multiplier = [0.5, -1.5]
    ref_net.torch_multiplier = torch.tensor([0.5, -1.5])
    with ref_net:
        pr = ref(hidden, enc, pooled, t, img_ids, txt_ids, guid)
        pr.square().sum().backward()
    with net:
        pn = nat.forward_native(hidden, enc, pooled, t, img_ids, txt_ids, guid)
        assert torch.allclose(pn, pr, rtol=1e-4, atol=1e-5)
        net.zero_grad_arena()
        nat.backward_native((2 * pn).detach())
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        err = (a.lora_up.weight.grad - b.lora_up.weight.grad).norm() / (b.lora_up.weight.grad.norm() + 1e-12)
        assert err < 2e-4, (a.lora_name, err.item())


def test_autograd_bridge_populates_param_grads():
    ref, ref_net, nat, net = build_pair()
    hidden, enc, pooled, t, img_ids, txt_ids, guid = inputs(B=1)
    with net:
        net.zero_grad_arena()
        (pred,) = nat(hidden, enc, pooled, t, img_ids, txt_ids, guid)
        pred.float().pow(2).mean().backward()
    gsum = sum(m.lora_up.weight.grad.abs().sum().item() for m in net.unet_loras)
    assert gsum > 0
    # views: Parameter.grad is the arena slice
    m0 = net.unet_loras[0]
    assert m0.lora_down.weight.grad.data_ptr() == net.arena_g[m0.off_down:].data_ptr()


def test_merge_in_equals_active_adapter_and_merge_out_restores():
    ref, ref_net, nat, net = build_pair()
    hidden, enc, pooled, t, img_ids, txt_ids, guid = inputs()
    with net:
        want = nat.forward_native(hidden, enc, pooled, t, img_ids, txt_ids, guid, save_for_backward=False)
    w0 = nat.transformer_blocks[0].attn.to_q.weight.detach().clone()
    net.merge_in(1.0, ops=ref_ops)
    assert net.is_merged_in and not torch.equal(w0, nat.transformer_blocks[0].attn.to_q.weight)
    with net:  # merged => adapters are skipped (toolkit/network_mixins.py:285-287)
        got = nat.forward_native(hidden, enc, pooled, t, img_ids, txt_ids, guid, save_for_backward=False)
    assert torch.allclose(got, want, rtol=1e-4, atol=2e-6 * float(want.abs().max()))  # fp32 summation-order noise on O(70) outputs
    lin = nat.transformer_blocks[0].attn.to_q
    assert torch.allclose(lin.weight_t, lin.weight.t(), rtol=0, atol=1e-6)
    net.merge_out(1.0, ops=ref_ops)
    assert torch.allclose(nat.transformer_blocks[0].attn.to_q.weight, w0, rtol=0, atol=1e-6)


def test_optimizer_state_exports_as_torch_adamw_state_dict():
    ref, ref_net, nat, net = build_pair(rank=4)
    net.arena_m.normal_()
    net.arena_v.uniform_()
    sd = net.optimizer_state_dict(step=7, lr=1e-4)
    opt = torch.optim.AdamW(net.prepare_optimizer_params()[0]["params"], lr=1e-4, eps=1e-6)
    opt.load_state_dict(sd)  # the reference's optimizer can resume from our optimizer.pt
    m0 = net.unet_loras[0]
    st = opt.state[m0.lora_down.weight]
    assert float(st["step"]) == 7 and torch.equal(st["exp_avg"], net.arena_m[m0.off_down:m0.off_down + m0.lora_down.weight.numel()].view_as(m0.lora_down.weight))
    m_copy = net.arena_m.clone()
    net.arena_m.zero_()
    assert net.load_optimizer_state_dict(opt.state_dict()) == 7
    for m in net.unet_loras:  # rank 4 lives in 16-wide padded blocks: every logical matrix is restored (the padding carries no state)
        for which in ("down", "up"):
            assert torch.equal(net.arena_view(net.arena_m, m, which), net.arena_view(m_copy, m, which)), (m.lora_name, which)
    assert m0.rank_pad == 16 and m0.blk_down == (16, m0.in_features) and m0.blk_up == (m0.out_features, 16)


def test_attached_network_stays_out_of_base_state_dict():
    ref, ref_net, nat, net = build_pair()
    assert not any("lora" in k or "network" in k for k in nat.state_dict().keys())
    assert sorted(nat.state_dict().keys()) == sorted(ref.state_dict().keys())


def test_fp8_weight_only_base_matches_oracle_with_dequantised_weights():
    ref, ref_net, nat, net = build_pair()
    nat.quantize_base_fp8()
    with torch.no_grad():  # the oracle multiplies with the dequantised weights (weight-only quantisation semantics)
        for (n, lin) in nat.named_modules():
            if getattr(lin, "qweight", None) is not None:
                dict(ref.named_modules())[n].weight.copy_(nat.dequantized_weight(lin))
    hidden, enc, pooled, t, img_ids, txt_ids, guid = inputs()
    with ref_net:
        pred_ref = ref(hidden, enc, pooled, t, img_ids, txt_ids, guid)
        pred_ref.square().sum().backward()
    with net:
        pred = nat.forward_native(hidden, enc, pooled, t, img_ids, txt_ids, guid)
        assert (pred - pred_ref).abs().max().item() <= 2e-6 * pred_ref.abs().max().item(), (pred - pred_ref).abs().max().item()  # fp32: summation order only (scale of the tensor: ~30)
        net.zero_grad_arena()
        nat.dgrad_census(reset=True)
        nat.backward_native((2 * pred).detach())
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        err = (a.lora_down.weight.grad - b.lora_down.weight.grad).norm() / (b.lora_down.weight.grad.norm() + 1e-12)
        assert err < 2e-4, (a.lora_name, err.item())
    # round 6: the quantised same-input groups keep the K-concatenated data-gradient GEMM (their e4m3 W^T expanded side by side into one scratch)
    assert nat.dgrad_census() == {"concat": CFG["num_single_layers"] + 2 * CFG["num_layers"], "fallback": 0}, nat.dgrad_census()


def test_merge_into_weight_only_fp8_base_requantises():
    """toolkit/network_mixins.py:452-459: merging into a quantised base dequantises, adds scale * up @ down and re-quantises, so
    the model stays quantised across merge / reset cycles."""
    ref, ref_net, nat, net = build_pair(rank=4)
    nat.quantize_base_fp8()
    lin = nat.transformer_blocks[0].attn.to_q
    m = lin.lora
    deq = lambda: lin.qweight.view(torch.float8_e4m3fn).float() * lin.wscale[:, None]  # noqa: E731
    w0 = deq()
    delta = 0.7 * m.scale * (m.lora_up.weight.detach() @ m.lora_down.weight.detach())
    net.merge_in(0.7, ops=ref_ops)
    assert net.is_merged_in and lin.qweight.dtype == torch.uint8 and lin.weight_t is None
    w1 = deq()
    step = (w0 + delta).abs().amax(dim=1, keepdim=True) / 448.0 * 32  # e4m3: 3 mantissa bits -> spacing <= amax / 14 at the top binade
    assert ((w1 - (w0 + delta)).abs() <= 0.5 * step + 1e-7).all()
    assert torch.allclose(lin.qweight_t, lin.qweight.t()) and not torch.equal(w1, w0)
    # exactly what a fresh quantisation of the merged weight gives
    q_ref = ((w0 + delta) / lin.wscale[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(lin.qweight, q_ref)
    net.merge_out(0.7, ops=ref_ops)
    assert ((deq() - w0).abs() <= step).all()  # two roundings away from the original at most
