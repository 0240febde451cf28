"""Full-size (BASELINE config 3 dimensions: d=3072, 24 heads, 4096+512 tokens) checks on the MI355X.

  * full width, reduced depth (1 double + 1 single block) against the fp32 oracle on the same inputs: exercises the
    256x256 GEMM tiles, N=18432 adaLN projections, segmented joint buffers and S=4608 attention exactly as the benchmark does;
  * size-independent properties on the full 19+38-block model: bitwise determinism of a step, zero-initialised lora_up
    == frozen base model (bitwise) with exactly-zero lora_down gradients, per-sample loss independent of batch mates.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def _flux(num_layers, num_single, rank=16, warm=True, seed=1234, network_type="lora"):
    import ai_toolkit_amd  # noqa: F401
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.flux import FluxTransformer2DModel
    from ai_toolkit_amd.lora import FusedLoRANetwork

    dev = "cuda"
    model = FluxTransformer2DModel(num_layers=num_layers, num_single_layers=num_single, dtype=bf, device=dev, ops=ops)
    g = torch.Generator(device=dev).manual_seed(seed)
    with torch.no_grad():
        for mod in model.modules():
            if mod.__class__.__name__ == "Linear":
                mod.weight.copy_((torch.randn(mod.weight.shape, device=dev, generator=g) * 0.02).to(bf))
                mod.bias.copy_((torch.randn(mod.bias.shape, device=dev, generator=g) * 0.01).to(bf))
    torch.manual_seed(seed)
    net = FusedLoRANetwork(model, lora_dim=rank, alpha=rank, network_type=network_type)
    if warm:
        with torch.no_grad():
            for m in net.unet_loras:
                if network_type == "lokr":
                    m.lokr_w2.normal_(0, 2e-3)  # lokr_w1 is kaiming-initialised, lokr_w2 zero: warm w2 so every gradient is non-trivial
                else:
                    m.lora_up.weight.normal_(0, 2e-3)
    net.apply_to()
    net.build_arena(dev, groups=model.lora_groups())
    net.refresh_shadows(ops)
    model.attach_network(net)
    model.prepare()
    return model, net, ops


def _batch(B, seed=42):
    g = torch.Generator(device="cuda").manual_seed(seed)
    lat = torch.randn(B, 16, 128, 128, device="cuda", generator=g).to(bf)
    emb = (torch.randn(B, 512, 4096, device="cuda", generator=g) * 0.1).to(bf)
    pooled = (torch.randn(B, 768, device="cuda", generator=g) * 0.1).to(bf)
    noise = torch.randn(B, 16, 128, 128, device="cuda", generator=g).to(bf)
    # timesteps whose t/1000 -> bf16 -> *1000 round trip is exact: the model (like the pinned diffusers) casts the timestep to
    # the bf16 model dtype before the *1000, which an fp32 oracle cannot mirror for arbitrary t
    ts = torch.tensor([500.0, 250.0, 125.0, 62.5][:B], device="cuda")
    return lat, emb, pooled, noise, ts


def test_full_width_two_blocks_vs_fp32_oracle():
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from oracle import flux_ref, lora_ref, train_ref

    model, net, ops = _flux(1, 1)
    ref = flux_ref.FluxTransformer2DModel(num_layers=1, num_single_layers=1).cuda()
    ref.load_state_dict({k: v.float() for k, v in model.state_dict().items()}, strict=True)
    ref_net = lora_ref.RefLoRANetwork(ref, 16).cuda()
    ref_net.torch_multiplier = ref_net.torch_multiplier.cuda()
    with torch.no_grad():
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            assert a.lora_name == b.lora_name
            b.lora_down.weight.copy_(a.lora_down.weight)
            b.lora_up.weight.copy_(a.lora_up.weight)
    ref_net.apply_to()
    lat, emb, pooled, noise, ts = _batch(1)
    oracle = train_ref.RefTrainStep(ref, ref_net, lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    loss32 = oracle.step(lat.float(), emb.float(), pooled.float(), noise.float(), ts).item()
    g32 = [p.grad for p in oracle.params]
    ours = FluxLoRATrainStep(model, net, ops, lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    loss = ours.step(lat, emb, pooled, noise=noise, timesteps=ts).item()
    assert abs(loss - loss32) <= 1e-3 * abs(loss32), (loss, loss32)
    mine = []
    for m in net.unet_loras:
        mine += [m.lora_down.weight.grad, m.lora_up.weight.grad]
    num = sum(((a - b) ** 2).sum().item() for a, b in zip(mine, g32))
    den = sum((b ** 2).sum().item() for b in g32)
    rel = math.sqrt(num / den)
    print(f"full-width 1+1 blocks: loss {loss:.6f} vs fp32 {loss32:.6f}; adapter-grad rel err {rel:.3e}")
    assert rel < 1.5e-2, rel
    worst = max(((a - b).norm() / (b.norm() + 1e-30)).item() for a, b in zip(mine, g32))
    assert worst < 0.08, worst


def test_full_width_lokr_two_blocks_vs_fp32_oracle():
    """LoKr at the real FLUX factor shapes (3072 -> 48x64, 12288 -> 96x128, 15360 -> 120x128, 18432 -> 128x144, 9216 -> 96x96)
    through the whole graph: Kronecker delta ahead of the ACCUM-epilogue GEMMs (plain, GELU, gate-residual, segmented joint
    buffers), the windowed data gradient of the single block's proj_out, factor gradients over millions of contraction rows."""
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from oracle import flux_ref, lora_ref, train_ref

    big = 9999999999
    model, net, ops = _flux(1, 1, rank=big, network_type="lokr")
    m0 = net.unet_loras[0]
    assert (m0.out_l, m0.in_m, m0.out_k, m0.in_n) == (128, 48, 144, 64), "norm1.linear 3072 -> 18432"
    ref = flux_ref.FluxTransformer2DModel(num_layers=1, num_single_layers=1).cuda()
    ref.load_state_dict({k: v.float() for k, v in model.state_dict().items()}, strict=True)
    ref_net = lora_ref.RefLoRANetwork(ref, big, network_type="lokr").cuda()
    ref_net.torch_multiplier = ref_net.torch_multiplier.cuda()
    with torch.no_grad():
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            assert a.lora_name == b.lora_name
            b.lokr_w1.copy_(a.lokr_w1)
            b.lokr_w2.copy_(a.lokr_w2)
    ref_net.apply_to()
    lat, emb, pooled, noise, ts = _batch(1)
    oracle = train_ref.RefTrainStep(ref, ref_net, lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    loss32 = oracle.step(lat.float(), emb.float(), pooled.float(), noise.float(), ts).item()
    g32 = [p.grad for p in oracle.params]
    ours = FluxLoRATrainStep(model, net, ops, lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    loss = ours.step(lat, emb, pooled, noise=noise, timesteps=ts).item()
    assert abs(loss - loss32) <= 1e-3 * abs(loss32), (loss, loss32)
    mine = []
    for m in net.unet_loras:
        mine += [m.lokr_w1.grad, m.lokr_w2.grad]
    num = sum(((a - b) ** 2).sum().item() for a, b in zip(mine, g32))
    den = sum((b ** 2).sum().item() for b in g32)
    rel = math.sqrt(num / den)
    worst = max(((a - b).norm() / (b.norm() + 1e-30)).item() for a, b in zip(mine, g32))
    print(f"full-width LoKr 1+1 blocks: loss {loss:.6f} vs fp32 {loss32:.6f}; factor-grad rel err {rel:.3e} (worst {worst:.3e})")
    assert rel < 2e-2 and worst < 0.1, (rel, worst)


def test_full_model_determinism_zero_adapter_and_batch_independence():
    from ai_toolkit_amd.trainer import FluxLoRATrainStep, make_ids

    model, net, ops = _flux(19, 38, warm=False)  # lora_up == 0 exactly as the reference initialises it
    assert len(net.unet_loras) == 494 and net.arena_p.numel() == 85_917_696
    lat, emb, pooled, noise, ts = _batch(2)
    step = FluxLoRATrainStep(model, net, ops, lr=1e-4, max_grad_norm=1.0)
    # --- zero adapter == base model, bitwise; lora_down grads exactly zero, lora_up grads not
    noisy = torch.empty(2, 4096, 64, dtype=bf, device="cuda")
    target = torch.empty_like(noisy)
    ops.flow_noise_pack(lat, noise, ts, noisy, target)
    img_ids, txt_ids = make_ids(128, 128, 512, "cuda")
    guid = torch.ones(2, device="cuda")
    base = model.forward_native(noisy, emb, pooled, ts / 1000, img_ids, txt_ids, guid, save_for_backward=False).clone()
    with net:
        with_adapter = model.forward_native(noisy, emb, pooled, ts / 1000, img_ids, txt_ids, guid, save_for_backward=False).clone()
    assert torch.equal(base, with_adapter)
    assert torch.isfinite(base.float()).all()
    # --- per-sample loss does not depend on batch mates
    p0 = net.arena_p.clone()
    step.lr = 0.0
    step.weight_decay = 0.0
    step.step(lat, emb, pooled, noise=noise, timesteps=ts)
    lps2 = step.loss_per_sample.clone()
    assert float(net.unet_loras[0].lora_down.weight.grad.abs().max()) == 0.0  # dA = (c dY B)^T X with B == 0
    assert float(net.unet_loras[0].lora_up.weight.grad.abs().max()) > 0.0
    step.step(lat[:1], emb[:1], pooled[:1], noise=noise[:1], timesteps=ts[:1])
    assert abs(step.loss_per_sample[0].item() - lps2[0].item()) <= 2e-6 * abs(lps2[0].item()), (step.loss_per_sample, lps2)
    assert torch.equal(net.arena_p, p0)  # lr = 0 and wd = 0 leave the adapter untouched
    # --- a real step is bitwise reproducible from the same state
    step.lr = 1e-4
    m0, v0 = net.arena_m.clone(), net.arena_v.clone()
    n0 = step.step_num
    l1 = step.step(lat, emb, pooled, noise=noise, timesteps=ts).clone()
    p1 = net.arena_p.clone()
    net.arena_p.copy_(p0)
    net.arena_m.copy_(m0)
    net.arena_v.copy_(v0)
    step.set_step_count(n0)  # host copy + the device-resident count of applied steps
    net.refresh_shadows(ops)
    l2 = step.step(lat, emb, pooled, noise=noise, timesteps=ts)
    assert torch.equal(l1, l2) and torch.equal(net.arena_p, p1)
