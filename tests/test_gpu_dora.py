"""DoRA on the MI355X through the C ABI: column-scale GEMM / GEMV epilogues, the norm and dz / d-magnitude kernels against the
oracle's functions of the same name, and one full DoRA train step against the fp32 oracle (reference DoRA semantics pinned on
CPU in tests/test_dora_cpu.py)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


@pytest.mark.parametrize("M,N,K,stage", [(512, 768, 256, 1), (8192, 4096, 256, 4), (300, 520, 192, 1)])
def test_gemm_col_scale_epilogue(M, N, K, stage):
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    b = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
    a2 = torch.randn(M, 16, generator=g).to(torch.bfloat16).cuda()
    b2 = (torch.randn(N, 16, generator=g) * 0.1).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).to(torch.bfloat16).cuda()
    cs = (1 + 0.2 * torch.randn(N, generator=g)).cuda()
    for flags in (0, ops.EPI_GELU):
        out, ref = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"), torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        kw = dict(bias=bias, a2=a2, b2=b2, flags=flags, col_scale=cs)
        aux, aux_r = torch.zeros_like(out), torch.zeros_like(out)
        ops.gemm_nt(a, b, out, aux_out=aux if flags else None, stage_mode=stage, tile_mode=2 if stage == 4 else None, **kw)
        ref_ops.gemm_nt(a, b, ref, aux_out=aux_r if flags else None, **kw)
        assert _rel(out, ref) < 4e-3, (flags, _rel(out, ref))
        if flags:
            assert _rel(aux, aux_r) < 4e-3


def test_gemv_col_scale_and_dora_kernels():
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    g = torch.Generator().manual_seed(3)
    Bm, N, K, R = 4, 1536, 3072, 16
    x = torch.randn(Bm, K, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).to(torch.bfloat16).cuda()
    T = torch.randn(Bm, R, generator=g).to(torch.bfloat16).cuda()
    bl = (torch.randn(N, R, generator=g) * 0.1).to(torch.bfloat16).cuda()
    cs = (1 + 0.2 * torch.randn(N, generator=g)).cuda()
    o, o_ref = torch.empty(Bm, N, dtype=torch.bfloat16, device="cuda"), torch.empty(Bm, N, dtype=torch.bfloat16, device="cuda")
    ops.gemv_nt(x, w, o, bias=bias, t=T, bl=bl, col_scale=cs)
    ref_ops.gemv_nt(x, w, o_ref, bias=bias, t=T, bl=bl, col_scale=cs)
    assert _rel(o, o_ref) < 4e-3
    # c = magnitude / ||W + s B A||
    A = (torch.randn(R, K, generator=g) / math.sqrt(R)).cuda()
    Bup = (torch.randn(N, R, generator=g) * 0.05).cuda().contiguous()
    mag = (w.float().norm(dim=1) * (1 + 0.05 * torch.randn(N, generator=g).cuda())).contiguous()
    w2 = w.float().pow(2).sum(1)
    tw = torch.empty(N, R, dtype=torch.bfloat16, device="cuda")
    ops.lora_down(w, A.to(torch.bfloat16), tw, scale=1.0, M=N)
    gram = torch.zeros(R, R, device="cuda")
    At = A.to(torch.bfloat16).t().contiguous()
    ops.lora_wgrad(At, At, gram, M=K)
    c, c_ref = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    ops.dora_colscale(w2, tw, Bup, gram, mag, 0.7, c)
    ref_ops.dora_colscale(w2, tw, Bup, gram, mag, 0.7, c_ref)
    assert torch.allclose(c, c_ref, rtol=1e-5, atol=1e-6)
    truth = mag / torch.linalg.norm(w.float() + 0.7 * Bup @ A.to(torch.bfloat16).float(), dim=1)
    assert torch.allclose(c, truth, rtol=2e-3), (c - truth).abs().max()
    # rank 80 (above one rank chunk): Gram matrix from global memory, rank-space operands in 64-rank chunks
    R8 = 80
    A8 = (torch.randn(R8, K, generator=g) / math.sqrt(R8)).cuda()
    B8 = (torch.randn(N, R8, generator=g) * 0.05).cuda().contiguous()
    tw8 = torch.empty(N, R8, dtype=torch.bfloat16, device="cuda")
    ops.lora_down(w, A8.to(torch.bfloat16), tw8, scale=1.0, M=N)
    gram8 = torch.zeros(R8, R8, device="cuda")
    At8 = A8.to(torch.bfloat16).t().contiguous()
    ops.lora_wgrad(At8, At8, gram8, M=K)
    c8, c8_ref = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    ops.dora_colscale(w2, tw8, B8, gram8, mag, 0.7, c8)
    ref_ops.dora_colscale(w2, tw8, B8, gram8, mag, 0.7, c8_ref)
    assert torch.allclose(c8, c8_ref, rtol=1e-5, atol=1e-6)
    truth8 = mag / torch.linalg.norm(w.float() + 0.7 * B8 @ A8.to(torch.bfloat16).float(), dim=1)
    assert torch.allclose(c8, truth8, rtol=2e-3), (c8 - truth8).abs().max()
    # dz / d magnitude
    M = 1000
    dy = torch.randn(M, N + 64, generator=g).to(torch.bfloat16).cuda()[:, :N]
    y = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    dz, dz_ref = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"), torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    dm, dm_ref = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
    ops.dora_bwd(dy, y, c, bias, mag, dz, dm, M=M)
    ref_ops.dora_bwd(dy, y, c, bias, mag, dz_ref, dm_ref, M=M)
    assert _rel(dz, dz_ref) < 3e-3 and _rel(dm, dm_ref) < 1e-4


@pytest.mark.parametrize("multiplier,rank", [(None, 16), ([1.0, 0.4], 16), (None, 80)], ids=["uniform", "per_sample", "uniform_r80"])
def test_dora_train_step_vs_fp32_oracle(multiplier, rank):
    """r80: above the 64 ranks of one skinny launch (rank chunks + aitk_dora_colscale's Gram matrix read through the caches).  per_sample: slider-style batch — the LoRA term takes each sample's multiplier, the DoRA weight the mean (toolkit/network_mixins.py:313-340;
    fused as a second un-scaled rank-r term, graph._DoraPS; pinned to the reference's own run on the CPU in tests/test_dora_cpu.py)."""
    import ai_toolkit_amd  # noqa: F401
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.flux import FluxTransformer2DModel
    from ai_toolkit_amd.lora import FusedLoRANetwork
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from oracle import flux_ref, lora_ref, train_ref
    from tests.test_gpu_e2e import CFG, _batch

    dev = "cuda"
    torch.manual_seed(0)
    ref = flux_ref.FluxTransformer2DModel(**CFG)
    flux_ref.init_synthetic_(ref, seed=1234, std=0.03)
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    ref = ref.to(dev)
    nat = FluxTransformer2DModel(**CFG, dtype=torch.bfloat16, device=dev, ops=ops)
    nat.load_state_dict({k: v.to(torch.bfloat16) for k, v in ref.state_dict().items()}, strict=True)
    torch.manual_seed(5)
    ref_net = lora_ref.RefLoRANetwork(ref, rank, network_type="dora").to(dev)
    torch.manual_seed(5)
    net = FusedLoRANetwork(nat, lora_dim=rank, network_type="dora")
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            up = torch.randn(b.lora_up.weight.shape, generator=g) * 0.02
            b.lora_up.weight.copy_(up)
            a.lora_up.weight.copy_(up)
            mg = b.magnitude.cpu() * (1 + 0.03 * torch.randn(b.magnitude.shape, generator=g))
            b.magnitude.copy_(mg)
            a.magnitude.copy_(mg)
    ref_net.torch_multiplier = ref_net.torch_multiplier.to(dev)
    ref_net.apply_to()
    net.apply_to()
    net.build_arena(dev, groups=nat.lora_groups())
    net.refresh_shadows(ops)
    nat.attach_network(net)
    nat.prepare()
    if multiplier is not None:
        net.multiplier = multiplier
        ref_net.torch_multiplier = torch.tensor(multiplier, device=dev)
    lat, emb, pooled, noise, ts = _batch(2)
    oracle = train_ref.RefTrainStep(ref, ref_net, lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    loss32 = oracle.step(lat.float(), emb.float(), pooled.float(), noise.float(), ts).item()
    g32 = {id(p): p.grad.clone() for p in oracle.params}
    ours = FluxLoRATrainStep(nat, net, ops, lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    loss = ours.step(lat, emb, pooled, noise=noise, timesteps=ts).item()
    assert abs(loss - loss32) <= 1.5e-3 * abs(loss32), (loss, loss32)
    num = den = 0.0
    num_m = den_m = 0.0
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        for pa, pb in ((a.lora_down.weight, b.lora_down.weight), (a.lora_up.weight, b.lora_up.weight)):
            num += ((pa.grad - g32[id(pb)]) ** 2).sum().item()
            den += (g32[id(pb)] ** 2).sum().item()
        num_m += ((a.magnitude.grad - g32[id(b.magnitude)]) ** 2).sum().item()
        den_m += (g32[id(b.magnitude)] ** 2).sum().item()
    e, e_m = math.sqrt(num / den), math.sqrt(num_m / den_m)
    print(f"dora loss ours {loss:.6f} fp32 {loss32:.6f}; grad rel err matrices {e:.3e} magnitude {e_m:.3e}")
    assert e < 2e-2 and e_m < 2e-2, (e, e_m)


def test_dora_over_the_weight_only_fp8_base_vs_fp32_oracle_on_the_dequantised_weights():
    """network.type dora + model.quantize (toolkit/models/DoRA.py:105-109: the norm is taken over weight.dequantize()): the quantised layers have
    released their bf16 copies, refresh_dora expands the e4m3 codes for its skinny pass and ||W_j||^2, the base GEMMs multiply with the same expansion.
    Oracle: the fp32 restatement on a model that holds the dequantised weights.  CPU twin: tests/test_dora_cpu.py."""
    import ai_toolkit_amd  # noqa: F401
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.flux import FluxTransformer2DModel
    from ai_toolkit_amd.lora import FusedLoRANetwork
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from oracle import flux_ref, lora_ref, train_ref
    from tests.test_gpu_e2e import CFG, _batch

    dev = "cuda"
    torch.manual_seed(0)
    ref = flux_ref.FluxTransformer2DModel(**CFG)
    flux_ref.init_synthetic_(ref, seed=1234, std=0.03)
    nat = FluxTransformer2DModel(**CFG, dtype=torch.bfloat16, device=dev, ops=ops)
    nat.load_state_dict({k: v.to(torch.bfloat16) for k, v in ref.state_dict().items()}, strict=True)
    nat.prepare()
    nat.quantize_base_fp8(release_bf16=True)
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(p.to(torch.bfloat16).float())
        mods = dict(ref.named_modules())
        for n, lin in nat.named_modules():
            if getattr(lin, "qweight", None) is not None:
                assert lin.weight.numel() == 0
                mods[n].weight.copy_(nat.dequantized_weight(lin).float().cpu())
    ref = ref.to(dev)
    torch.manual_seed(5)
    ref_net = lora_ref.RefLoRANetwork(ref, 16, network_type="dora").to(dev)
    torch.manual_seed(5)
    net = FusedLoRANetwork(nat, lora_dim=16, network_type="dora")
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            assert torch.allclose(a.magnitude.cpu(), b.magnitude.cpu(), rtol=1e-5, atol=0)  # both from the dequantised weight
            up = torch.randn(b.lora_up.weight.shape, generator=g) * 0.02
            b.lora_up.weight.copy_(up)
            a.lora_up.weight.copy_(up)
            mg = b.magnitude.cpu() * (1 + 0.03 * torch.randn(b.magnitude.shape, generator=g))
            b.magnitude.copy_(mg)
            a.magnitude.copy_(mg)
    ref_net.torch_multiplier = ref_net.torch_multiplier.to(dev)
    ref_net.apply_to()
    net.apply_to()
    net.build_arena(dev, groups=nat.lora_groups())
    net.refresh_shadows(ops)
    nat.attach_network(net)
    lat, emb, pooled, noise, ts = _batch(2)
    oracle = train_ref.RefTrainStep(ref, ref_net, lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    loss32 = oracle.step(lat.float(), emb.float(), pooled.float(), noise.float(), ts).item()
    g32 = {id(p): p.grad.clone() for p in oracle.params}
    ours = FluxLoRATrainStep(nat, net, ops, lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    loss = ours.step(lat, emb, pooled, noise=noise, timesteps=ts).item()
    assert abs(loss - loss32) <= 1.5e-3 * abs(loss32), (loss, loss32)
    num = den = num_m = den_m = 0.0
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        for pa, pb in ((a.lora_down.weight, b.lora_down.weight), (a.lora_up.weight, b.lora_up.weight)):
            num += ((pa.grad - g32[id(pb)]) ** 2).sum().item()
            den += (g32[id(pb)] ** 2).sum().item()
        num_m += ((a.magnitude.grad - g32[id(b.magnitude)]) ** 2).sum().item()
        den_m += (g32[id(b.magnitude)] ** 2).sum().item()
    e, e_m = math.sqrt(num / den), math.sqrt(num_m / den_m)
    print(f"dora over fp8 base: loss ours {loss:.6f} fp32 {loss32:.6f}; grad rel err matrices {e:.3e} magnitude {e_m:.3e}")
    assert e < 2e-2 and e_m < 2e-2, (e, e_m)
