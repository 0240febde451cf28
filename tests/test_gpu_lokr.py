"""LoKr on the MI355X through the C ABI: aitk_kron_apply (per-token A . X . B^T, every mode the graph uses) against the oracle's
function of the same name, and one full LoKr train step against the fp32 autograd oracle (reference LoKr semantics are pinned
on CPU in tests/test_lokr_cpu.py against vectors produced by the reference's LokrModule)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


SHAPES = [  # a_in, b_in, a_out, b_out, M
    (48, 64, 48, 64, 1000),    # FLUX 3072 -> 3072
    (48, 64, 96, 128, 777),    # 3072 -> 12288
    (120, 128, 48, 64, 300),   # 15360 -> 3072 (single-block proj_out; a_in padded to 128)
    (48, 64, 128, 144, 4),     # adaLN 3072 -> 18432 on B rows
    (96, 128, 48, 64, 260),    # 12288 -> 3072
    (128, 144, 48, 64, 9),     # 18432 features per row: longer than the register prefetch window (direct staging path)
    (16, 16, 24, 32, 64),      # tiny-model shapes (pads in every dimension)
    (32, 40, 16, 16, 130),
    (80, 112, 32, 48, 200),    # Wan 8960 -> 1536
]


@pytest.mark.parametrize("a_in,b_in,a_out,b_out,M", SHAPES)
def test_kron_apply_full_product_and_data_gradient_form(a_in, b_in, a_out, b_out, M):
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    g = torch.Generator().manual_seed(a_in * 7 + b_out)
    x = torch.randn(M, a_in * b_in, generator=g).to(BF).cuda()
    A = (torch.randn(a_out, a_in, generator=g) / math.sqrt(a_in)).to(BF).cuda()
    Bm = (torch.randn(b_out, b_in, generator=g) / math.sqrt(b_in)).to(BF).cuda()
    kw = dict(a_in=a_in, b_in=b_in, a_out=a_out, b_out=b_out)
    out, ref = torch.empty(M, a_out * b_out, dtype=BF, device="cuda"), torch.empty(M, a_out * b_out, dtype=BF, device="cuda")
    ops.kron_apply(x, A, Bm, out, scale=0.7, **kw)
    ref_ops.kron_apply(x, A, Bm, ref, scale=0.7, **kw)
    assert _rel(out, ref) < 4e-3, _rel(out, ref)
    # equals F.linear(x, kron(A, B)) — the un-factorised definition (toolkit/models/lokr.py make_kron)
    if a_in * b_in * a_out * b_out <= 3072 * 3072:
        dense = x.float() @ torch.kron(A.float(), Bm.float()).t() * 0.7
        assert _rel(out, dense) < 8e-3
    # transposed output, accumulate, column window
    outT, refT = torch.empty_like(out), torch.empty_like(ref)
    ops.kron_apply(x, A, Bm, outT, transpose_out=True, **kw)
    ref_ops.kron_apply(x, A, Bm, refT, transpose_out=True, **kw)
    assert _rel(outT, refT) < 4e-3
    base = torch.randn(M, a_out * b_out, generator=g).to(BF).cuda()
    acc, acc_r = base.clone(), base.clone()
    ops.kron_apply(x, A, Bm, acc, accumulate=True, **kw)
    ref_ops.kron_apply(x, A, Bm, acc_r, accumulate=True, **kw)
    assert _rel(acc, acc_r) < 4e-3
    n = a_out * b_out
    c0, nc = (n // 16) * 8, (n // 32) * 8
    wide = torch.zeros(M, nc + 24, dtype=BF, device="cuda")
    win, win_r = wide[:, 8:8 + nc], torch.zeros(M, nc, dtype=BF, device="cuda")
    ops.kron_apply(x, A, Bm, win, col0=c0, ncols=nc, **kw)
    ref_ops.kron_apply(x, A, Bm, win_r, col0=c0, ncols=nc, **kw)
    assert _rel(win, win_r) < 4e-3 and float(wide[:, :8].abs().max()) == 0 and float(wide[:, 8 + nc:].abs().max()) == 0


@pytest.mark.parametrize("a_in,b_in,a_out,b_out,M", SHAPES[:4] + SHAPES[5:7])
def test_kron_apply_identity_factor_modes(a_in, b_in, a_out, b_out, M):
    """the intermediates of the factor gradients: (I, B) transposed, (A, I), and the pure per-token transpose."""
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    g = torch.Generator().manual_seed(b_in * 3 + a_out)
    x = torch.randn(M, a_in * b_in, generator=g).to(BF).cuda()
    A = (torch.randn(a_out, a_in, generator=g) / math.sqrt(a_in)).to(BF).cuda()
    Bm = (torch.randn(b_out, b_in, generator=g) / math.sqrt(b_in)).to(BF).cuda()
    for Aop, Bop, ao, bo, tr in ((None, Bm, a_in, b_out, True), (None, Bm, a_in, b_out, False), (A, None, a_out, b_in, False),
                                 (A, None, a_out, b_in, True), (None, None, a_in, b_in, True), (None, None, a_in, b_in, False)):
        out = torch.empty(M, ao * bo, dtype=BF, device="cuda")
        ref = torch.empty_like(out)
        kw = dict(a_in=a_in, b_in=b_in, a_out=ao, b_out=bo, transpose_out=tr, scale=1.3)
        ops.kron_apply(x, Aop, Bop, out, **kw)
        ref_ops.kron_apply(x, Aop, Bop, ref, **kw)
        assert _rel(out, ref) < 4e-3, (Aop is None, Bop is None, tr, _rel(out, ref))
    # identity x identity without scale is a bit-exact (transposing) copy
    out = torch.empty(M, a_in * b_in, dtype=BF, device="cuda")
    ops.kron_apply(x, None, None, out, a_in=a_in, b_in=b_in, a_out=a_in, b_out=b_in, transpose_out=True)
    assert torch.equal(out.view(M, b_in, a_in), x.view(M, a_in, b_in).transpose(1, 2))


def test_kron_apply_segmented_rows():
    """rows of the image stream inside the joint [B, S, C] buffer (x_seg) and a segmented destination (out_seg)."""
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    g = torch.Generator().manual_seed(1)
    B, S, St, a_in, b_in, a_out, b_out = 3, 40, 8, 16, 16, 16, 16
    C = a_in * b_in
    joint = torch.randn(B, S, C, generator=g).to(BF).cuda()
    A = (torch.randn(a_out, a_in, generator=g) / 4).to(BF).cuda()
    Bm = (torch.randn(b_out, b_in, generator=g) / 4).to(BF).cuda()
    Si = S - St
    seg = (Si, S * C)
    xv = joint.view(B * S, C)[St:St + Si]
    dst, dst_r = torch.zeros(B, S, C, dtype=BF, device="cuda"), torch.zeros(B, S, C, dtype=BF, device="cuda")
    kw = dict(a_in=a_in, b_in=b_in, a_out=a_out, b_out=b_out, x_seg=seg, out_seg=seg, M=B * Si)
    ops.kron_apply(xv, A, Bm, dst.view(B * S, C)[St:St + Si], **kw)
    ref_ops.kron_apply(xv, A, Bm, dst_r.view(B * S, C)[St:St + Si], **kw)
    assert _rel(dst, dst_r) < 4e-3 and float(dst[:, :St].abs().max()) == 0
    want = torch.einsum("bmqo,pq->bmpo", torch.einsum("bmqs,os->bmqo", joint[:, St:].float().view(B, Si, a_in, b_in), Bm.float()), A.float())
    assert _rel(dst[:, St:], want.reshape(B, Si, C)) < 8e-3


@pytest.mark.parametrize("factor", [-1, 4])
def test_two_stage_lokr_train_step_vs_fp32_oracle(monkeypatch, factor):
    """lora.check_kron_fits "two_stage" forced on the tiny model: W2 through aitk_gemm_nt over (token x factor-index) rows, lokr_w1 mixed in by
    aitk_kron_apply (B = identity) on the narrower side — forward delta, data gradient (incl. the windowed proj_out one) and both factor gradients."""
    from ai_toolkit_amd import lora as L

    real = L.check_kron_fits

    def forced(name, in_m, in_n, out_l, out_k):
        real(name, in_m, in_n, out_l, out_k)
        return "two_stage"

    monkeypatch.setattr(L, "check_kron_fits", forced)
    test_lokr_train_step_vs_fp32_oracle(factor, expect_two_stage=True)


def test_two_stage_lokr_at_flux_width_with_factor_4():
    """The case the two-stage form exists for: d = 3072 (24 heads), `network.lokr_factor: 4` -> lokr_w1 4 x 4, W2 768 x 768 / 3072 x 768 / 768 x 3840 ...
    — none fits the per-token kernel (1.2 MiB of LDS).  One double + one single block at FLUX width, a short sequence, vs the fp32 oracle."""
    test_lokr_train_step_vs_fp32_oracle(4, expect_two_stage=True, cfg_over=dict(num_attention_heads=24, num_layers=1, num_single_layers=1), hw=(8, 8), n_txt=16, tol=3e-2, check_update=False)


@pytest.mark.parametrize("factor", [-1, 4, 8])
def test_lokr_train_step_vs_fp32_oracle(factor, expect_two_stage=False, cfg_over=None, hw=(16, 12), n_txt=40, tol=2e-2, check_update=True):
    """factor -1: the default factorisation; 4 / 8 (`network.lokr_factor`): lokr_w1 is 4 x 4 / 8 x 8, below the 16-column granule of
    aitk_lora_wgrad — graph._skinny_tn reduces zero-padded copies and adds the valid block."""
    import ai_toolkit_amd  # noqa: F401
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.flux import FluxTransformer2DModel
    from ai_toolkit_amd.lora import FusedLoRANetwork
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from oracle import flux_ref, lora_ref, train_ref
    from tests.test_gpu_e2e import CFG as CFG3

    CFG = dict(CFG3, num_attention_heads=2)  # d = 256: every Kronecker factor of the model is a multiple of 8
    CFG.update(cfg_over or {})
    dev, big = "cuda", 9999999999
    torch.manual_seed(0)
    ref = flux_ref.FluxTransformer2DModel(**CFG)
    flux_ref.init_synthetic_(ref, seed=1234, std=0.03)
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(p.to(BF).float())
    ref = ref.to(dev)
    nat = FluxTransformer2DModel(**CFG, dtype=BF, device=dev, ops=ops)
    nat.load_state_dict({k: v.to(BF) for k, v in ref.state_dict().items()}, strict=True)
    torch.manual_seed(5)
    ref_net = lora_ref.RefLoRANetwork(ref, big, network_type="lokr", lokr_factor=factor).to(dev)
    torch.manual_seed(5)
    net = FusedLoRANetwork(nat, lora_dim=big, alpha=big, network_type="lokr", lokr_factor=factor)
    if factor > 0:
        assert any(tuple(m.lokr_w1.shape) == (factor, factor) for m in net.unet_loras)
    assert any(m.kron_two_stage for m in net.unet_loras) == expect_two_stage
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            assert torch.equal(a.lokr_w1, b.lokr_w1.cpu())
            w2 = torch.randn(b.lokr_w2.shape, generator=g) * 0.03
            b.lokr_w2.copy_(w2)
            a.lokr_w2.copy_(w2)
    ref_net.torch_multiplier = ref_net.torch_multiplier.to(dev)
    ref_net.apply_to()
    net.apply_to()
    net.build_arena(dev, groups=nat.lora_groups())
    net.refresh_shadows(ops)
    nat.attach_network(net)
    nat.prepare()
    gb = torch.Generator().manual_seed(5)
    Bn, (Hl, Wl) = 2, hw
    lat = torch.randn(Bn, 16, Hl, Wl, generator=gb).to(BF).to(dev)
    emb = (torch.randn(Bn, n_txt, CFG["joint_attention_dim"], generator=gb) * 0.5).to(BF).to(dev)
    pooled = (torch.randn(Bn, CFG["pooled_projection_dim"], generator=gb) * 0.5).to(BF).to(dev)
    noise = torch.randn(Bn, 16, Hl, Wl, generator=gb).to(BF).to(dev)
    ts = torch.tensor([700.0, 250.0], device=dev)
    oracle = train_ref.RefTrainStep(ref, ref_net, lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    loss32 = oracle.step(lat.float(), emb.float(), pooled.float(), noise.float(), ts).item()
    g32 = {id(p): p.grad.clone() for p in oracle.params}
    ours = FluxLoRATrainStep(nat, net, ops, lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    loss = ours.step(lat, emb, pooled, noise=noise, timesteps=ts).item()
    assert abs(loss - loss32) <= 1.5e-3 * abs(loss32), (loss, loss32)
    num = den = 0.0
    worst = 0.0
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        for pa, pb in ((a.lokr_w1, b.lokr_w1), (a.lokr_w2, b.lokr_w2)):
            d2 = ((pa.grad - g32[id(pb)]) ** 2).sum().item()
            n2 = (g32[id(pb)] ** 2).sum().item()
            num, den = num + d2, den + n2
            worst = max(worst, math.sqrt(d2 / (n2 + 1e-30)))
    e = math.sqrt(num / den)
    print(f"lokr loss ours {loss:.6f} fp32 {loss32:.6f}; factor-gradient rel err {e:.3e} (worst layer {worst:.3e})")
    assert e < tol, e
    if not check_update:  # (synthetic N(0, 0.03^2) weights at d = 3072 give a loss of ~5: an AdamW step of 1e-3 on 768 x 768 factors overshoots there)
        return
    # a real update moves the loss and keeps the adapter finite
    ours2 = FluxLoRATrainStep(nat, net, ops, lr=1e-3, max_grad_norm=1.0)
    l0 = ours2.step(lat, emb, pooled, noise=noise, timesteps=ts).item()
    l1 = ours2.step(lat, emb, pooled, noise=noise, timesteps=ts).item()
    assert math.isfinite(l1) and l1 < l0, (l0, l1)


def test_kron_merge_kernel():
    """aitk_kron_merge (LoKr merge_in) on the FLUX factor shapes, weight and transposed-copy orientation, vs torch.kron in fp32."""
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    g = torch.Generator().manual_seed(3)
    for (ar, ac, br, bc) in ((48, 48, 64, 64), (96, 48, 128, 64), (48, 120, 64, 128), (24, 16, 32, 16)):
        W = (torch.randn(ar * br, ac * bc, generator=g) * 0.02).to(BF).cuda()
        A = torch.randn(ar, ac, generator=g).cuda() * 0.1
        Bm = torch.randn(br, bc, generator=g).cuda() * 0.1
        want = ref_ops.kron_merge(W.clone(), A, Bm, 0.7)
        got = ops.kron_merge(W.clone(), A, Bm, 0.7)
        # same fp32 sum, different association (alpha * A first): at most one bf16 ulp apart
        assert ((got.float() - want.float()).abs() <= 2.0 ** -7 * want.float().abs() + 1e-6).all(), (ar, ac, br, bc)
        assert _rel(got, want) < 2e-3
        # transposed copy: kron(A, B)^T = kron(A^T, B^T)
        Wt = W.t().contiguous()
        got_t = ops.kron_merge(Wt, A.t().contiguous(), Bm.t().contiguous(), 0.7)
        assert torch.equal(got_t, got.t())  # both orientations round identically


def test_lowrank_lokr_compose_and_pair_gradient_kernels():
    """shadow kind 3 (W2 = a @ b composed in fp32 -> bf16, both orientations) and aitk_lokr_lowrank_grad against fp32 torch."""
    import ai_toolkit_amd  # noqa: F401
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    g = torch.Generator().manual_seed(3)
    for O, I, r in ((64, 64, 16), (128, 96, 4), (24, 40, 8)):
        a = torch.randn(O, r, generator=g).cuda()
        b = torch.randn(r, I, generator=g).cuda()
        arena = torch.cat((torch.zeros(7), a.flatten().cpu(), b.flatten().cpu())).cuda()
        entries = [(7, O, I, 3, 0, O * I, 0, r)]
        sh = torch.zeros(2 * O * I, dtype=BF, device="cuda")
        sh_ref = torch.zeros_like(sh)
        ops.refresh_shadows(arena, sh, ops.make_shadow_table(entries, "cuda"))
        ref_ops.refresh_shadows(arena, sh_ref, ref_ops.make_shadow_table(entries, "cuda"))
        torch.cuda.synchronize()
        # fp32 summation order differs (fma chain vs torch matmul): identical up to one bf16 rounding of a value that sat on a boundary
        assert _rel(sh.float(), sh_ref.float()) < 2e-3
        assert torch.equal(sh[:O * I].view(O, I).t(), sh[O * I:].view(I, O))
        dw = torch.randn(O, I, generator=g).cuda()
        ga0, gb0 = torch.randn(O, r, generator=g).cuda(), torch.randn(r, I, generator=g).cuda()
        for acc in (False, True):
            ga, gb = ga0.clone(), gb0.clone()
            ops.lokr_lowrank_grad(dw, a, b, ga, gb, accumulate=acc)
            ra, rb = ga0.clone(), gb0.clone()
            ref_ops.lokr_lowrank_grad(dw, a, b, ra, rb, accumulate=acc)
            torch.cuda.synchronize()
            assert torch.allclose(ga, ra, rtol=1e-4, atol=1e-4) and torch.allclose(gb, rb, rtol=1e-4, atol=1e-4), (O, I, r, acc)


def test_lowrank_lokr_train_step_vs_fp32_oracle():
    """`lokr_full_rank: false` (lora_dim 4 < max(out_k, in_n) / 2 for every layer): loss and the gradients of lokr_w1, lokr_w2_a, lokr_w2_b."""
    import ai_toolkit_amd  # noqa: F401
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.flux import FluxTransformer2DModel
    from ai_toolkit_amd.lora import FusedLoRANetwork
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from oracle import flux_ref, lora_ref, train_ref
    from tests.test_gpu_e2e import CFG as CFG3

    CFG = dict(CFG3, num_attention_heads=2)
    dev, r = "cuda", 4
    torch.manual_seed(0)
    ref = flux_ref.FluxTransformer2DModel(**CFG)
    flux_ref.init_synthetic_(ref, seed=1234, std=0.03)
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(p.to(BF).float())
    ref = ref.to(dev)
    nat = FluxTransformer2DModel(**CFG, dtype=BF, device=dev, ops=ops)
    nat.load_state_dict({k: v.to(BF) for k, v in ref.state_dict().items()}, strict=True)
    torch.manual_seed(5)
    ref_net = lora_ref.RefLoRANetwork(ref, r, network_type="lokr").to(dev)
    torch.manual_seed(5)
    net = FusedLoRANetwork(nat, lora_dim=r, alpha=r, network_type="lokr")
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            assert not a.use_w2 and torch.equal(a.lokr_w1, b.lokr_w1.cpu()) and torch.equal(a.lokr_w2_a, b.lokr_w2_a.cpu())
            wb = torch.randn(b.lokr_w2_b.shape, generator=g) * 0.1
            b.lokr_w2_b.copy_(wb)
            a.lokr_w2_b.copy_(wb)
    ref_net.torch_multiplier = ref_net.torch_multiplier.to(dev)
    ref_net.apply_to()
    net.apply_to()
    net.build_arena(dev, groups=nat.lora_groups())
    net.refresh_shadows(ops)
    nat.attach_network(net)
    nat.prepare()
    gb = torch.Generator().manual_seed(5)
    Bn, Hl, Wl, n_txt = 2, 16, 12, 40
    lat = torch.randn(Bn, 16, Hl, Wl, generator=gb).to(BF).to(dev)
    emb = (torch.randn(Bn, n_txt, CFG["joint_attention_dim"], generator=gb) * 0.5).to(BF).to(dev)
    pooled = (torch.randn(Bn, CFG["pooled_projection_dim"], generator=gb) * 0.5).to(BF).to(dev)
    noise = torch.randn(Bn, 16, Hl, Wl, generator=gb).to(BF).to(dev)
    ts = torch.tensor([700.0, 250.0], device=dev)
    oracle = train_ref.RefTrainStep(ref, ref_net, lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    loss32 = oracle.step(lat.float(), emb.float(), pooled.float(), noise.float(), ts).item()
    g32 = {id(p): p.grad.clone() for p in oracle.params}
    ours = FluxLoRATrainStep(nat, net, ops, lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    loss = ours.step(lat, emb, pooled, noise=noise, timesteps=ts).item()
    assert abs(loss - loss32) <= 1.5e-3 * abs(loss32), (loss, loss32)
    num = den = 0.0
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        for k in ("lokr_w1", "lokr_w2_a", "lokr_w2_b"):
            gr = g32[id(getattr(b, k))]
            num += ((getattr(a, k).grad - gr) ** 2).sum().item()
            den += (gr ** 2).sum().item()
    e = math.sqrt(num / den)
    print(f"low-rank lokr loss ours {loss:.6f} fp32 {loss32:.6f}; factor-gradient rel err {e:.3e}")
    assert e < 2e-2, e
    ours2 = FluxLoRATrainStep(nat, net, ops, lr=1e-3, max_grad_norm=1.0)
    l0 = ours2.step(lat, emb, pooled, noise=noise, timesteps=ts).item()
    l1 = ours2.step(lat, emb, pooled, noise=noise, timesteps=ts).item()
    assert math.isfinite(l1) and l1 < l0, (l0, l1)
