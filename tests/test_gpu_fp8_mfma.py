"""W8A8 base GEMM on the MX-scaled fp8 MFMA (AitkGemmArgs.b_scale_mode 3; BASELINE config 5's "CDNA4 fp8 MFMA base"): kernel-level
parity of aitk_quant_rows_fp8 and of the persistent fp8 8-phase GEMM (with the bf16 LoRA slab, epilogues, grouped launches, ragged
shapes) against the oracle table, whose arithmetic is the exact product of the same e4m3 codes accumulated in fp32."""
import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def _rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


@pytest.mark.parametrize("M,K,seg,colmul", [(300, 3072, False, False), (1027, 256, False, True), (512, 15360, False, False), (96, 384, True, True)])
def test_quant_rows_fp8_kernel_vs_oracle(M, K, seg, colmul):
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    g = torch.Generator().manual_seed(M + K)
    cm = (torch.rand(K, generator=g) * 0.01 + 1e-4).cuda() if colmul else None
    if seg:  # rows of two segments inside a wider joint buffer (the image half of the attention output)
        buf = (torch.randn(2, 80, K + 64, generator=g) * 3).to(bf).cuda()
        x, x_seg = buf[0, 32:, :K], (48, 80 * (K + 64))
        M = 96
    else:
        x, x_seg = (torch.randn(M, K, generator=g) * torch.rand(M, 1, generator=g) * 5).to(bf).cuda(), None
    outs = []
    for o_ in (ops, ref_ops):
        q = torch.zeros(M, K, dtype=torch.uint8, device="cuda")
        rs = torch.zeros(M, device="cuda")
        o_.quant_rows_fp8(x, q, rs, col_mul=cm, x_seg=x_seg, M=M)
        outs.append((q, rs))
    assert torch.equal(outs[0][1], outs[1][1])
    bad = (outs[0][0] != outs[1][0]).float().mean().item()
    assert bad == 0.0, bad
    # de-quantised values reproduce the input to e4m3 precision (3 mantissa bits: <= 2^-4 relative on normal numbers)
    deq = outs[0][0].view(torch.float8_e4m3fn).float() * outs[0][1][:, None]
    ref = ref_ops._seg_view(x, x_seg, M).float() * (cm[None, :] if cm is not None else 1.0)
    assert _rel(deq, ref) < 0.04


def _operands(M, N, K, K2, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(M, K, generator=g) * 1.5).to(bf).cuda()
    w = (torch.randn(N, K, generator=g) * 0.02).to(bf).cuda()
    ws = (w.float().abs().amax(1) / 448.0).contiguous()
    wq = (w.float() / ws[:, None]).to(torch.float8_e4m3fn).view(torch.uint8).contiguous()
    a2 = (torch.randn(M, K2, generator=g) * 0.3).to(bf).cuda() if K2 else None
    b2 = (torch.randn(N, K2, generator=g) * 0.05).to(bf).cuda() if K2 else None
    bias = (torch.randn(N, generator=g) * 0.1).to(bf).cuda()
    return x, wq, ws, a2, b2, bias


@pytest.mark.parametrize("M,N,K,K2", [(512, 512, 3072, 48), (4608, 3072, 3072, 48), (1000, 520, 400, 0), (260, 264, 128, 16), (2048, 3072, 12288, 96),
                                      (300, 256, 144, 48)])
def test_w8a8_gemm_kernel_vs_oracle(M, N, K, K2):
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    x, wq, ws, a2, b2, bias = _operands(M, N, K, K2, M + N + K)
    xq = torch.empty(M, K, dtype=torch.uint8, device="cuda")
    xs = torch.empty(M, device="cuda")
    ops.quant_rows_fp8(x, xq, xs)
    outs = []
    for o_ in (ops, ref_ops):
        out = torch.full((M, N), float("nan"), dtype=bf, device="cuda")
        o_.gemm_nt(xq, wq, out, bias=bias, a2=a2, b2=b2, a_scale=xs, b_scale=ws, b_scale_mode=3)
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0].float()).all()
    assert _rel(outs[0], outs[1]) < 3e-3, _rel(outs[0], outs[1])  # one bf16 rounding of the output; the fp8 products themselves are exact
    # and the quantised product is the bf16 product to fp8 precision
    full = torch.empty(M, N, dtype=bf, device="cuda")
    ref_ops.gemm_nt(x, (wq.view(torch.float8_e4m3fn).float() * ws[:, None]).to(bf), full, bias=bias, a2=a2, b2=b2)
    assert _rel(outs[0], full) < 0.05, _rel(outs[0], full)


def test_w8a8_gemm_epilogues_dgrad_form_and_grouped_launch():
    """GELU (+ saved pre-activation), gate-residual, DGELU and ACCUM epilogues; b_scale = None with the weight scale folded into the
    quantised operand (the data-gradient form); two problems in one grouped launch == two single launches, bitwise."""
    from ai_toolkit_amd import _capi, ops
    from oracle import ref_ops

    M, N, K, K2 = 1536, 1024, 2048, 48
    x, wq, ws, a2, b2, bias = _operands(M, N, K, K2, 5)
    g = torch.Generator().manual_seed(9)
    res = torch.randn(M, N, generator=g).to(bf).cuda()
    gate = torch.randn(3, N, generator=g).to(bf).cuda()
    u_in = torch.randn(M, N, generator=g).to(bf).cuda()
    xq, xs = torch.empty(M, K, dtype=torch.uint8, device="cuda"), torch.empty(M, device="cuda")
    ops.quant_rows_fp8(x, xq, xs)
    for flags, kw in ((_capi.EPI_GELU, dict(aux=True)), (_capi.EPI_GATE_RES, dict(aux=True, aux_in=res, gate=gate, gate_rows=512)),
                      (_capi.EPI_DGELU, dict(aux_in=u_in)), (_capi.EPI_ACCUM, dict())):
        outs = []
        for o_ in (ops, ref_ops):
            out = res.clone() if flags == _capi.EPI_ACCUM else torch.empty(M, N, dtype=bf, device="cuda")
            aux = torch.empty(M, N, dtype=bf, device="cuda") if kw.get("aux") else None
            o_.gemm_nt(xq, wq, out, bias=bias, a2=a2, b2=b2, a_scale=xs, b_scale=ws, b_scale_mode=3, flags=flags, aux_out=aux,
                       aux_in=kw.get("aux_in"), gate=kw.get("gate"), gate_rows=kw.get("gate_rows", 0))
            outs.append((out, aux))
        assert _rel(outs[0][0], outs[1][0]) < 4e-3, (flags, _rel(outs[0][0], outs[1][0]))
        if outs[0][1] is not None:
            assert _rel(outs[0][1], outs[1][1]) < 4e-3, flags
    # data-gradient form: dX = rowscale * (Q(dY * wscale) . Wq^T-codes): contraction over the output channels
    dy = (torch.randn(M, N, generator=g) * 0.1).to(bf).cuda()
    wqt = wq.t().contiguous()  # [K, N] bytes: rows = input channels
    dyq, dys = torch.empty(M, N, dtype=torch.uint8, device="cuda"), torch.empty(M, device="cuda")
    ops.quant_rows_fp8(dy, dyq, dys, col_mul=ws)
    dx, dx_ref = torch.empty(M, K, dtype=bf, device="cuda"), torch.empty(M, K, dtype=bf, device="cuda")
    ops.gemm_nt(dyq, wqt, dx, a_scale=dys, b_scale=None, b_scale_mode=3)
    ref_ops.gemm_nt(dyq, wqt, dx_ref, a_scale=dys, b_scale=None, b_scale_mode=3)
    assert _rel(dx, dx_ref) < 3e-3
    w_deq = (wq.view(torch.float8_e4m3fn).float() * ws[:, None])
    assert _rel(dx, dy.float() @ w_deq) < 0.05
    # grouped launch
    x2, wq2, ws2, a22, b22, bias2 = _operands(640, N, K, K2, 6)
    xq2, xs2 = torch.empty(640, K, dtype=torch.uint8, device="cuda"), torch.empty(640, device="cuda")
    ops.quant_rows_fp8(x2, xq2, xs2)
    single = [torch.empty(M, N, dtype=bf, device="cuda"), torch.empty(640, N, dtype=bf, device="cuda")]
    ops.gemm_nt(xq, wq, single[0], bias=bias, a2=a2, b2=b2, a_scale=xs, b_scale=ws, b_scale_mode=3)
    ops.gemm_nt(xq2, wq2, single[1], bias=bias2, a2=a22, b2=b22, a_scale=xs2, b_scale=ws2, b_scale_mode=3)
    grouped = [torch.empty(M, N, dtype=bf, device="cuda"), torch.empty(640, N, dtype=bf, device="cuda")]
    with ops.recording() as la:
        ops.gemm_nt(xq, wq, grouped[0], bias=bias, a2=a2, b2=b2, a_scale=xs, b_scale=ws, b_scale_mode=3)
    with ops.recording() as lb:
        ops.gemm_nt(xq2, wq2, grouped[1], bias=bias2, a2=a22, b2=b22, a_scale=xs2, b_scale=ws2, b_scale_mode=3)
    ops.replay_paired(la, lb)
    torch.cuda.synchronize()
    assert torch.equal(single[0], grouped[0]) and torch.equal(single[1], grouped[1])


def _step_both_modes(model, net, ops, batch):
    """one step (no optimizer movement) of the SAME quantised model in weight-only mode and in W8A8 mode -> (loss, adapter gradients) each"""
    from ai_toolkit_amd.trainer import FluxLoRATrainStep

    lat, emb, pooled, noise, ts = batch
    kw = dict(lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    out = {}
    for mode in (False, True):
        model.fp8_mfma = mode
        l = FluxLoRATrainStep(model, net, ops, **kw).step(lat, emb, pooled, noise=noise, timesteps=ts).item()
        out[mode] = (l, [g.detach().clone() for m in net.unet_loras for g in (m.lora_down.weight.grad, m.lora_up.weight.grad)])
    return out


def _rel_lists(a, b):
    import math

    num = sum(((x.float() - y.float()) ** 2).sum().item() for x, y in zip(a, b))
    den = sum((y.float() ** 2).sum().item() for y in b)
    return math.sqrt(num / max(den, 1e-300))


def test_w8a8_train_step_vs_weight_only_small_model():
    from ai_toolkit_amd import ops
    from tests.test_gpu_e2e import _batch, _build

    ref, ref_net, nat, net = _build(rank=32)
    nat.quantize_base_fp8()
    o = _step_both_modes(nat, net, ops, _batch(2))
    e_l = abs(o[True][0] - o[False][0]) / abs(o[False][0])
    e_g = _rel_lists(o[True][1], o[False][1])
    print(f"PARITY W8A8 vs weight-only fp8 (FLUX 2+3 blocks, r32): loss {o[True][0]:.6f} vs {o[False][0]:.6f} (rel {e_l:.2e}); adapter-gradient rel diff {e_g:.3e}")
    assert e_l < 2e-2 and e_g < 0.15, (e_l, e_g)


def test_w8a8_train_step_vs_weight_only_full_depth_r32_at_1024():
    """BASELINE config 5 at size: 19 + 38 blocks, 4096 + 512 tokens, r32; the W8A8 step against the weight-only step of the same
    quantised model (which tests/test_gpu_parity_r3.py holds to the fp32 oracle)."""
    from tests.test_gpu_fullsize import _batch, _flux

    model, net, ops = _flux(19, 38, rank=32)
    model.quantize_base_fp8(release_bf16=True)
    o = _step_both_modes(model, net, ops, _batch(1))
    e_l = abs(o[True][0] - o[False][0]) / abs(o[False][0])
    e_g = _rel_lists(o[True][1], o[False][1])
    worst = max(_rel_lists([a], [b]) for a, b in zip(o[True][1], o[False][1]))
    print(f"PARITY W8A8 vs weight-only fp8, full depth 19+38 @1024^2 r32 B=1: loss {o[True][0]:.6f} vs {o[False][0]:.6f} (rel {e_l:.2e}); "
          f"adapter-gradient rel diff {e_g:.3e} (worst module {worst:.3e})")
    assert e_l < 3e-2 and e_g < 0.25, (e_l, e_g)
