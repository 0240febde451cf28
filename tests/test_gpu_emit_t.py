"""AITK_EPI_EMIT_T on the HIP kernels: the BIAS | GELU launch of the persistent GEMM that also leaves the column-tile partials of the consumer layer's
lora_down product, aitk_lora_down_raw / aitk_lora_t_finish, and the FLUX step with the route switched on.
  * the launch's own outputs (gelu(u), u) are BIT-IDENTICAL to the plain BIAS | GELU launch (the emission only reads what the epilogue stores);
  * T from the partials equals aitk_lora_down on the stored GELU output to fp32 summation order (the slab's hi part to 1 bf16 ulp, hi + lo to 1e-5);
  * single and grouped (image + text stream) launches, ragged row counts (aspect-ratio buckets: the last row tile stores what exists), LoRA slab on the producer, a column window of a wider lora_down matrix + one raw tile
    (the single blocks' proj_out over [attn | gelu(mlp)]);
  * the train step with model.emit_t: loss and adapter gradients against the step without it (reference semantic: toolkit/network_mixins.py:309-321)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def _rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


def _slab_value(T, rp=16):
    return T[:, :rp].float() + T[:, rp:2 * rp].float()


@pytest.mark.parametrize("M,N,K,slab", [(4608, 3072, 256, True), (7168, 12288, 512, True), (512, 1024, 128, False), (4464, 3072, 256, True), (300, 1024, 128, False), (136, 512, 128, True)])
@pytest.mark.parametrize("R", [16, 32])
def test_emitting_gelu_launch_matches_plain_launch_and_lora_down(M, N, K, slab, R):
    from ai_toolkit_amd import ops
    from ai_toolkit_amd._capi import EPI_GELU

    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(bf)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(bf)
    bias = (torch.randn(N, device="cuda", generator=g) * 0.1).to(bf)
    a2 = (torch.randn(M, 48, device="cuda", generator=g) * 0.1).to(bf) if slab else None
    b2 = (torch.randn(N, 48, device="cuda", generator=g) * 0.05).to(bf) if slab else None
    A = torch.randn(R, N, device="cuda", generator=g) * 0.05
    a_hi = A.to(bf)
    a_lo = (A - a_hi.float()).to(bf)
    h0, u0 = torch.empty(M, N, dtype=bf, device="cuda"), torch.empty(M, N, dtype=bf, device="cuda")
    ops.gemm_nt(x, w, h0, bias=bias, a2=a2, b2=b2, flags=EPI_GELU, aux_out=u0, stage_mode=4)
    h1, u1 = torch.empty(M, N, dtype=bf, device="cuda"), torch.empty(M, N, dtype=bf, device="cuda")
    nt = N // 256
    partial = torch.full((nt + 1, M, R), float("nan"), device="cuda")
    ops.gemm_nt(x, w, h1, bias=bias, a2=a2, b2=b2, flags=EPI_GELU, aux_out=u1, emit_t=(a_hi, a_lo, partial, 0), stage_mode=4)
    torch.cuda.synchronize()
    assert torch.equal(h0, h1) and torch.equal(u0, u1)
    assert torch.isfinite(partial[:nt]).all() and torch.isnan(partial[nt]).all()
    # every tile against the fp32 product of the stored values
    want = torch.stack([h1[:, t * 256:(t + 1) * 256].float() @ (a_hi.float() + a_lo.float())[:, t * 256:(t + 1) * 256].t() for t in range(nt)])
    assert _rel(partial[:nt], want) < 2e-6, _rel(partial[:nt], want)
    # T as the consumer's GEMM takes it: finish vs aitk_lora_down on the stored GELU output (scale + per-sample multipliers)
    rpb = (M + 3) // 4
    mult = torch.tensor([1.0, 0.4, 2.0, 0.7], device="cuda")
    T_ref, T = torch.empty(M, 3 * R, dtype=bf, device="cuda"), torch.empty(M, 3 * R, dtype=bf, device="cuda")
    ops.lora_down(h1, a_hi, T_ref, scale=0.5, mult=mult, rows_per_batch=rpb, p_lo=a_lo, split=R)
    ops.lora_t_finish(partial, nt, T, scale=0.5, mult=mult, rows_per_batch=rpb, split=R)
    assert torch.equal(T[:, :R], T[:, 2 * R:3 * R])
    assert _rel(_slab_value(T, R), _slab_value(T_ref, R)) < 1e-5, _rel(_slab_value(T, R), _slab_value(T_ref, R))
    # one more tile from aitk_lora_down_raw (the attention third of a single block's proj_out input) rides along at either rank
    xa = (torch.randn(M, 256, device="cuda", generator=g) * 0.5).to(bf)
    Aa = torch.randn(R, 256, device="cuda", generator=g) * 0.05
    ops.lora_down_raw(xa, Aa.to(bf), partial[nt], p_lo=(Aa - Aa.to(bf).float()).to(bf))
    torch.cuda.synchronize()
    want_raw = xa.float() @ (Aa.to(bf).float() + (Aa - Aa.to(bf).float()).to(bf).float()).t()
    assert _rel(partial[nt], want_raw) < 2e-6


def test_column_window_plus_raw_tile_and_grouped_launch():
    from ai_toolkit_amd import ops
    from ai_toolkit_amd._capi import EPI_GELU

    g = torch.Generator(device="cuda").manual_seed(3)
    d, K = 1024, 256
    # (a) proj_out of a single block: input [attn (d) | gelu(mlp) (4d)], lora_down [16, 5d]
    M = 2304
    x = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(bf)
    w = (torch.randn(4 * d, K, device="cuda", generator=g) * 0.05).to(bf)
    bias = (torch.randn(4 * d, device="cuda", generator=g) * 0.1).to(bf)
    A = torch.randn(16, 5 * d, device="cuda", generator=g) * 0.05
    a_hi = A.to(bf)
    a_lo = (A - a_hi.float()).to(bf)
    cat = (torch.randn(M, 5 * d, device="cuda", generator=g) * 0.5).to(bf)
    u = torch.empty(M, 4 * d, dtype=bf, device="cuda")
    nt = 4 * d // 256
    partial = torch.zeros(nt + 1, M, 16, device="cuda")
    ops.gemm_nt(x, w, cat[:, d:], bias=bias, flags=EPI_GELU, aux_out=u, emit_t=(a_hi[:, d:], a_lo[:, d:], partial, 0), stage_mode=4)
    ops.lora_down_raw(cat[:, :d], a_hi[:, :d], partial[nt], p_lo=a_lo[:, :d])
    T, T_ref = torch.empty(M, 48, dtype=bf, device="cuda"), torch.empty(M, 48, dtype=bf, device="cuda")
    ops.lora_t_finish(partial, nt + 1, T, scale=1.0, split=16)
    ops.lora_down(cat, a_hi, T_ref, scale=1.0, p_lo=a_lo, split=16)
    assert _rel(_slab_value(T), _slab_value(T_ref)) < 1e-5
    # (b) the image and the text stream of a double block in ONE grouped launch: each problem fills its own slab
    Mi, Mt, N = 4096, 512, 4 * d
    xs = [(torch.randn(m, K, device="cuda", generator=g) * 0.5).to(bf) for m in (Mi, Mt)]
    ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.05).to(bf) for _ in range(2)]
    bs = [(torch.randn(N, device="cuda", generator=g) * 0.1).to(bf) for _ in range(2)]
    As = [torch.randn(16, N, device="cuda", generator=g) * 0.05 for _ in range(2)]
    hs = [torch.empty(m, N, dtype=bf, device="cuda") for m in (Mi, Mt)]
    us = [torch.empty(m, N, dtype=bf, device="cuda") for m in (Mi, Mt)]
    ps = [torch.full((N // 256, m, 16), float("nan"), device="cuda") for m in (Mi, Mt)]
    recs = []
    for i in range(2):
        ah = As[i].to(bf)
        al = (As[i] - ah.float()).to(bf)
        with ops.recording() as launches:
            ops.gemm_nt(xs[i], ws[i], hs[i], bias=bs[i], flags=EPI_GELU, aux_out=us[i], emit_t=(ah, al, ps[i], 0))
        recs.append((launches, ah, al))
    ops.replay_paired(recs[0][0], recs[1][0])
    torch.cuda.synchronize()
    for i in range(2):
        ah, al = recs[i][1], recs[i][2]
        want = torch.stack([hs[i][:, t * 256:(t + 1) * 256].float() @ (ah.float() + al.float())[:, t * 256:(t + 1) * 256].t() for t in range(N // 256)])
        assert _rel(ps[i], want) < 2e-6, (i, _rel(ps[i], want))
        h_ref, u_ref = torch.empty_like(hs[i]), torch.empty_like(us[i])
        ops.gemm_nt(xs[i], ws[i], h_ref, bias=bs[i], flags=EPI_GELU, aux_out=u_ref, stage_mode=4)
        assert torch.equal(h_ref, hs[i]) and torch.equal(u_ref, us[i])


def test_contract_refusals():
    from ai_toolkit_amd import ops
    from ai_toolkit_amd._capi import EPI_GELU

    M, N, K = 512, 640, 128  # ragged COLUMN tiles: a wave's 64 columns must lie inside the matrix (any row count is fine: stores are predicated per row)
    x, w = torch.zeros(M, K, dtype=bf, device="cuda"), torch.zeros(N, K, dtype=bf, device="cuda")
    bias, h, u = torch.zeros(N, dtype=bf, device="cuda"), torch.empty(M, N, dtype=bf, device="cuda"), torch.empty(M, N, dtype=bf, device="cuda")
    a = torch.zeros(16, N, dtype=bf, device="cuda")
    with pytest.raises(AssertionError):
        ops.gemm_nt(x, w, h, bias=bias, flags=EPI_GELU, aux_out=u, emit_t=(a, a, torch.zeros(2, M, 16, device="cuda"), 0))


def test_flux_step_with_emission_matches_the_step_without():
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from tests.test_gpu_e2e import _batch, _build

    res = {}
    for emit in (False, True):
        _, _, nat, net = _build()
        nat.emit_t = emit
        n_fin = [0]
        orig = ops.lora_t_finish

        def counted(*a, **k):
            n_fin[0] += 1
            return orig(*a, **k)

        ops.lora_t_finish = counted
        try:
            step = FluxLoRATrainStep(nat, net, ops, lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
            lat, emb, pooled, noise, ts = _batch(2)  # 96 image + 40 text tokens per sample: ragged row tiles in every stream
            loss = step.step(lat, emb, pooled, noise=noise, timesteps=ts).item()
        finally:
            ops.lora_t_finish = orig
        res[emit] = (loss, net.arena_g.clone(), n_fin[0])
    assert res[False][2] == 0
    (l0, g0, _), (l1, g1, n1) = res[False], res[True]
    assert n1 == 2 * 2 + 3  # two streams of each double block + every single block (tests/test_gpu_e2e.CFG: 2 + 3 blocks)
    assert abs(l1 - l0) <= 2e-4 * abs(l0), (l0, l1)
    assert _rel(g1, g0) < 5e-3, _rel(g1, g0)
