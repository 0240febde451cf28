"""aitk_gemm_nt_grouped: two problems (same N, K, K2, flags; different operands and row counts) in one persistent launch must give
bitwise the results of two aitk_gemm_nt calls — every output tile is computed by the same code on the same operands — and the FLUX
double block with its image / text launches merged (graph.FusedGraphBase._paired) must be bitwise the sequential graph."""

import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _problem(M, N, K, r, g, flags=0, gate=False):
    from ai_toolkit_amd import ops

    x = torch.randn(M, K, generator=g).to(BF).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF).cuda()
    kw = dict(bias=torch.randn(N, generator=g).to(BF).cuda(), flags=flags)
    if r:
        kw.update(a2=torch.randn(M, r, generator=g).to(BF).cuda(), b2=(torch.randn(N, r, generator=g) * 0.1).to(BF).cuda())
    if gate:  # gate-residual epilogue (to_out / ff.net.2 of a double block)
        rows = M // 2
        kw.update(flags=flags | 16, aux_in=torch.randn(M, N, generator=g).to(BF).cuda(), gate=torch.randn(2, N, generator=g).to(BF).cuda(),
                  gate_rows=rows, aux_out=torch.empty(M, N, dtype=BF, device="cuda"))
    return x, w, kw


@pytest.mark.parametrize("M1,M2,N,K,r,gate", [(28672, 3584, 3072, 3072, 48, False), (4096, 512, 3072, 3072, 48, True),
                                              (16384, 2048, 12288, 3072, 48, False), (8192, 1000, 3072, 12288, 0, True),
                                              (1024, 256, 768, 256, 16, False)])
def test_grouped_launch_is_bitwise_two_single_launches(M1, M2, N, K, r, gate):
    from ai_toolkit_amd import ops

    g = torch.Generator().manual_seed(M1 + N)
    pa, pb = _problem(M1, N, K, r, g, gate=gate), _problem(M2, N, K, r, g, gate=gate)
    ref, got = [], []
    for x, w, kw in (pa, pb):
        o = torch.full((x.shape[0], N), float("nan"), dtype=BF, device="cuda")
        ops.gemm_nt(x, w, o, **kw)
        ref.append((o, kw["aux_out"].clone() if gate else None))
    recs = []
    for x, w, kw in (pa, pb):
        o = torch.full((x.shape[0], N), float("nan"), dtype=BF, device="cuda")
        if gate:
            kw["aux_out"].fill_(float("nan"))
        with ops.recording() as rec:
            ops.gemm_nt(x, w, o, **kw)
        assert [e[0] for e in rec if e[0] != "_keepalive"] == ["aitk_gemm_nt"]
        recs.append(rec)
        got.append(o)
    ops.replay_paired(*recs)
    torch.cuda.synchronize()
    for (o_ref, aux_ref), o, (x, w, kw) in zip(ref, got, (pa, pb)):
        assert torch.equal(o_ref, o)
        if gate:
            assert torch.equal(aux_ref, kw["aux_out"])


def test_mismatched_problems_fall_back_to_two_launches():
    from ai_toolkit_amd import ops

    g = torch.Generator().manual_seed(1)
    xa, wa, kwa = _problem(4096, 3072, 3072, 48, g)
    xb, wb, kwb = _problem(2048, 1536, 3072, 48, g)  # different N: not groupable
    oa, ob = torch.empty(4096, 3072, dtype=BF, device="cuda"), torch.empty(2048, 1536, dtype=BF, device="cuda")
    ra, rb = torch.empty_like(oa), torch.empty_like(ob)
    ops.gemm_nt(xa, wa, ra, **kwa)
    ops.gemm_nt(xb, wb, rb, **kwb)
    with ops.recording() as r1:
        ops.gemm_nt(xa, wa, oa, **kwa)
    with ops.recording() as r2:
        ops.gemm_nt(xb, wb, ob, **kwb)
    ops.replay_paired(r1, r2)
    torch.cuda.synchronize()
    assert torch.equal(oa, ra) and torch.equal(ob, rb)


def test_flux_step_with_merged_stream_launches_is_bitwise_the_sequential_graph():
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from tests.test_gpu_e2e import _batch, _build

    kw = dict(lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    _, _, nat_a, net_a = _build()
    _, _, nat_b, net_b = _build()
    nat_a.pair_streams, nat_b.pair_streams = True, False
    sa, sb = FluxLoRATrainStep(nat_a, net_a, ops, **kw), FluxLoRATrainStep(nat_b, net_b, ops, **kw)
    for k in range(2):
        lat, emb, pooled, noise, ts = _batch(2, seed=90 + k)
        la = sa.step(lat, emb, pooled, noise=noise, timesteps=ts).clone()
        ga = net_a.arena_g.clone()
        lb = sb.step(lat, emb, pooled, noise=noise, timesteps=ts).clone()
        assert torch.equal(la, lb) and torch.equal(ga, net_b.arena_g) and torch.equal(net_a.arena_p, net_b.arena_p), k


def test_fp8_base_step_with_merged_stream_launches_is_bitwise_the_sequential_graph():
    """weight-only fp8 base: each layer's weight is expanded into a bf16 scratch right before its GEMM; the two merged streams use
    separate scratch buffers (graph._dq_slot), so the merged order must still give the sequential result bit for bit."""
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from tests.test_gpu_e2e import _batch, _build

    kw = dict(lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    _, _, nat_a, net_a = _build(rank=32)
    _, _, nat_b, net_b = _build(rank=32)
    nat_a.quantize_base_fp8()
    nat_b.quantize_base_fp8()
    nat_a.pair_streams, nat_b.pair_streams = True, False
    sa, sb = FluxLoRATrainStep(nat_a, net_a, ops, **kw), FluxLoRATrainStep(nat_b, net_b, ops, **kw)
    for k in range(2):
        lat, emb, pooled, noise, ts = _batch(2, seed=95 + k)
        la = sa.step(lat, emb, pooled, noise=noise, timesteps=ts).clone()
        ga = net_a.arena_g.clone()
        lb = sb.step(lat, emb, pooled, noise=noise, timesteps=ts).clone()
        assert torch.equal(la, lb) and torch.equal(ga, net_b.arena_g) and torch.equal(net_a.arena_p, net_b.arena_p), k
