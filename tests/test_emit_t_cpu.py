"""AITK_EPI_EMIT_T on the host logic (oracle kernel table): with `model.emit_t` the GELU launches of ff.net.0.proj / ff_context.net.0.proj / proj_mlp leave the
column-tile partials of the NEXT layer's lora_down product (toolkit/network_mixins.py:309-321: lora_down on the layer input, here the GELU output) and
aitk_lora_t_finish sums them; the single blocks' proj_out adds the attention half of its input as one more tile.  Same prediction, same gradients as the
graph that calls aitk_lora_down on the stored GELU output — and the route is really taken (launch names counted)."""
import pytest
import torch

import ai_toolkit_amd  # noqa: F401
from oracle import ref_ops
from tests.test_host_graph_cpu import build_pair, inputs


@pytest.fixture(autouse=True)
def _any_row_count(monkeypatch):
    """the HIP kernel emits for whole 256-row tiles only and the oracle table says the same; the tiny models of this file have a few dozen rows"""
    monkeypatch.setattr(ref_ops, "EMIT_T_ROW_TILE", 1)


def _run(nat, net, emit):
    nat.emit_t = emit
    calls = {"lora_t_finish": 0, "lora_down_raw": 0, "emit_gemm": 0}
    orig = {k: getattr(ref_ops, k) for k in ("lora_t_finish", "lora_down_raw", "gemm_nt")}

    def count(name):
        def f(*a, **k):
            calls[name] += 1
            return orig[name](*a, **k)
        return f

    def gemm(*a, **k):
        if k.get("emit_t") is not None:
            calls["emit_gemm"] += 1
        return orig["gemm_nt"](*a, **k)

    ref_ops.lora_t_finish, ref_ops.lora_down_raw, ref_ops.gemm_nt = count("lora_t_finish"), count("lora_down_raw"), gemm
    try:
        hidden, enc, pooled, timestep, img_ids, txt_ids, guidance = inputs()
        net.zero_grad_arena()
        with net:
            pred = nat.forward_native(hidden, enc, pooled, timestep, img_ids, txt_ids, guidance)
            g = torch.Generator().manual_seed(5)
            nat.backward_native(torch.randn(pred.shape, generator=g))
    finally:
        for k, v in orig.items():
            setattr(ref_ops, k, v)
    return pred.clone(), net.arena_g.clone(), calls


@pytest.mark.parametrize("rank", [16, 32, 8])
def test_emitted_t_equals_the_separate_lora_down_launch(rank):
    ref, ref_net, nat, net = build_pair(rank=rank)
    p0, g0, c0 = _run(nat, net, False)
    p1, g1, c1 = _run(nat, net, True)
    n_dbl, n_sgl = len(nat.transformer_blocks), len(nat.single_transformer_blocks)
    assert c0 == {"lora_t_finish": 0, "lora_down_raw": 0, "emit_gemm": 0}
    assert c1 == {"lora_t_finish": 2 * n_dbl + n_sgl, "lora_down_raw": n_sgl, "emit_gemm": 2 * n_dbl + n_sgl}, c1
    # fp32 table: only the summation order over the column tiles differs
    assert (p1 - p0).abs().max().item() <= 1e-5 * p0.abs().max().item()
    assert (g1 - g0).abs().max().item() <= 2e-5 * g0.abs().max().item()


def test_emission_is_skipped_where_it_cannot_apply():
    # rank 4 (rank_pad 16 holds — emission runs), DoRA consumers and dropout do not take it
    for kw in (dict(rank=16, network_type="dora"),):
        ref, ref_net, nat, net = build_pair(**kw)
        _, _, c = _run(nat, net, True)
        assert c["emit_gemm"] == 0 and c["lora_t_finish"] == 0, (kw, c)
    ref, ref_net, nat, net = build_pair(rank=16)
    net.dropout = 0.1
    for m in net.unet_loras:
        m.dropout = 0.1
    net.train()
    _, _, c = _run(nat, net, True)
    assert c["emit_gemm"] == 0
    # per-sample multipliers ride in the finish pass
    ref, ref_net, nat, net = build_pair(rank=16)
    net.multiplier = [1.0, 0.4]
    p0, g0, _ = _run(nat, net, False)
    p1, g1, c = _run(nat, net, True)
    assert c["emit_gemm"] > 0 and (p1 - p0).abs().max().item() <= 1e-5 * p0.abs().max().item() and (g1 - g0).abs().max().item() <= 2e-5 * g0.abs().max().item()
