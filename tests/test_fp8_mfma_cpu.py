"""W8A8 mode of the fp8 base (quantize_base_fp8(mfma=True); BASELINE config 5's "CDNA4 fp8 MFMA base") on the host graph with the oracle
kernel table: plumbing (per-token activation quantisation in front of every base GEMM, weight scale folded into dY for the data
gradient, operand cache) and the size of the deviation from the weight-only arithmetic the reference's quantisers define."""
import torch

import ai_toolkit_amd  # noqa: F401
from oracle import ref_ops
from tests.test_host_graph_cpu import build_pair, inputs


def _run(mfma, counter=None):
    ref, ref_net, nat, net = build_pair()
    nat.quantize_base_fp8(mfma=mfma)
    hidden, enc, pooled, t, img_ids, txt_ids, guid = inputs()
    orig = ref_ops.quant_rows_fp8
    if counter is not None:
        def counting(*a, **k):
            counter.append(a[0].shape)
            return orig(*a, **k)
        ref_ops.quant_rows_fp8 = counting
    try:
        with net:
            pred = nat.forward_native(hidden, enc, pooled, t, img_ids, txt_ids, guid)
            n_fwd = len(counter) if counter is not None else 0
            net.zero_grad_arena()
            nat.backward_native((2 * pred).detach())
    finally:
        ref_ops.quant_rows_fp8 = orig
    grads = [g.clone() for m in net.unet_loras for g in (m.lora_down.weight.grad, m.lora_up.weight.grad)]
    return pred, grads, nat, n_fwd


def _rel(a, b):
    num = sum(((x - y) ** 2).sum().item() for x, y in zip(a, b))
    den = sum((y ** 2).sum().item() for y in b)
    return (num / den) ** 0.5


def test_w8a8_mode_tracks_the_weight_only_base_within_fp8_precision():
    p_w, g_w, _, _ = _run(False)
    calls = []
    p_m, g_m, nat, n_fwd = _run(True, calls)
    assert nat.fp8_mfma
    # e4m3 keeps 3 mantissa bits: per-product rounding noise ~2^-5 rms on the activation operand, averaged over the contraction
    e_pred = ((p_m - p_w).norm() / p_w.norm()).item()
    e_grad = _rel(g_m, g_w)
    assert e_pred < 0.06, e_pred
    assert e_grad < 0.15, e_grad
    # forward: one quantisation per distinct GEMM input — double block: xn (q, k, v share it), attention output, xn2, h per stream;
    # single block: xn (q, k, v, proj_mlp share it) and the [attn | mlp] concatenation
    cfg = nat.config
    assert n_fwd == cfg["num_layers"] * 2 * 4 + cfg["num_single_layers"] * 2, (n_fwd, len(calls))
    assert len(calls) > n_fwd  # backward quantises the output gradients


def test_quant_rows_oracle_round_trip_and_column_multiplier():
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(37, 64, generator=g) * 3).to(torch.bfloat16)
    cm = torch.rand(64, generator=g) + 0.5
    q, rs = torch.zeros(37, 64, dtype=torch.uint8), torch.zeros(37)
    ref_ops.quant_rows_fp8(x, q, rs, col_mul=cm)
    v = x.float() * cm
    assert torch.allclose(rs, v.abs().amax(1) / 448.0)
    deq = q.view(torch.float8_e4m3fn).float() * rs[:, None]
    assert ((deq - v).abs() <= 2.0 ** -4 * v.abs().amax(1, keepdim=True) + 1e-9).all()
    assert float(deq.abs().max()) <= float(v.abs().max()) * (1 + 1e-6)
