"""`FluxTransformer2DModel.recompute_gelu`: the GELU outputs are dropped after the forward pass and the lora_down gradients of ff.net.2 /
proj_out are formed from the saved pre-activation (aitk_lora_wgrad2: [attention output | gelu(u)] in two parts).  The values are the same
(GELU of the rounded pre-activation is what the forward pass fed the layer), so everything must match the default graph and the oracle's
autograd; the tape must no longer hold the GELU outputs."""
import torch

import ai_toolkit_amd  # noqa: F401
from tests.test_host_graph_cpu import build_pair, inputs


def _run(recompute, network_type="lora"):
    ref, ref_net, nat, net = build_pair(network_type=network_type)
    nat.recompute_gelu = recompute
    hidden, enc, pooled, t, img_ids, txt_ids, guid = inputs()
    with ref_net:
        pred_ref = ref(hidden, enc, pooled, t, img_ids, txt_ids, guid)
        w = torch.randn(pred_ref.shape, generator=torch.Generator().manual_seed(11))
        (pred_ref * w).sum().backward()
    with net:
        pred = nat.forward_native(hidden, enc, pooled, t, img_ids, txt_ids, guid)
        net.zero_grad_arena()
        nat.backward_native(w)
    grads = [(a.lora_down.weight.grad.clone(), a.lora_up.weight.grad.clone()) for a in net.unet_loras]
    refg = [(b.lora_down.weight.grad.clone(), b.lora_up.weight.grad.clone()) for b in ref_net.unet_loras]
    return pred, pred_ref, grads, refg, [a.lora_name for a in net.unet_loras]


def test_recompute_gelu_equals_the_default_graph_and_the_oracle():
    pred0, pred_ref, g0, refg, names = _run(False)
    pred1, _, g1, _, _ = _run(True)
    assert torch.equal(pred0, pred1)
    assert torch.allclose(pred1, pred_ref, rtol=1e-4, atol=1e-5)
    for (d0, u0), (d1, u1), (dr, ur), nm in zip(g0, g1, refg, names):
        # fp32 on the CPU table: gelu(u) recomputed == gelu(u) stored, so the two graphs agree to the last bit of the accumulation order
        assert torch.allclose(d0, d1, rtol=1e-6, atol=1e-7) and torch.equal(u0, u1), nm
        assert ((d1 - dr).norm() / (dr.norm() + 1e-12)).item() < 2e-4, nm


def test_recompute_gelu_drops_the_gelu_outputs_from_the_tape(monkeypatch):
    from ai_toolkit_amd.graph import _ActInput
    from oracle import ref_ops

    ref, ref_net, nat, net = build_pair()
    nat.recompute_gelu = True
    seen = []
    real = ref_ops.lora_wgrad

    def spy(s, g, out, **kw):
        seen.append((None if g is None else tuple(g.shape), None if kw.get("g2") is None else tuple(kw["g2"].shape), kw.get("g2_act")))
        return real(s, g, out, **kw)

    monkeypatch.setattr(ref_ops, "lora_wgrad", spy)
    hidden, enc, pooled, t, img_ids, txt_ids, guid = inputs()
    with net:
        pred = nat.forward_native(hidden, enc, pooled, t, img_ids, txt_ids, guid)
        net.zero_grad_arena()
        nat.backward_native(torch.ones_like(pred))
    two_part = [x for x in seen if x[1] is not None]
    n_dbl, n_sgl = len(nat.transformer_blocks), len(nat.single_transformer_blocks)
    assert len(two_part) == 2 * n_dbl + n_sgl and all(x[2] == "gelu" for x in two_part)
    assert sum(1 for x in two_part if x[0] is None) == 2 * n_dbl  # ff.net.2 of both streams: gelu(u) alone
    assert sum(1 for x in two_part if x[0] is not None) == n_sgl  # proj_out: [attention output | gelu(u)]
    # DoRA reads its input tensor in more places: the outputs stay on the tape there
    ref2, ref_net2, nat2, net2 = build_pair(network_type="dora")
    nat2.recompute_gelu = True
    seen.clear()
    with net2:
        pred = nat2.forward_native(hidden, enc, pooled, t, img_ids, txt_ids, guid)
        net2.zero_grad_arena()
        nat2.backward_native(torch.ones_like(pred))
    assert not [x for x in seen if x[1] is not None] and _ActInput is not None
