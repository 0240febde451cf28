"""Device-side failure handling of the train loop on the HIP kernels (CPU twin with the reference citations: tests/test_failure_handling_cpu.py):
the loss guard of aitk_mse_loss_grad and the skipped optimizer launch of aitk_adamw_ema_step against the oracle table, and a planted NaN
batch through the whole fused step."""
import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def test_loss_guard_kernel_matches_the_oracle_table():
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    g = torch.Generator().manual_seed(0)
    B, T, F = 3, 96, 64
    pred = (torch.randn(B, T, F, generator=g)).to(bf).cuda()
    target = (torch.randn(B, T, F, generator=g)).to(bf).cuda()

    def run(table, pred_, max_loss):
        dp = torch.full_like(pred_, 7.0)
        lps, loss = torch.zeros(B, device="cuda"), torch.zeros(1, device="cuda")
        guard = torch.zeros(8, dtype=torch.int32, device="cuda")
        table.mse_loss_grad(pred_, target, dp, lps, loss, guard=guard, max_loss=max_loss)
        return dp, loss, guard.tolist()

    l_plain = run(ops, pred, None)[1].item()
    for pr, ml, want_gate in ((pred, None, 0), (pred, 2.0 * l_plain, 0), (pred, 0.5 * l_plain, 1)):
        dp_o, l_o, g_o = run(ops, pr, ml)
        dp_r, l_r, g_r = run(ref_ops, pr, ml)
        assert g_o == g_r and g_o[6] == want_gate, (g_o, g_r)
        assert abs(l_o.item() - l_r.item()) <= 1e-5 * abs(l_r.item())
        assert torch.allclose(dp_o.float(), dp_r.float(), rtol=1e-2, atol=1e-7)
        if want_gate:
            assert not dp_o.any() and abs(l_o.item() - ml) < 1e-6
    for bad_value in (float("nan"), float("inf")):
        bad = pred.clone()
        bad[1, 5, 9] = bad_value
        dp_o, l_o, g_o = run(ops, bad, None)
        assert l_o.item() == 0.0 and not dp_o.any() and g_o[:3] == [1, 1, 0] and g_o[6] == 1
        assert g_o == run(ref_ops, bad, None)[2]


def test_guarded_optimizer_launch_skips_on_device_and_counts_applied_steps():
    from ai_toolkit_amd import ops
    from oracle import ref_ops

    n = 3 * 4096 + 777
    g = torch.Generator().manual_seed(1)
    p0 = torch.randn(n, generator=g).cuda()
    grads = [torch.randn(n, generator=g).cuda() * 0.1 for _ in range(3)]
    kw = dict(lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.01, max_norm=1.0, ema_decay=0.99)

    def run(table, seq, guarded):
        p, m, v, e = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), p0.clone()
        guard = torch.zeros(8, dtype=torch.int32, device="cuda") if guarded else None
        norm = torch.zeros(1, device="cuda")
        step = 0
        for kind, gr in seq:
            gr = gr.clone()
            if kind == "nan":
                gr[1234] = float("nan")
            if kind == "gated" and guarded:
                guard[0] = 1
            step += 1
            table.adamw_ema_step(p, gr, m, v, step=step, ema=e, norm_out=norm, **kw, **(dict(guard=guard, n_micro=1) if guarded else {}))
        return p, m, v, e, (guard.tolist() if guarded else None)

    healthy = [("ok", grads[0]), ("ok", grads[1]), ("ok", grads[2])]
    a, b = run(ops, healthy, True), run(ops, healthy, False)
    assert all(torch.equal(x, y) for x, y in zip(a[:4], b[:4]))  # device-derived bias corrections == the host's
    assert a[4][3:6] == [3, 0, 0]
    seq = [("ok", grads[0]), ("nan", grads[1]), ("gated", grads[1]), ("ok", grads[2])]
    o, r = run(ops, seq, True), run(ref_ops, seq, True)
    assert o[4] == r[4] and o[4][3:6] == [2, 2, 0]
    for x, y in zip(o[:4], r[:4]):
        assert torch.isfinite(x).all() and torch.allclose(x, y, rtol=2e-5, atol=1e-7)
    # p, m, v after [ok, skip, skip, ok] == after [ok, ok] (the EMA saw two extra updates)
    two = run(ops, [("ok", grads[0]), ("ok", grads[2])], True)
    assert all(torch.equal(x, y) for x, y in zip(o[:3], two[:3]))


def test_planted_nan_batch_leaves_the_adapter_state_bit_identical_and_training_continues():
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from oracle.pairs import batch as _batch, build as _build

    _, _, nat, net = _build()
    _, _, nat2, net2 = _build()
    kw = dict(lr=1e-3, weight_decay=0.01, max_grad_norm=1.0, ema_decay=0.99)
    step, twin = FluxLoRATrainStep(nat, net, ops, **kw), FluxLoRATrainStep(nat2, net2, ops, **kw)
    b0, b1, b2 = _batch(2, seed=50), _batch(2, seed=51), _batch(2, seed=52)
    for s in (step, twin):
        s.step(b0[0], b0[1], b0[2], noise=b0[3], timesteps=b0[4])
    p0, m0, v0, e0, sh0 = (t.clone() for t in (net.arena_p, net.arena_m, net.arena_v, net.arena_ema, net.arena_shadow))
    bad = b1[0].clone()
    bad[1, 3, 2, 2] = float("nan")
    loss = step.step(bad, b1[1], b1[2], noise=b1[3], timesteps=b1[4])
    assert loss.item() == 0.0
    assert torch.equal(net.arena_p, p0) and torch.equal(net.arena_m, m0) and torch.equal(net.arena_v, v0) and torch.equal(net.arena_shadow, sh0)
    assert torch.allclose(net.arena_ema, e0 - (1.0 - 0.99) * (e0 - p0), rtol=0, atol=1e-7)
    c = step.guard_counters()
    assert c["nonfinite_losses"] == 1 and c["steps_skipped"] == 1 and c["steps_applied"] == 1 and c["last_step_skipped"]
    l_a = step.step(b2[0], b2[1], b2[2], noise=b2[3], timesteps=b2[4])
    l_b = twin.step(b2[0], b2[1], b2[2], noise=b2[3], timesteps=b2[4])
    assert torch.equal(l_a, l_b) and torch.equal(net.arena_p, net2.arena_p) and torch.equal(net.arena_m, net2.arena_m) and torch.equal(net.arena_v, net2.arena_v)
    # max_loss on the device: a bound under the loss reports the bound and passes ZERO gradients on — the optimizer still steps on them like
    # torch.optim.AdamW on zero .grad tensors (decoupled weight decay only: p *= 1 - lr * wd; moments stay 0; the step is counted as applied)
    _, _, nat3, net3 = _build()
    ml = 0.5 * l_b.item()
    capped = FluxLoRATrainStep(nat3, net3, ops, max_loss=ml, **kw)
    p3 = net3.arena_p.clone()
    assert abs(capped.step(b2[0], b2[1], b2[2], noise=b2[3], timesteps=b2[4]).item() - ml) < 1e-6
    want = p3 * (1.0 - kw["lr"] * kw["weight_decay"])
    assert float((net3.arena_p - want).abs().max()) <= 1e-7 * float(want.abs().max()) and not net3.arena_m.any() and not net3.arena_v.any()
    c3 = capped.guard_counters()
    assert c3["clamped_losses"] == 1 and c3["steps_applied"] == 1 and c3["steps_skipped"] == 0
