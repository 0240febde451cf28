"""Failure handling of the reference's train loop on the fused step, host logic on the oracle kernel table:

  * non-finite loss: `if not torch.isfinite(loss): loss = torch.zeros_like(loss).requires_grad_(True)` (SDTrainer.py:2221-2224) — the
    micro-batch leaves no gradient; with no gradient on any parameter `optimizer.step()` skips every parameter (torch/optim: `if p.grad is
    None: continue`: no weight decay, no moment decay, no step count) while `ema.update()` still runs (SDTrainer.py:2285-2293);
  * `train.max_loss`: `loss = torch.clamp(loss, max=max_loss)` (SDTrainer.py:1049-1050) — above the bound the reported loss is max_loss and
    the derivative is 0.

The fused step decides both on the device (a guard buffer, no host sync) and must land where that torch sequence lands."""
import torch

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd.trainer import FluxLoRATrainStep
from oracle import ref_ops
from tests.test_host_graph_cpu import build_pair
from tests.test_train_step_cpu import batch

KW = dict(lr=1e-3, weight_decay=0.01, max_grad_norm=0.5, ema_decay=0.9)


def _state(net):
    return [t.clone() for t in (net.arena_p, net.arena_m, net.arena_v, net.arena_ema)]


def test_nan_batch_skips_the_update_like_adamw_without_grads_and_the_next_step_is_unaffected():
    _, _, nat, net = build_pair(rank=4)
    _, _, nat2, net2 = build_pair(rank=4)
    step, twin = FluxLoRATrainStep(nat, net, ref_ops, **KW), FluxLoRATrainStep(nat2, net2, ref_ops, **KW)
    b0, b1, b2 = batch(2, seed=10), batch(2, seed=11), batch(2, seed=12)
    for s in (step, twin):
        s.step(*b0[:3], noise=b0[3], timesteps=b0[4])
    assert all(torch.equal(a, b) for a, b in zip(_state(net), _state(net2)))
    p0, m0, v0, e0 = _state(net)
    bad = b1[0].clone()
    bad[0, 0, 0, 0] = float("nan")
    loss = step.step(bad, b1[1], b1[2], noise=b1[3], timesteps=b1[4])
    assert loss.item() == 0.0  # the reference reports zeros_like(loss)
    assert torch.equal(net.arena_p, p0) and torch.equal(net.arena_m, m0) and torch.equal(net.arena_v, v0)
    # toolkit/ema.py:126-139 ran anyway: s -= (1 - d) (s - p) with p unchanged
    want = e0 - (1.0 - 0.9) * (e0 - p0)
    assert torch.allclose(net.arena_ema, want, rtol=0, atol=1e-7)
    c = step.guard_counters()
    assert c == {"nonfinite_losses": 1, "clamped_losses": 0, "steps_applied": 1, "steps_skipped": 1, "last_step_skipped": True}
    assert torch.isfinite(net.arena_p).all() and torch.isfinite(net.arena_shadow.float()).all()
    # the following good step == the twin that never saw the bad batch (AdamW's bias corrections count APPLIED steps); the EMA differs by
    # the extra update above, nothing else
    step.step(*b2[:3], noise=b2[3], timesteps=b2[4])
    twin.step(*b2[:3], noise=b2[3], timesteps=b2[4])
    assert torch.equal(net.arena_p, net2.arena_p) and torch.equal(net.arena_m, net2.arena_m) and torch.equal(net.arena_v, net2.arena_v)
    assert step.guard_counters()["steps_applied"] == 2 and not step.guard_counters()["last_step_skipped"]


def test_skipped_step_equals_the_torch_sequence_on_parameters_without_grad():
    """the same thing said with torch's own optimizer: zero_grad(set_to_none) -> (no backward) -> clip -> AdamW.step -> ema.update"""
    _, _, nat, net = build_pair(rank=4)
    step = FluxLoRATrainStep(nat, net, ref_ops, **KW)
    b0, b1 = batch(2, seed=10), batch(2, seed=11)
    step.step(*b0[:3], noise=b0[3], timesteps=b0[4])
    p = torch.nn.Parameter(net.arena_p.clone())
    opt = torch.optim.AdamW([p], lr=KW["lr"], eps=1e-6, weight_decay=KW["weight_decay"])
    opt.state[p] = {"step": torch.tensor(1.0), "exp_avg": net.arena_m.clone(), "exp_avg_sq": net.arena_v.clone()}
    ema = net.arena_ema.clone()
    opt.zero_grad(set_to_none=True)
    torch.nn.utils.clip_grad_norm_([p], KW["max_grad_norm"])
    opt.step()
    ema.sub_((ema - p.detach()) * (1.0 - KW["ema_decay"]))
    bad = b1[3].clone()
    bad[1, 3, 2, 1] = float("inf")
    step.step(*b1[:3], noise=bad, timesteps=b1[4])
    assert torch.equal(net.arena_p, p.detach()) and torch.equal(net.arena_m, opt.state[p]["exp_avg"]) and torch.equal(net.arena_v, opt.state[p]["exp_avg_sq"])
    assert float(opt.state[p]["step"]) == 1.0
    assert torch.allclose(net.arena_ema, ema, rtol=0, atol=1e-7)


def test_max_loss_clamp_reports_the_bound_and_passes_no_gradient():
    _, _, nat, net = build_pair(rank=4)
    _, _, nat2, net2 = build_pair(rank=4)
    b = batch(2, seed=10)
    free = FluxLoRATrainStep(nat2, net2, ref_ops, **KW)
    l_free = free.step(*b[:3], noise=b[3], timesteps=b[4]).item()
    p_init = torch.nn.Parameter(torch.zeros(1))  # noqa: F841
    # (a) bound above the loss: bitwise the unguarded step
    _, _, nat3, net3 = build_pair(rank=4)
    hi = FluxLoRATrainStep(nat3, net3, ref_ops, max_loss=10.0 * l_free, **KW)
    assert hi.step(*b[:3], noise=b[3], timesteps=b[4]).item() == l_free
    assert torch.equal(net3.arena_p, net2.arena_p) and torch.equal(net3.arena_ema, net2.arena_ema)
    # (b) bound below: loss == max_loss and the clamp passes ZERO gradients back (not none): the optimizer still steps, like torch.optim.AdamW on
    # zero .grad tensors — decoupled weight decay p *= 1 - lr * wd, moments stay 0, the step count advances (a clamped step is not skipped)
    p0 = net.arena_p.clone()
    lo = FluxLoRATrainStep(nat, net, ref_ops, max_loss=0.5 * l_free, **KW)
    assert abs(lo.step(*b[:3], noise=b[3], timesteps=b[4]).item() - 0.5 * l_free) < 1e-7
    assert not net.arena_m.any() and not net.arena_v.any() and not net.arena_g.any()
    tw = torch.nn.Parameter(p0.clone())
    tw.grad = torch.zeros_like(tw)
    torch.optim.AdamW([tw], lr=KW["lr"], eps=KW.get("eps", 1e-6), weight_decay=KW.get("weight_decay", 0.01)).step()
    assert torch.equal(net.arena_p, tw.detach())
    assert lo.guard_counters() == {"nonfinite_losses": 0, "clamped_losses": 1, "steps_applied": 1, "steps_skipped": 0, "last_step_skipped": False}


def test_gated_micro_batch_of_an_accumulation_list_contributes_nothing_the_others_train():
    """gradient_accumulation (SDTrainer.py:2243-2293): one of two micro-batches above max_loss -> the step is the other micro-batch's, and the
    reported loss is max_loss + its loss"""
    b1, b2 = batch(2, seed=10), batch(2, seed=11)
    _, _, nat, net = build_pair(rank=4)
    _, _, nat2, net2 = build_pair(rank=4)
    only2 = FluxLoRATrainStep(nat2, net2, ref_ops, **KW)
    l2 = only2.step(*b2[:3], noise=b2[3], timesteps=b2[4]).item()
    ml = 3.0 * l2  # the second micro-batch stays under the bound, the weighted first one far above it
    acc = FluxLoRATrainStep(nat, net, ref_ops, max_loss=ml, **KW)
    big = torch.tensor([1e4, 1e4])  # per-sample loss weights push the first micro-batch over the bound
    total = acc.step_list([dict(latents=b1[0], prompt_embeds=b1[1], pooled_embeds=b1[2], noise=b1[3], timesteps=b1[4], loss_weight=big),
                           dict(latents=b2[0], prompt_embeds=b2[1], pooled_embeds=b2[2], noise=b2[3], timesteps=b2[4])]).item()
    assert abs(total - (ml + l2)) <= 1e-5 * (ml + l2)
    assert torch.equal(net.arena_p, net2.arena_p) and torch.equal(net.arena_m, net2.arena_m)
    assert acc.guard_counters()["steps_applied"] == 1 and acc.guard_counters()["clamped_losses"] == 1


def test_guard_off_is_the_previous_arithmetic_and_guard_on_does_not_change_a_healthy_run():
    b = [batch(2, seed=10 + k) for k in range(3)]
    outs = []
    for guard in (True, False):
        _, _, nat, net = build_pair(rank=4)
        st = FluxLoRATrainStep(nat, net, ref_ops, nonfinite_guard=guard, **KW)
        assert (st.guard is not None) == guard
        for x in b:
            st.step(*x[:3], noise=x[3], timesteps=x[4])
        outs.append(_state(net))
    assert all(torch.equal(a, c) for a, c in zip(*outs))
