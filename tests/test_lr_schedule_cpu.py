"""Closed-form LR schedules vs the torch.optim.lr_scheduler classes the reference's factory instantiates
(toolkit/scheduler.py:6-59), and the schedule driving the fused step's lr argument."""
import pytest
import torch

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd.lr_schedule import LRSchedule

S = torch.optim.lr_scheduler
CASES = [
    ("constant", dict(total_iters=40), lambda o: S.ConstantLR(o, factor=1.0, total_iters=40)),
    ("constant", dict(factor=0.25, total_iters=7), lambda o: S.ConstantLR(o, factor=0.25, total_iters=7)),
    ("linear", dict(start_factor=0.1, end_factor=1.0, total_iters=15), lambda o: S.LinearLR(o, start_factor=0.1, end_factor=1.0, total_iters=15)),
    ("cosine", dict(total_iters=40, eta_min=1e-6), lambda o: S.CosineAnnealingLR(o, T_max=40, eta_min=1e-6)),
    ("cosine_with_restarts", dict(total_iters=10), lambda o: S.CosineAnnealingWarmRestarts(o, T_0=10)),
    ("cosine_with_restarts", dict(total_iters=6, T_mult=2, eta_min=1e-5), lambda o: S.CosineAnnealingWarmRestarts(o, T_0=6, T_mult=2, eta_min=1e-5)),
    ("step", dict(step_size=9, gamma=0.5), lambda o: S.StepLR(o, step_size=9, gamma=0.5)),
    ("constant_with_warmup", dict(num_warmup_steps=12),
     lambda o: S.LambdaLR(o, lambda t: float(t) / float(max(1.0, 12)) if t < 12 else 1.0)),  # diffusers.optimization form
]


@pytest.mark.parametrize("name,kw,make", CASES)
def test_schedule_equals_torch_scheduler(name, kw, make):
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=3e-4)
    ref = make(opt)
    ours = LRSchedule(name, 3e-4, **kw)
    assert ours.get_last_lr()[0] == pytest.approx(ref.get_last_lr()[0], rel=1e-9)
    for t in range(45):
        opt.step()
        ref.step()
        assert ours.step() == pytest.approx(ref.get_last_lr()[0], rel=1e-6, abs=1e-12), (name, t)


def test_unknown_name_raises_like_reference():
    with pytest.raises(ValueError):
        LRSchedule("polynomial_decay_with_magic", 1e-4)
