"""Learning-rate schedules of the reference's factory (toolkit/scheduler.py:6-59) in closed form.

The fused AdamW kernel takes `lr` as a launch argument, so there is no torch optimizer object to hang a
torch.optim.lr_scheduler on; this module reproduces the value those schedulers hand to the optimizer at every step:

  constant ................ ConstantLR(factor=1.0 default, total_iters)        toolkit/scheduler.py:28-32
  linear .................. LinearLR(start_factor, end_factor, total_iters)   34-38
  cosine .................. CosineAnnealingLR(T_max=total_iters, eta_min)     10-15
  cosine_with_restarts .... CosineAnnealingWarmRestarts(T_0=total_iters, T_mult, eta_min)   16-21
  step .................... StepLR(step_size, gamma)                          22-26
  constant_with_warmup .... diffusers get_constant_schedule_with_warmup(num_warmup_steps)   39-45

The trainer builds the schedule with total_iters = steps (jobs/process/BaseSDTrainProcess.py:2231-2242) and calls
`.step()` once per train-loop iteration, also on accumulation iterations (SDTrainer.py:2298-2300).
"""
import math


class LRSchedule:
    def __init__(self, name, base_lr, **kwargs):
        self.name = name or "constant"
        self.base_lr = float(base_lr)
        self.kw = dict(kwargs)
        self.last_epoch = 0
        if self.name == "constant_with_warmup" and "num_warmup_steps" not in self.kw:
            self.kw["num_warmup_steps"] = 1000  # the reference's default (with a printed warning)
        if self.name not in ("constant", "linear", "cosine", "cosine_with_restarts", "step", "constant_with_warmup"):
            raise ValueError("Scheduler must be cosine, cosine_with_restarts, step, linear or constant")

    def lr_at(self, t):
        k, lr = self.kw, self.base_lr
        if self.name == "constant":
            return lr * (k.get("factor", 1.0) if t < k.get("total_iters", 5) else 1.0)
        if self.name == "linear":
            s, e, n = k.get("start_factor", 1.0 / 3), k.get("end_factor", 1.0), k.get("total_iters", 5)
            return lr * (s + (e - s) * min(t, n) / n)
        if self.name == "cosine":
            T, lo = k.get("T_max", k.get("total_iters")), k.get("eta_min", 0.0)
            return lo + (lr - lo) * (1 + math.cos(math.pi * t / T)) / 2
        if self.name == "cosine_with_restarts":
            T0, mult, lo = k.get("T_0", k.get("total_iters")), k.get("T_mult", 1), k.get("eta_min", 0.0)
            if mult == 1:
                Ti, tc = T0, t % T0
            else:
                n = int(math.log(t / T0 * (mult - 1) + 1, mult))
                tc = t - T0 * (mult ** n - 1) / (mult - 1)
                Ti = T0 * mult ** n
            return lo + (lr - lo) * (1 + math.cos(math.pi * tc / Ti)) / 2
        if self.name == "step":
            return lr * k.get("gamma", 0.1) ** (t // k["step_size"])
        w = k["num_warmup_steps"]  # constant_with_warmup
        return lr * (float(t) / float(max(1.0, w)) if t < w else 1.0)

    def get_last_lr(self):
        return [self.lr_at(self.last_epoch)]

    def step(self, epoch=None):
        self.last_epoch = self.last_epoch + 1 if epoch is None else int(epoch)
        return self.lr_at(self.last_epoch)

    def state_dict(self):
        return {"last_epoch": self.last_epoch, "base_lr": self.base_lr}

    def load_state_dict(self, sd):
        self.last_epoch = int(sd["last_epoch"])
