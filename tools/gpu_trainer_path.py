"""The step the REFERENCE'S TRAINER runs over the plug-in, on the HIP kernels at full FLUX.1-dev size, timed beside the fused train step.

    python tools/gpu_trainer_path.py [--batches 1,7] [--steps 5] [--out gpurun_out/trainer_path.json]

Per batch size, same process, same box:
  trainer_path        tools/trainer_harness.TrainerLoop.hook_train_loop — SDTrainer.hook_train_loop's sequence over an ADOPTED network
                      (get_noise_prediction through the autograd bridge, torch MSE, loss.backward(), clip_grad_norm_, torch.optim.AdamW(eps=1e-6)
                      .step(), zero_grad(set_to_none=True), ema.update(), loss.item()), with the optimizer / EMA served by the arena kernels
                      (ai_toolkit_amd/adopt.py, the default);
  trainer_path_torch  the same loop with AITK_FUSE_TRAINER_STEP=0: torch's foreach AdamW over 988 views + toolkit/ema.py's Python loop;
  fused_step          FluxLoRATrainStep.step (the headline's step: one launch sequence, clip + AdamW + EMA in one kernel, no host sync).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def timed(fn, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps, out


def run_trainer_path(dev, batches, steps, warm=2, rank=16, log=print):
    import bench
    from ai_toolkit_amd import adopt
    from ai_toolkit_amd.plugin import Flux1MI355Model
    from tools.trainer_harness import TrainerLoop

    model, _, ops = bench.build_flux(dev, rank=rank, attach=False)
    sd = Flux1MI355Model(str(dev), model=model, dtype=torch.bfloat16)
    loop = TrainerLoop(sd, rank=rank, lr=1e-4, weight_decay=0.01, max_grad_norm=1.0, ema_decay=0.99, device=dev)
    with torch.no_grad():  # "warm" adapter like the headline's
        for m in loop.network.unet_loras:
            m.lora_up.weight.normal_(0, 1e-3)
    res = {}
    for B in batches:
        lat, emb, pooled = bench.make_batch(dev, B, seed=42)
        one = lambda: loop.hook_train_loop(lat, emb, pooled)  # noqa: E731
        for mode, env in (("trainer_path", "1"), ("trainer_path_torch", "0")):
            os.environ["AITK_FUSE_TRAINER_STEP"] = env
            s0 = dict(adopt.STATS)
            for _ in range(warm):
                one()
            ms, loss = timed(one, steps)
            d = {k: adopt.STATS[k] - s0[k] for k in s0}
            res.setdefault(str(B), {})[mode] = {"ms_per_step": ms, "images_per_s": B * 1e3 / ms, "loss": loss, "calls": d}
            log(f"B={B} {mode}: {ms:.1f} ms/step  {B * 1e3 / ms:.3f} img/s  loss {loss:.5f}  {d}")
        del lat, emb, pooled
    os.environ["AITK_FUSE_TRAINER_STEP"] = "1"
    res["adapters"] = len(loop.network.unet_loras)
    res["parameters"] = len(loop.params)
    res["peak_mem_GiB"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)
    del loop, sd, model
    gc.collect()
    torch.cuda.empty_cache()
    return res


def run_fused(dev, batches, steps, warm=2, rank=16, log=print):
    import bench
    from ai_toolkit_amd.trainer import FluxLoRATrainStep

    model, net, ops = bench.build_flux(dev, rank=rank)
    step = FluxLoRATrainStep(model, net, ops, lr=1e-4, weight_decay=0.01, max_grad_norm=1.0, ema_decay=0.99, timestep_type="linear", seed=1000)
    res = {}
    for B in batches:
        lat, emb, pooled = bench.make_batch(dev, B, seed=42)
        one = lambda: step.step(lat, emb, pooled)  # noqa: E731
        for _ in range(warm):
            one()
        ms, loss = timed(one, steps)
        res[str(B)] = {"ms_per_step": ms, "images_per_s": B * 1e3 / ms, "loss": float(loss.item())}
        log(f"B={B} fused_step: {ms:.1f} ms/step  {B * 1e3 / ms:.3f} img/s")
        del lat, emb, pooled
    del step, model, net
    gc.collect()
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,7")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default="both", choices=["both", "trainer", "fused"])
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    batches = [int(x) for x in a.batches.split(",")]
    out = {"device": torch.cuda.get_device_name(0), "steps": a.steps}
    if a.only in ("both", "trainer"):
        out["trainer"] = run_trainer_path(dev, batches, a.steps)
    if a.only in ("both", "fused"):
        out["fused_step"] = run_fused(dev, batches, a.steps)
    if "trainer" in out and "fused_step" in out:
        for B in batches:
            f = out["fused_step"][str(B)]["ms_per_step"]
            out["trainer"][str(B)]["gap_vs_fused_step"] = {k: v["ms_per_step"] / f - 1.0 for k, v in out["trainer"][str(B)].items() if isinstance(v, dict) and "ms_per_step" in v}
    print(json.dumps(out, indent=1))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
