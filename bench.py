"""Headline benchmark: train images/sec, FLUX.1-dev LoRA r16 @1024^2 (BASELINE.json), N MI355X data-parallel.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one full LoRA training step (timestep/noise mix, DiT forward with the adapter fused into every wrapped Linear,
flow-matching MSE, backward, gradient all-reduce, clip + AdamW + EMA, shadow refresh) on synthetic data of the real
shape with random-init weights of the real architecture (19 double + 38 single blocks, d=3072, 24 heads, 494 adapters).
Inputs are resident in HBM before the timed region.  Weak scaling: per-GPU batch fixed.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

FLOP_PER_IMAGE = 163.6e12  # fwd 74.4 + bwd 89.2 TFLOP, no recompute, LoRA/embedders excluded
PEAK_BF16 = 2500.0  # TFLOP/s dense (MI355X_MICROARCH.md)


def build_flux(dev, rank=16, num_layers=19, num_single=38, ema=True, fp8_base=False, network_type="lora"):
    import ai_toolkit_amd  # noqa: F401
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.flux import FluxTransformer2DModel
    from ai_toolkit_amd.lora import FusedLoRANetwork

    model = FluxTransformer2DModel(num_layers=num_layers, num_single_layers=num_single, dtype=torch.bfloat16, device=dev, ops=ops)
    g = torch.Generator(device=dev).manual_seed(1234)
    with torch.no_grad():
        for mod in model.modules():
            if mod.__class__.__name__ == "Linear":
                mod.weight.copy_((torch.randn(mod.weight.shape, device=dev, generator=g) * 0.02).to(torch.bfloat16))
    torch.manual_seed(1234)
    if network_type == "lokr":  # full Kronecker factors (the reference's lokr_full_rank default), not the headline metric
        net = FusedLoRANetwork(model, lora_dim=9999999999, alpha=9999999999, network_type="lokr")
    else:
        net = FusedLoRANetwork(model, lora_dim=rank, network_type=network_type)
    with torch.no_grad():  # "warm" adapter so dA != 0 from step 0
        for m in net.unet_loras:
            m.lora_up.weight.normal_(0, 1e-3)  # LoKr: lokr_w1 takes the `up` slot and is already non-zero; w2 below
            if network_type == "lokr":
                m.lokr_w2.normal_(0, 1e-3)
    net.apply_to()
    net.build_arena(dev, ema=ema, groups=model.lora_groups())
    net.refresh_shadows(ops)
    model.attach_network(net)
    if fp8_base:  # BASELINE config 5: e4m3 weight-only base (per-output-channel scale) + bf16/fp32 adapter
        model.quantize_base_fp8(release_bf16=True)
    model.prepare()
    return model, net, ops


def gemm_roofline(step_fn, ops_mod):
    """One extra instrumented step: HIP events (torch's current stream = the stream every kernel is launched on) around
    each launch of the dominant kernel (gemm_nt); algorithmic FLOPs = 2 M N (K + K2) per launch."""
    recs = []
    orig = ops_mod.gemm_nt

    def timed(a, b, out, **kw):
        M = kw.get("M") or a.shape[0]
        N, K = b.shape
        K2 = kw["a2"].shape[1] if kw.get("a2") is not None else 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(a, b, out, **kw)
        e1.record()
        recs.append((e0, e1, 2.0 * M * N * (K + K2), (M, N, K, K2, kw.get("flags", 0))))
        return r

    ops_mod.gemm_nt = timed
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        ops_mod.gemm_nt = orig
    ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in recs)
    fl = sum(f for _, _, f, _ in recs)
    by_shape = {}
    for e0, e1, f, key in recs:
        d = by_shape.setdefault(key, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += e0.elapsed_time(e1)
        d[2] += f
    census = sorted(({"M": k[0], "N": k[1], "K": k[2], "K2": k[3], "flags": k[4], "calls": v[0], "ms": round(v[1], 2),
                      "tflops": round(v[2] / v[1] / 1e9, 1)} for k, v in by_shape.items()), key=lambda r: -r["ms"])
    return {"launches": len(recs), "gemm_ms_per_step": ms, "gemm_flop_per_step": fl, "avg_launch_us": 1e3 * ms / max(1, len(recs)),
            "tflops": fl / ms / 1e9 if ms > 0 else 0.0, "census": census}


def cpu_baseline():
    """Oracle ('port') timed on this host's cores on a bounded sample: 1 double + 1 single FLUX.1-dev block at full
    width / full sequence (B=1, fp32), fwd + bwd + AdamW; extrapolated linearly to 19 + 38 blocks."""
    from oracle import flux_ref, lora_ref, train_ref

    torch.manual_seed(0)
    times = {}
    for kind, (nl, ns) in (("double", (1, 0)), ("single", (0, 1))):
        m = flux_ref.FluxTransformer2DModel(num_layers=nl, num_single_layers=ns)
        flux_ref.init_synthetic_(m, std=0.02)
        net = lora_ref.RefLoRANetwork(m, 16)
        net.apply_to()
        st = train_ref.RefTrainStep(m, net, lr=1e-4)
        g = torch.Generator().manual_seed(1)
        lat = torch.randn(1, 16, 128, 128, generator=g)
        emb = torch.randn(1, 512, 4096, generator=g) * 0.1
        pooled = torch.randn(1, 768, generator=g) * 0.1
        noise = torch.randn(1, 16, 128, 128, generator=g)
        ts = torch.tensor([500.0])
        t0 = time.time()
        st.step(lat, emb, pooled, noise, ts)
        times[kind] = time.time() - t0
        del m, net, st
    full = 19 * times["double"] + 38 * times["single"]
    return {"value": 1.0 / full, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 double ({times['double']:.1f}s) + 1 single ({times['single']:.1f}s) FLUX.1-dev block, 1024^2, B=1, fp32, "
                      "fwd+bwd+AdamW incl. embedders/head each; extrapolated x19/x38"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("AITK_BENCH_BATCH", "0")),
                    help="per-GPU batch; 0 = 7 when the GPU has >= 252 GiB free (244 GiB peak; 7 x 4608 rows = 126 row tiles make the "
                         "N=3072 GEMMs 5.9 tile rounds on 256 CUs instead of 3.4 at batch 4), else 4 (157 GiB)")
    ap.add_argument("--rank", type=int, default=16)
    ap.add_argument("--fp8-base", action="store_true", help="BASELINE config 5 variant (not the headline metric): fp8 e4m3 base weights")
    ap.add_argument("--network", default="lora", choices=["lora", "dora", "lokr"], help="adapter type (headline metric: lora)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    if world > 1 or os.environ.get("AITK_BENCH_FORCE_PG"):  # FORCE_PG: 1-rank RCCL group, exercises the collective path on one GPU
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        pg = dist.group.WORLD

    from ai_toolkit_amd.trainer import FluxLoRATrainStep

    model, net, ops = build_flux(dev, rank=args.rank, fp8_base=args.fp8_base, network_type=args.network)
    step = FluxLoRATrainStep(model, net, ops, lr=1e-4, weight_decay=0.01, max_grad_norm=1.0, ema_decay=0.99,
                             timestep_type="linear", process_group=pg, seed=1000 + rank)
    B = args.batch
    if B <= 0:
        avail_gib = (torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev)) / 2 ** 30  # free + what this process holds
        B = 7 if (avail_gib >= 252 and not args.fp8_base and args.network == "lora") else 4
    gen = torch.Generator(device=dev).manual_seed(42 + rank)
    lat = torch.randn(B, 16, 128, 128, device=dev, generator=gen).to(torch.bfloat16)
    emb = (torch.randn(B, 512, 4096, device=dev, generator=gen) * 0.1).to(torch.bfloat16)
    pooled = (torch.randn(B, 768, device=dev, generator=gen) * 0.1).to(torch.bfloat16)

    def one():
        return step.step(lat, emb, pooled)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    final_loss = float(loss.item())
    ips = world * B * args.steps / dt
    out = {
        "metric": f"train images/sec, FLUX.1-dev LoRA r{args.rank} @1024^2" + (" (fp8 e4m3 weight-only base)" if args.fp8_base else "")
                  + (f" [adapter: {args.network}]" if args.network != "lora" else ""),
        "value": ips, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16 (fp8 e4m3 weight-only base, expanded per layer to bf16 before its GEMM)" if args.fp8_base else "bf16", "data": "synthetic (random-init FLUX.1-dev architecture, N(0,1) latents, 0.1*N(0,1) text embeds)",
        "config": {"workload": "FLUX.1-dev DiT LoRA r16, 1024x1024 (4096 img + 512 txt tokens), bf16, AdamW+EMA, clip 1.0",
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}", "adapters": len(net.unet_loras),
                   "lora_params": net.arena_p.numel(), "grad_checkpointing": False},
        "final_loss": final_loss,
        "step_mfma_frac": FLOP_PER_IMAGE * (ips / world) / (PEAK_BF16 * 1e12),
    }
    rf = None
    if not args.no_roofline:
        rf = gemm_roofline(one, ops)  # every rank runs the instrumented step (it contains the gradient all-reduce)
    peak_mem = torch.cuda.max_memory_allocated() / 2 ** 30
    out["config"]["peak_mem_GiB"] = round(peak_mem, 1)
    if rank == 0:
        if rf is not None:
            if os.environ.get("AITK_GEMM_CENSUS"):  # per-shape breakdown of the instrumented step (not part of the JSON line)
                with open(os.environ["AITK_GEMM_CENSUS"], "w") as fh:
                    json.dump(rf["census"], fh, indent=0)
            out["roofline"] = {"bound": "mfma", "kernel": "aitk_gemm_nt: gemm_nt_8phase_kernel (big problems) + gemm_nt_kernel<1,128,128> "
                                                         "(LoRA-fused bf16 GEMM, all launches of one step)",
                               "achieved": rf["tflops"], "peak": PEAK_BF16, "unit": "TFLOP/s", "frac": rf["tflops"] / PEAK_BF16,
                               "traffic": None, "launches_per_step": rf["launches"], "avg_launch_us": rf["avg_launch_us"],
                               "gemm_ms_per_step": rf["gemm_ms_per_step"]}
            # memory-side bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction +
            # WRITE_SIZE, profiles/r01_pmc_v2/summary.json) for the dominant shape 18432x3072x3072 (+r16 slab)
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_v2", "summary.json")
            if os.path.exists(pmc):
                by_m = json.load(open(pmc)).get("gemm_nt_8phase_kernel_by_M", {})
                g8 = by_m.get(str(B * 4608))  # the most frequent launch of a step: (B * 4608) x 3072 x 3072 (+r16 slab)
                if g8 is not None:
                    out["roofline"]["traffic"] = g8["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_note"] = (f"PMC, launch {B * 4608}x3072x3072+r16: memory-side bytes incl. Infinity-Cache hits; "
                                                       f"algorithmic {g8['algorithmic_bytes']} B")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
