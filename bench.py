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
PEAK_FP8 = 5000.0  # TFLOP/s dense fp8 MFMA (MX-scaled K = 64 / 128 forms; measured ceiling 4.65 PF, MI355X_MICROARCH.md)


def build_flux(dev, rank=16, num_layers=19, num_single=38, ema=True, fp8_base=False, network_type="lora", fp8_mfma=False, attach=True):
    """attach=False: the bare native model (frozen base, prepared) with no adapter network — what a plug-in's load_model leaves for the
    reference's trainer to build its own network over (leg `trainer_path`)."""
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
    if not attach:
        for p in model.parameters():
            p.requires_grad_(False)
        if fp8_base:
            model.quantize_base_fp8(release_bf16=True, mfma=fp8_mfma)
        model.prepare()
        return model, None, ops
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
        model.quantize_base_fp8(release_bf16=True, mfma=fp8_mfma)  # mfma: W8A8 on the MX-scaled fp8 MFMA (per-token e4m3 activations)
    model.prepare()
    return model, net, ops


def build_unet(dev, kind="sdxl", rank=8, conv_rank=0):
    """BASELINE config 2 (SDXL UNet LoRA r8) / config 1 architecture (SD1.5) with synthetic weights: W ~ N(0, 1/fan_in)."""
    import math

    import ai_toolkit_amd  # noqa: F401
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.lora import FusedLoRANetwork
    from ai_toolkit_amd.unet import SD15_CONFIG, SDXL_CONFIG, UNet2DConditionModel

    model = UNet2DConditionModel(**(SDXL_CONFIG if kind == "sdxl" else SD15_CONFIG), dtype=torch.bfloat16, device=dev, ops=ops)
    g = torch.Generator(device=dev).manual_seed(1234)
    with torch.no_grad():
        for m in model.modules():
            w = getattr(m, "weight", None)
            if w is not None and w.dim() >= 2:
                w.copy_((torch.randn(w.shape, device=dev, generator=g) / math.sqrt(w[0].numel())).to(torch.bfloat16))
    torch.manual_seed(1234)
    net = FusedLoRANetwork(model, lora_dim=rank, alpha=rank, target_lin_modules=("Transformer2DModel",), is_transformer=False,
                           peft_format=False, transformer_only=False, base_model_version="sdxl" if kind == "sdxl" else "sd1",
                           conv_lora_dim=conv_rank or None, conv_alpha=conv_rank or None)  # --conv-rank: network.conv (3x3-conv adapters too)
    with torch.no_grad():
        for m in net.unet_loras:
            m.lora_up.weight.normal_(0, 1e-3)
    net.apply_to()
    net.build_arena(dev, ema=True, groups=model.lora_groups())
    net.refresh_shadows(ops)
    model.attach_network(net)
    model.prepare()
    return model, net, ops


def bench_unet(args, dev):
    """Secondary line (not the headline metric): train images/sec of the UNet LoRA step — BASELINE config 2 (SDXL r8 @1024^2) or the
    SD1.5 architecture at 512^2 — with the MFMA throughput of its GEMM + implicit-GEMM-conv launches measured by events in one extra step."""
    from ai_toolkit_amd.trainer import UNetLoRATrainStep

    kind = args.model
    model, net, ops = build_unet(dev, kind, rank=args.rank if args.rank != 16 else (8 if kind == "sdxl" else 4), conv_rank=args.conv_rank)
    step = UNetLoRATrainStep(model, net, ops, lr=1e-4, weight_decay=0.01, max_grad_norm=1.0, ema_decay=0.99, seed=1000)
    # batch sweep on one MI355X (profiles/r02_bench_unet_*.json): SDXL 28.4 / 31.7 / 31.1 img/s at B = 8 / 12 / 16 (12 x 1024 tokens = 240 tiles
    # of 256^2 at the 32x32 level: 94 % of one tile round); SD1.5 105 / 139 / 159 img/s at B = 8 / 16 / 32
    B = args.batch if args.batch > 0 else (12 if kind == "sdxl" else 32)
    side = 128 if kind == "sdxl" else 64
    gen = torch.Generator(device=dev).manual_seed(42)
    lat = torch.randn(B, 4, side, side, device=dev, generator=gen).to(torch.bfloat16)
    ctx = (torch.randn(B, 77, model.config["cross_attention_dim"], device=dev, generator=gen) * 0.5).to(torch.bfloat16)
    pooled = (torch.randn(B, 1280, device=dev, generator=gen) * 0.5).to(torch.bfloat16) if kind == "sdxl" else None

    def one():
        return step.step(lat, ctx, pooled)

    for _ in range(args.warmup):
        one()
    dt_eager, per, loss = timed_steps(one, args.steps, torch.cuda.synchronize)
    dt, mode = dt_eager, "eager launches"
    graph = None
    if not args.no_graph:
        # the same step with forward + loss + backward replayed as one hipGraph (trainer.step_graphed): the UNet step is ~5 400 small
        # launches and host-bound when launched from Python one by one
        try:
            def one_g():
                return step.step_graphed(latents=lat, prompt_embeds=ctx, pooled_embeds=pooled)

            for _ in range(max(args.warmup, 2)):
                one_g()
            dt_g, per_g, loss_g = timed_steps(one_g, args.steps, torch.cuda.synchronize)
            graph = {"images_per_s": B * args.steps / dt_g, "ms_per_step": 1e3 * dt_g / args.steps}
            if dt_g < dt:
                dt, per, loss, mode = dt_g, per_g, loss_g, "hipGraph replay (forward + loss + backward), optimizer eager"
        except torch.OutOfMemoryError as ex:  # the only tolerated failure of this leg (second memory pool); anything else is a bug: raise
            graph = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    recs = []
    ops._gemm_hook = lambda e0, e1, flops, shapes: recs.append((e0, e1, flops))
    try:
        one()
        torch.cuda.synchronize()
    finally:
        ops._gemm_hook = None
    ms = sum(a.elapsed_time(b) for a, b, _ in recs)
    fl = sum(f for _, _, f in recs)
    name = "SDXL UNet LoRA r8 @1024^2" if kind == "sdxl" else "SD1.5 UNet LoRA r4 @512^2"
    if args.conv_rank:
        name += f" + conv-LoRA r{args.conv_rank} (network.conv)"
    out = {"metric": f"train images/sec, {name}", "value": B * args.steps / dt, "unit": "images/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "bf16", "data": "synthetic (random-init architecture, N(0,1) latents, 0.5*N(0,1) text states)",
           "config": {"workload": f"{name}, eps-prediction DDPM, 77 text tokens, bf16, AdamW+EMA, clip 1.0 (BASELINE config {2 if kind == 'sdxl' else 1} "
                                  "architecture; not the headline metric)", "per_gpu_batch": B, "adapters": len(net.unet_loras),
                      "lora_params": net.arena_p.numel(), "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                      "launch_mode": mode},
           "launch_modes": {"eager": {"images_per_s": B * args.steps / dt_eager, "ms_per_step": 1e3 * dt_eager / args.steps}, "graph": graph},
           "final_loss": float(loss.item()),
           "step_ms": {"median": _pct(per, 0.5), "p10": _pct(per, 0.1), "p90": _pct(per, 0.9), "n": len(per)},
           "roofline": {"bound": "mfma", "kernel": "aitk_gemm_nt (LoRA-fused token GEMMs + implicit-GEMM 3x3 convolutions, all launches of one step)",
                        "achieved": fl / ms / 1e9 if ms > 0 else 0.0, "peak": PEAK_BF16, "unit": "TFLOP/s", "frac": (fl / ms / 1e9) / PEAK_BF16 if ms > 0 else 0.0,
                        "traffic": None, "launches_per_step": len(recs), "gemm_conv_ms_per_step": ms, "gemm_conv_tflop_per_step": fl / 1e12}}
    if kind == "sd15" and not args.no_cpu_baseline:
        # BASELINE configs[0] is the reference's own CPU-runnable case: SD1.5 UNet LoRA r4 @512^2, fp32, batch 1.  The oracle runs exactly
        # that, full size (859.5 M parameters, 192 adapters), on this host's cores: one warm-up step + one timed step.
        out["cpu_baseline"] = cpu_baseline_sd15()
    print(json.dumps(out), flush=True)


def cpu_baseline_sd15():
    from oracle import lora_ref, train_ref, unet_ref

    torch.manual_seed(0)
    m = unet_ref.UNet2DConditionModel(**unet_ref.SD15)
    unet_ref.init_synthetic_(m, seed=11)
    net = lora_ref.RefLoRANetwork(m, 4, target=("Transformer2DModel",), kohya_unet=True, alpha=4.0)
    net.apply_to()
    st = train_ref.RefUNetTrainStep(m, net, lr=1e-4)
    g = torch.Generator().manual_seed(1)
    lat, noise = torch.randn(1, 4, 64, 64, generator=g), torch.randn(1, 4, 64, 64, generator=g)
    ctx = torch.randn(1, 77, 768, generator=g) * 0.5
    ts = torch.tensor([500])
    st.step(lat, ctx, None, noise, ts)
    t0 = time.time()
    st.step(lat, ctx, None, noise, ts)
    dt = time.time() - t0
    return {"value": 1.0 / dt, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"full-size SD1.5 UNet (859.5 M parameters, 192 LoRA r4 adapters), 512^2, B=1, fp32, one complete step "
                      f"(fwd + bwd + clip + AdamW) in {dt:.1f} s after one warm-up step"}


def gemm_roofline(step_fn, ops_mod):
    """One extra instrumented step: HIP events (torch's current stream = the stream every kernel is launched on) around
    each launch of the dominant kernel (aitk_gemm_nt / aitk_gemm_nt_grouped, hooked where ops.py invokes the C entry point);
    algorithmic FLOPs = 2 M N (K + K2) per problem."""
    recs = []

    def hook(e0, e1, flops, shapes):
        recs.append((e0, e1, flops, shapes))

    ops_mod._gemm_hook = hook
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        ops_mod._gemm_hook = None
    ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in recs)
    fl = sum(f for _, _, f, _ in recs)
    by_shape = {}
    for e0, e1, f, key in recs:
        d = by_shape.setdefault(key, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += e0.elapsed_time(e1)
        d[2] += f
    def label(k):  # one problem (M, N, K, K2, flags) or a grouped pair of them
        probs = k if isinstance(k[0], tuple) else (k,)
        return {"M": "+".join(str(q[0]) for q in probs), "N": probs[0][1], "K": probs[0][2], "K2": probs[0][3], "flags": probs[0][4]}

    census = sorted((dict(label(k), calls=v[0], ms=round(v[1], 2), tflops=round(v[2] / v[1] / 1e9, 1)) for k, v in by_shape.items()),
                    key=lambda r: -r["ms"])
    return {"launches": len(recs), "gemm_ms_per_step": ms, "gemm_flop_per_step": fl, "avg_launch_us": 1e3 * ms / max(1, len(recs)),
            "tflops": fl / ms / 1e9 if ms > 0 else 0.0, "census": census}


def attention_roofline(step_fn, ops_mod, peak=PEAK_BF16):
    """One extra instrumented step: HIP events around every aitk_attn_fwd / aitk_attn_bwd call (the step's second kernel family, 35 % of the
    headline step); algorithmic FLOPs = 2 (forward) / 4 (backward) matmuls of 2 S Skv d per (batch, head).  aitk_attn_bwd's time covers its
    three kernels (delta, dK/dV, dQ)."""
    recs = []
    ops_mod._attn_hook = lambda name, e0, e1, fl, shape: recs.append((name, e0, e1, fl, shape))
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        ops_mod._attn_hook = None
    out = {}
    for which in ("aitk_attn_fwd", "aitk_attn_bwd"):
        sel = [r for r in recs if r[0] == which]
        ms = sum(r[1].elapsed_time(r[2]) for r in sel)
        fl = sum(r[3] for r in sel)
        if sel and ms > 0:
            out[which] = {"calls": len(sel), "ms_per_step": ms, "achieved": fl / ms / 1e9, "frac": fl / ms / 1e9 / peak,
                          "shape_BHSSkvD": list(sel[0][4])}
    ms = sum(r[1].elapsed_time(r[2]) for r in recs)
    fl = sum(r[3] for r in recs)
    out.update({"bound": "mfma", "kernel": "aitk_attn_fwd (attn_fwd_kernel) + aitk_attn_bwd (attn_delta_kernel, attn_bwd_dkdv_ws_kernel emitting dS, attn_bwd_dq_ds8_kernel; AITK_ATTN_DS=0: attn_bwd_dq_kernel), all calls of one step",
                "achieved": fl / ms / 1e9 if ms > 0 else 0.0, "peak": peak, "unit": "TFLOP/s", "frac": (fl / ms / 1e9 / peak) if ms > 0 else 0.0,
                "attn_ms_per_step": ms, "note": "algorithmic flops (2 fwd + 4 bwd matmuls); the default backward executes 5 (S and dP once, in the dK/dV pass, which hands its bf16 dS to the dQ product through HBM); with AITK_ATTN_DS=0 it executes 7"})
    return out


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


BUCKETS = [(1024, 1024), (832, 1216), (1216, 832), (896, 1152), (1152, 896)]  # BASELINE.md §2 bucket mix (W x H pixels)


def _pmc_summaries():
    """Committed rocprofv3 --pmc summaries under profiles/, newest round first: r<NN>_pmc_summary.json (rounds 5+, tools/pmc_round_summary.py)
    and r<NN>_pmc*/summary.json (rounds 1-4)."""
    import glob
    import re

    found = glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc*summary*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc*", "summary.json"))

    def key(path):
        rel = os.path.relpath(path, os.path.join(ROOT, "profiles"))
        m = re.match(r"r(\d\d)", rel)
        return (int(m.group(1)) if m else 0, os.path.getmtime(path), rel)

    return sorted(set(found), key=key, reverse=True)


def _pct(xs, q):
    xs = sorted(xs)
    if not xs:
        return None
    k = (len(xs) - 1) * q
    lo, hi = int(k), min(int(k) + 1, len(xs) - 1)
    return xs[lo] + (xs[hi] - xs[lo]) * (k - lo)


def make_batch(dev, B, width=1024, height=1024, seed=42):
    gen = torch.Generator(device=dev).manual_seed(seed)
    lat = torch.randn(B, 16, height // 8, width // 8, device=dev, generator=gen).to(torch.bfloat16)
    emb = (torch.randn(B, 512, 4096, device=dev, generator=gen) * 0.1).to(torch.bfloat16)
    pooled = (torch.randn(B, 768, device=dev, generator=gen) * 0.1).to(torch.bfloat16)
    return lat, emb, pooled


def timed_steps(fn, n, sync):
    """n calls of fn bracketed by `sync` (barrier + device sync) on both sides; per-step GPU durations from events recorded at
    the step boundaries on the launch stream (no host sync inside the region)."""
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    sync()
    t0 = time.perf_counter()
    evs[0].record()
    out = None
    for i in range(n):
        out = fn()
        evs[i + 1].record()
    sync()
    dt = time.perf_counter() - t0
    per = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
    return dt, per, out


def gpu_comparator(dev, rank, steps=3):
    """The reference-equivalent PyTorch-ROCm EAGER step on this GPU, in this run: the oracle's plain-PyTorch FLUX.1-dev (bf16,
    F.scaled_dot_product_attention) + the oracle restatement of the reference LoRA modules (fp32 adapter on an fp32 activation copy,
    toolkit/network_mixins.py:304-342) + autograd + clip_grad_norm_ + torch.optim.AdamW; B = 1, no gradient checkpointing (the
    reference's default recomputes every block and would be slower still)."""
    from oracle import flux_ref, lora_ref

    torch.manual_seed(0)
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(dev):
            model = flux_ref.FluxTransformer2DModel()
    finally:
        torch.set_default_dtype(torch.float32)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.Linear):
                m.weight.normal_(0, 0.02)
                if m.bias is not None:
                    m.bias.zero_()
    for p in model.parameters():
        p.requires_grad_(False)
    net = lora_ref.RefLoRANetwork(model, rank).to(dev)
    net.torch_multiplier = net.torch_multiplier.to(dev)
    with torch.no_grad():
        for m in net.unet_loras:
            m.lora_up.weight.normal_(0, 1e-3)
    net.apply_to()
    params = [p for m in net.unet_loras for p in (m.lora_down.weight, m.lora_up.weight)]
    opt = torch.optim.AdamW(params, lr=1e-4, eps=1e-6, weight_decay=0.01)
    lat, emb, pooled = make_batch(dev, 1)
    img_ids, txt_ids = flux_ref.make_ids(128, 128, 512, dev)
    guid = torch.ones(1, device=dev)

    def step():
        noise = torch.randn_like(lat)
        t = torch.rand(1, device=dev)
        noisy = ((1 - t.view(1, 1, 1, 1)) * lat.float() + t.view(1, 1, 1, 1) * noise.float()).to(torch.bfloat16)
        opt.zero_grad(set_to_none=True)
        with net:
            pred = flux_ref.unpack_latents(model(flux_ref.pack_latents(noisy), emb, pooled, t, img_ids, txt_ids, guid), 128, 128)
            loss = torch.nn.functional.mse_loss(pred.float(), (noise.float() - lat.float()))
            loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        return loss

    step()
    dt, per, _ = timed_steps(step, steps, torch.cuda.synchronize)
    return {"value": steps / dt, "unit": "images/s", "kind": "PyTorch-ROCm eager, same GPU (oracle modules: bf16 base, fp32 LoRA, autograd, "
            "clip_grad_norm_, torch AdamW; no gradient checkpointing)", "per_gpu_batch": 1, "steps": steps, "ms_per_step": 1e3 * dt / steps}


def _run_leg(out, failed, name, fn):
    """An optional leg of the JSON line: whatever it raises is recorded under its own key and in `failed_legs` — the headline `value`
    measured before it must survive a side leg's failure (hipGraph capture error, a missing test module, out of memory, ...)."""
    try:
        out[name] = fn()
    except Exception as ex:  # noqa: BLE001 - recorded, never silent: the line says which leg failed and why
        out[name] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        failed.append(name)
        if isinstance(ex, torch.OutOfMemoryError):
            torch.cuda.empty_cache()


def _persist_headline(out):
    """The measured headline object on disk before any optional leg runs (AITK_BENCH_HEADLINE_FILE, default under the system temp
    directory): a crash the per-leg handler cannot catch (a fault inside a kernel, a killed process) still leaves the number."""
    import tempfile

    path = os.environ.get("AITK_BENCH_HEADLINE_FILE") or os.path.join(tempfile.gettempdir(), "aitk_bench_headline.json")
    try:
        with open(path, "w") as fh:
            json.dump(out, fh)
    except OSError:
        pass


def _smi_index(dev_index):
    """rocm-smi's index of torch device `dev_index`: the two orders differ under HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES remapping, so the
    device is looked up by its PCI bus address (rocm-smi --showbus).  Returns (index, how)."""
    import re
    import shutil
    import subprocess

    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}".lower()
        txt = subprocess.run([smi, "--showbus"], capture_output=True, text=True, timeout=5).stdout
        for m in re.finditer(r"GPU\[(\d+)\]\s*:\s*PCI Bus:\s*([0-9A-Fa-f:.]+)", txt):
            if m.group(2).lower().startswith(want):
                return int(m.group(1)), f"pci {want}"
    except Exception:  # noqa: BLE001 - telemetry is best effort
        pass
    return dev_index, "torch device index (PCI lookup unavailable)"


def allreduce_standalone(step, net, pg, world, dev, iters=5):
    """The two gradient all-reduce pieces of a step on an otherwise idle GPU (after the timed region): per-piece duration from events on the
    launch stream around a blocking collective, and the ring bus bandwidth 2 (P - 1) / P x bytes / time it corresponds to — what
    `allreduce_ms_exposed` is to be read against."""
    import torch.distributed as dist

    g = net.arena_g
    n_mat = getattr(net, "n_mat", g.numel())
    ranges = [(step._split, n_mat), (0, step._split)] + ([(n_mat, g.numel())] if g.numel() > n_mat else [])
    bf = step.allreduce_dtype == "bf16"
    bufs = [torch.zeros(b - a, dtype=torch.bfloat16 if bf else torch.float32, device=dev) for a, b in ranges]
    res = []
    for (a, b), buf in zip(ranges, bufs):
        dist.all_reduce(buf, group=pg)  # warm-up (communicator / channel set-up for this size)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            dist.all_reduce(buf, group=pg)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        nbytes = buf.numel() * buf.element_size()
        res.append({"arena_range": [a, b], "bytes": nbytes, "ms": ms,
                    "bus_GBps": (2 * (world - 1) / world) * nbytes / (ms * 1e6) if world > 1 and ms > 0 else None})
    return {"pieces": res, "total_ms": sum(r["ms"] for r in res), "note": "blocking all-reduce of each piece on an idle GPU, order of issue "
            "(single-stream adapters first); bus_GBps = ring traffic per GPU / time"}


def xgmi_topology():
    """Link type and hop count between the GPUs of this node as rocm-smi reports them (`--showtopotype`, `--showtopohops`): xGMI is
    point-to-point, a ring all-reduce is bound by the slowest link it crosses (7 links x ~153 GB/s per GPU on an 8-GPU MI355X board)."""
    import re
    import shutil
    import subprocess

    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    out = {}
    try:
        t = subprocess.run([smi, "--showtopotype"], capture_output=True, text=True, timeout=10).stdout
        kinds = re.findall(r"\b(XGMI|PCIE)\b", t)
        out["link_types"] = {k: kinds.count(k) for k in sorted(set(kinds))}
        h = subprocess.run([smi, "--showtopohops"], capture_output=True, text=True, timeout=10).stdout
        hops = [int(x) for line in h.splitlines() if line.startswith("GPU") for x in re.findall(r"\s(\d+)(?=\s|$)", line)]
        out["max_hops"] = max(hops) if hops else None
    except Exception as ex:  # noqa: BLE001 - telemetry is best effort
        out["error"] = f"{type(ex).__name__}: {ex}"[:120]
    return out


def _flush_c_stdio():
    import ctypes

    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


def self_launch(n):
    """`python bench.py --gpus N` from a cold shell: one child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment, the same contract torch.distributed.run provides), rendezvous on 127.0.0.1.  Only rank 0 prints the JSON line; the
    children's stderr passes through.  Returns the exit code (first failing rank's, 0 when all succeed)."""
    import socket
    import subprocess

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if os.environ.get("AITK_BENCH_ONE_DEVICE") and have >= 1:
        have = n  # control-flow rehearsal of the N-rank run on one GPU (every rank on device 0, gloo transport): not a measurement
    if have < n:
        sys.stderr.write(f"bench.py --gpus {n}: {have} GPU(s) visible on this host, {n} needed\n")
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        live = list(procs)
        while live:  # poll every child: the rank that dies is rarely the one a sequential wait() would be sitting on
            time.sleep(0.2)
            for pr in list(live):
                code = pr.poll()
                if code is None:
                    continue
                live.remove(pr)
                if code != 0 and rc == 0:
                    rc = code
                    for other in live:  # a dead rank leaves the others blocked in a collective: stop exactly the children we started
                        other.terminate()
    except KeyboardInterrupt:
        for pr in procs:
            if pr.poll() is None:
                pr.terminate()
        rc = 130
    return rc


def parity_leg(dev):
    """`parity` block of the JSON line (N = 1): one step of a small FLUX (2 double + 3 single blocks, 3 heads of 128, 96 image + 40 text
    tokens, B = 2, LoRA r16) on the HIP path against the oracle on identical weights / inputs, measured in this run — the oracle is the
    CHECKER here (fp32 = truth, ref16 = the reference's arithmetic: bf16 modules + fp32 adapter), never the thing timed.  The
    full-size numbers (19+38 blocks @1024^2, SDXL, SD1.5, Wan config 4, fp8 r32) come from `pytest -m gpu` (tests/test_gpu_parity_r*.py)."""
    import math

    from ai_toolkit_amd import ops
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from oracle import train_ref
    from oracle.pairs import batch as _batch, build as _build

    def rel(a, b):
        num = sum(((x.float() - y.float()) ** 2).sum().item() for x, y in zip(a, b))
        return math.sqrt(num / max(sum((y.float() ** 2).sum().item() for y in b), 1e-300))

    ref, ref_net, nat, net = _build(rank=16, dev=str(dev))
    lat, emb, pooled, noise, ts = _batch(2, dev=str(dev))
    kw = dict(lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    oracle = train_ref.RefTrainStep(ref, ref_net, **kw)
    l32 = oracle.step(lat.float(), emb.float(), pooled.float(), noise.float(), ts).item()
    g32 = [p.grad.clone() for p in oracle.params]
    ref.to(torch.bfloat16)
    l16 = oracle.step(lat, emb, pooled, noise, ts, dtype=torch.bfloat16).item()
    g16 = [p.grad.clone() for p in oracle.params]
    lo = FluxLoRATrainStep(nat, net, ops, **kw).step(lat, emb, pooled, noise=noise, timesteps=ts).item()
    go = []
    for m in net.unet_loras:
        go += [m.lora_down.weight.grad, m.lora_up.weight.grad]
    # ref16_self: the SAME reference arithmetic on a second backend — the oracle's bf16 step on the host CPU's kernels vs
    # the run above on the GPU (rocBLAS), identical weights / adapter state / inputs.  If two executions of the reference's own code differ by
    # as much as the HIP path differs from either, north_star's 1e-3 on LoRA deltas is not attainable by the reference itself.
    from oracle.pairs import cpu_twin

    ref.float()
    twin, twin_net = cpu_twin(ref, ref_net, 16)
    cpu = [t.cpu() for t in (lat, emb, pooled, noise, ts)]
    o_cpu = train_ref.RefTrainStep(twin, twin_net, **kw)
    l32c = o_cpu.step(cpu[0].float(), cpu[1].float(), cpu[2].float(), cpu[3].float(), cpu[4]).item()
    g32c = [p.grad.clone().to(dev) for p in o_cpu.params]
    twin.to(torch.bfloat16)
    l16c = o_cpu.step(*cpu[:4], cpu[4], dtype=torch.bfloat16).item()
    g16c = [p.grad.clone().to(dev) for p in o_cpu.params]
    return {"config": "FLUX 2+3 blocks, 3x128 heads, 96 img + 40 txt tokens, B=2, LoRA r16 (same seeds / weights / inputs on every path)",
            "tolerance": "north_star: 1e-3 relative on the bf16 loss and on LoRA deltas",
            "loss_rel": abs(lo - l32) / abs(l32), "grad_rel": rel(go, g32), "ref16_floor": rel(g16, g32),
            "ref16_loss_rel": abs(l16 - l32) / abs(l32), "grad_rel_vs_ref16": rel(go, g16),
            "ref16_self": {"grad_rel_gpu_vs_cpu": rel(g16, g16c), "loss_rel_gpu_vs_cpu": abs(l16 - l16c) / abs(l16c),
                           "fp32_grad_rel_gpu_vs_cpu": rel(g32, g32c), "grad_rel_ours_vs_ref16_cpu": rel(go, g16c),
                           "what": "the oracle's bf16 step (reference arithmetic: bf16 modules + fp32 adapter) on the GPU vs the same code on "
                                   "the host CPU; fp32_* = the same pair in fp32 (control: summation order only)"},
            "note": "grad_rel = adapter-gradient error of the HIP path vs the fp32 oracle (relative Frobenius over all 2 x adapters "
                    "matrices); ref16_floor = the same for the reference's own bf16 arithmetic.  Full-size cases: DESIGN.md section 7"}


def build_wan(dev, layers=30, rank=16):
    """BASELINE config 4 architecture (Wan2.1-T2V-1.3B: 30 blocks, d = 1536, ffn 8960) with synthetic weights, LoRA r16 on the reference's
    ['blocks'] filter (toolkit/models/wan21/wan21.py:330, 735)."""
    import ai_toolkit_amd  # noqa: F401
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.lora import FusedLoRANetwork
    from ai_toolkit_amd.wan import WanTransformer3DModel

    torch.manual_seed(0)
    model = WanTransformer3DModel(num_layers=layers, dtype=torch.bfloat16, device=dev, ops=ops)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "scale_shift_table" in n:
                p.normal_(0, 1.0 / 1536 ** 0.5)
            elif p.ndim >= 2:
                p.normal_(0, 0.02)
            elif n.endswith("bias"):
                p.normal_(0, 0.01)
    net = FusedLoRANetwork(model, lora_dim=rank, target_lin_modules=("WanTransformer3DModel",), transformer_block_names=["blocks"],
                           base_model_version="wan_2.1")
    net.apply_to()
    net.build_arena(dev, groups=model.lora_groups())
    with torch.no_grad():
        for m in net.unet_loras:
            m.lora_up.weight.normal_(0, 0.01)
    net.refresh_shadows(ops)
    model.attach_network(net)
    model.prepare()
    return model, net, ops


def secondary_configs(dev, steps=3):
    """The other BASELINE.json configs on this GPU, a few steps each, so that their rates are observed by whoever runs `python bench.py` and not
    only by the builder: config 2 (SDXL UNet LoRA r8 @1024^2), config 1's architecture (SD1.5 r4 @512^2), config 4's per-GPU shape
    (Wan2.1-T2V-1.3B r16, 13 x 64 x 64 latents), config 5 (FLUX r32 on the fp8 base, W8A8 on the fp8 MFMA).  Synthetic weights / data, full
    train step (noise .. AdamW + EMA), eager launches.  Never part of `value`."""
    import gc

    from ai_toolkit_amd.trainer import FluxLoRATrainStep, UNetLoRATrainStep, WanLoRATrainStep

    out = {}

    def timed(fn, B, ops=None, peak=PEAK_BF16, flop_per_unit=None):
        fn()
        dt, _, _ = timed_steps(fn, steps, torch.cuda.synchronize)
        r = {"value": B * steps / dt, "ms_per_step": 1e3 * dt / steps, "per_gpu_batch": B, "steps": steps,
             "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
        if ops is not None:  # one more, instrumented step: the GEMM (+ implicit-GEMM convolution) launches against the MFMA peak of their operand type
            try:
                rf = gemm_roofline(fn, ops)
                r["roofline"] = {"bound": "mfma", "kernel": "aitk_gemm_nt / aitk_gemm_nt_grouped, all launches of one step", "achieved": rf["tflops"], "peak": peak,
                                 "unit": "TFLOP/s", "frac": rf["tflops"] / peak, "gemm_ms_per_step": rf["gemm_ms_per_step"], "launches_per_step": rf["launches"]}
            except Exception as ex:  # noqa: BLE001
                r["roofline"] = {"error": f"{type(ex).__name__}: {ex}"[:200]}
        if flop_per_unit:
            r["step_frac_of_bf16_peak"] = flop_per_unit * r["value"] / (PEAK_BF16 * 1e12)
        return r

    def unet(kind):
        model, net, ops = build_unet(dev, kind, rank=8 if kind == "sdxl" else 4)
        st = UNetLoRATrainStep(model, net, ops, lr=1e-4, weight_decay=0.01, max_grad_norm=1.0, ema_decay=0.99, seed=1000)
        B, side = (12, 128) if kind == "sdxl" else (32, 64)
        gen = torch.Generator(device=dev).manual_seed(42)
        lat = torch.randn(B, 4, side, side, device=dev, generator=gen).to(torch.bfloat16)
        ctx = (torch.randn(B, 77, model.config["cross_attention_dim"], device=dev, generator=gen) * 0.5).to(torch.bfloat16)
        pooled = (torch.randn(B, 1280, device=dev, generator=gen) * 0.5).to(torch.bfloat16) if kind == "sdxl" else None
        r = timed(lambda: st.step(lat, ctx, pooled), B, ops)
        r.update(unit="images/s", workload="SDXL UNet LoRA r8 @1024^2 (config 2)" if kind == "sdxl" else "SD1.5 UNet LoRA r4 @512^2 (config 1 architecture)")
        return r

    def wan():
        model, net, ops = build_wan(dev)
        st = WanLoRATrainStep(model, net, ops, lr=1e-4, seed=1)
        B = 4
        lat = torch.randn(B, 16, 13, 64, 64, device=dev).to(torch.bfloat16)
        txt = (torch.randn(B, 512, 4096, device=dev) * 0.3).to(torch.bfloat16)
        r = timed(lambda: st.step(lat, txt), B, ops)
        r.update(unit="videos/s", workload="Wan2.1-T2V-1.3B LoRA r16, 49 x 512 x 512 clip = 13 x 64 x 64 latents, 13 312 tokens (config 4, per-GPU shape)")
        return r

    def flux_w8a8():
        model, net, ops = build_flux(dev, rank=32, fp8_base=True, fp8_mfma=True)
        st = FluxLoRATrainStep(model, net, ops, lr=1e-4, weight_decay=0.01, max_grad_norm=1.0, ema_decay=0.99, timestep_type="linear", seed=1000)
        B = 7 if (torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev)) / 2 ** 30 >= 232 else 4
        lat, emb, pooled = make_batch(dev, B, seed=42)
        r = timed(lambda: st.step(lat, emb, pooled), B, ops, PEAK_FP8, FLOP_PER_IMAGE)
        r.update(unit="images/s", workload="FLUX.1-dev LoRA r32 on the fp8 e4m3 base, W8A8 on v_mfma_scale_f32_32x32x64_f8f6f4 (config 5; opt-in mode: "
                                           "per-token e4m3 activations are not the reference's weight-only arithmetic)")
        return r

    def flux_fp8_weight_only():
        """BASELINE config 5 in the REFERENCE'S arithmetic (toolkit/util/quantize.py:43-75 -> optimum-quanto qfloat8 weights, bf16 activations):
        e4m3 weights with a per-output-channel scale, expanded to bf16 on the way to LDS, bf16 MFMA, bf16 / fp32 rank-32 adapter"""
        model, net, ops = build_flux(dev, rank=32, fp8_base=True, fp8_mfma=False)
        st = FluxLoRATrainStep(model, net, ops, lr=1e-4, weight_decay=0.01, max_grad_norm=1.0, ema_decay=0.99, timestep_type="linear", seed=1000)
        B = 7 if (torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev)) / 2 ** 30 >= 232 else 4
        lat, emb, pooled = make_batch(dev, B, seed=42)
        r = timed(lambda: st.step(lat, emb, pooled), B, ops, PEAK_BF16, FLOP_PER_IMAGE)
        r.update(unit="images/s", workload="FLUX.1-dev LoRA r32 over the weight-only fp8 e4m3 base (config 5 as the reference computes it: "
                                           "weights dequantised to bf16 inside the GEMM's staging, bf16 MFMA, bf16 activations)")
        return r

    for name, fn in (("config2_sdxl", lambda: unet("sdxl")), ("config1_sd15", lambda: unet("sd15")), ("config4_wan21", wan),
                     ("config5_flux_fp8_weight_only", flux_fp8_weight_only), ("config5_flux_fp8_w8a8", flux_w8a8)):
        torch.cuda.reset_peak_memory_stats()
        _run_leg(out, [], name, fn)
        gc.collect()
        torch.cuda.empty_cache()
    return out


def _rel_lists(a, b):
    import math

    num = sum(((x.float() - y.float()) ** 2).sum().item() for x, y in zip(a, b))
    return math.sqrt(num / max(sum((y.float() ** 2).sum().item() for y in b), 1e-300))


def parity_full_depth_ours(model, net, ops, dev):
    """HIP half of `parity.full_depth`: the benchmarked model itself (19 double + 38 single blocks, d = 3072, 4096 + 512 tokens, 494 adapters in
    the state the timed steps left them), one B = 1 step with fixed noise / timestep and a zero learning rate -> loss + every adapter
    gradient; returns what the oracle half needs (the frozen base weights by reference, the adapter matrices and gradients as copies)."""
    from ai_toolkit_amd.trainer import FluxLoRATrainStep

    g = torch.Generator(device=dev).manual_seed(777)
    lat = torch.randn(1, 16, 128, 128, device=dev, generator=g).to(torch.bfloat16)
    emb = (torch.randn(1, 512, 4096, device=dev, generator=g) * 0.1).to(torch.bfloat16)
    pooled = (torch.randn(1, 768, device=dev, generator=g) * 0.1).to(torch.bfloat16)
    noise = torch.randn(1, 16, 128, 128, device=dev, generator=g).to(torch.bfloat16)
    ts = torch.tensor([500.0], device=dev)
    st = FluxLoRATrainStep(model, net, ops, lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    lo = st.step(lat, emb, pooled, noise=noise, timesteps=ts).item()
    grads, adapters = [], []
    for m in net.unet_loras:
        grads += [m.lora_down.weight.grad.detach().clone(), m.lora_up.weight.grad.detach().clone()]
        adapters.append((m.lora_name, m.lora_down.weight.detach().clone(), m.lora_up.weight.detach().clone()))
    sd = {k: v for k, v in model.state_dict().items()}  # references: the bf16 base weights outlive the fused model's other copies
    return {"_loss": lo, "_grads": grads, "_adapters": adapters, "_state": sd, "_batch": (lat, emb, pooled, noise, ts), "_rank": net.lora_dim}


def parity_full_depth_oracle(h, dev):
    """Oracle half of `parity.full_depth` (the CHECKER, outside every timed region): the eager oracle on the same base weights, adapter state
    and inputs — first the reference's own arithmetic (bf16 modules + fp32 adapter, toolkit/network_mixins.py:309-321), then fp32 = truth with
    the blocks under activation checkpointing (SDTrainer.py:2226-2238 does the same to bound memory)."""
    from torch.utils.checkpoint import checkpoint

    from oracle import flux_ref, lora_ref, train_ref

    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(dev):
            ref = flux_ref.FluxTransformer2DModel()
    finally:
        torch.set_default_dtype(torch.float32)
    ref.load_state_dict(h.pop("_state"), strict=True)
    for p in ref.parameters():
        p.requires_grad_(False)
    ref_net = lora_ref.RefLoRANetwork(ref, h["_rank"]).to(dev)
    ref_net.torch_multiplier = ref_net.torch_multiplier.to(dev)
    with torch.no_grad():
        for (name, down, up), b in zip(h["_adapters"], ref_net.unet_loras):
            assert name == b.lora_name, (name, b.lora_name)
            b.lora_down.weight.copy_(down)
            b.lora_up.weight.copy_(up)
    ref_net.apply_to()
    for blk in list(ref.transformer_blocks) + list(ref.single_transformer_blocks):
        f = blk.forward
        blk.forward = (lambda *a, _f=f: checkpoint(_f, *a, use_reentrant=False))
    lat, emb, pooled, noise, ts = h["_batch"]
    oracle = train_ref.RefTrainStep(ref, ref_net, lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    l16 = oracle.step(lat, emb, pooled, noise, ts, dtype=torch.bfloat16).item()
    g16 = [p.grad.clone() for p in oracle.params]
    ref.float()
    torch.cuda.empty_cache()
    l32 = oracle.step(lat.float(), emb.float(), pooled.float(), noise.float(), ts).item()
    g32 = [p.grad.clone() for p in oracle.params]
    go, lo = h["_grads"], h["_loss"]
    worst_o = max(_rel_lists([a], [b]) for a, b in zip(go, g32))
    worst_16 = max(_rel_lists([a], [b]) for a, b in zip(g16, g32))
    return {"config": f"FLUX.1-dev 19 + 38 blocks, d = 3072, 4096 img + 512 txt tokens, B = 1, LoRA r{h['_rank']} on {len(h['_adapters'])} Linears: the "
                      "benchmarked model after the timed steps, one step with fixed noise / timestep; oracle fp32 under activation checkpointing",
            "loss_ours": lo, "loss_fp32": l32, "loss_rel": abs(lo - l32) / abs(l32), "ref16_loss_rel": abs(l16 - l32) / abs(l32),
            "grad_rel": _rel_lists(go, g32), "ref16_floor": _rel_lists(g16, g32), "grad_rel_vs_ref16": _rel_lists(go, g16),
            "worst_module_grad_rel": worst_o, "ref16_worst_module": worst_16,
            "note": "grad_rel / ref16_floor as in the small case above; north_star's 1e-3 holds for the loss, not for adapter gradients of "
                    "any bf16-operand execution (DESIGN.md section 7: measured floor ~6e-3)"}


def dvfs_leg(one, dev_index, steps=3, poll_here=True):
    """Shader clock and socket power while the step runs (rank 0, AFTER the timed region — never part of `value`): MI355X clocks to its
    1400-W budget, so the MFMA-bound kernels of this step run well below the 2400 MHz that PEAK_BF16 is quoted at
    (profiles/r03_clock_power_by_kernel.txt: the 8-phase GEMM alone sits at 1.50-1.75 GHz at 1400 W).  Polls rocm-smi from a thread while
    `steps` more steps run; returns None when rocm-smi is not there.  Under data parallelism EVERY rank runs the extra steps (they contain
    the gradient all-reduce); only the rank with poll_here polls."""
    import re
    import shutil
    import subprocess
    import threading

    smi = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
    how = None
    if smi is not None and poll_here:
        dev_index, how = _smi_index(dev_index)
    if smi is None or not poll_here:
        for _ in range(steps + 1):  # the same number of steps as the polling rank
            one()
        torch.cuda.synchronize()
        return None
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                r = subprocess.run([smi, "--showclocks", "--showpower", "-d", str(dev_index)], capture_output=True, text=True, timeout=5).stdout
                c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", r)
                w = re.search(r"Power \(W\): ([\d.]+)", r)
                if c and w:
                    samples.append((int(c.group(1)), float(w.group(1))))
            except Exception:  # noqa: BLE001 - telemetry is best effort
                return
            stop.wait(0.05)

    cap = None
    try:
        r = subprocess.run([smi, "--showmaxpower", "-d", str(dev_index)], capture_output=True, text=True, timeout=5).stdout
        m = re.search(r"Max Graphics Package Power \(W\): ([\d.]+)", r)
        cap = float(m.group(1)) if m else None
    except Exception:  # noqa: BLE001
        pass
    one()
    torch.cuda.synchronize()
    th = threading.Thread(target=poll, daemon=True)
    th.start()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop.set()
    th.join(timeout=6)
    if len(samples) < 4:
        return None
    clk, pw = sorted(c for c, _ in samples), sorted(w for _, w in samples)
    return {"sclk_mhz": {"median": clk[len(clk) // 2], "min": clk[0], "max": clk[-1]}, "power_w": {"median": pw[len(pw) // 2], "max": pw[-1]},
            "power_cap_w": cap, "samples": len(samples), "ms_per_step_while_polling": 1e3 * dt / steps, "smi_device": dev_index, "smi_device_by": how,
            "note": "rocm-smi polled from a thread during extra steps after the timed region; the roofline peak is quoted at 2400 MHz"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("AITK_BENCH_BATCH", "0")),
                    help="per-GPU batch; 0 = 7 when the GPU has >= 252 GiB free (244 GiB peak; 7 x 4608 rows = 126 row tiles make the "
                         "N=3072 GEMMs 5.9 tile rounds on 256 CUs instead of 3.4 at batch 4), else 4 (157 GiB)")
    ap.add_argument("--rank", type=int, default=16)
    ap.add_argument("--fp8-base", action="store_true", help="BASELINE config 5 variant (not the headline metric): fp8 e4m3 base weights")
    ap.add_argument("--fp8-mfma", action="store_true", help="with --fp8-base (implied): run the base GEMMs W8A8 on the MX-scaled fp8 MFMA "
                    "(v_mfma_scale_f32_32x32x64_f8f6f4; activations quantised per token to e4m3) instead of expanding the weights to bf16")
    ap.add_argument("--network", default="lora", choices=["lora", "dora", "lokr"], help="adapter type (headline metric: lora)")
    ap.add_argument("--model", default="flux", choices=["flux", "sdxl", "sd15"], help="flux = the headline metric; sdxl / sd15 = the UNet path "
                    "(BASELINE configs 2 / 1 architectures), single GPU")
    ap.add_argument("--conv-rank", type=int, default=0, help="UNet bench: also wrap the 3x3 convolutions / time_emb_proj / conv_shortcut of the "
                    "ResNet blocks and samplers (the reference's network.conv) at this rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="UNet bench: skip the hipGraph-replay leg")
    ap.add_argument("--recompute-gelu", action="store_true", help="FLUX: drop the GELU outputs after the forward pass (the lora_down gradients "
                    "rebuild them from the pre-activation inside aitk_lora_wgrad2): 6 GB less per image (196 vs 238 GiB peak at B = 7), -0.7 %% step time, "
                    "bit-identical gradients (profiles/r04_recompute_gelu.json); chosen automatically when B = 7 would not fit otherwise")
    ap.add_argument("--allreduce-dtype", default=os.environ.get("AITK_ALLREDUCE_DTYPE", "fp32"), choices=["fp32", "bf16"],
                    help="transport of the LoRA-gradient all-reduce (N > 1): fp32 = parity with one rank on the concatenated batch (default), "
                         "bf16 = half the bytes on the xGMI links, one bf16 rounding of each rank's gradient and of the sum (SURVEY.md section 8e)")
    ap.add_argument("--rccl-channels", type=int, default=int(os.environ.get("AITK_RCCL_CHANNELS", "0")),
                    help="N > 1: pin RCCL to this many channels (NCCL_MIN_NCHANNELS = NCCL_MAX_NCHANNELS; one channel = one workgroup = one CU "
                         "taken from the persistent 256-workgroup compute kernels while a collective runs); 0 = RCCL's own choice")
    ap.add_argument("--no-dvfs", action="store_true", help="skip the clock / power telemetry leg (3 extra steps under rocm-smi polling)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the few-step runs of BASELINE configs 1, 2, 4 and 5 behind the headline (about 80 s)")
    ap.add_argument("--no-extras", action="store_true", help="skip the batch sweep, the bucketed run and the same-GPU eager comparator")
    args = ap.parse_args()
    if args.fp8_mfma:
        args.fp8_base = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))  # plain `python bench.py --gpus N`: spawn the N ranks ourselves
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with nproc-per-node {args.gpus} (or unset WORLD_SIZE to self-launch)")
    # A run with a process group keeps stdout for the ONE JSON line: RCCL prints a version banner (and gloo its connection lines) on
    # file descriptor 1 from C — everything written to fd 1 from here on goes to stderr, the JSON line is written to the saved descriptor
    json_fd = None
    if world > 1 or os.environ.get("AITK_BENCH_FORCE_PG"):
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
    one_device = bool(os.environ.get("AITK_BENCH_ONE_DEVICE"))  # rehearsal: all ranks share device 0 (needs AITK_BENCH_BACKEND=gloo)
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    rccl_hi_prio = False
    if world > 1 or os.environ.get("AITK_BENCH_FORCE_PG"):  # FORCE_PG: 1-rank RCCL group, exercises the collective path on one GPU
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        backend = os.environ.get("AITK_BENCH_BACKEND", "nccl")  # nccl = RCCL; gloo only for the one-device rehearsal
        if args.rccl_channels > 0:  # must be in the environment before the communicator is created
            os.environ["NCCL_MIN_NCHANNELS"] = os.environ["NCCL_MAX_NCHANNELS"] = str(args.rccl_channels)
        if backend == "nccl":
            # the collective's kernels go to a HIGH-PRIORITY stream: the compute kernels of the step are persistent (one workgroup per CU,
            # ~1 ms each), so an all-reduce kernel can only start when a compute kernel ends — with priority it is dispatched ahead of the
            # next queued compute kernel instead of behind the whole queue
            pg_opts = None
            try:
                pg_opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            except Exception:  # noqa: BLE001 - older torch builds: default stream priority
                pass
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, **({"pg_options": pg_opts} if pg_opts is not None else {}))
            rccl_hi_prio = pg_opts is not None
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        pg = dist.group.WORLD
        # preflight: the collective must actually span the ranks the line will claim.  One all-reduce of ones; every rank
        # prints what it saw FIRST (stderr), and a job whose communicator is smaller than --gpus stops here instead of timing replicas.
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe, group=pg)
        seen = int(round(probe.item()))
        sys.stderr.write(json.dumps({"rccl.ranks": seen, "expected": max(args.gpus, world), "rank": rank, "backend": backend,
                                     "device": torch.cuda.get_device_name(dev)}) + "\n")
        sys.stderr.flush()
        if seen != world or (args.gpus > 1 and seen != args.gpus):
            raise SystemExit(f"preflight: the all-reduce spans {seen} rank(s), expected {max(args.gpus, world)}: refusing to report a multi-GPU number")

    if args.model != "flux":
        if json_fd is not None:  # the UNet legs are single-GPU and print their own line
            os.dup2(json_fd, 1)
        bench_unet(args, dev)
        return

    from ai_toolkit_amd.trainer import FluxLoRATrainStep

    model, net, ops = build_flux(dev, rank=args.rank, fp8_base=args.fp8_base, network_type=args.network, fp8_mfma=args.fp8_mfma)
    model.recompute_gelu = bool(args.recompute_gelu)
    step = FluxLoRATrainStep(model, net, ops, lr=1e-4, weight_decay=0.01, max_grad_norm=1.0, ema_decay=0.99,
                             timestep_type="linear", process_group=pg, seed=1000 + rank, allreduce_dtype=args.allreduce_dtype)
    B = args.batch
    if B <= 0:
        avail_gib = (torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev)) / 2 ** 30  # free + what this process holds
        # bf16 base: 245 GiB peak at B = 7; the fp8 base holds 23.8 GB less (weights as bytes, bf16 copies released): 7 fits from 232 GiB
        need = 239 if args.fp8_base else 259  # incl. the 6.6-GiB dS scratch of the 5-matmul attention backward (ops.ATTN_DS)
        B = 7 if (avail_gib >= need and args.network == "lora") else 4
        # between the two: B = 7 still fits once the GELU outputs are dropped after the forward pass (recompute_gelu: 196 GiB peak at B = 7
        # against 238, -0.7 % step time, bit-identical gradients; profiles/r04_recompute_gelu.json) — better than falling to B = 4
        need_rg = 197 if args.fp8_base else 217
        if B == 4 and args.network == "lora" and avail_gib >= need_rg:
            B = 7
            model.recompute_gelu = True
        if world > 1:  # every rank must step the same shard size (global batch = B * world): take the smallest choice
            bt = torch.tensor([B], device=dev, dtype=torch.int64)
            torch.distributed.all_reduce(bt, op=torch.distributed.ReduceOp.MIN)
            B = int(bt.item())
        if world > 1:  # ... and the same graph
            rg = torch.tensor([int(model.recompute_gelu)], device=dev, dtype=torch.int64)
            torch.distributed.all_reduce(rg, op=torch.distributed.ReduceOp.MAX)
            model.recompute_gelu = bool(rg.item())
        batch_note = (f"auto: {avail_gib:.0f} GiB available on rank {rank}, {need} GiB needed for per-GPU batch 7" +
                      (f" ({need_rg} with recompute_gelu)" if model.recompute_gelu and not args.recompute_gelu else "") +
                      ("" if B == 7 else " -> per-GPU batch 4 (the B = 7 rate needs a GPU with nothing else resident)"))
    else:
        batch_note = "--batch / AITK_BENCH_BATCH"
    lat, emb, pooled = make_batch(dev, B, seed=42 + rank)

    def one():
        return step.step(lat, emb, pooled)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one()
    step.collect_dp_timing = world > 1 or pg is not None
    dt, per_step_ms, loss = timed_steps(one, args.steps, barrier)
    per_rank_ms = None
    if world > 1:
        tr = torch.zeros(world, device=dev, dtype=torch.float64)
        tr[rank] = dt
        torch.distributed.all_reduce(tr, op=torch.distributed.ReduceOp.SUM)  # every rank's own wall time for the K steps
        per_rank_ms = [1e3 * x / args.steps for x in tr.tolist()]
        dt = max(tr.tolist())  # the job is as slow as its slowest rank
    final_loss = float(loss.item())
    ips = world * B * args.steps / dt
    workload = (f"FLUX.1-dev DiT {args.network.upper() if args.network != 'lora' else 'LoRA'} r{args.rank}, 1024x1024 (4096 img + 512 txt "
                "tokens), bf16" + (" activations, W8A8 base GEMMs on the MX-scaled fp8 MFMA (per-token e4m3 activations x per-channel e4m3 weights), "
                                   "bf16 / fp32 adapter" if args.fp8_mfma else " activations over a weight-only fp8 e4m3 base" if args.fp8_base else "")
                + ", AdamW+EMA, clip 1.0")
    out = {
        "metric": f"train images/sec, FLUX.1-dev LoRA r{args.rank} @1024^2" + (" (fp8 e4m3 base, W8A8 fp8 MFMA)" if args.fp8_mfma else
                                                                                 " (fp8 e4m3 weight-only base)" if args.fp8_base else "")
                  + (f" [adapter: {args.network}]" if args.network != "lora" else ""),
        "value": ips, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("fp8 e4m3 x e4m3 base GEMMs (fp32 accumulate) + bf16 everything else" if args.fp8_mfma else
                  "bf16 (fp8 e4m3 weight-only base, expanded per layer to bf16 before its GEMM)" if args.fp8_base else "bf16"), "data": "synthetic (random-init FLUX.1-dev architecture, N(0,1) latents, 0.1*N(0,1) text embeds)",
        "config": {"workload": workload,
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}", "adapters": len(net.unet_loras),
                   "lora_params": net.arena_p.numel(), "grad_checkpointing": False, "recompute_gelu": bool(model.recompute_gelu),
                   "adapter_precision": "fp32 master + split bf16 (hi+lo) shadows on MFMA", "batch_choice": batch_note},
        "final_loss": final_loss,
        "step_ms": {"median": _pct(per_step_ms, 0.5), "p10": _pct(per_step_ms, 0.1), "p90": _pct(per_step_ms, 0.9), "n": len(per_step_ms),
                    "note": "GPU time between step boundaries (events on the launch stream), rank 0"},
        "step_mfma_frac": FLOP_PER_IMAGE * (ips / world) / (PEAK_BF16 * 1e12),
    }
    if pg is not None:
        esz = 2 if args.allreduce_dtype == "bf16" else 4
        out["rccl"] = {"ranks": torch.distributed.get_world_size(pg), "backend": torch.distributed.get_backend(pg),
                       "allreduce_dtype": args.allreduce_dtype, "allreduce_bytes_per_step": esz * net.arena_g.numel(), "pieces": 2,
                       "high_priority_stream": rccl_hi_prio, "channels": args.rccl_channels or "rccl default",
                       "env": {k: os.environ[k] for k in ("NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS", "NCCL_ALGO", "NCCL_PROTO", "RCCL_MSCCL_ENABLE",
                                                          "HSA_ENABLE_IPC_MODE_LEGACY") if k in os.environ},
                       "ms_per_step_by_rank": per_rank_ms}
        tmp = {}
        _run_leg(tmp, [], "standalone", lambda: allreduce_standalone(step, net, pg, world, dev))  # every rank takes part
        out["rccl"]["allreduce_standalone"] = tmp["standalone"]
        if rank == 0:
            out["rccl"]["topology"] = xgmi_topology()
        if one_device:
            out["rccl"]["rehearsal"] = "AITK_BENCH_ONE_DEVICE: every rank on device 0 — control flow only, `value` is NOT a multi-GPU measurement"
            out["data"] += " [one-device rehearsal, not a measurement]"
    if step.collect_dp_timing and step.dp_wait_events:
        waits = [a.elapsed_time(b) for a, b in step.dp_wait_events]
        alone = (out.get("rccl", {}).get("allreduce_standalone") or {}).get("total_ms")
        if alone:
            # the line's own answer to "did RCCL starve beside the persistent compute kernels": the all-reduce needs `total_ms` on an idle GPU;
            # hidden_fraction = how much of that disappeared behind the double-block backward, exposed_over_standalone > 1 = it ran SLOWER
            # than alone AND none of it was hidden (starved for CUs)
            out["rccl"]["hidden_fraction"] = max(0.0, 1.0 - _pct(waits, 0.5) / alone)
            out["rccl"]["exposed_over_standalone"] = _pct(waits, 0.5) / alone
        out["allreduce_ms_exposed"] = {"median": _pct(waits, 0.5), "p90": _pct(waits, 0.9), "n": len(waits),
                                       "note": "launch-stream time between the end of backward and the optimizer kernel, i.e. the part "
                                               "of the gradient all-reduce not hidden behind the double-block backward (rank 0)"}
    step.collect_dp_timing = False
    failed_legs = []
    if rank == 0:
        _persist_headline(out)  # the measured headline is on disk before anything else runs
    rf = None
    if not args.no_roofline:
        tmp = {}
        _run_leg(tmp, failed_legs, "roofline", lambda: gemm_roofline(one, ops))  # every rank runs the instrumented step (it contains the gradient all-reduce)
        rf = tmp["roofline"] if "error" not in tmp["roofline"] else None
        if rf is None:
            out["roofline"] = tmp["roofline"]
        _run_leg(out, failed_legs, "roofline_attention", lambda: attention_roofline(one, ops, PEAK_BF16))
    peak_mem = torch.cuda.max_memory_allocated() / 2 ** 30
    out["config"]["peak_mem_GiB"] = round(peak_mem, 1)
    if not args.no_dvfs:
        tmp = {}
        _run_leg(tmp, failed_legs, "dvfs", lambda: dvfs_leg(one, local_rank, poll_here=rank == 0))
        if tmp["dvfs"] is not None:
            out["dvfs"] = tmp["dvfs"]
    if world > 1:
        torch.distributed.barrier()

    extras = world == 1 and not args.no_extras and not args.fp8_base and args.network == "lora"
    full_parity = None
    if extras:
        # ---- batch sweep (single bucket): the headline B plus 1 and 4 (SURVEY.md section 8d asked for B in {1, 2, 4})
        del lat, emb, pooled
        sweep = {str(B): {"images_per_s": ips, "ms_per_step": 1e3 * dt / args.steps}}

        def leg_sweep():
            for b2 in (1, 4):
                if b2 == B:
                    continue
                l2, e2, p2 = make_batch(dev, b2, seed=43)
                fn = lambda: step.step(l2, e2, p2)  # noqa: E731
                fn()
                d2, _, _ = timed_steps(fn, 3, barrier)
                sweep[str(b2)] = {"images_per_s": b2 * 3 / d2, "ms_per_step": 1e3 * d2 / 3}
            return sweep

        _run_leg(out, failed_legs, "batch_sweep", leg_sweep)

        # ---- hipGraph replay of the same step (trainer.step_graphed): the ~5 000 launches of forward + backward leave the Python
        # host path.  Measured at B = 1, the reference's default batch size, where the eager launch sequence is closest to host-bound.
        def leg_graph():
            torch.cuda.empty_cache()
            l2, e2, p2 = make_batch(dev, 1, seed=43)
            fn = lambda: step.step_graphed(latents=l2, prompt_embeds=e2, pooled_embeds=p2)  # noqa: E731
            try:
                fn()
                fn()
                d2, _, _ = timed_steps(fn, 5, barrier)
            finally:
                step._graphs.clear()
                step._graph_pool = None
            return {"per_gpu_batch": 1, "images_per_s": 5 / d2, "ms_per_step": 1e3 * d2 / 5,
                    "eager_images_per_s": sweep["1"]["images_per_s"] if "1" in sweep else None,
                    "note": "forward + loss + backward replayed as one hipGraph per bucket shape; clip/AdamW/EMA launched eagerly"}

        _run_leg(out, failed_legs, "graph_replay", leg_graph)
        bb = min(B, 4)

        # ---- bucketed run (BASELINE.json configs[2] "1024x1024 buckets"): the five resolutions of BASELINE.md section 2 cycled so the
        # sequence length changes every step (toolkit/config_modules.py:1095-1113, toolkit/data_loader.py:718, 749-756)
        def leg_bucketed():
            torch.cuda.empty_cache()
            batches = [make_batch(dev, bb, w, h, seed=50 + i) for i, (w, h) in enumerate(BUCKETS)]
            for bt in batches:  # first touch of every shape (allocator, RoPE tables) outside the timed region
                step.step(*bt)
            order = [batches[i % len(batches)] for i in range(2 * len(batches))]
            it = iter(order)
            db, per_b, _ = timed_steps(lambda: step.step(*next(it)), len(order), barrier)
            return {"images_per_s": bb * len(order) / db, "per_gpu_batch": bb, "steps": len(order),
                    "buckets": [f"{w}x{h}" for w, h in BUCKETS],
                    "ms_per_step_by_bucket": {f"{w}x{h}": round((per_b[i] + per_b[i + 5]) / 2, 2) for i, (w, h) in enumerate(BUCKETS)},
                    "single_bucket_same_batch_images_per_s": sweep.get(str(bb), {}).get("images_per_s")}

        _run_leg(out, failed_legs, "bucketed", leg_bucketed)

        # ---- uncached-latent mode (north_star "VAE latent encode" inside the hot loop; jobs/process/BaseSDTrainProcess.py:1133 calls
        # sd.encode_images when the batch carries no cached latents): the FLUX.1 AutoencoderKL encoder runs on the batch's images in
        # front of every step (the whole batch in one launch sequence), same batch size as the bucketed leg
        def leg_uncached():
            from ai_toolkit_amd import vae as nvae

            torch.cuda.empty_cache()
            enc = nvae.AutoencoderKLEncoder(dtype=torch.bfloat16, device=dev, ops=ops)
            gv = torch.Generator(device=dev).manual_seed(7)
            with torch.no_grad():
                for name, prm in enc.named_parameters():
                    if name.endswith("weight") and prm.dim() > 1:
                        prm.copy_((torch.randn(prm.shape, device=dev, generator=gv) * prm[0].numel() ** -0.5).to(prm.dtype))
            enc.prepare()
            imgs = torch.rand(bb, 3, 1024, 1024, device=dev, generator=gv) * 2 - 1
            _, e2, p2 = make_batch(dev, bb, seed=44)
            vb = int(os.environ.get("AITK_BENCH_VAE_BATCH", str(bb)))  # images per encoder launch sequence

            def fn_vae():
                lat2 = torch.cat([enc.encode_images(imgs[i:i + vb], generator=gv) for i in range(0, bb, vb)])
                return step.step(lat2, e2, p2)

            fn_vae()
            dv, _, _ = timed_steps(fn_vae, 3, barrier)
            return {"images_per_s": bb * 3 / dv, "ms_per_step": 1e3 * dv / 3, "per_gpu_batch": bb,
                    "cached_latents_same_batch_images_per_s": sweep.get(str(bb), {}).get("images_per_s"),
                    "note": f"FLUX.1 VAE encoder (1024x1024 -> 16x128x128, {vb} image(s) per launch sequence) + train step"}

        _run_leg(out, failed_legs, "uncached_latents", leg_uncached)

        # ---- the HIP half of the full-depth parity leg: THIS model — 19 + 38 blocks, 494 adapters, whatever state the
        # timed steps left the adapter in — steps one B = 1 batch with fixed noise / timestep; the oracle half runs after the model is released
        def leg_full_parity_ours():
            return parity_full_depth_ours(model, net, ops, dev)

        tmp = {}
        _run_leg(tmp, failed_legs, "parity_full_depth", leg_full_parity_ours)
        full_parity = tmp.get("parity_full_depth")
    if rank == 0:
        if rf is not None:
            if os.environ.get("AITK_GEMM_CENSUS"):  # per-shape breakdown of the instrumented step (not part of the JSON line)
                with open(os.environ["AITK_GEMM_CENSUS"], "w") as fh:
                    json.dump(rf["census"], fh, indent=0)
            out["roofline"] = {"bound": "mfma", "kernel": ("aitk_gemm_nt / aitk_gemm_nt_grouped: gemm_nt_8phase_f8_kernel, gemm_nt_8phase_f8_grouped_kernel (W8A8 on "
                                                          "v_mfma_scale_f32_32x32x64_f8f6f4 + bf16 LoRA slab, all launches of one step)" if args.fp8_mfma else
                                                          "aitk_gemm_nt / aitk_gemm_nt_grouped: gemm_nt_8phase_kernel, gemm_nt_8phase_grouped_kernel "
                                                          "(image+text stream of the double blocks in one launch), gemm_nt_kernel<1,128,128> "
                                                          "(LoRA-fused bf16 GEMM, all launches of one step)"),
                               "achieved": rf["tflops"], "peak": PEAK_FP8 if args.fp8_mfma else PEAK_BF16, "unit": "TFLOP/s",
                               "frac": rf["tflops"] / (PEAK_FP8 if args.fp8_mfma else PEAK_BF16),
                               "traffic": None, "launches_per_step": rf["launches"], "avg_launch_us": rf["avg_launch_us"],
                               "gemm_ms_per_step": rf["gemm_ms_per_step"]}
            if out.get("dvfs"):  # the same rate against the MFMA peak at the clock the step sustained (step median: the GEMMs alone clock lower)
                clk = out["dvfs"]["sclk_mhz"]["median"]
                out["roofline"]["frac_at_step_clock"] = out["roofline"]["achieved"] / (out["roofline"]["peak"] * clk / 2400.0)
            # memory-side bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction +
            # WRITE_SIZE) for the dominant shape (B * 4608) x 3072 x 3072 (+ LoRA slab): newest round first
            for pmc in _pmc_summaries():
                by_m = json.load(open(pmc)).get("gemm_nt_8phase_f8_kernel_by_M" if args.fp8_mfma else "gemm_nt_8phase_kernel_by_M", {})
                g8 = by_m.get(str(B * 4608))  # the most frequent launch of a step
                if g8 is not None:
                    out["roofline"]["traffic"] = g8["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_note"] = (f"PMC ({os.path.basename(os.path.dirname(pmc))}), launch {B * 4608}x3072x3072+LoRA slab: "
                                                       f"memory-side bytes incl. Infinity-Cache hits; algorithmic {g8['algorithmic_bytes']} B")
                    break
    if extras:
        # ---- same-GPU eager comparator, measured in this run (our step state released first: the eager path needs ~78 GiB at B = 1)
        del step, model, net, one
        import gc

        gc.collect()
        torch.cuda.empty_cache()
        # ---- the step the REFERENCE'S TRAINER runs over the plug-in: SDTrainer.hook_train_loop's sequence over an ADOPTED
        # network on a fresh bare model — get_noise_prediction through the autograd bridge, torch MSE + the mid-step isfinite sync,
        # loss.backward(), clip_grad_norm_, torch.optim.AdamW(eps=1e-6).step(), zero_grad(set_to_none=True), ema.update(), loss.item() —
        # timed at B = 1 (the reference's default batch size) and at the headline's B, with the optimizer / EMA served by the arena kernels
        # (`trainer_path`, the default of ai_toolkit_amd/adopt.py) and left to torch's foreach AdamW + the EMA class's Python loop
        # (`trainer_path_torch`), beside this run's fused-step numbers for the same batch sizes
        def leg_trainer_path():
            from tools.gpu_trainer_path import run_trainer_path

            res = run_trainer_path(dev, sorted({1, B}), steps=3, warm=2, rank=args.rank, log=lambda *_: None)
            fused = {str(B): 1e3 * dt / args.steps, **{k: v["ms_per_step"] for k, v in (out.get("batch_sweep") or {}).items() if isinstance(v, dict) and "ms_per_step" in v}}
            for b2 in sorted({1, B}):
                ent = res.get(str(b2))
                if ent and str(b2) in fused:
                    ent["fused_step_ms"] = fused[str(b2)]
                    ent["gap_vs_fused_step"] = {k: v["ms_per_step"] / fused[str(b2)] - 1.0 for k, v in ent.items() if isinstance(v, dict) and "ms_per_step" in v}
            res["note"] = ("tools/trainer_harness.TrainerLoop.hook_train_loop = extensions_built_in/sd_trainer/SDTrainer.py:2243-2318 over an adopted network "
                           "(stand-in trainer-side objects with the reference's protocol; every kernel of the step is the HIP library's or one of the torch ops the "
                           "reference's loop issues itself); 3 timed steps after 2, host clock around the loop (it syncs every step like the reference: loss.item())")
            return res

        _run_leg(out, failed_legs, "trainer_path", leg_trainer_path)
        gc.collect()
        torch.cuda.empty_cache()
        _run_leg(out, failed_legs, "gpu_comparator", lambda: gpu_comparator(dev, args.rank))
        gc.collect()
        torch.cuda.empty_cache()
        _run_leg(out, failed_legs, "parity", lambda: parity_leg(dev))
        gc.collect()
        torch.cuda.empty_cache()
        if isinstance(full_parity, dict) and "error" not in full_parity:
            tmp = {}
            _run_leg(tmp, failed_legs, "parity_full_depth", lambda: parity_full_depth_oracle(full_parity, dev))
            full_parity = tmp["parity_full_depth"]
        if isinstance(out.get("parity"), dict) and full_parity is not None:
            out["parity"]["full_depth"] = full_parity
        elif full_parity is not None:
            out["parity_full_depth"] = full_parity
        full_parity = None
        gc.collect()
        torch.cuda.empty_cache()
        if not args.no_secondary:
            _run_leg(out, failed_legs, "secondary_configs", lambda: secondary_configs(dev))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        _run_leg(out, failed_legs, "cpu_baseline", cpu_baseline)
    if pg is not None:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if failed_legs:
        out["failed_legs"] = failed_legs  # each failed leg carries its own {"error": ...}; the headline fields above are unaffected
    if rank == 0:
        _flush_c_stdio()
        sys.stdout.flush()
        if json_fd is not None:
            os.write(json_fd, (json.dumps(out) + "\n").encode())
        else:
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
