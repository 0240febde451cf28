"""Evidence table for the FLUX.1-dev B = 7 step: kernel x calls/step x ms/step x share x achieved rate, from a rocprofv3
kernel-stats CSV of `bench.py --no-extras` and the bench line of the same tree.
    python tools/kernel_table.py profiles/r02_rocprof_kernel_stats_j_b7.csv profiles/r02_bench_j_b7.log > profiles/r02_kernel_table.md
MFMA kernels: executed flops / time (attention: 4 B H S^2 D per matmul pair -> fwd 4, dK/dV 8, dQ 6 B H S^2 D; GEMM: 2 M N (K + K2) summed
over the launches of one step, taken from the bench line's event-timed census).  HBM-bound kernels: share of the step only (their
standalone GB/s are in profiles/r02_lora_down_split_u6.json and profiles/r01_lora_skinny_bench.log)."""
import csv
import json
import sys

B, S, H, D = 7, 4608, 24, 128
PAIR = B * H * S * S * D  # one S x S x D contraction over all heads, multiply-add counted once


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    bench = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    steps = next(int(r["Calls"]) for r in rows if "attn_fwd_kernel" in r["Name"]) / 57.0
    gemm_flops = bench["roofline"]["achieved"] * 1e12 * bench["roofline"]["gemm_ms_per_step"] * 1e-3
    fam = [("gemm_nt_8phase_grouped_et_kernel", "LoRA-fused GEMM, grouped, BIAS + GELU launches that also emit the next layer's lora_down partials (AITK_EPI_EMIT_T)", None),
           ("gemm_nt_8phase_et_kernel", "LoRA-fused GEMM, BIAS + GELU launches that also emit the next layer's lora_down partials (AITK_EPI_EMIT_T)", None),
           ("gemm_nt_8phase_grouped_kernel", "LoRA-fused GEMM, persistent 8-phase, image+text stream grouped", None),
           ("gemm_nt_8phase_kernel", "LoRA-fused GEMM, persistent 8-phase", None),
           ("gemm_nt_kernel", "LoRA-fused GEMM, 128x128 / other", None),
           ("attn_bwd_dkdv", "attention backward dK, dV (4 matmuls; wave-specialised kernel at head_dim 128: 8 waves, two per SIMD; round 6: also emits its bf16 dS, 7.1 GB per launch)", 8 * PAIR),
           ("attn_bwd_dq_ds", "attention backward dQ = dS K as a product of its own (1 matmul; streams the dS the dK/dV pass emitted: HBM-bound)", 2 * PAIR),
           ("attn_bwd_dq_kernel", "attention backward dQ (3 matmuls, recomputes S, dP)", 6 * PAIR),
           ("attn_fwd_kernel", "attention forward", 4 * PAIR),
           ("lora_down16_kernel", "LoRA skinny down (T = x A^T, dT = dy B), split precision", 0),
           ("lora_wgrad_fused", "LoRA backward, one read of dY: dT partials + lora_up gradient (aitk_lora_bwd_fused, 128 columns per workgroup)", 0),
           ("lora_bwd_fused_ct", "LoRA backward, one read of dY: dT partials + lora_up gradient (aitk_lora_bwd_fused, 2 / 4 column tiles per workgroup)", 0),
           ("lora_bwd_finish2", "the two finish passes of aitk_lora_bwd_fused as one launch (lora_up gradient chunks -> arena, dT column partials -> slab)", 0),
           ("lora_wgrad_finish_multi", "up to eight weight-gradient finish passes as one launch (deferred finishes, ABI 12)", 0),
           ("lora_dt_finish", "dT finish (sum of the column partials -> [hi | lo | hi] slab)", 0),
           ("lora_wgrad", "LoRA weight gradients (lora_down gradient; + finish passes)", 0),
           ("ln_mod_", "adaLN LayerNorm fwd / bwd", 0), ("qkv_post", "QK-RMSNorm + RoPE fwd / bwd", 0), ("gate_bwd", "gate backward", 0),
           ("colsum_finish", "column-sum finish", 0), ("gemv_nt", "adaLN / embedder GEMV", 0), ("attn_delta", "attention delta", 0),
           ("adamw_ema", "clip + AdamW + EMA", 0), ("refresh_shadows", "split-precision shadow refresh", 0)]
    agg = {}
    tot = 0.0
    for r in rows:
        t = float(r["TotalDurationNs"]) / steps / 1e6
        key = next((f for f in fam if f[0] in r["Name"]), None)
        if key is None or "at::native" in r["Name"]:
            key = ("other", "torch init / copies (model construction, not the step)", 0)
        a = agg.setdefault(key[0], [key[1], key[2], 0.0, 0.0])
        a[2] += int(r["Calls"]) / steps
        a[3] += t
        tot += t
    gemm_ms = sum(v[3] for k, v in agg.items() if k.startswith("gemm_nt"))
    print(f"# FLUX.1-dev LoRA r16 @1024^2, B = 7: kernel table ({sys.argv[1]}, {steps:.0f} profiled steps; bench line {sys.argv[2]}: "
          f"{bench['value']:.2f} img/s, step {bench['step_ms']['median']:.0f} ms)\n")
    print("| kernel | what | launches / step | ms / step | share | executed TFLOP/s | of 2500 |")
    print("|---|---|---|---|---|---|---|")
    for k, (what, fl, calls, ms) in sorted(agg.items(), key=lambda kv: -kv[1][3]):
        rate = ""
        frac = ""
        if fl:
            tf = fl * (calls if k.startswith("attn") else 1) / (ms * 1e-3) / 1e12
            rate, frac = f"{tf:.0f}", f"{tf / 2500:.2f}"
        print(f"| `{k}` | {what} | {calls:.0f} | {ms:.1f} | {100 * ms / tot:.1f} % | {rate} | {frac} |")
    print(f"\nAll GEMM kernels together: {gemm_ms:.1f} ms / step under the profiler for {gemm_flops / 1e12:.0f} TFLOP per step "
          f"(event-timed census of the bench line) = {gemm_flops / gemm_ms / 1e9:.0f} TFLOP/s = {gemm_flops / gemm_ms / 1e9 / 2500:.3f} of the dense bf16 peak; "
          f"the bench line's own event timing of the same launches: {bench['roofline']['achieved']:.0f} TFLOP/s over {bench['roofline']['gemm_ms_per_step']:.1f} ms.")
    print(f"Sum of kernel time per step {tot:.1f} ms (incl. model construction kernels of the profiled process).")


if __name__ == "__main__":
    main()
