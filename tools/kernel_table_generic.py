"""Kernel table of any bench.py configuration from a rocprofv3 kernel-stats CSV (secondary configs: SDXL, SD1.5, Wan):
    python tools/kernel_table_generic.py <stats.csv> <bench_profiled.log> > profiles/rNN_kernel_table_<config>.md
Steps profiled = launches of adamw_ema_kernel (one per optimizer step, warm-up included).  Kernels are grouped by the name in front of the first
'<' / '(' so that template instances add up; torch's own kernels (model construction, batch generation) are one row."""
import csv
import json
import re
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    bench = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    if "metric" not in bench:  # tools/gpu_wan_bench.py's line
        bench = {"metric": bench["workload"], "value": bench["samples_per_s"], "unit": "videos/s", "ms_per_step": 1e3 * bench["s_per_step"],
                 "config": {"per_gpu_batch": int(bench["workload"].rsplit("B=", 1)[-1])}}
    steps = next((int(r["Calls"]) for r in rows if "adamw_ema_kernel" in r["Name"]), None)
    if steps is None:  # (the optimizer kernel can fall below the CSV's cut) one shadow refresh per step + one at set-up
        steps = next(int(r["Calls"]) for r in rows if "refresh_shadows_kernel" in r["Name"]) - 1
    agg = {}
    for r in rows:
        name = r["Name"]
        key = "torch / runtime kernels (construction, batch generation, copies)" if ("at::native" in name or "rocclr" in name) else re.split(r"[<(]", name.replace("void ", ""))[0]
        a = agg.setdefault(key, [0, 0.0])
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
    tot = sum(v[1] for v in agg.values()) / steps / 1e6
    print(f"# {bench['metric']}: kernel table ({sys.argv[1]}, {steps} profiled steps; bench line of the same run: {bench['value']:.2f} {bench['unit']}, "
          f"{bench['ms_per_step']:.1f} ms / step, per-GPU batch {bench['config'].get('per_gpu_batch')})\n")
    print("| kernel | launches / step | ms / step | share of kernel time | avg us |")
    print("|---|---|---|---|---|")
    for k, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        ms = ns / steps / 1e6
        if ms < 0.002 * tot:
            continue
        print(f"| `{k}` | {c / steps:.0f} | {ms:.2f} | {100 * ms / tot:.1f} % | {ns / c / 1e3:.1f} |")
    r = bench.get("roofline") or {}
    tail = f"Sum of kernel time per step {tot:.1f} ms against a step of {bench['ms_per_step']:.1f} ms under the profiler."
    if r.get("achieved") is not None:
        tail += f"  GEMM launches of one step (event-timed in the bench line): {r['achieved']:.0f} TFLOP/s = {r.get('frac', 0.0):.3f} of the dense bf16 peak"
        tail += f" over {r['gemm_ms_per_step']:.1f} ms." if r.get("gemm_ms_per_step") is not None else "."
    print("\n" + tail)

main()
