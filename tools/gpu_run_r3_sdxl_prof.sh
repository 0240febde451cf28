#!/bin/bash
# rocprofv3 kernel stats of the SDXL UNet LoRA step (BASELINE config 2) and of the SD1.5 step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for m in sdxl sd15; do
  (cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r3_prof_$m" -o $m --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --model $m --steps 6 --warmup 2 --no-extras --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r3_bench_prof_$m.log" 2>&1)
  echo "prof $m rc=$?"; grep '^{' gpurun_out/r3_bench_prof_$m.log | tail -1 | cut -c1-200
  f=$(find gpurun_out/r3_prof_$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r3_kernel_stats_$m.csv && head -30 "$f" | cut -c1-140
  rm -rf gpurun_out/r3_prof_$m
done
