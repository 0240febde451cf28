#!/bin/bash
# round 4: kernel stats of the FLUX step at per-GPU batch 1 (the reference's default batch size): where does the per-step fixed cost sit?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r4_prof_b1" -o b1 --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --batch 1 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r04_bench_b1_profiled.log" 2>&1)
grep '^{' gpurun_out/r04_bench_b1_profiled.log | tail -1 | cut -c1-200
f=$(find gpurun_out/r4_prof_b1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r04_rocprof_kernel_stats_b1.csv && head -30 "$f" | cut -c1-160
rm -rf gpurun_out/r4_prof_b1
