#!/bin/bash
# round-2 GPU pass V: kernel stats of the headline step on the final tree (rocprofv3), SDXL bench lines with and without conv-LoRA
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2v_prof" -o b7 --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 2 --no-extras --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r2v_bench_prof.log" 2>&1)
echo "prof rc=$?"; grep '^{' gpurun_out/r2v_bench_prof.log | tail -1 | cut -c1-200
f=$(find gpurun_out/r2v_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-160
rm -f $(find gpurun_out/r2v_prof -name "*kernel_trace.csv")
timeout 300 python bench.py --model sdxl --steps 8 --warmup 3 --no-graph > gpurun_out/r2v_bench_sdxl.log 2>&1; echo "sdxl rc=$?"; grep '^{' gpurun_out/r2v_bench_sdxl.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['adapters'], 'adapters', round(d['roofline']['achieved'],1))"
timeout 300 python bench.py --model sdxl --conv-rank 8 --steps 8 --warmup 3 --no-graph > gpurun_out/r2v_bench_sdxl_conv.log 2>&1; echo "sdxl conv rc=$?"; grep '^{' gpurun_out/r2v_bench_sdxl_conv.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['adapters'], 'adapters', round(d['roofline']['achieved'],1))"
