#!/bin/bash
# round-3 evidence pass: rocprofv3 kernel stats of the headline step (bf16) and of the config-5 W8A8 step, PMC passes (SQ activity + MFMA busy,
# FETCH_SIZE, WRITE_SIZE in separate runs; no trace domains alongside --pmc) over tools/gpu_pmc_target.py
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r3_prof" -o b7 --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 2 --no-extras --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r3_bench_prof.log" 2>&1)
echo "prof bf16 rc=$?"; grep '^{' gpurun_out/r3_bench_prof.log | tail -1 | cut -c1-160
f=$(find gpurun_out/r3_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r3_kernel_stats_b7.csv && head -12 "$f" | cut -c1-150
rm -rf gpurun_out/r3_prof
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r3_prof8" -o f8 --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --fp8-mfma --rank 32 --steps 6 --warmup 2 --no-extras --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r3_bench_prof_f8.log" 2>&1)
echo "prof f8 rc=$?"; grep '^{' gpurun_out/r3_bench_prof_f8.log | tail -1 | cut -c1-160
f=$(find gpurun_out/r3_prof8 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r3_kernel_stats_f8_r32_b7.csv && head -12 "$f" | cut -c1-150
rm -rf gpurun_out/r3_prof8
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=p$i; i=$((i+1))
  (cd /tmp && AITK_PMC_M=32256 AITK_PMC_K2=48 timeout 300 rocprofv3 --pmc $c -d "$GRAFT_REPO_ROOT/gpurun_out/r3_pmc/$tag" -o $tag --output-format csv -- python "$GRAFT_REPO_ROOT/tools/gpu_pmc_target.py" > "$GRAFT_REPO_ROOT/gpurun_out/r3_pmc_$tag.log" 2>&1)
  echo "pmc $tag rc=$?"
done
python tools/pmc_round_summary.py gpurun_out/r3_pmc gpurun_out/r3_pmc_summary.json 32256 48 | tail -40
find gpurun_out/r3_pmc -name "*.csv" -size +2M -delete
