#!/bin/bash
# round 4 evidence pass on the final tree: full GPU test tier, the driver's bench command, rocprofv3 kernel stats of the headline step, PMC passes
# (SQ activity + MFMA busy, FETCH_SIZE, WRITE_SIZE in separate runs; no trace domains alongside --pmc) over tools/gpu_pmc_target.py
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/r04_pytest_gpu_full_final2.log 2>&1; tail -25 gpurun_out/r04_pytest_gpu_full_final2.log
timeout 900 python bench.py > gpurun_out/r04_bench_default_final2.json 2> gpurun_out/r04_bench_default_final2.err; tail -2 gpurun_out/r04_bench_default_final2.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_bench_default_final2.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "gemm", d["roofline"].get("achieved"), d["roofline"].get("frac"), "failed", d.get("failed_legs"))
print("sweep", d.get("batch_sweep"), "uncached", d.get("uncached_latents"), "bucketed", (d.get("bucketed") or {}).get("images_per_s"))
print("secondary", json.dumps(d.get("secondary_configs"))[:1500])
print("parity full", json.dumps((d.get("parity") or {}).get("full_depth"))[:700])
PY
(cd /tmp && AITK_GEMM_CENSUS="$GRAFT_REPO_ROOT/gpurun_out/r04_census_final2.json" timeout 500 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r4_prof" -o b7 --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 2 --no-extras --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r04_bench_b7_profiled_final2.log" 2>&1)
echo "prof bf16 rc=$?"; grep '^{' gpurun_out/r04_bench_b7_profiled_final2.log | tail -1 | cut -c1-160
f=$(find gpurun_out/r4_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r04_rocprof_kernel_stats_b7_final2.csv && head -14 "$f" | cut -c1-150
rm -rf gpurun_out/r4_prof
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=p$i; i=$((i+1))
  (cd /tmp && AITK_PMC_M=32256 AITK_PMC_K2=48 AITK_PMC_F8=0 timeout 300 rocprofv3 --pmc $c -d "$GRAFT_REPO_ROOT/gpurun_out/r4_pmc2/$tag" -o $tag --output-format csv -- python "$GRAFT_REPO_ROOT/tools/gpu_pmc_target.py" > "$GRAFT_REPO_ROOT/gpurun_out/r4_pmc2_$tag.log" 2>&1)
  echo "pmc $tag rc=$?"
done
python tools/pmc_round_summary.py gpurun_out/r4_pmc2 gpurun_out/r04_pmc_summary_final2.json 32256 48 | tail -45
find gpurun_out/r4_pmc2 -name "*.csv" -size +2M -delete
