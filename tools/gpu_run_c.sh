#!/bin/bash
# round-2 GPU pass C: full GPU suite on the final tree, headline bench, SDXL bench, VAE bench, PMC passes (separate runs, counters only)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -s --durations=12 -p no:cacheprovider > gpurun_out/r2c_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "PARITY|passed|failed|FAILED" gpurun_out/r2c_pytest.log | cut -c1-400 | head -30
timeout 700 python bench.py > gpurun_out/r2c_bench.log 2>&1; echo "bench rc=$?"; tail -c 2500 gpurun_out/r2c_bench.log
timeout 400 python bench.py --model sdxl --steps 6 --warmup 2 > gpurun_out/r2c_bench_sdxl.log 2>&1; echo "sdxl rc=$?"; tail -c 1200 gpurun_out/r2c_bench_sdxl.log
timeout 300 python tools/gpu_vae_bench.py > gpurun_out/r2c_vae_bench.log 2>&1; echo "vae rc=$?"; tail -5 gpurun_out/r2c_vae_bench.log
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES"; do
  tag=$(echo $c | cut -d' ' -f1)
  (cd /tmp && AITK_PMC_M=32256 AITK_PMC_K2=48 timeout 300 rocprofv3 --pmc $c -d "$GRAFT_REPO_ROOT/gpurun_out/r2c_pmc/$tag" -o $tag --output-format csv -- python "$GRAFT_REPO_ROOT/tools/gpu_pmc_target.py" > "$GRAFT_REPO_ROOT/gpurun_out/r2c_pmc_$tag.log" 2>&1)
  echo "pmc $tag rc=$?"
done
find gpurun_out/r2c_pmc -name "*counter_collection.csv" | head
