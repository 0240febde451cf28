#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_unet.py -q -s -p no:cacheprovider > gpurun_out/r2b2_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "PARITY|passed|failed|FAILED|Error|assert" gpurun_out/r2b2_pytest.log | head -40
timeout 600 python bench.py --model sdxl --steps 6 --warmup 2 > gpurun_out/r2b2_bench_sdxl.log 2>&1; echo "sdxl rc=$?"; tail -c 1800 gpurun_out/r2b2_bench_sdxl.log
timeout 300 python bench.py --model sd15 --steps 6 --warmup 2 > gpurun_out/r2b2_bench_sd15.log 2>&1; echo "sd15 rc=$?"; tail -c 1500 gpurun_out/r2b2_bench_sd15.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2b2_prof_sdxl" -o sdxl -- python "$GRAFT_REPO_ROOT/bench.py" --model sdxl --steps 3 --warmup 1 > "$GRAFT_REPO_ROOT/gpurun_out/r2b2_prof_sdxl.log" 2>&1); echo "prof rc=$?"
