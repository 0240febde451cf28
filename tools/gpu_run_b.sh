#!/bin/bash
# round-2 GPU pass B: UNet kernels + path, parity ablation, SDXL / SD1.5 bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/gpu_check4.py > gpurun_out/r2b_check4.log 2>&1; echo "check4 rc=$?"; tail -25 gpurun_out/r2b_check4.log
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_e2e.py tests/test_gpu_parity_r2.py::test_tiny_step_four_way_parity -q -s -p no:cacheprovider > gpurun_out/r2b_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "PARITY|passed|failed|FAILED|Error" gpurun_out/r2b_pytest.log | head -30
timeout 400 python tools/gpu_parity_ablate.py > gpurun_out/r2b_ablate.log 2>&1; echo "ablate rc=$?"; cat gpurun_out/r2b_ablate.log | tail -14
timeout 600 python bench.py --model sdxl --steps 6 --warmup 2 > gpurun_out/r2b_bench_sdxl.log 2>&1; echo "sdxl rc=$?"; tail -c 1800 gpurun_out/r2b_bench_sdxl.log
timeout 300 python bench.py --model sd15 --steps 6 --warmup 2 > gpurun_out/r2b_bench_sd15.log 2>&1; echo "sd15 rc=$?"; tail -c 1500 gpurun_out/r2b_bench_sd15.log
