#!/bin/bash
# round 3, second batch: native 64-wide head attention (SDXL), ranks above 64 / conv ranks above 16, bench stdout order
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest -q -x tests/test_gpu_r3_kernels.py tests/test_gpu_conv_lora.py tests/test_gpu_e2e.py tests/test_gpu_unet.py tests/test_gpu_wan.py \
  -m gpu -s > gpurun_out/r03_pytest_b.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r03_pytest_b.log
tail -5 gpurun_out/r03_pytest_b.log
timeout 600 python bench.py --model sdxl --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03_bench_sdxl_native.json 2> gpurun_out/r03_bench_sdxl_native.err
echo "sdxl rc=$?"; tail -c 600 gpurun_out/r03_bench_sdxl_native.json
AITK_BENCH_FORCE_PG=1 timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r03_bench_rccl1_b.json 2> gpurun_out/r03_bench_rccl1_b.err
echo "rccl1 rc=$?"; tail -n 1 gpurun_out/r03_bench_rccl1_b.json | cut -c1-200
