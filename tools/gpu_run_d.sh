#!/bin/bash
# round-2 GPU pass D: hipGraph replay tests + UNet benches with the graph leg
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_fp8.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r2d_pytest.log 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/r2d_pytest.log | cut -c1-300
timeout 400 python bench.py --model sdxl --steps 6 --warmup 2 > gpurun_out/r2d_bench_sdxl.log 2>&1; echo "sdxl rc=$?"; tail -c 1500 gpurun_out/r2d_bench_sdxl.log
timeout 300 python bench.py --model sd15 --steps 6 --warmup 2 > gpurun_out/r2d_bench_sd15.log 2>&1; echo "sd15 rc=$?"; tail -c 1500 gpurun_out/r2d_bench_sd15.log
