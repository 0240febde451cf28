#!/bin/bash
# round-2 GPU pass H: UNet batch sweep beyond 8 rows (GEMV chunking)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for b in 8 12 16; do
  timeout 300 python bench.py --model sdxl --steps 6 --warmup 2 --batch $b > gpurun_out/r2h_sdxl_b$b.log 2>&1
  echo "sdxl B=$b rc=$? $(tail -1 gpurun_out/r2h_sdxl_b$b.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), 'img/s', d['launch_modes'], '; gemm+conv', round(d['roofline']['achieved'],1), 'TF; mem', d['config']['peak_mem_GiB'])" 2>&1)"
done
for b in 8 16 32; do
  timeout 300 python bench.py --model sd15 --steps 6 --warmup 2 --batch $b > gpurun_out/r2h_sd15_b$b.log 2>&1
  echo "sd15 B=$b rc=$? $(tail -1 gpurun_out/r2h_sd15_b$b.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), 'img/s', d['launch_modes'], '; gemm+conv', round(d['roofline']['achieved'],1), 'TF; mem', d['config']['peak_mem_GiB'])" 2>&1)"
done
timeout 300 python -m pytest tests/test_gpu_unet.py -m gpu -q -x -p no:cacheprovider -k "four_way or three_steps" 2>&1 | tail -3
