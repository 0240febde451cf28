#!/bin/bash
# round 4, call E: XCD start stagger of the persistent GEMM (A/B on the headline step) + Cout = 128 convolutions on the 8-phase kernel (VAE A/B)
mkdir -p gpurun_out
export TMPDIR=/tmp
for sg in 0 1 2 4 0; do
  AITK_GEMM_STAGGER=$sg timeout 300 python bench.py --steps 8 --warmup 3 --no-extras --no-cpu-baseline --no-dvfs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stagger $sg', round(d['value'],4), round(d['ms_per_step'],1), round(d['roofline']['achieved'],1), round(d['roofline']['gemm_ms_per_step'],1))"
done | tee gpurun_out/r04_ab_xcd_stagger.txt
AITK_CONV8_MIN_N=128 AITK_VAE_BENCH_OUT=gpurun_out/r04_vae_bench_conv8_n128.json timeout 300 python tools/gpu_vae_bench.py 2>/dev/null | tail -1
AITK_VAE_BENCH_OUT=gpurun_out/r04_vae_bench_conv8_n256.json timeout 300 python tools/gpu_vae_bench.py 2>/dev/null | tail -1
