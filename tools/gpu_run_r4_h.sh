#!/bin/bash
# round 4, run H: tile-switch variants of the persistent 8-phase GEMM (AITK_GEMM8_EV) — bit-exactness, time(K) fits, FLUX shapes, s_memtime trace, step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/gpu_gemm8_ev.py check sweep trace > gpurun_out/r04_gemm8_ev.log 2>&1; echo "ev tool rc=$?"; grep -v '^{"trace' gpurun_out/r04_gemm8_ev.log | cut -c1-400 | tail -40
for ev in 0 2 1 0 2; do
  AITK_GEMM8_EV=$ev timeout 300 python bench.py --steps 8 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r04_bench_ev$ev.log 2>&1
  echo "ev=$ev $(grep '^{' gpurun_out/r04_bench_ev$ev.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('gemm_ms_per_step'), d['roofline'].get('achieved'))")" | tee -a gpurun_out/r04_ab_gemm8_ev_step.txt
done
