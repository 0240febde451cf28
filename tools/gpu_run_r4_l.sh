#!/bin/bash
# round 4, run L: merged-phase K loop of the persistent GEMM (AITK_GEMM8_PH=4: 4 barrier intervals of 16 MFMAs per K-tile instead of 8 of 8) — bit-exactness,
# time(K) fits, FLUX shapes, trace, step A/B.  The first command runs under a short timeout: a barrier mismatch would hang.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
AITK_EVS=1,2 timeout 120 python tools/gpu_gemm8_ev.py check > gpurun_out/r04_gemm8_ph4_check.log 2>&1; rc=$?; echo "check rc=$rc"; tail -6 gpurun_out/r04_gemm8_ph4_check.log | cut -c1-300
if [ $rc -ne 0 ]; then echo "bit-exactness failed or hung: stopping"; exit 0; fi
AITK_EVS=1,2 timeout 400 python tools/gpu_gemm8_ev.py sweep trace > gpurun_out/r04_gemm8_ph4.log 2>&1; echo "sweep rc=$?"; grep -v '^{"trace' gpurun_out/r04_gemm8_ph4.log | cut -c1-420 | tail -14
grep '^{"trace' gpurun_out/r04_gemm8_ph4.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    if d['wg']==0: print(d['trace_epi'][:9], 'variant',d['ev'],'wave',d['wave'], d['ticks_10ns'])
"
for ph in 8 4 8 4; do
  AITK_GEMM8_PH=$ph timeout 300 python bench.py --steps 8 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r04_bench_ph$ph.log 2>&1
  echo "ph=$ph $(grep '^{' gpurun_out/r04_bench_ph$ph.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('gemm_ms_per_step'), d['roofline'].get('achieved'), d.get('final_loss'))")" | tee -a gpurun_out/r04_ab_gemm8_ph4_step.txt
done
