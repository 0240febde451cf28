#!/bin/bash
# round 4, run J: fast epilogue forms of the persistent 8-phase GEMM (AITK_GEMM8_FE) — bit-exactness vs the generic form, time(K) fits, FLUX shapes, trace, step A/B,
# the GEMM / conv / grouped / graph GPU tests on the new default
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/gpu_gemm8_ev.py check sweep trace > gpurun_out/r04_gemm8_fe2.log 2>&1; echo "fe tool rc=$?"; grep -v '^{"trace' gpurun_out/r04_gemm8_fe2.log | cut -c1-420 | tail -40
grep '^{"trace' gpurun_out/r04_gemm8_fe2.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    if d['wg']==0: print(d['trace_epi'][:9], 'fe',d['ev'],'wave',d['wave'], d['ticks_10ns'])
"
for fe in 0 1 0 1; do
  AITK_GEMM8_FE=$fe timeout 300 python bench.py --steps 8 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r04_bench2_fe$fe.log 2>&1
  echo "fe=$fe $(grep '^{' gpurun_out/r04_bench2_fe$fe.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('gemm_ms_per_step'), d['roofline'].get('achieved'))")" | tee -a gpurun_out/r04_ab_gemm8_fe2_step.txt
done
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_grouped_gemm.py tests/test_gpu_conv8.py tests/test_gpu_graph.py tests/test_gpu_e2e.py -q -m gpu -x > gpurun_out/r04_pytest_fe2.log 2>&1; tail -3 gpurun_out/r04_pytest_fe2.log
