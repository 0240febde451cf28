#!/bin/bash
# round 4, call G: wave-specialised dK/dV kernel v3 — bit-identity, trace, isolated timing, step A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_attn_ws.py -x -q -m gpu > gpurun_out/r04_pytest_attn_ws.log 2>&1; tail -8 gpurun_out/r04_pytest_attn_ws.log
timeout 200 python tools/gpu_attn_ws_trace.py 2>&1 | grep -v amdgpu | grep "tile 9\|tile 10" | tee gpurun_out/r04_attn_ws_trace.txt
AITK_AB_B=7 timeout 400 python tools/gpu_attn_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_attn_ab_ws.txt
if grep -q passed gpurun_out/r04_pytest_attn_ws.log && ! grep -q failed gpurun_out/r04_pytest_attn_ws.log; then
for ws in 0 1 0 1; do
  AITK_ATTN_DKDV_WS=$ws timeout 300 python bench.py --steps 8 --warmup 3 --no-extras --no-cpu-baseline --no-dvfs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ws $ws', round(d['value'],4), round(d['ms_per_step'],1), round(d['final_loss'],6))"
done | tee gpurun_out/r04_ab_attn_ws_step.txt
fi
