#!/bin/bash
# round 4, second half: the persistent GEMM's tile switch (profiles/r04_gemm8_tile_switch.md).  In their first versions the same three steps also switched variants that
# no longer exist (rolled epilogue / counted tile-top wait: AITK_GEMM8_EV; within-XCD start stagger: AITK_GEMM8_STAGGER; merged-phase K loop: AITK_GEMM8_PH) — their logs
# are kept under profiles/ (r04_gemm8_ev_rolled_epilogue.log, r04_gemm8_stagger_within_xcd.log, r04_gemm8_merged_phases*).
#   1. what bounds the epilogue's store burst (standalone probe)
#   2. generic vs fast epilogue forms: bit-exactness, time(K) fits, FLUX shapes, s_memtime trace of the tile switch
#   3. step A/B, then the GEMM / conv / graph tests on the default build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/epilogue_store tools/probes/epilogue_store.hip 2>/dev/null && timeout 200 /tmp/epilogue_store > gpurun_out/r04_probe_epilogue_store.txt 2>&1
timeout 600 python tools/gpu_gemm8_ev.py check sweep trace > gpurun_out/r04_gemm8_fast_epilogue.log 2>&1; echo "tool rc=$?"; grep -v '^{"trace' gpurun_out/r04_gemm8_fast_epilogue.log | cut -c1-420 | tail -30
for fe in 0 1 0 1; do
  AITK_GEMM8_FE=$fe timeout 300 python bench.py --steps 8 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r04_bench_fe$fe.log 2>&1
  echo "fe=$fe $(grep '^{' gpurun_out/r04_bench_fe$fe.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('gemm_ms_per_step'), d['roofline'].get('achieved'))")" | tee -a gpurun_out/r04_ab_gemm8_fast_epilogue_step.txt
done
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_grouped_gemm.py tests/test_gpu_conv8.py tests/test_gpu_graph.py tests/test_gpu_e2e.py -q -m gpu -x > gpurun_out/r04_pytest_fast_epilogue.log 2>&1; tail -3 gpurun_out/r04_pytest_fast_epilogue.log
