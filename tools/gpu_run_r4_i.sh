#!/bin/bash
# round 4, run I: what bounds the GEMM epilogue's store burst (tools/probes/epilogue_store.hip) and the within-XCD start stagger of the persistent GEMM
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/epilogue_store tools/probes/epilogue_store.hip 2>/dev/null && timeout 200 /tmp/epilogue_store > gpurun_out/r04_probe_epilogue_store.txt 2>&1
echo "probe rc=$?"; cat gpurun_out/r04_probe_epilogue_store.txt | cut -c1-200
timeout 300 python tools/gpu_gemm8_ev.py stagger > gpurun_out/r04_gemm8_stagger.log 2>&1; echo "stagger rc=$?"; grep '^{' gpurun_out/r04_gemm8_stagger.log | cut -c1-500
for st in 0 400 1600 0; do
  AITK_GEMM8_STAGGER=$st timeout 300 python bench.py --steps 8 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r04_bench_stag$st.log 2>&1
  echo "stagger=$st $(grep '^{' gpurun_out/r04_bench_stag$st.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('gemm_ms_per_step'), d['roofline'].get('achieved'))")" | tee -a gpurun_out/r04_ab_gemm8_stagger_step.txt
done
