#!/bin/bash
# round-2 GPU pass M: fp8 base with merged stream launches + B = 7
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_grouped_gemm.py tests/test_gpu_fp8.py tests/test_gpu_vae.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4
B="--steps 4 --warmup 2 --no-extras --no-cpu-baseline"
for cfg in "--fp8-base --rank 32" "--fp8-base --rank 32 --batch 4"; do
  timeout 300 python bench.py $cfg $B > gpurun_out/r2m_fp8.log 2>&1
  echo "[$cfg] rc=$? $(grep '^{' gpurun_out/r2m_fp8.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],3), d['unit'], 'B', d['config'].get('per_gpu_batch'), round(d['ms_per_step'],1), 'ms; gemm', round(d.get('roofline',{}).get('achieved',0),1), 'TF; mem', d['config'].get('peak_mem_GiB'))" 2>&1 | cut -c1-300)"
  grep '^{' gpurun_out/r2m_fp8.log | tail -1 >> gpurun_out/r2m_fp8_lines.json
done
timeout 200 python bench.py $B > gpurun_out/r2m_bf16.log 2>&1; echo "bf16 $(grep '^{' gpurun_out/r2m_bf16.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],3))")"
