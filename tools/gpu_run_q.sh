#!/bin/bash
# round-2 GPU pass Q: packed fp32 softmax arithmetic in the forward / dQ attention kernels vs the previous build (same box)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/gpu_attn_ab.py 2>&1 | grep -v amdgpu.ids | tail -6
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -k "attn or attention or four_way or tiny or step or padded" 2>&1 | tail -3
B="--steps 6 --warmup 2 --no-extras --no-cpu-baseline"
for lib in prev new prev new; do
  if [ $lib = prev ]; then export AITK_LIB_PATH=$GRAFT_REPO_ROOT/ai-toolkit_amd/libaitk_abl_attn_prev.so; else unset AITK_LIB_PATH; fi
  timeout 300 python bench.py $B > gpurun_out/r2q_bench_$lib.log 2>&1
  echo "$lib $(grep '^{' gpurun_out/r2q_bench_$lib.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],3), 'img/s', round(d['step_ms']['median'],1))")"
done
