#!/bin/bash
# round-2 GPU pass G: full suite after the recording keep-alive fix; lora_down U = 6 (4 waves/SIMD) vs U = 8 A/B; default bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r2g_pytest.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r2g_pytest.log | cut -c1-300
B="--steps 6 --warmup 2 --no-extras --no-cpu-baseline"
run() { tag=$1; shift; env "$@" timeout 300 python bench.py $B > gpurun_out/r2g_bench_$tag.log 2>&1; echo "$tag rc=$? $(tail -1 gpurun_out/r2g_bench_$tag.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],3), 'img/s', round(d['step_ms']['median'],1), 'ms; gemm', round(d['roofline']['achieved'],1), 'TF', round(d['roofline']['gemm_ms_per_step'],1), 'ms', d['roofline']['launches_per_step'], 'launches')" 2>&1)"; }
run u6 AITK_LORA_DOWN_U=6
run u8 AITK_LORA_DOWN_U=8
run u6_again AITK_LORA_DOWN_U=6
timeout 200 python tools/gpu_lora_down_split_bench.py > gpurun_out/r2g_skinny_u6.log 2>&1; echo "skinny u6 rc=$?"; tail -12 gpurun_out/r2g_skinny_u6.log
AITK_LORA_DOWN_U=8 timeout 200 python tools/gpu_lora_down_split_bench.py > gpurun_out/r2g_skinny_u8.log 2>&1; echo "skinny u8 rc=$?"; tail -12 gpurun_out/r2g_skinny_u8.log
