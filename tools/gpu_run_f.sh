#!/bin/bash
# round-2 GPU pass F: grouped 8-phase launch — full suite, then same-box A/B of stream pairing and of the big-tile threshold
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r2f_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r2f_pytest.log | cut -c1-300
B="--steps 6 --warmup 2 --no-extras --no-cpu-baseline"
run() { tag=$1; shift; env "$@" timeout 300 python bench.py $B > gpurun_out/r2f_bench_$tag.log 2>&1; echo "$tag rc=$? $(tail -1 gpurun_out/r2f_bench_$tag.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],3), 'img/s', round(d['step_ms']['median'],1), 'ms; gemm', round(d['roofline']['achieved'],1), 'TF', round(d['roofline']['gemm_ms_per_step'],1), 'ms', d['roofline']['launches_per_step'], 'launches')" 2>&1)"; }
run pair_t192 AITK_PAIR_STREAMS=1
run nopair_t192 AITK_PAIR_STREAMS=0
run nopair_t128 AITK_PAIR_STREAMS=0 AITK_BIG_TILES_MIN=128
run pair_t128 AITK_PAIR_STREAMS=1 AITK_BIG_TILES_MIN=128
run pair_t192_again AITK_PAIR_STREAMS=1
for t in 192 128 96; do
  AITK_BIG_TILES_MIN=$t timeout 200 python bench.py --model sdxl --steps 6 --warmup 2 --no-graph > gpurun_out/r2f_sdxl_t$t.log 2>&1
  echo "sdxl t$t rc=$? $(tail -1 gpurun_out/r2f_sdxl_t$t.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), 'img/s; gemm+conv', round(d['roofline']['achieved'],1), 'TF', round(d['roofline']['gemm_conv_ms_per_step'],1), 'ms')" 2>&1)"
done
