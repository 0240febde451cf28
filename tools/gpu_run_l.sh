#!/bin/bash
# round-2 GPU pass L: the other configurations on the final tree — 1-rank RCCL code path, fp8 base (config 5), Wan2.1 (config 4), DoRA, LoKr
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
B="--steps 4 --warmup 2 --no-extras --no-cpu-baseline"
run() { tag=$1; shift; "$@" > gpurun_out/r2l_$tag.log 2>&1; echo "$tag rc=$? $(tail -1 gpurun_out/r2l_$tag.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],3), d['unit'], 'B', d['config'].get('per_gpu_batch'), round(d['ms_per_step'],1), 'ms; gemm', round(d.get('roofline',{}).get('achieved',0),1), 'TF; mem', d['config'].get('peak_mem_GiB'), d.get('allreduce_ms_exposed'))" 2>&1 | cut -c1-300)"; }
run rccl1 env AITK_BENCH_FORCE_PG=1 timeout 300 python bench.py $B
run fp8_r32 timeout 300 python bench.py --fp8-base --rank 32 $B
run dora timeout 300 python bench.py --network dora $B
run lokr timeout 300 python bench.py --network lokr $B
timeout 300 python tools/gpu_wan_bench.py > gpurun_out/r2l_wan.log 2>&1; echo "wan rc=$?"; tail -3 gpurun_out/r2l_wan.log | cut -c1-400
