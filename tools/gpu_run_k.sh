#!/bin/bash
# round-2 GPU pass K: attention kernels that skip the zero-padded part of narrow heads (AitkAttnArgs.Dv) — suite + UNet benches
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r2k_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r2k_pytest.log | cut -c1-300
for m in sdxl sd15; do
  timeout 300 python bench.py --model $m --steps 8 --warmup 3 > gpurun_out/r2k_bench_$m.log 2>&1
  echo "$m rc=$? $(tail -1 gpurun_out/r2k_bench_$m.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), 'img/s B', d['config']['per_gpu_batch'], d['launch_modes'], '; gemm+conv', round(d['roofline']['achieved'],1), 'TF')" 2>&1)"
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2k_prof_sdxl" -o r2k_sdxl -- python "$GRAFT_REPO_ROOT/bench.py" --model sdxl --steps 4 --warmup 2 --no-graph > "$GRAFT_REPO_ROOT/gpurun_out/r2k_prof_sdxl.log" 2>&1)
echo "prof sdxl rc=$?"
