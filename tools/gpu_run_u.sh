#!/bin/bash
# round-2 GPU pass U: validation of the tree (all GPU tests, smoke, default bench line)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r2u_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r2u_pytest.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2u_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r2u_smoke.log
timeout 900 python bench.py > gpurun_out/r2u_bench.log 2>&1; echo "bench rc=$?"; grep '^{' gpurun_out/r2u_bench.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],3), 'img/s', d['step_ms'], 'gemm', round(d['roofline']['achieved'],1), round(d['roofline']['frac'],3), 'sweep', {k: round(v['images_per_s'],2) for k,v in d['batch_sweep'].items()}, 'uncached', d.get('uncached_latents'), 'cmp', d['gpu_comparator'].get('value'), 'cpu', d['cpu_baseline']['value'])"
