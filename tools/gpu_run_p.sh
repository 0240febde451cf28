#!/bin/bash
# round-2 GPU pass P: last validation of the tree + instruction-mix counters of the attention / GEMM kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r2p_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r2p_pytest.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2p_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r2p_smoke.log
timeout 900 python bench.py > gpurun_out/r2p_bench.log 2>&1; echo "bench rc=$?"; grep '^{' gpurun_out/r2p_bench.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],3), 'img/s', d['step_ms'], 'gemm', round(d['roofline']['achieved'],1), round(d['roofline']['frac'],3), 'graph', d.get('graph_replay'), 'cmp', d['gpu_comparator']['value'], 'cpu', d['cpu_baseline']['value'])"
timeout 400 python bench.py --model sd15 --steps 8 --warmup 3 > gpurun_out/r2p_bench_sd15.log 2>&1; echo "sd15 rc=$?"; grep '^{' gpurun_out/r2p_bench_sd15.log | tail -1 | cut -c1-400
timeout 400 python bench.py --model sdxl --steps 8 --warmup 3 > gpurun_out/r2p_bench_sdxl.log 2>&1; echo "sdxl rc=$?"; grep '^{' gpurun_out/r2p_bench_sdxl.log | tail -1 | cut -c1-300
c="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES"
(cd /tmp && AITK_PMC_M=32256 AITK_PMC_K2=48 timeout 300 rocprofv3 --pmc $c -d "$GRAFT_REPO_ROOT/gpurun_out/r2p_pmc/insts" -o insts --output-format csv -- python "$GRAFT_REPO_ROOT/tools/gpu_pmc_target.py" > "$GRAFT_REPO_ROOT/gpurun_out/r2p_pmc_insts.log" 2>&1)
echo "pmc insts rc=$?"; find gpurun_out/r2p_pmc -name "*counter_collection.csv" | head -2
