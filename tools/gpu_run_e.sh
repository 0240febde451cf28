#!/bin/bash
# round-2 GPU pass E: full GPU suite on the final tree, default bench (incl. hipGraph leg), kernel-trace stats of FLUX and SDXL benches
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -s --durations=12 -p no:cacheprovider > gpurun_out/r2e_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|lokr loss|low-rank" gpurun_out/r2e_pytest.log | cut -c1-300 | head -20
timeout 700 python bench.py > gpurun_out/r2e_bench.log 2>&1; echo "bench rc=$?"; tail -c 3500 gpurun_out/r2e_bench.log
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2e_prof" -o r2e -- python "$GRAFT_REPO_ROOT/bench.py" --steps 4 --warmup 2 --no-extras --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/r2e_prof.log" 2>&1)
echo "prof rc=$?"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2e_prof_sdxl" -o r2e_sdxl -- python "$GRAFT_REPO_ROOT/bench.py" --model sdxl --steps 4 --warmup 2 --no-graph > "$GRAFT_REPO_ROOT/gpurun_out/r2e_prof_sdxl.log" 2>&1)
echo "prof sdxl rc=$?"
find gpurun_out/r2e_prof gpurun_out/r2e_prof_sdxl -name "*kernel_stats*" | head
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2e_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r2e_smoke.log
