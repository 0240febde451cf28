#!/bin/bash
# round-2 GPU pass J: validation of the final tree — full GPU suite, smoke, default bench line, kernel-trace stats (FLUX, SDXL)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -s --durations=10 -p no:cacheprovider > gpurun_out/r2j_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED" gpurun_out/r2j_pytest.log | cut -c1-300 | head
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2j_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r2j_smoke.log
timeout 900 python bench.py > gpurun_out/r2j_bench.log 2>&1; echo "bench rc=$?"; tail -c 4500 gpurun_out/r2j_bench.log
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2j_prof" -o r2j -- python "$GRAFT_REPO_ROOT/bench.py" --steps 4 --warmup 2 --no-extras --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/r2j_prof.log" 2>&1)
echo "prof rc=$?"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2j_prof_sdxl" -o r2j_sdxl -- python "$GRAFT_REPO_ROOT/bench.py" --model sdxl --steps 4 --warmup 2 --no-graph > "$GRAFT_REPO_ROOT/gpurun_out/r2j_prof_sdxl.log" 2>&1)
echo "prof sdxl rc=$?"; tail -c 600 gpurun_out/r2j_prof_sdxl.log
