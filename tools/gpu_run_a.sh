#!/bin/bash
# round-2 GPU pass A: full GPU suite, default bench (with extras), kernel-trace stats of a short bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s --durations=15 -p no:cacheprovider > gpurun_out/r2a_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2a_pytest.log
tail -5 gpurun_out/r2a_pytest.log
timeout 600 python bench.py > gpurun_out/r2a_bench.log 2>&1
echo "bench rc=$?"; tail -c 3000 gpurun_out/r2a_bench.log
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2a_prof" -o r2a -- python "$GRAFT_REPO_ROOT/bench.py" --steps 4 --warmup 2 --no-extras --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/r2a_prof.log" 2>&1)
echo "prof rc=$?"
find gpurun_out/r2a_prof -name "*stats*" | head
