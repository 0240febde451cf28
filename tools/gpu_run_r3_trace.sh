#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
(cd /tmp && timeout 500 rocprofv3 --kernel-trace -d /tmp/r3_trace -o t --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-roofline --no-dvfs > /tmp/r3_trace.log 2>&1)
echo "trace rc=$?"
f=$(find /tmp/r3_trace -name "*kernel_trace.csv" | head -1)
ls -la "$f"; head -1 "$f"
python tools/trace_neighbors.py "$f" | tee gpurun_out/r03_trace_neighbors.txt | head -70
