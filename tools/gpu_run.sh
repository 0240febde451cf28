#!/bin/bash
# ONE runner for every gpurun call:  gpurun --timeout T -- 'bash tools/gpu_run.sh TAG task [task ...]'
# Outputs go to gpurun_out/<TAG>_*.  Tasks (each under its own timeout so a hang cannot take the box's whole budget):
#   tests:<pytest args>        pytest -m gpu on the given files / node ids / -k expression ("," separates words) -> <TAG>_pytest_<n>.log
#   alltests                   the whole -m gpu tier                                                      -> <TAG>_pytest_gpu_full.log
#   bench[:<args>]             python bench.py <args> ("," separates words)                              -> <TAG>_bench_<n>.json/.err
#   prof[:<args>]              rocprofv3 --kernel-trace --stats around bench.py --no-extras <args>        -> <TAG>_rocprof_kernel_stats.csv
#   pmc                        three --pmc passes (SQ activity, FETCH_SIZE, WRITE_SIZE) over tools/gpu_pmc_target.py -> <TAG>_pmc_summary.json
#   py:<script>[,<args>]       python <script> <args>                                                     -> <TAG>_<script>.log
#   env:<NAME>=<value>         export for the following tasks
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG="$1"; shift
n=0
for task in "$@"; do
  n=$((n+1))
  kind="${task%%:*}"; rest=""; [ "$task" != "$kind" ] && rest="${task#*:}"
  args="${rest//,/ }"
  case "$kind" in
    env) export "$rest"; echo "[run] export $rest";;
    tests) timeout 1500 python -m pytest $args -q -m gpu --durations=8 > "gpurun_out/${TAG}_pytest_$n.log" 2>&1; echo "[run] tests rc=$?"; tail -30 "gpurun_out/${TAG}_pytest_$n.log" | cut -c1-600;;
    alltests) timeout 2400 python -m pytest tests -q -m gpu --durations=15 > "gpurun_out/${TAG}_pytest_gpu_full.log" 2>&1; echo "[run] alltests rc=$?"; tail -25 "gpurun_out/${TAG}_pytest_gpu_full.log" | cut -c1-400;;
    bench) timeout 1200 python bench.py $args > "gpurun_out/${TAG}_bench_$n.json" 2> "gpurun_out/${TAG}_bench_$n.err"; echo "[run] bench rc=$?"; tail -2 "gpurun_out/${TAG}_bench_$n.err" | cut -c1-400
       python - "gpurun_out/${TAG}_bench_$n.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print("value", d.get("value"), "ms", d.get("ms_per_step"), "gemm", r.get("achieved"), r.get("frac"), "failed", d.get("failed_legs"), "loss", d.get("final_loss"))
    for k in ("batch_sweep", "secondary_configs"):
        if d.get(k): print(k, json.dumps(d[k])[:900])
    if d.get("parity"): print("parity", json.dumps(d["parity"])[:1800])
except Exception as e:
    print("no bench line:", e)
PY
       ;;
    prof) (cd /tmp && AITK_GEMM_CENSUS="$GRAFT_REPO_ROOT/gpurun_out/${TAG}_census.json" timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/_prof" -o b --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 2 --no-extras --no-cpu-baseline $args > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_profiled.log" 2>&1)
       echo "[run] prof rc=$?"; grep '^{' "gpurun_out/${TAG}_bench_profiled.log" | tail -1 | cut -c1-200
       f=$(find gpurun_out/_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "gpurun_out/${TAG}_rocprof_kernel_stats.csv" && head -22 "$f" | cut -c1-160
       rm -rf gpurun_out/_prof;;
    pmc) i=0
       for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
         t=p$i; i=$((i+1))
         (cd /tmp && AITK_PMC_M=32256 AITK_PMC_K2=48 AITK_PMC_F8=0 timeout 300 rocprofv3 --pmc $c -d "$GRAFT_REPO_ROOT/gpurun_out/_pmc/$t" -o $t --output-format csv -- python "$GRAFT_REPO_ROOT/tools/gpu_pmc_target.py" > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$t.log" 2>&1); echo "[run] pmc $t rc=$?"
       done
       python tools/pmc_round_summary.py gpurun_out/_pmc "gpurun_out/${TAG}_pmc_summary.json" 32256 48 | tail -40
       rm -rf gpurun_out/_pmc;;
    py) script="${rest%%,*}"; sargs=""; [ "$rest" != "$script" ] && sargs="${rest#*,}"; sargs="${sargs//,/ }"
       b=$(basename "$script" .py)
       timeout 1500 python "$script" $sargs > "gpurun_out/${TAG}_${b}_$n.log" 2>&1; echo "[run] py $script rc=$?"; grep -v "amdgpu.ids" "gpurun_out/${TAG}_${b}_$n.log" | tail -12 | cut -c1-2500;;
    *) echo "[run] unknown task $task";;
  esac
done
