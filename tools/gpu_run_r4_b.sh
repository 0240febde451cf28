#!/bin/bash
# round 4, call B: the dQ-atomics probe + the default bench line with the new legs (parity.full_depth, per-leg handlers)
mkdir -p gpurun_out
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_dq tools/probes/atomic_dq.hip 2>/dev/null && timeout 120 /tmp/atomic_dq > gpurun_out/r04_probe_atomic_dq.txt 2>&1
cat gpurun_out/r04_probe_atomic_dq.txt
timeout 900 python bench.py > gpurun_out/r04_bench_default_a.json 2> gpurun_out/r04_bench_default_a.err
tail -c 3000 gpurun_out/r04_bench_default_a.err | tail -5
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_bench_default_a.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "gemm", d["roofline"].get("achieved"), d["roofline"].get("frac"))
print("failed", d.get("failed_legs"))
print("parity", json.dumps(d.get("parity"), indent=0)[:1500])
print({k: (v if not isinstance(v, dict) else "...") for k, v in d.items() if k not in ("parity",)}.keys())
PY
