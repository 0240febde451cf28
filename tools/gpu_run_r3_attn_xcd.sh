#!/bin/bash
# XCD-aware workgroup order in the attention kernels: correctness, same-box A/B is not possible from one tree -> timing + FETCH_SIZE vs round-3 records
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest -q tests/test_gpu_kernels.py tests/test_gpu_wan.py tests/test_gpu_r3_kernels.py tests/test_gpu_unet.py -m gpu -k "attn or attention or unet or sdxl or sd15" 2>&1 | tail -4
timeout 300 python tools/gpu_attn_ab.py 2>&1 | grep -v amdgpu.ids
for v in 1 2; do timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'])"; done
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$GRAFT_REPO_ROOT/gpurun_out/r3_pmc_xcd" -o x --output-format csv -- python "$GRAFT_REPO_ROOT/tools/gpu_pmc_target.py" > "$GRAFT_REPO_ROOT/gpurun_out/r3_pmc_xcd.log" 2>&1)
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r3_pmc_xcd/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for fn in f:
    for r in csv.DictReader(open(fn)):
        if r.get("Counter_Name") == "FETCH_SIZE":
            acc[r["Kernel_Name"].split("(")[0][:40]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("FETCH_SIZE", k, len(v), "launches, mean KB", sum(v) / len(v), "-> corrected MB", sum(v) / len(v) * 2 / 1024)
PY
find gpurun_out/r3_pmc_xcd -name "*.csv" -size +2M -delete
