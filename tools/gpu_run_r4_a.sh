#!/bin/bash
# round 4, call A: K-concatenated dgrad A/B on the headline step + the GPU tests it touches + recompute_gelu tests
mkdir -p gpurun_out
export TMPDIR=/tmp
AITK_CONCAT_DGRAD=0 AITK_GEMM_CENSUS=gpurun_out/r04_census_concat0.json timeout 400 python bench.py --steps 8 --warmup 3 --no-extras --no-cpu-baseline --no-dvfs > gpurun_out/r04_bench_concat0.json 2> gpurun_out/r04_bench_concat0.err
AITK_GEMM_CENSUS=gpurun_out/r04_census_concat1.json timeout 400 python bench.py --steps 8 --warmup 3 --no-extras --no-cpu-baseline --no-dvfs > gpurun_out/r04_bench_concat1.json 2> gpurun_out/r04_bench_concat1.err
timeout 900 python -m pytest tests/test_gpu_recompute_gelu.py tests/test_gpu_e2e.py tests/test_gpu_grouped_gemm.py tests/test_gpu_fullsize.py tests/test_gpu_wan.py tests/test_gpu_unet.py -x -q -m gpu > gpurun_out/r04_pytest_a.log 2>&1
tail -5 gpurun_out/r04_pytest_a.log
python - <<'PY'
import json
for t in ("concat0", "concat1"):
    try:
        d = json.loads(open(f"gpurun_out/r04_bench_{t}.json").read().strip().splitlines()[-1])
        print(t, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["gemm_ms_per_step"], d["roofline"]["launches_per_step"])
    except Exception as e:
        print(t, "ERR", e)
PY
