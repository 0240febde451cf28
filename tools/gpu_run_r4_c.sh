#!/bin/bash
# round 4, call C: DP paths (bf16 transport kernels, 2-rank one-device bf16 step, 1-rank RCCL group through the N > 1 bench path, 2-rank rehearsal) + plug-in GPU tests
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dp_rccl.py tests/test_gpu_plugin.py -x -q -m gpu > gpurun_out/r04_pytest_c.log 2>&1
tail -5 gpurun_out/r04_pytest_c.log
AITK_BENCH_FORCE_PG=1 timeout 400 python bench.py --no-extras --no-cpu-baseline --steps 4 --warmup 2 --allreduce-dtype bf16 > gpurun_out/r04_bench_rccl_1rank_bf16.json 2> gpurun_out/r04_bench_rccl_1rank_bf16.err
tail -3 gpurun_out/r04_bench_rccl_1rank_bf16.err
AITK_BENCH_ONE_DEVICE=1 AITK_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --batch 1 --no-extras --no-cpu-baseline --no-dvfs --steps 3 --warmup 1 --allreduce-dtype bf16 > gpurun_out/r04_bench_rehearsal_2rank_one_device.json 2> gpurun_out/r04_bench_rehearsal_2rank.err
tail -3 gpurun_out/r04_bench_rehearsal_2rank.err
python - <<'PY'
import json
for f in ("r04_bench_rccl_1rank_bf16", "r04_bench_rehearsal_2rank_one_device"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["n_gpus"], json.dumps(d.get("rccl"))[:1200], d.get("allreduce_ms_exposed"), d.get("failed_legs"))
    except Exception as e:
        print(f, "ERR", e)
PY
