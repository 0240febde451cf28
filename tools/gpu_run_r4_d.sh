#!/bin/bash
# round 4, call D: conv mode of the 8-phase kernel (kernel tests, the graphs that use it, same-box A/B on the VAE encoder and the SDXL step) + outlier-statistics parity
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_conv8.py -x -q -m gpu -s > gpurun_out/r04_pytest_conv8.log 2>&1; tail -12 gpurun_out/r04_pytest_conv8.log
timeout 900 python -m pytest tests/test_gpu_vae.py tests/test_gpu_unet.py tests/test_gpu_conv_lora.py -x -q -m gpu > gpurun_out/r04_pytest_d.log 2>&1; tail -4 gpurun_out/r04_pytest_d.log
AITK_CONV8=0 AITK_VAE_BENCH_OUT=gpurun_out/r04_vae_bench_conv8_off.json timeout 300 python tools/gpu_vae_bench.py 2>/dev/null | tail -1
AITK_VAE_BENCH_OUT=gpurun_out/r04_vae_bench_conv8_on.json timeout 300 python tools/gpu_vae_bench.py 2>/dev/null | tail -1
AITK_CONV8=0 timeout 400 python bench.py --model sdxl --steps 6 --warmup 2 --no-graph > gpurun_out/r04_bench_sdxl_conv8_off.json 2>/dev/null
timeout 400 python bench.py --model sdxl --steps 6 --warmup 2 --no-graph > gpurun_out/r04_bench_sdxl_conv8_on.json 2>/dev/null
python - <<'PY'
import json
for t in ("off", "on"):
    try:
        d = json.loads(open(f"gpurun_out/r04_bench_sdxl_conv8_{t}.json").read().strip().splitlines()[-1])
        print("sdxl conv8", t, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["gemm_conv_ms_per_step"])
    except Exception as e:
        print(t, "ERR", e)
PY
timeout 900 python tools/gpu_outlier_parity.py --full > gpurun_out/r04_outlier_parity.log 2>&1; grep OUTLIER gpurun_out/r04_outlier_parity.log; tail -3 gpurun_out/r04_outlier_parity.log
