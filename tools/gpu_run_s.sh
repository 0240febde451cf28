#!/bin/bash
# round-2 GPU pass S: conv-LoRA kernel pieces + train step with conv adapters; regression of the conv path after the K-slab / split-slab change
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv_lora.py "tests/test_gpu_unet.py::test_conv_forward_and_data_gradient" tests/test_gpu_wan_vae.py -m gpu -q -p no:cacheprovider -s 2>&1 | grep -v "amdgpu.ids\|MIOpen" | tail -40 > gpurun_out/r2s_pytest.log
tail -30 gpurun_out/r2s_pytest.log | cut -c1-400
