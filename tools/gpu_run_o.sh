#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f ai-toolkit_amd/libaitk_abl_attn_b128.so ai-toolkit_amd/libaitk_abl_attn_noexp.so ai-toolkit_amd/libaitk_abl_attn_nomfma.so ai-toolkit_amd/libaitk_abl_attn_nolds.so
timeout 600 python tools/gpu_attn_ab.py 2>&1 | grep -v amdgpu.ids | tail -6
AITK_LIB_PATH=$GRAFT_REPO_ROOT/ai-toolkit_amd/libaitk_abl_attn_pipe.so timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k attn 2>&1 | tail -3
