#!/bin/bash
# round-2 GPU pass T: software-pipelined dK/dV attention-backward kernel: parity + same-box A/B against the un-pipelined kernel and the experiment builds
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/gpu_attn_ab.py 2>&1 | grep -v amdgpu.ids | tail -16
