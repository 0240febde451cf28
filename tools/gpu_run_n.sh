#!/bin/bash
# round-2 GPU pass N: where does the attention time go?  Timing-only ablation libraries (tools/build_abl_attn.py), per-kernel times by rocprofv3
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python tools/gpu_attn_ab.py 2>&1 | grep -v amdgpu.ids | tail -12
cd /tmp
for tag in product b128 noexp nomfma nolds; do
  lib="$GRAFT_REPO_ROOT/ai-toolkit_amd/libaitk_abl_attn_$tag.so"
  [ "$tag" = product ] && lib="$GRAFT_REPO_ROOT/ai-toolkit_amd/libaitk_mi355.so"
  AITK_LIB_PATH=$lib timeout 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2n_prof_$tag" -o r2n_$tag -- python "$GRAFT_REPO_ROOT/tools/gpu_attn_ab.py" child > "$GRAFT_REPO_ROOT/gpurun_out/r2n_prof_$tag.log" 2>&1
  echo "prof $tag rc=$?"
done
