#!/bin/bash
# round-2 GPU pass R: Wan2.1 video-VAE kernels / encoder graph parity, the 2-D conv path after the conv_t3d change, full-size clip timing + kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_wan_vae.py tests/test_gpu_vae.py -m gpu -q -p no:cacheprovider -s 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r2r_pytest.log
tail -8 gpurun_out/r2r_pytest.log
timeout 300 python tools/gpu_wan_vae_bench.py 2>&1 | grep -v amdgpu.ids | tail -3
(cd /tmp && AITK_WANVAE_SHAPE=49,512,512 timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2r_wanvae_prof" -o wv --output-format csv -- python "$GRAFT_REPO_ROOT/tools/gpu_wan_vae_bench.py" > "$GRAFT_REPO_ROOT/gpurun_out/r2r_wanvae_prof.log" 2>&1)
echo "prof rc=$?"; find gpurun_out/r2r_wanvae_prof -name "*kernel_stats.csv" | head -2
f=$(find gpurun_out/r2r_wanvae_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200
