#!/bin/bash
# round-2 GPU pass W: last validation of the tree (all GPU tests + smoke) after the ABI 3 rebuild
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r2w_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r2w_pytest.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2w_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r2w_smoke.log
