#!/bin/bash
# bins sweep + sharded tests + same-box A/B against lib_prev (forward stage table)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bins_sweep.py tests/test_gpu_sharded.py -q -m gpu --tb=short 2>&1 | tail -40 > gpurun_out/r4_bins.log
echo "== bins sweep + sharded"; tail -25 gpurun_out/r4_bins.log
bash tools/gpu_ab_prev.sh
