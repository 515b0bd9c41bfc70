#!/bin/bash
# fixed-capacity exchange: kernel / piece / policy tests + the multi-rank bench path over gloo on one GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_next_rows.py -q -x -k "fixed or sharded or multi_rank or route" --durations=5 2>&1 | tail -25
