#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --tb=short -x 2>&1 | tail -15
timeout 300 python tools/gpu_exp1.py depth_digit_bits=8 depth_digit_bits=11
