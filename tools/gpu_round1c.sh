#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
export OMP_NUM_THREADS=16
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu --tb=short 2>&1 | tail -60 > gpurun_out/c_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c_smoke.log 2>&1
tail -40 gpurun_out/c_pytest.log; tail -3 gpurun_out/c_smoke.log
