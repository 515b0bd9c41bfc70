#!/bin/bash
# the whole GPU test suite with per-test durations
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --tb=short --durations=15 2>&1 | tail -80 > gpurun_out/suite.log
tail -60 gpurun_out/suite.log
