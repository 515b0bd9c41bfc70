#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python bench.py ) > gpurun_out/bench_default.log 2>&1
tail -c 3000 gpurun_out/bench_default.log | tail -5 | cut -c1-1500
