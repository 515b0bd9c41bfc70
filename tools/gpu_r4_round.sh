#!/bin/bash
# bins sweep + sharded tests, same-box A/B against lib_prev, then the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
bash tools/gpu_r4_quick.sh
bash tools/gpu_bench_default.sh
