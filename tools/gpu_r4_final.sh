#!/bin/bash
# round 4, final measurement call: profiles (kernel trace, PMC passes, default bench line with the blend timeline) + the GPU test suite + fuzz
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
ROUND_TAG=r04 SKIP_AB=1 SKIP_TRAIN=1 SKIP_MODEL=1 bash tools/gpu_round_bundle.sh
echo "== fuzz render"; FUZZ_SECONDS=${FUZZ_SECONDS:-150} timeout 300 python tools/gpu_fuzz_render.py 400 31 2>&1 | tail -1 | cut -c1-3000
echo "== fuzz bins"; timeout 200 python tools/gpu_fuzz_bins.py 300 7 2>&1 | tail -1 | cut -c1-600
