#!/bin/bash
# round 4, session 2, experiment call 4: phase split of the blend waves' lifetime, whole-operator fuzz with a time budget
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== wave trace"; timeout 300 python tools/gpu_wave_trace.py 2 2>&1 | tail -1 | cut -c1-5000
echo "== fuzz render"; FUZZ_SECONDS=${FUZZ_SECONDS:-200} timeout 400 python tools/gpu_fuzz_render.py 400 21 2>&1 | tail -1 | cut -c1-5000
