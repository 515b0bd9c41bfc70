#!/bin/bash
# round 4, session 2, experiment call 2: whole-operator fuzz (bars with magnitudes), per-wave trace of the blend launches, scheduling model
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== parity subset"; timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "forward_parity or backward_parity or bit_reproducible or snug" 2>&1 | tail -3
echo "== wave trace"; timeout 300 python tools/gpu_wave_trace.py 2>&1 | tail -1 | cut -c1-6000
echo "== fuzz render 1"; timeout 420 python tools/gpu_fuzz_render.py ${FUZZ_FRAMES:-300} 11 2>&1 | tail -1 | cut -c1-5000
echo "== tail model"; timeout 300 python tools/gpu_tail_model.py 2>&1 | tail -1 | cut -c1-4000
