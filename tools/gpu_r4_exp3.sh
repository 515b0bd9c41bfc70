#!/bin/bash
# round 4, session 2, experiment call 3: launch order of the blend backward (tiles by sum / by heaviest half / half tiles), fuzz with conditioning-aware bars
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== sweep train"; timeout 300 python tools/gpu_opt_sweep.py --train --frames 60 --reps 4 "bwd_heavy_first=1" "bwd_heavy_first=2" "bwd_heavy_first=3" "bwd_heavy_first=0" 2>&1 | grep '^{' | cut -c1-900
echo "== wave trace"; timeout 400 python tools/gpu_wave_trace.py 3,2,1 2>&1 | tail -1 | cut -c1-7000
echo "== backward parity"; timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -q -m gpu -x -k "backward or band or sharded or bit_reproducible" 2>&1 | tail -3
echo "== fuzz render"; timeout 500 python tools/gpu_fuzz_render.py ${FUZZ_FRAMES:-300} 21 2>&1 | tail -1 | cut -c1-4000
