#!/bin/bash
# round 4, session 2, experiment call 1: whole-operator fuzz, tail model of the blend launches, option / occupancy sweeps
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
L="$PWD/gaussian-splatting_amd"
echo "== fuzz render"; timeout 420 python tools/gpu_fuzz_render.py ${FUZZ_FRAMES:-250} 1 2>&1 | tail -2 | cut -c1-3000
echo "== tail model"; timeout 240 python tools/gpu_tail_model.py 2>&1 | tail -1 | cut -c1-3000
echo "== sweep forward, default lib"; timeout 240 python tools/gpu_opt_sweep.py "" "preprocess_grid_cap=768" "preprocess_grid_cap=1536" "preprocess_grid_cap=2048" 2>&1 | grep '^{' | cut -c1-700
cp gpurun_out/opt_sweep.json gpurun_out/opt_sweep_lib.json
echo "== sweep forward, lib_occ3f"; GSR_LIB="$L/lib_occ3f/libgsr_hip.so" timeout 240 python tools/gpu_opt_sweep.py "" "preprocess_grid_cap=768" "preprocess_grid_cap=1536" 2>&1 | grep '^{' | cut -c1-700
cp gpurun_out/opt_sweep.json gpurun_out/opt_sweep_occ3f.json
echo "== sweep forward again, default lib"; timeout 240 python tools/gpu_opt_sweep.py "" "preprocess_grid_cap=768" 2>&1 | grep '^{' | cut -c1-700
echo "== sweep train, default lib"; timeout 240 python tools/gpu_opt_sweep.py --train --frames 60 "" "bwd_heavy_first=0" 2>&1 | grep '^{' | cut -c1-900
cp gpurun_out/opt_sweep_train.json gpurun_out/opt_sweep_train_lib.json
echo "== sweep train, lib_occ3fb"; GSR_LIB="$L/lib_occ3fb/libgsr_hip.so" timeout 240 python tools/gpu_opt_sweep.py --train --frames 60 "" "preprocess_grid_cap=768" 2>&1 | grep '^{' | cut -c1-900
cp gpurun_out/opt_sweep_train.json gpurun_out/opt_sweep_train_occ3fb.json
