#!/bin/bash
# split-SH loader A/B (lib_ab = branch-free, lib_x = conditional loads): per-kernel durations of the sparse leg, same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
for lib in lib_ab lib_x; do
  GSR_LIB="$GRAFT_REPO_ROOT/gaussian-splatting_amd/$lib/libgsr_hip.so" bash tools/gpu_kstats.sh 2>&1 | grep -E "preprocess_|bucket_|emit_|ssim" | sed "s/^/$lib rep $rep: /"
done
done
