#!/bin/bash
# Build the library of an EARLIER revision next to the current one, for same-box A/B runs (box-to-box variance on the GPU
# pool is larger than most single-kernel gains, so a change is only ever judged against its predecessor on ONE box):
#     bash tools/build_prev_lib.sh [git-rev]          # default HEAD  ->  gaussian-splatting_amd/lib_prev/libgsr_hip.so
#     gpurun -- 'LIBS="lib lib_prev" bash tools/gpu_ab.sh'
# lib_prev/ is git-ignored (*.so) and travels to the GPU box with the snapshot.
set -e
REV="${1:-HEAD}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
git -C "$ROOT" archive "$REV" gaussian-splatting_amd/csrc gaussian-splatting_amd/build.py include | tar -x -C "$TMP"
( cd "$TMP/gaussian-splatting_amd" && python build.py > "$TMP/build.log" 2>&1 ) || { tail -20 "$TMP/build.log"; exit 1; }
mkdir -p "$ROOT/gaussian-splatting_amd/lib_prev"
cp "$TMP/gaussian-splatting_amd/lib/libgsr_hip.so" "$ROOT/gaussian-splatting_amd/lib_prev/libgsr_hip.so"
echo "built $(git -C "$ROOT" rev-parse --short "$REV") -> gaussian-splatting_amd/lib_prev/libgsr_hip.so"
