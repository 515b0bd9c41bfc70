#!/bin/bash
# Round-5 bring-up of the candidates that round 4 left written, CPU-verified and UNMEASURED, in one gpurun call:
#   * lib_next  = the product sources with -DGSR_FWD_TL_DECAY=1 -DGSR_FWD_COMPACT=1 (forward blend: 24 -> 22 VALU and 11 -> 4 SALU per step; csrc/render_fwd.hip)
#   * lib_ab    = measurement build; option emit_scatter_mode=1 (level-1 scatter ranking row segments; csrc/ab/emit_scatter_segments.inc)
# Build both HERE first (no GPU needed), they travel with the snapshot:
#   GSR_OUT=lib_next GSR_EXTRA_FLAGS="-DGSR_FWD_TL_DECAY=1 -DGSR_FWD_COMPACT=1" python gaussian-splatting_amd/build.py ; GSR_AB=1 python gaussian-splatting_amd/build.py
#   gpurun --timeout 1500 -- 'bash tools/gpu_r5_candidates.sh'
# 1. parity + full-size suites on lib_next (bit-exact bins, image bars, n_contrib); 2. same-box interleaved A/B lib vs lib_next (3 x 2 bench runs);
# 3. the segment-ranking scatter's bring-up test, then its A/B inside lib_ab.  Adopt a candidate only if the STEP gains on this box (DESIGN 9 "Boxes").
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
NEXT="$PWD/gaussian-splatting_amd/lib_next/libgsr_hip.so"
if [ -f "$NEXT" ]; then
  echo "== parity suites on lib_next"
  GSR_LIB="$NEXT" timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_reference_glue.py -q -m gpu --tb=short -x 2>&1 | tail -15 | tee gpurun_out/r5_next_pytest.log
  echo "== A/B lib vs lib_next"
  LIBS="lib lib_next" bash tools/gpu_ab.sh | tee gpurun_out/r5_ab_next.log
else
  echo "lib_next not built (see the header)"
fi
if [ -f gaussian-splatting_amd/lib_ab/libgsr_hip.so ]; then
  echo "== segment-ranking scatter: bring-up test"
  GSR_LIB="$PWD/gaussian-splatting_amd/lib_ab/libgsr_hip.so" timeout 600 python -m pytest tests/test_gpu_bins_sweep.py -q -m gpu --tb=short -k "segments" 2>&1 | tail -8 | tee gpurun_out/r5_segments_pytest.log
  echo "== A/B emit_scatter_mode"
  bash tools/gpu_ab.sh "emit_scatter_mode=0" "emit_scatter_mode=1" | tee gpurun_out/r5_ab_segments.log
fi
