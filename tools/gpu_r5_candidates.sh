#!/bin/bash
# Round-5 bring-up of the candidates that round 4 left written, CPU-verified and UNMEASURED, in one gpurun call:
#   * lib_next_fwd = the product sources with -DGSR_FWD_TL_DECAY=1 -DGSR_FWD_COMPACT=1 (forward blend: 24 -> 22 VALU and 11 -> 4 SALU per step; csrc/render_fwd.hip)
#   * lib_next_bins = the product sources with -DGSR_MATCH_BITOP3=1 (digit matching of ds_scatter / ds_segsort / emit_scatter / bucket_scatter and the LSD fallback's rs_scatter: 8 -> 4 VALU per bit; csrc/gsr_wave.h)
#   * lib_next_bwd = the product sources with -DGSR_BWD_DPP_FUSE=1 (blend backward: every cross-lane add one v_add_f32_dpp, 103 -> 95 VALU per step; csrc/render_bwd.hip)
#   * lib_ab    = measurement build; option emit_scatter_mode=1 (level-1 scatter ranking row segments; csrc/ab/emit_scatter_segments.inc)
# CPU pins of the candidate forms first:  GSR_TEST_CANDIDATES=1 python -m pytest tests -q -m "not gpu" -k "candidate or bitop3"
# Build both HERE first (no GPU needed), they travel with the snapshot:
#   GSR_OUT=lib_next_fwd GSR_EXTRA_FLAGS="-DGSR_FWD_TL_DECAY=1 -DGSR_FWD_COMPACT=1" python gaussian-splatting_amd/build.py
#   GSR_OUT=lib_next_bins GSR_EXTRA_FLAGS="-DGSR_MATCH_BITOP3=1" python gaussian-splatting_amd/build.py
#   GSR_OUT=lib_next_bwd GSR_EXTRA_FLAGS="-DGSR_BWD_DPP_FUSE=1" python gaussian-splatting_amd/build.py ; GSR_AB=1 python gaussian-splatting_amd/build.py
#   gpurun --timeout 2400 -- 'bash tools/gpu_r5_candidates.sh'
# 1. parity + full-size suites on both candidate libraries (bit-exact bins, image bars, n_contrib, gradients); 2. same-box interleaved A/B of the three libraries
# (3 x 4 bench runs: the forward candidate shows in ms per frame AND in the train legs, the backward candidate in the train legs only);
# 3. the segment-ranking scatter's bring-up test, then its A/B inside lib_ab.  Adopt a candidate only if the STEP gains on this box (DESIGN 9 "Boxes").
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
LIBS_AB="lib"
for cand in lib_next_fwd lib_next_bins lib_next_bwd; do
  NEXT="$PWD/gaussian-splatting_amd/$cand/libgsr_hip.so"
  if [ -f "$NEXT" ]; then
    echo "== parity suites on $cand"
    GSR_LIB="$NEXT" timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_bins_sweep.py tests/test_gpu_reference_glue.py -q -m gpu --tb=short -x 2>&1 | tail -15 | tee gpurun_out/r5_${cand}_pytest.log
    LIBS_AB="$LIBS_AB $cand"
  else
    echo "$cand not built (see the header)"
  fi
done
echo "== A/B $LIBS_AB"
LIBS="$LIBS_AB" bash tools/gpu_ab.sh | tee gpurun_out/r5_ab_next.log
if [ -f gaussian-splatting_amd/lib_ab/libgsr_hip.so ]; then
  echo "== segment-ranking scatter: bring-up test"
  GSR_LIB="$PWD/gaussian-splatting_amd/lib_ab/libgsr_hip.so" timeout 600 python -m pytest tests/test_gpu_bins_sweep.py -q -m gpu --tb=short -k "segments" 2>&1 | tail -8 | tee gpurun_out/r5_segments_pytest.log
  echo "== A/B emit_scatter_mode"
  bash tools/gpu_ab.sh "emit_scatter_mode=0" "emit_scatter_mode=1" | tee gpurun_out/r5_ab_segments.log
fi
