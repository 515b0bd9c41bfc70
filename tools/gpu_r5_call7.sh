#!/bin/bash
# Round 5, seventh GPU call: the two ranking kernels of the tile sort (emit_scatter, bucket_scatter: four waves per SIMD, half of their wave-cycles waiting)
# at five / six waves per SIMD (amdgpu_waves_per_eu: 96 / 80 VGPRs with 6-25 of them spilled) -- same-box interleaved A/B, forward only.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d.get("train_iters_per_s"), d["stage_ms"], (d.get("gpu_event_ms") or {}).get("forward"))
PY
}
for rep in 1 2 3; do
  for lib in lib lib_occ5 lib_occ6; do
    GSR_LIB="$PWD/gaussian-splatting_amd/$lib/libgsr_hip.so" timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-in-flight --densify-iters 0 --train-steps 0 > gpurun_out/r5c7_${lib}_$rep.log 2>&1
    line "$lib rep $rep:" gpurun_out/r5c7_${lib}_$rep.log
  done
done | tee gpurun_out/r5c7_ab_occ.log
GSR_LIB="$PWD/gaussian-splatting_amd/lib_occ5/libgsr_hip.so" timeout 300 python -m pytest tests/test_gpu_bins_sweep.py -q -m gpu -x 2>&1 | tail -2
