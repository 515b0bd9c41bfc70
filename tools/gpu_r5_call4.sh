#!/bin/bash
# Round 5, fourth GPU call: the robust key range of the depth bucket sort against the true range -- (1) same-box interleaved A/B on the bench frame
# (no outliers: nothing may get slower), (2) the bench frame with 24 far outliers (the cliff the robust range removes).
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d.get("train_iters_per_s"), d["stage_ms"], (d.get("gpu_event_ms") or {}).get("forward"))
PY
}
for rep in 1 2 3 4; do
  for lib in lib lib_truerange; do
    GSR_LIB="$PWD/gaussian-splatting_amd/$lib/libgsr_hip.so" timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-in-flight --densify-iters 0 --train-steps 0 > gpurun_out/r5c4_${lib}_$rep.log 2>&1
    line "$lib rep $rep:" gpurun_out/r5c4_${lib}_$rep.log
  done
done | tee gpurun_out/r5c4_ab_range.log
for lib in lib lib_truerange lib lib_truerange; do
  GSR_LIB="$PWD/gaussian-splatting_amd/$lib/libgsr_hip.so" timeout 300 python tools/gpu_outlier_probe.py 2>&1 | tail -1
done | tee gpurun_out/r5c4_outliers.log
timeout 600 python -m pytest tests/test_gpu_bins_sweep.py -q -m gpu -x 2>&1 | tail -3
