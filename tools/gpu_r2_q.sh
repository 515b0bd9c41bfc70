#!/bin/bash
# occupancy experiment: per-Gaussian kernels at 3 waves/SIMD (spilling) vs 2 (lib_x vs lib_ab), same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
for lib in lib_ab lib_x; do
  GSR_LIB=gaussian-splatting_amd/$lib/libgsr_hip.so timeout 300 python bench.py --no-other-configs --no-cpu-baseline > gpurun_out/r2q_bench_${lib}_$rep.log 2>&1
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r2q_bench_${lib}_$rep.log") if l.startswith("{")][-1])
print("$lib rep $rep:", d["value"], d["ms_per_step"], d["train_iters_per_s"], d["stage_ms"])
PY
done
done
