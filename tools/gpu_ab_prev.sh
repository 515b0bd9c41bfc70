#!/bin/bash
# same-box A/B of the current library against gaussian-splatting_amd/lib_prev (tools/build_prev_lib.sh <rev>), interleaved runs,
# forward only + stage table.  The older library may have an older ABI: the symbols both share are bound.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp GSR_ALLOW_ABI_MISMATCH=1
for rep in 1 2 3; do
  for lib in lib lib_prev; do
    GSR_LIB="$PWD/gaussian-splatting_amd/$lib/libgsr_hip.so" timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-in-flight --densify-iters 0 --train-steps ${TRAIN_STEPS:-0} $BENCH_EXTRA > gpurun_out/abp_${lib}_$rep.log 2>&1
    python - "$lib" "$rep" "gpurun_out/abp_${lib}_$rep.log" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
    print(f"{sys.argv[1]:8s} rep {sys.argv[2]}:", d["value"], d["ms_per_step"], d.get("train_iters_per_s"), {k: v for k, v in d["stage_ms"].items()})
except Exception as e:
    print("bench failed", e); print(open(sys.argv[3]).read()[-1500:])
PY
  done
done
