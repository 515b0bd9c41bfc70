#!/bin/bash
# A/B of library options on ONE box (box-to-box variance is larger than most single-kernel gains):
#   gpurun -- 'bash tools/gpu_ab.sh "depth_sort_mode=1" "depth_sort_mode=2"'            (options of the measurement build)
#   GSR_LIB=gaussian-splatting_amd/lib_ab/libgsr_hip.so is set automatically when a lib_ab build exists.
# Each configuration runs bench.py twice, interleaved; one line per run: Mpix/s, ms/frame, train it/s, stage table.
#   LIBS="lib lib_prev" bash tools/gpu_ab.sh          compares LIBRARIES instead (tools/build_prev_lib.sh builds lib_prev from
#                                                       an earlier revision); three interleaved runs each
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -n "$LIBS" ]; then
  for rep in 1 2 3; do
    for lib in $LIBS; do
      GSR_LIB="$PWD/gaussian-splatting_amd/$lib/libgsr_hip.so" timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-in-flight --no-pmc --densify-iters 0 > gpurun_out/ab_${lib}_$rep.log 2>&1
      python - "$lib" "$rep" "gpurun_out/ab_${lib}_$rep.log" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
print(f"{sys.argv[1]} rep {sys.argv[2]}:", d["value"], d["ms_per_step"], d["train_iters_per_s"], d["stage_ms"])
PY
    done
  done
  exit 0
fi
[ -f gaussian-splatting_amd/lib_ab/libgsr_hip.so ] && export GSR_LIB="$PWD/gaussian-splatting_amd/lib_ab/libgsr_hip.so"
for rep in 1 2; do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    opts=""; for o in ${cfg//,/ }; do opts="$opts --opt $o"; done
    timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-in-flight --no-pmc --densify-iters 0 $opts > gpurun_out/ab_${i}_$rep.log 2>&1
    python - "$cfg" "$rep" "gpurun_out/ab_${i}_$rep.log" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
print(f"{sys.argv[1]} rep {sys.argv[2]}:", d["value"], d["ms_per_step"], d["train_iters_per_s"], d["stage_ms"])
PY
  done
done
