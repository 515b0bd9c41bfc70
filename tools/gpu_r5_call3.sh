#!/bin/bash
# Round 5, third GPU call: the default bench line (does the reordered JSON parse, what is in its last 4 KB), the kernel table of the merged sources,
# and a same-box interleaved A/B of compile-time tuning builds: GSR_FWD_CHECK_EVERY (termination ballot of the forward walk every N steps; 8 was tuned on
# round 3's walk, the compacted walk's step is cheaper) and GSR_TS_ITEMS (instances per workgroup of the fused emission / tile sort).
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d.get("train_iters_per_s"), d["stage_ms"], (d.get("gpu_event_ms") or {}).get("forward"))
PY
}
echo "== A/B tuning builds (forward only)"
for rep in 1 2 3; do
  for lib in lib lib_chk4 lib_chk6 lib_chk12 lib_chk16 lib_ts2048; do
    [ -f gaussian-splatting_amd/$lib/libgsr_hip.so ] || continue
    GSR_LIB="$PWD/gaussian-splatting_amd/$lib/libgsr_hip.so" timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-in-flight --densify-iters 0 --train-steps 0 > gpurun_out/r5c3_${lib}_$rep.log 2>&1
    line "$lib rep $rep:" gpurun_out/r5c3_${lib}_$rep.log
  done
done | tee gpurun_out/r5c3_ab_tuning.log
echo "== kernel table"
bash tools/gpu_kstats.sh 2>&1 | tail -40 | tee gpurun_out/r5c3_kstats.log
echo "== default line"
( time timeout 900 python bench.py > gpurun_out/r5c3_bench_default.json 2> gpurun_out/r5c3_bench_default.err ) 2>&1 | tail -4
tail -c 4096 gpurun_out/r5c3_bench_default.json
tail -3 gpurun_out/r5c3_bench_default.err
