#!/bin/bash
# experiment: forward frames alternating between HIP streams (independent views, e.g. render.py's loop) + the two-thread lease test
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "concurrent_forward or debug_mode" -x 2>&1 | tail -5
for s in 1 2 3 1 2; do
  timeout 300 python bench.py --streams $s --no-other-configs --no-cpu-baseline --no-full-loop --densify-iters 0 --train-steps 0 --steps 100 --warmup 20 > gpurun_out/streams_$s.log 2>&1
  python - $s gpurun_out/streams_$s.log <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    print("streams", sys.argv[1], d["value"], d["ms_per_step"], d["config"].get("frame_streams"))
except Exception as e:
    print("failed", e); print(open(sys.argv[2]).read()[-1200:])
PY
done
