#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for it in 4096 8192; do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --train-steps 0 --sort-items $it > gpurun_out/s_$it.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/s_$it.log").read().strip().splitlines()[-1]); print($it, d["ms_per_step"], d["stage_ms"])
PY
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "forward_parity or full_size or 65536" 2>&1 | tail -3
