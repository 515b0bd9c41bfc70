#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/gpu_diag.py > gpurun_out/b_diag.log 2>&1
for v in 0 1 2 3; do
  timeout 300 python bench.py --steps 30 --warmup 5 --variant $v --no-cpu-baseline --train-steps 0 > gpurun_out/b_bench_v$v.log 2>&1
done
cat gpurun_out/b_diag.log
for v in 0 1 2 3; do python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/b_bench_v$v.log").read().strip().splitlines()[-1]); print("variant $v", d["value"], "Mpix/s", d["ms_per_step"], "ms", d["stage_ms"])
except Exception as e:
    print("variant $v failed", e); print(open("gpurun_out/b_bench_v$v.log").read()[-2000:])
PY
done
