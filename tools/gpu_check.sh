#!/bin/bash
# new rows first (split SH, sparse Adam, kNN, 2-rank bench), then the parity suite, then the bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_next_rows.py -q -m gpu --tb=short -x 2>&1 | tail -60 > gpurun_out/c_next.log
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu --tb=short 2>&1 | tail -40 > gpurun_out/c_pytest.log
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-full-loop > gpurun_out/c_bench.log 2>&1
timeout 300 python tools/gpu_knn_time.py > gpurun_out/c_knn.log 2>&1
echo "== next rows"; tail -30 gpurun_out/c_next.log
echo "== parity"; tail -15 gpurun_out/c_pytest.log
echo "== knn"; tail -8 gpurun_out/c_knn.log
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/c_bench.log").read().strip().splitlines()[-1]); print(d["value"], "Mpix/s", d["ms_per_step"], "ms; train", d["train_iters_per_s"], "it/s (ssim)", d.get("train_iters_per_s_l1"), "l1", d["stage_ms"])
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/c_bench.log").read()[-3000:])
PY
