#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2d_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2d_pytest.log
tail -5 gpurun_out/r2d_pytest.log
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --train-steps 0 > gpurun_out/r2d_bench.log 2>&1
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r2d_bench.log") if l.startswith("{")][-1])
print("default:", d["value"], d["stage_ms"])
PY
BENCH_EXTRA="--train-steps 0" bash tools/gpu_kstats.sh 2>&1 | head -20
