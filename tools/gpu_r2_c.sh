#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2c_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2c_pytest.log
tail -5 gpurun_out/r2c_pytest.log
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r2c_bench.log 2>&1
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r2c_bench.log") if l.startswith("{")][-1])
print("default:", d["value"], d["train_iters_per_s"], d["train_iters_per_s_l1"], d["stage_ms"])
PY
timeout 600 python tools/gpu_lane_stats.py 0.012 300 > gpurun_out/r2c_lane.log 2>&1; tail -3 gpurun_out/r2c_lane.log
timeout 600 python tools/gpu_lane_stats.py 0.006 300 > gpurun_out/r2c_lane6.log 2>&1; tail -3 gpurun_out/r2c_lane6.log
bash tools/gpu_kstats.sh 2>&1 | head -24
bash tools/gpu_pmc_sq.sh > /dev/null 2>&1
python tools/pmc_sq_summary.py $(find gpurun_out/prof_sq gpurun_out/prof_sq2 -name "*counter_collection.csv") > gpurun_out/r2c_sq.csv 2>&1
head -12 gpurun_out/r2c_sq.csv
