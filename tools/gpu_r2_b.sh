#!/bin/bash
# round 2, GPU call B: quad backward (default lib) tests + A/B against the tile kernel (lib_ab), kernel-level profile
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2b_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2b_pytest.log
tail -12 gpurun_out/r2b_pytest.log
GSR_LIB=$PWD/gaussian-splatting_amd/lib_ab/libgsr_hip.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2b_pytest_ab.log 2>&1
echo "pytest_ab rc=$?" >> gpurun_out/r2b_pytest_ab.log
tail -5 gpurun_out/r2b_pytest_ab.log
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r2b_bench.log 2>&1
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r2b_bench.log") if l.startswith("{")][-1])
print("default:", d["value"], d["train_iters_per_s"], d["train_iters_per_s_l1"], d["stage_ms"])
PY
GSR_LIB=$PWD/gaussian-splatting_amd/lib_ab/libgsr_hip.so timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --bwd-variant 4 > gpurun_out/r2b_bench_v4.log 2>&1
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r2b_bench_v4.log") if l.startswith("{")][-1])
print("bwd v4:", d["value"], d["train_iters_per_s"], d["train_iters_per_s_l1"], d["stage_ms"])
PY
bash tools/gpu_kstats.sh 2>&1 | tail -45
