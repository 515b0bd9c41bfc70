#!/bin/bash
# first-pass histogram inside preprocess + one-round rs_scan + hoisted scatter loads: parity, then A/B timing on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2p_pytest.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/r2p_pytest.log
GSR_LIB=gaussian-splatting_amd/lib_ab/libgsr_hip.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r2p_pytest_ab.log 2>&1
echo "pytest AB rc=$?"; tail -2 gpurun_out/r2p_pytest_ab.log
for rep in 1 2; do
for m in 1 0; do
  timeout 300 python bench.py --no-other-configs --no-cpu-baseline --opt first_hist_in_preprocess=$m > gpurun_out/r2p_bench_${m}_$rep.log 2>&1
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r2p_bench_${m}_$rep.log") if l.startswith("{")][-1])
print("first_hist $m rep $rep:", d["value"], d["ms_per_step"], d["train_iters_per_s"], d["stage_ms"])
PY
done
done
