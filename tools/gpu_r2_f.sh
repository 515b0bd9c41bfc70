#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
# default build: new tests (reference glue, sparse adam vectorised)
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2f_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2f_pytest.log
tail -6 gpurun_out/r2f_pytest.log
# onesweep depth sort under the parity tests (bounded by timeout in case of a protocol bug)
GSR_OPTIONS=depth_sort_mode=1 timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2f_pytest_os.log 2>&1
echo "pytest_os rc=$?" >> gpurun_out/r2f_pytest_os.log
tail -6 gpurun_out/r2f_pytest_os.log
for mode in 0 1; do
GSR_OPTIONS=depth_sort_mode=$mode timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --train-steps 0 > gpurun_out/r2f_bench_os$mode.log 2>&1
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r2f_bench_os$mode.log") if l.startswith("{")][-1])
print("depth_sort_mode $mode:", d["value"], d["stage_ms"], d["roofline"]["frac"] if d["roofline"] else None)
PY
done
GSR_OPTIONS=depth_sort_mode=1 BENCH_EXTRA="--train-steps 0" bash tools/gpu_kstats.sh 2>&1 | head -22
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2f_bench_full.log 2>&1
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r2f_bench_full.log") if l.startswith("{")][-1])
print("full:", d["value"], d["train_iters_per_s"], d["train_iters_per_s_sparse_adam"], d["train_iters_per_s_l1"], d["forward_builds_ms"], d["blend_work"])
print(d["roofline"]); print(d["roofline_train"])
PY
