#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2m_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2m_pytest.log
tail -4 gpurun_out/r2m_pytest.log
( time timeout 600 python bench.py ) > gpurun_out/r2m_bench.log 2>&1
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r2m_bench.log") if l.startswith("{")][-1])
print("default:", d["value"], d["train_iters_per_s"], d["train_iters_per_s_sparse_adam"], d["train_iters_per_s_l1"], d["stage_ms"])
print(d["other_configs_forward"]); print(d["roofline_train"]["achieved"], d["roofline_train"]["frac"], d["retimed"])
PY
tail -3 gpurun_out/r2m_bench.log | grep real
timeout 900 python tools/train_run.py > gpurun_out/r2m_train_ref.log 2>&1
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r2m_train_ref.log") if l.startswith("{")][-1])
print("train_run:", d["value"], d["final_P"], d["peak_device_memory_bytes"], [ (w["until_iter"], w["iters_per_s"], w["mem_allocated"]) for w in d["windows"]])
PY
