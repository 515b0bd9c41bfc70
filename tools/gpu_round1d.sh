#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu --tb=short 2>&1 | tail -40 > gpurun_out/d_pytest.log
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --compare-torch-adam > gpurun_out/d_bench.log 2>&1
tail -30 gpurun_out/d_pytest.log
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/d_bench.log").read().strip().splitlines()[-1]); print(d["value"], "Mpix/s", d["ms_per_step"], "ms; train", d["train_iters_per_s"], "it/s (ssim)", d.get("train_iters_per_s_l1"), "l1", d.get("train_iters_per_s_l1_torch_adam"), "l1+torchadam", d["stage_ms"])
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/d_bench.log").read()[-3000:])
PY
