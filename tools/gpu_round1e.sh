#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --tb=short -x 2>&1 | tail -15 > gpurun_out/e_pytest.log
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/e_bench.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_e" -o r1e -- python "$OLDPWD/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --train-steps 10 > "$OLDPWD/gpurun_out/e_prof.log" 2>&1 )
tail -12 gpurun_out/e_pytest.log
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/e_bench.log").read().strip().splitlines()[-1]); print(d["value"], "Mpix/s", d["ms_per_step"], "ms; train", d["train_iters_per_s"], "it/s", d["train_ms_per_iter"], "ms", d["stage_ms"])
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/e_bench.log").read()[-3000:])
PY
find gpurun_out/prof_e -name "*kernel_stats*" | head; f=$(find gpurun_out/prof_e -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" | cut -c1-200
