#!/bin/bash
# split preprocess (geometry + colour) and colour/sort overlap: parity under each mode, then A/B timing on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for m in 1 2; do
  GSR_OPTIONS="color_overlap=$m" timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_reference_glue.py -m gpu -x -q -k "not variant" > gpurun_out/r2n_pytest_$m.log 2>&1
  echo "mode $m pytest rc=$?"; tail -2 gpurun_out/r2n_pytest_$m.log
done
for rep in 1 2; do
for m in 0 1 2; do
  timeout 300 python bench.py --no-other-configs --no-cpu-baseline --opt color_overlap=$m > gpurun_out/r2n_bench_${m}_$rep.log 2>&1
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r2n_bench_${m}_$rep.log") if l.startswith("{")][-1])
print("mode $m rep $rep:", d["value"], d["ms_per_step"], d["train_iters_per_s"], d["stage_ms"])
PY
done
done
