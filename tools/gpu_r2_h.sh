#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
GSR_OPTIONS=render_bwd_variant=6 timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_reference_glue.py -m gpu -x -q > gpurun_out/r2h_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2h_pytest.log
tail -4 gpurun_out/r2h_pytest.log
for opt in render_bwd_variant=0 render_bwd_variant=6; do
GSR_OPTIONS=$opt timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r2h_bench_$opt.log 2>&1
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r2h_bench_$opt.log") if l.startswith("{")][-1])
print("$opt:", d["value"], d["train_iters_per_s"], d["train_iters_per_s_sparse_adam"], d["train_iters_per_s_l1"], d["stage_ms"], d["blend_work"])
PY
done
GSR_OPTIONS=render_bwd_variant=6 BENCH_EXTRA="--train-steps 8" bash tools/gpu_kstats.sh 2>&1 | head -24
