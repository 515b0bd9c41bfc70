#!/bin/bash
# batched loads everywhere (per-Gaussian kernels, SSIM staging, scan kernels, scatter's rectangle gather): parity, bench, kernel table
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2w_pytest.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/r2w_pytest.log
GSR_LIB=gaussian-splatting_amd/lib_ab/libgsr_hip.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r2w_pytest_ab.log 2>&1
echo "pytest AB rc=$?"; tail -1 gpurun_out/r2w_pytest_ab.log
for rep in 1 2; do
  timeout 300 python bench.py --no-other-configs --no-cpu-baseline > gpurun_out/r2w_bench_$rep.log 2>&1
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r2w_bench_$rep.log") if l.startswith("{")][-1])
print("rep $rep:", d["value"], d["ms_per_step"], "ssim", d["train_iters_per_s"], "sparse", d["train_iters_per_s_sparse_adam"], "l1", d["train_iters_per_s_l1"], d["stage_ms"])
PY
done
bash tools/gpu_kstats.sh 2>&1 | head -34
