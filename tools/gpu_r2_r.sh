#!/bin/bash
# LDS-DMA staging of the SH block in the per-Gaussian kernels: parity (product lib, default sh_dma=3), then A/B timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_reference_glue.py tests/test_gpu_next_rows.py -m gpu -x -q > gpurun_out/r2r_pytest.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/r2r_pytest.log
for rep in 1 2; do
for cfg in "lib 3" "lib 0" "lib 1" "lib 2" "lib_x 3" "lib_x 0"; do
  set -- $cfg
  GSR_LIB=gaussian-splatting_amd/$1/libgsr_hip.so timeout 300 python bench.py --no-other-configs --no-cpu-baseline --opt sh_dma=$2 > gpurun_out/r2r_bench_$1_$2_$rep.log 2>&1
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r2r_bench_$1_$2_$rep.log") if l.startswith("{")][-1])
s=d["stage_ms"]
print("$1 sh_dma=$2 rep $rep:", d["value"], d["ms_per_step"], d["train_iters_per_s"], "pre", s["preprocess"], "pre_bwd", s["preprocess_bwd"])
PY
done
done
