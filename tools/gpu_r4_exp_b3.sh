#!/bin/bash
# experiment: the per-Gaussian backward at three waves per SIMD (preprocess.hip built -fno-slp-vectorize, which brings preprocess_bwd_kernel
# from 210 to 176 VGPRs, + amdgpu_waves_per_eu(3,3) on it: 168 VGPRs, nothing spilled; with the SLP vectoriser on the same attribute spills 36).
# Read-out: the preprocess_bwd stage (library HIP events) of the train step, grid caps 768 / 1024 / 1536 (three resident workgroups per CU
# make 768 one full round).  lib_b3 = this build; its forward kernels carry -fno-slp-vectorize too and are not what is being judged.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
run() {   # lib tag extra-args
  GSR_LIB="$R/gaussian-splatting_amd/$1/libgsr_hip.so" timeout 100 python bench.py --steps 10 --warmup 3 --train-steps 30 --no-other-configs --no-cpu-baseline --no-in-flight --no-full-loop --densify-iters 0 --min-warm-seconds 0.3 $3 > gpurun_out/b3_$2.log 2>&1
  python - "$2" "gpurun_out/b3_$2.log" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    s = d["stage_ms"]
    print(f"{sys.argv[1]:16s} preprocess_bwd {s.get('preprocess_bwd')}  preprocess {s.get('preprocess')}  train {d.get('train_iters_per_s')} it/s  l1 {d.get('train_iters_per_s_l1')}  sparse {d.get('train_iters_per_s_sparse_adam')}")
except Exception as e:
    print(sys.argv[1], "bench failed", e); print(open(sys.argv[2]).read()[-600:])
PY
}
for rep in 1 2; do
  run lib      lib_$rep ""
  run lib_b3   b3_1024_$rep ""
  run lib_b3   b3_768_$rep "--opt preprocess_grid_cap=768"
  run lib_b3   b3_1536_$rep "--opt preprocess_grid_cap=1536"
  run lib      lib_768_$rep "--opt preprocess_grid_cap=768"
done
echo "A/B done at $SECONDS s"
GSR_LIB="$R/gaussian-splatting_amd/lib_b3/libgsr_hip.so" timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_next_rows.py -m gpu -q -x -k "backward or grad or train or reproducible" > gpurun_out/b3_pytest.log 2>&1; echo "b3 pytest rc=$? at $SECONDS s"; tail -3 gpurun_out/b3_pytest.log | cut -c1-200
