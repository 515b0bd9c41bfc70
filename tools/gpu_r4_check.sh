#!/bin/bash
# round 4: the binning chain first (bit-exact bins in every regime, both depth sorts, both rectangle modes), then the whole
# GPU suite, then an interleaved A/B of the two depth sorts on the bench frame
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bins_sweep.py -q -m gpu --tb=short 2>&1 | tail -60 > gpurun_out/r4_bins.log
echo "== bins sweep"; tail -40 gpurun_out/r4_bins.log
if grep -q "failed\|error" gpurun_out/r4_bins.log; then echo "BINS FAILED"; fi
timeout 1500 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -60 > gpurun_out/r4_pytest.log
echo "== whole gpu suite"; tail -30 gpurun_out/r4_pytest.log
GSR_LIB= bash -c 'unset GSR_LIB; for rep in 1 2; do for m in 1 2; do
  timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-in-flight --densify-iters 0 --opt depth_sort_mode=$m > gpurun_out/r4_ab_${m}_$rep.log 2>&1
  python - "$m" "$rep" "gpurun_out/r4_ab_${m}_$rep.log" <<PY
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
    print(f"depth_sort_mode={sys.argv[1]} rep {sys.argv[2]}:", d["value"], d["ms_per_step"], d.get("train_iters_per_s"), d["stage_ms"])
except Exception as e:
    print("bench failed", e); print(open(sys.argv[3]).read()[-2000:])
PY
done; done'
