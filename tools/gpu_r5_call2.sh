#!/bin/bash
# Round 5, second GPU call: (1) the tests that are new or changed this round (band-pipelined forward, robust depth-bucket range through the bins sweep,
# fp64 adjudication on the clustered whole frame, smoke()'s tighter gradient bar); (2) same-box interleaved A/B of option fwd_bands (level-2 sort + blend
# band by band on two HIP streams) and of render_fwd_lds_pad (resident waves of the forward blend capped through LDS).
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bins_sweep.py -q -m gpu --tb=short -x 2>&1 | tail -12 | tee gpurun_out/r5c2_pytest_parity.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu --tb=short -x -k "clustered_whole_frame or config1_1M_1080p_whole_frame" 2>&1 | tail -12 | tee gpurun_out/r5c2_pytest_whole.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/r5c2_smoke.log
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d.get("train_iters_per_s"), d["stage_ms"], (d.get("gpu_event_ms") or {}).get("forward"))
PY
}
echo "== A/B fwd_bands (forward + train legs)"
for rep in 1 2 3; do
  for nb in 1 2 3 4; do
    timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-in-flight --densify-iters 0 --opt fwd_bands=$nb > gpurun_out/r5c2_bands${nb}_$rep.log 2>&1
    line "fwd_bands=$nb rep $rep:" gpurun_out/r5c2_bands${nb}_$rep.log
  done
done | tee gpurun_out/r5c2_ab_bands.log
echo "== A/B render_fwd_lds_pad (forward only)"
for rep in 1 2; do
  for pad in 0 2000 4000 7000 10000; do
    timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-in-flight --densify-iters 0 --train-steps 0 --opt render_fwd_lds_pad=$pad > gpurun_out/r5c2_pad${pad}_$rep.log 2>&1
    line "render_fwd_lds_pad=$pad rep $rep:" gpurun_out/r5c2_pad${pad}_$rep.log
  done
done | tee gpurun_out/r5c2_ab_pad.log
