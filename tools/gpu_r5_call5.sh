#!/bin/bash
# Round 5, fifth GPU call: randomised cross-checks of the merged sources (bins incl. the new "outliers" depth kind; the whole operator forward +
# backward against the oracle), the new structural case of the bins sweep, the whole-frame comparison at configs[4].
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bins_sweep.py -q -m gpu -x -k "depth_outliers" 2>&1 | tail -3
( time timeout 420 python tools/gpu_fuzz_bins.py 400 51 ) 2>&1 | tail -4 | cut -c1-600; cp gpurun_out/fuzz_bins.json gpurun_out/r05_fuzz_bins.json 2>/dev/null
( time FUZZ_BIG=1 timeout 300 python tools/gpu_fuzz_bins.py 24 52 ) 2>&1 | tail -4 | cut -c1-600; cp gpurun_out/fuzz_bins.json gpurun_out/r05_fuzz_bins_big.json 2>/dev/null
( time timeout 420 python tools/gpu_fuzz_render.py 160 53 ) 2>&1 | tail -4 | cut -c1-600; cp gpurun_out/fuzz_render.json gpurun_out/r05_fuzz_render.json 2>/dev/null
( time timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "config4_6M_1080p_whole_frame" ) 2>&1 | tail -6
python - <<'PY'
import json
d = json.load(open("gpurun_out/parity_report.json"))
for k, v in d.items():
    if "6M" in k and "whole" in k:
        print(k, json.dumps(v))
PY
