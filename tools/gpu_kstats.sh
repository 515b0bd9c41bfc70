#!/bin/bash
# quick per-kernel duration table of a short bench run (rocprofv3 --kernel-trace): steady-state median / p10 / p90 per kernel
# (warm-up dispatches dropped, tools/kernel_trace_stats.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
cd /tmp
rm -rf "$R/gpurun_out/prof_k"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_k" -o r1 -- python "$R/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-in-flight --densify-iters 0 --train-steps 10 --min-warm-seconds 0.2 $BENCH_EXTRA > "$R/gpurun_out/k_prof.log" 2>&1
cd "$R"
python tools/kernel_trace_stats.py gpurun_out/prof_k gpurun_out/kernel_stats_steady.csv 0.25
python - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/kernel_stats_steady.csv")))[:34]:
    print(f"{r['Name'][:56]:56s} {r['CallsAfterWarmup']:>5s} median {float(r['MedianNs'])/1e3:8.1f}  p10 {float(r['P10Ns'])/1e3:8.1f}  p90 {float(r['P90Ns'])/1e3:8.1f} us  {float(r['Percentage']):5.1f}%")
PY
find gpurun_out/prof_k -name "*kernel_trace*" -delete
tail -1 gpurun_out/k_prof.log | cut -c1-400
