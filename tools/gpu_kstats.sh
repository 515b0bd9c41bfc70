#!/bin/bash
# quick per-kernel duration table of a short bench run (rocprofv3 --kernel-trace --stats)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
cd /tmp
rm -rf "$R/gpurun_out/prof_k"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_k" -o r1 -- python "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --densify-iters 0 --train-steps 8 --min-warm-seconds 0.2 $BENCH_EXTRA > "$R/gpurun_out/k_prof.log" 2>&1
cd "$R"
python - <<'PY'
import csv, glob, re
f = glob.glob("gpurun_out/prof_k/**/*kernel_stats.csv", recursive=True)
if not f:
    print("no stats file", glob.glob("gpurun_out/prof_k/**/*", recursive=True)); raise SystemExit
for r in list(csv.DictReader(open(f[0])))[:32]:
    n = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
    print(f"{n[:58]:58s} {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f}%")
PY
find gpurun_out/prof_k -name "*kernel_trace*" -delete
tail -2 gpurun_out/k_prof.log | cut -c1-600
