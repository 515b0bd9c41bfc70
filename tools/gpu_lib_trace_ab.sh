#!/bin/bash
# Same-box kernel-level A/B of LIBRARIES: for every library in $LIBS (directories under gaussian-splatting_amd/) a short bench.py run (forward + the
# train legs) under rocprofv3 --kernel-trace; per library: steady-state kernel medians (tools/kernel_trace_stats.py) and the GPU timeline of the last
# half of the run (tools/kernel_trace_timeline.py: busy / idle share, gaps) -- tells a slower KERNEL from time lost BETWEEN kernels.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
for lib in ${LIBS:-lib lib_prev}; do
  cd /tmp
  rm -rf "$R/gpurun_out/prof_ab_$lib"
  GSR_LIB="$R/gaussian-splatting_amd/$lib/libgsr_hip.so" timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/prof_ab_$lib" -o r1 -- python "$R/bench.py" --steps 30 --warmup 5 --train-steps 40 --no-cpu-baseline --no-other-configs --no-in-flight --no-pmc --densify-iters 0 --min-warm-seconds 0.2 > "$R/gpurun_out/trace_ab_$lib.log" 2>&1
  cd "$R"
  python tools/kernel_trace_stats.py gpurun_out/prof_ab_$lib gpurun_out/kernel_stats_$lib.csv 0.25 > /dev/null
  echo "== $lib"; tail -1 gpurun_out/trace_ab_$lib.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['train_iters_per_s'], d['train_iters_per_s_sparse_adam'])"
  python tools/kernel_trace_timeline.py gpurun_out/prof_ab_$lib gpurun_out/kernel_timeline_$lib.json 0.3 | head -16 | cut -c1-400
  find gpurun_out/prof_ab_$lib -name "*kernel_trace*" -delete
done
python - <<'PY'
import csv, os
libs = os.environ.get("LIBS", "lib lib_prev").split()
tabs = {l: {r["Name"]: float(r["MedianNs"]) / 1e3 for r in csv.DictReader(open(f"gpurun_out/kernel_stats_{l}.csv"))} for l in libs}
names = [n for n in tabs[libs[0]] if not n.startswith("at::") and tabs[libs[0]][n] > 3.0]
print("kernel".ljust(52), "  ".join(l.rjust(10) for l in libs))
for n in names:
    print(n[:52].ljust(52), "  ".join(f"{tabs[l].get(n, float('nan')):10.1f}" for l in libs))
PY
