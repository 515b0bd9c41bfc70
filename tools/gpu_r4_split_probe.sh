#!/bin/bash
# why is the train step slower with the three-way build although every library stage is as fast or faster?  (1) train it/s against the
# backward's grid cap, (2) per-kernel medians (rocprofv3 --kernel-trace) of the train step with both libraries
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
B="--steps 5 --warmup 2 --train-steps 40 --no-other-configs --no-cpu-baseline --no-in-flight --no-full-loop --densify-iters 0 --min-warm-seconds 0.3"
one() {  # lib tag opts
  GSR_LIB="$R/gaussian-splatting_amd/$1/libgsr_hip.so" timeout 60 python bench.py $B $3 > gpurun_out/pr_$2.log 2>&1
  python - "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(f"gpurun_out/pr_{sys.argv[1]}.log") if l.startswith("{")][-1])
    print(f"{sys.argv[1]:14s} train {d['train_iters_per_s']}  l1 {d.get('train_iters_per_s_l1')}  sparse {d.get('train_iters_per_s_sparse_adam')}  pre_bwd {d['stage_ms'].get('preprocess_bwd')}")
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
one lib_prev prev ""
one lib new_1536 ""
one lib new_1024 "--opt preprocess_bwd_grid_cap=1024"
one lib new_768 "--opt preprocess_bwd_grid_cap=768"
one lib new_2304 "--opt preprocess_bwd_grid_cap=2304"
one lib new_all1536 "--opt preprocess_grid_cap=1536"
one lib_prev prev2 ""
echo "caps done at $SECONDS s"
cd /tmp
for lib in lib_prev lib; do
  GSR_LIB="$R/gaussian-splatting_amd/$lib/libgsr_hip.so" timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_$lib" -o r1 -- python "$R/bench.py" --steps 5 --warmup 2 --train-steps 60 --no-other-configs --no-cpu-baseline --no-in-flight --no-full-loop --densify-iters 0 --min-warm-seconds 0.2 > "$R/gpurun_out/pr_prof_$lib.log" 2>&1
  python "$R/tools/kernel_trace_stats.py" "$R/gpurun_out/prof_$lib" "$R/gpurun_out/kstats_$lib.csv" 0.25 > /dev/null 2>&1
  rm -rf "$R/gpurun_out/prof_$lib"
done
cd "$R"
python - <<'PY'
import csv
a = {r["Name"]: r for r in csv.DictReader(open("gpurun_out/kstats_lib_prev.csv"))}
b = {r["Name"]: r for r in csv.DictReader(open("gpurun_out/kstats_lib.csv"))}
print(f"{'kernel':52s} {'before us':>10s} {'after us':>10s}  calls")
for n in sorted(set(a) | set(b), key=lambda n: -float((a.get(n) or b.get(n))["TotalDurationNs"])):
    x, y = a.get(n), b.get(n)
    fx = float(x["MedianNs"]) / 1e3 if x else float("nan")
    fy = float(y["MedianNs"]) / 1e3 if y else float("nan")
    if max(fx if x else 0, fy if y else 0) >= 4.0:
        print(f"{n[:52]:52s} {fx:10.1f} {fy:10.1f}  {(x or y)['CallsAfterWarmup']}")
PY
echo "done at $SECONDS s"
