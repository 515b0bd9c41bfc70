"""Per-kernel duration statistics from a rocprofv3 --kernel-trace run (the per-dispatch *kernel_trace.csv), computed on the GPU box
before the trace is deleted (it is too large to merge back):

    python tools/kernel_trace_stats.py gpurun_out/prof_stats gpurun_out/kernel_stats_steady.csv [skip_fraction]

VERDICT r02 weak #3: the `--stats` table averages EVERY dispatch, warm-up launches included (render_fwd: mean 160.6 us, min 136.8,
max 806 -- the mean did not reproduce bench.py's HIP-event figure to better than 10 %).  Here the first `skip_fraction` (default
0.25) of every kernel's dispatches is dropped and the median, p10, p90 are reported beside the mean, so that `roofline.kernel_ms`
can be reproduced from profiles/ to a couple of per cent."""
import collections
import csv
import glob
import os
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
files = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
if not files:
    print("no kernel_trace.csv under", src)
    raise SystemExit(1)
dur = collections.defaultdict(list)
for f in files:
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        n = re.sub(r"\(.*", "", n)
        dur[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v[int(len(v) * skip):]) for v in dur.values()) or 1
with open(dst, "w") as out:
    w = csv.writer(out)
    w.writerow(["Name", "Calls", "CallsAfterWarmup", "TotalDurationNs", "MeanNs", "MedianNs", "P10Ns", "P90Ns", "MinNs", "MaxNs", "Percentage"])
    for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1][int(len(kv[1]) * skip):])):
        s = sorted(v[int(len(v) * skip):])
        if not s:
            continue
        q = lambda f: s[min(len(s) - 1, int(f * len(s)))]
        w.writerow([n, len(v), len(s), sum(s), round(sum(s) / len(s), 1), q(0.5), q(0.1), q(0.9), s[0], s[-1], round(100.0 * sum(s) / tot, 2)])
print("wrote", dst, len(dur), "kernels")
