"""GPU timeline summary of a rocprofv3 --kernel-trace run: over the LAST `frac` of the dispatches (steady state) -- wall span, time with
at least one kernel running (union of the intervals), idle time between kernels, and per kernel: dispatches, total and mean duration,
share of the span.  Tells a host-bound loop (large idle share) from a GPU-bound one.

    python tools/kernel_trace_timeline.py <rocprof dir> <out.json> [frac=0.5] [iterations in that window, for per-iteration figures]
"""
import collections
import csv
import glob
import json
import os
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
iters = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
files = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        n = re.sub(r"\(.*", "", n)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
rows = rows[int(len(rows) * (1.0 - frac)):]
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
for s, e, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
# idle time by (kernel that ended, kernel that started next) -- where the GPU waits, and for whom
pair_gap = collections.defaultdict(lambda: [0, 0])
prev_end, prev_name = rows[0][1], rows[0][2]
for s_, e_, n_ in rows[1:]:
    if s_ > prev_end:
        g_ = pair_gap[(prev_name[:40], n_[:40])]
        g_[0] += 1
        g_[1] += s_ - prev_end
    if e_ > prev_end:
        prev_end, prev_name = e_, n_
per = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    per[n][0] += 1
    per[n][1] += e - s
gaps.sort()
out = {"dispatches": len(rows), "span_ms": span / 1e6, "busy_ms": busy / 1e6, "idle_ms": (span - busy) / 1e6, "idle_share": (span - busy) / span,
       "gaps": {"count": len(gaps), "median_us": gaps[len(gaps) // 2] / 1e3 if gaps else 0, "p90_us": gaps[int(len(gaps) * 0.9)] / 1e3 if gaps else 0,
                "max_us": gaps[-1] / 1e3 if gaps else 0, "over_20us_total_ms": sum(g for g in gaps if g > 20000) / 1e6,
                "over_20us_count": sum(1 for g in gaps if g > 20000)},
       "kernels": [{"name": n, "dispatches": c, "total_ms": round(t / 1e6, 4), "mean_us": round(t / c / 1e3, 2), "share_of_span": round(t / span, 4),
                    **({"ms_per_iteration": round(t / 1e6 / iters, 4), "launches_per_iteration": round(c / iters, 2)} if iters else {})}
                   for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])]}
out["idle_by_kernel_pair"] = [{"after": a_, "before": b_, "count": c_, "idle_ms": round(t_ / 1e6, 4), "mean_us": round(t_ / c_ / 1e3, 2)}
                              for (a_, b_), (c_, t_) in sorted(pair_gap.items(), key=lambda kv: -kv[1][1])[:25]]
if iters:
    out["span_ms_per_iteration"] = span / 1e6 / iters
    out["busy_ms_per_iteration"] = busy / 1e6 / iters
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("kernels", "idle_by_kernel_pair")}))
for g_ in out["idle_by_kernel_pair"][:14]:
    print(f"  idle {g_['idle_ms']:8.3f} ms  n {g_['count']:5d}  mean {g_['mean_us']:7.1f} us   after {g_['after']:40s} -> {g_['before']}")
for k in out["kernels"][:40]:
    print(f"{k['name'][:60]:60s} n {k['dispatches']:6d}  total {k['total_ms']:9.3f} ms  mean {k['mean_us']:8.1f} us  {100 * k['share_of_span']:5.1f} %" + (f"  {k['ms_per_iteration']:.4f} ms/it" if iters else ""))
