"""Turn a rocprofv3 rocpd sqlite database (default output format of this image's rocprofv3) into the per-kernel
stats table that `--stats` would print: name, calls, total ns, average ns, percentage.  Usage:
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.csv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc").fetchall()
print("Name,Calls,TotalDurationNs,AverageNs,Percentage")
for n, c, t, a, p in rows:
    n = n.split("(")[0].replace("(anonymous namespace)::", "")
    print(f"\"{n}\",{c},{int(t)},{a:.1f},{p:.2f}")
