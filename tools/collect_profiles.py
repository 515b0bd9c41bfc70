"""Turn the output of tools/gpu_profile.sh (gpurun_out/p_bench_default.log, prof_stats/, prof_fetch/, prof_write/) into the
committed summaries under profiles/:
    rNN_bench_default.json        the default `python bench.py` JSON line
    rNN_kernel_stats.csv          rocprofv3 --kernel-trace --stats per-kernel table
    rNN_pmc_per_kernel.csv        FETCH_SIZE / WRITE_SIZE per kernel and launch (separate passes)
    pmc_latest.json               the same, keyed by kernel name -- bench.py reads roofline.traffic from here
Usage: python tools/collect_profiles.py r01"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
# where the summaries go: profiles/ (run locally on merged-back raw files) or, on the GPU box, a directory under gpurun_out/ -- the raw
# counter files of five rocprofv3 passes exceed what gpurun merges back (64 MiB), so tools/gpu_profile.sh summarises there and deletes them
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")
os.makedirs(OUT, exist_ok=True)


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)


line = None
if os.path.exists(os.path.join(G, "p_bench_default.log")):      # (tools/gpu_profile.sh takes the bench line AFTER the counter passes)
    line = [ln for ln in open(os.path.join(G, "p_bench_default.log")) if ln.startswith("{")][-1]
    json.loads(line)
    open(os.path.join(OUT, f"{tag}_bench_default.json"), "w").write(line)

steady = os.path.join(G, "kernel_stats_steady.csv")
if os.path.exists(steady):
    # tools/kernel_trace_stats.py: warm-up dispatches dropped, median / p10 / p90 beside the mean
    rows = list(csv.DictReader(open(steady)))
    open(os.path.join(OUT, f"{tag}_kernel_stats.csv"), "w").write(open(steady).read())
else:
    stats = glob.glob(os.path.join(G, "prof_stats", "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(OUT, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])

agg = collections.defaultdict(lambda: {"FETCH_SIZE": [0, 0.0], "WRITE_SIZE": [0, 0.0]})
for d in ("prof_fetch", "prof_write"):
    for fcsv in glob.glob(os.path.join(G, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fcsv)):
            c = r["Counter_Name"]
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                a = agg[short(r["Kernel_Name"])][c]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
# SQ counters (two passes of tools/gpu_profile.sh): per-kernel means per launch
sq = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in ("prof_sq", "prof_sq2"):
    for fcsv in glob.glob(os.path.join(G, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fcsv)):
            a = sq[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
SQ_NAMES = ['SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_WAVE_CYCLES',
            'SQ_BUSY_CYCLES', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_ANY',
            'SQ_WAIT_INST_ANY', 'SQ_LDS_BANK_CONFLICT']
if sq:
    with open(os.path.join(OUT, f"{tag}_pmc_sq_per_kernel.csv"), "w") as f:
        f.write("kernel," + ",".join(SQ_NAMES) + "\n")
        key = lambda k: -sq[k].get('SQ_WAVE_CYCLES', [1, 0])[1] / max(1, sq[k].get('SQ_WAVE_CYCLES', [1, 0])[0])
        for k in sorted(sq, key=key):
            if k.startswith("at::") or k.startswith("__amd"):
                continue
            f.write('"' + k + '",' + ",".join(f"{sq[k][n][1] / sq[k][n][0]:.0f}" if n in sq[k] else "" for n in SQ_NAMES) + "\n")

kern = {}
for k, v in agg.items():
    if k.startswith("at::") or k.startswith("__amd") or not v["FETCH_SIZE"][0] or not v["WRITE_SIZE"][0]:
        continue
    fk = v["FETCH_SIZE"][1] / v["FETCH_SIZE"][0]
    wk = v["WRITE_SIZE"][1] / v["WRITE_SIZE"][0]
    kern[k] = {"FETCH_SIZE_KB_per_launch": round(fk, 1), "launches_FETCH_SIZE": v["FETCH_SIZE"][0],
               "WRITE_SIZE_KB_per_launch": round(wk, 1), "launches_WRITE_SIZE": v["WRITE_SIZE"][0],
               "hbm_bytes_corrected": int((2 * fk + wk) * 1024)}
    for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
        if k in sq and n in sq[k]:
            kern[k][n] = int(sq[k][n][1] / sq[k][n][0])
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --steps 10 --warmup 2 "
                 "--train-steps 5; correction per MI355X_MICROARCH.md HBM section: bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                 "(FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950; gather-heavy kernels are over-corrected by up "
                 "to 2x on the read side)",
       "kernels": kern}
json.dump(out, open(os.path.join(OUT, "pmc_latest.json"), "w"), indent=1)
with open(os.path.join(OUT, f"{tag}_pmc_per_kernel.csv"), "w") as f:
    f.write("kernel,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,hbm_bytes_per_launch_corrected\n")
    for k in sorted(kern, key=lambda k: -kern[k]["hbm_bytes_corrected"]):
        f.write(f"\"{k}\",{kern[k]['FETCH_SIZE_KB_per_launch']},{kern[k]['WRITE_SIZE_KB_per_launch']},{kern[k]['hbm_bytes_corrected']}\n")
print("bench:", json.loads(line)["value"] if line else None, "Mpix/s;", len(rows), "kernels in stats;", len(kern), "kernels with PMC")
for k in ("render_fwd_wave_bf<true, 1, false>", "render_fwd_wave_bf<true, 1, true>", "render_bwd_half", "preprocess_fwd_kernel<false>", "adam_kernel"):
    if k in kern:
        print(" ", k, kern[k]["hbm_bytes_corrected"] / 1e6, "MB/launch")
