#!/bin/bash
# instruction counts of ds_segsort on the uniform and the crowd frame of tools/gpu_depth_distribution_probe.py (where does a 3x longer launch spend it?)
cd /tmp; export TMPDIR=/tmp
for c in uniform crowd; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_$c
  PROBE_CASES=$c rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o r -- python $GRAFT_REPO_ROOT/tools/gpu_depth_distribution_probe.py > /tmp/out_$c.log 2>&1
  python3 - $GRAFT_REPO_ROOT/gpurun_out/pmc_$c $c <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "ds_segsort" in k or "ds_scatter" in k or "ds_hist" in k:
        name = "ds_segsort" if "ds_segsort" in k else "ds_scatter" if "ds_scatter" in k else "ds_hist"
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"]); n[(name, r["Counter_Name"])] += 1
for name in acc:
    print(sys.argv[2], name, {c: round(v / n[(name, c)]) for c, v in acc[name].items()})
PY
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_$c
done
