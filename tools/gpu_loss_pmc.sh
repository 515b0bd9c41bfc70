#!/bin/bash
# SQ + memory counter passes on the SSIM kernels alone (tools/gpu_loss_only.py); options of the library as arguments ("ssim_variant=1")
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
cd /tmp
python "$R/tools/gpu_loss_only.py" 50 "$@"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_INSTS_VALU"; do
  i=$((i+1))
  rm -rf "$R/gpurun_out/loss_pmc_$i"
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$R/gpurun_out/loss_pmc_$i" -o r1 -- python "$R/tools/gpu_loss_only.py" 6 "$@" > "$R/gpurun_out/loss_pmc_$i.log" 2>&1
done
cd "$R"
python - <<'PY'
import collections, csv, glob, re
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/loss_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        if "ssim" not in k and "loss" not in k:
            continue
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(agg):
    print(k)
    for n in sorted(agg[k]):
        print(f"    {n:32s} {agg[k][n][1] / agg[k][n][0]:16.0f}")
PY
find gpurun_out -path "*loss_pmc_*" -name "*kernel_trace*" -delete
