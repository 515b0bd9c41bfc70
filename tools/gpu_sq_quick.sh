#!/bin/bash
# SQ counter passes of a forward-only bench run -> per-kernel means for the binning kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
cd /tmp
B="--steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-in-flight --densify-iters 0 --train-steps 0 --min-warm-seconds 0.2"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d "$R/gpurun_out/prof_sq" -o r1 -- python "$R/bench.py" $B > "$R/gpurun_out/p_prof_sq.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU --output-format csv -d "$R/gpurun_out/prof_sq2" -o r1 -- python "$R/bench.py" $B > "$R/gpurun_out/p_prof_sq2.log" 2>&1
cd "$R"
python tools/pmc_sq_summary.py $(find gpurun_out/prof_sq gpurun_out/prof_sq2 -name "*counter_collection.csv") > gpurun_out/sq_quick.csv
find gpurun_out/prof_sq gpurun_out/prof_sq2 -name "*kernel_trace*" -delete
find gpurun_out/prof_sq gpurun_out/prof_sq2 -name "*counter_collection.csv" -delete
column -s, -t gpurun_out/sq_quick.csv | cut -c1-260 | head -24
