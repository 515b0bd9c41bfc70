#!/bin/bash
# per-kernel durations of the loss kernels alone (tools/gpu_loss_only.py under rocprofv3 --kernel-trace --stats)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
cd /tmp
rm -rf "$R/gpurun_out/loss_k"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/loss_k" -o r1 -- python "$R/tools/gpu_loss_only.py" 100 "$@" > "$R/gpurun_out/loss_k.log" 2>&1
cd "$R"
python tools/kernel_trace_stats.py gpurun_out/loss_k gpurun_out/loss_kernel_stats.csv 0.1 > /dev/null
grep -i "ssim\|loss_mean\|Name" gpurun_out/loss_kernel_stats.csv | cut -d, -f1,2,5,6,7,8
find gpurun_out/loss_k -name "*kernel_trace*" -delete
