#!/bin/bash
# Where does an iteration of a GROWN scene go?    One call:
#   1. the configs[2] stand-in with the growth settings run to iteration $ITERS (default 20000), state saved on the box, then tools/train_run.py's
#      --timeline pass (wall clock / host enqueue time / GPU events per phase / the library's stage events);
#   2. the same loop continued from the saved state for 300 iterations under rocprofv3 --kernel-trace: per-kernel table + idle share
#      (tools/kernel_trace_timeline.py over the last 200 iterations).
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
ITERS=${ITERS:-20000}
TAG=${TAG:-grown}
timeout 900 python tools/train_run.py --grad-threshold 0.00002 --iters $ITERS --save-state /tmp/state_$TAG.pt --timeline 200 --tag _timeline_$TAG $TRAIN_EXTRA > gpurun_out/train_timeline_$TAG.log 2>&1
grep TIMELINE gpurun_out/train_timeline_$TAG.log | cut -c1-3000
cd /tmp
rm -rf "$R/gpurun_out/prof_train_$TAG"
timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/prof_train_$TAG" -o t -- python "$R/tools/train_run.py" --grad-threshold 0.00002 --load-state /tmp/state_$TAG.pt --iters $((ITERS + 300)) --tag _traced_$TAG $TRAIN_EXTRA > "$R/gpurun_out/train_traced_$TAG.log" 2>&1
cd "$R"
tail -1 gpurun_out/train_traced_$TAG.log | cut -c1-300
# the traced run renders the 32 ground-truth views first (32 forwards), then 300 iterations: the last 60 % of the dispatches are >= 180 iterations
python tools/kernel_trace_timeline.py gpurun_out/prof_train_$TAG gpurun_out/train_kernel_timeline_$TAG.json 0.6 180
find gpurun_out/prof_train_$TAG -name "*kernel_trace*" -delete
