#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/r2e_bench.log 2>&1
tail -1 gpurun_out/r2e_bench.log | cut -c1-6000
timeout 300 python tools/train_run.py --iters 1500 --P0 50000 --P-target 300000 > gpurun_out/r2e_train_small.log 2>&1
tail -3 gpurun_out/r2e_train_small.log | cut -c1-1500
timeout 900 python tools/train_run.py > gpurun_out/r2e_train.log 2>&1
tail -1 gpurun_out/r2e_train.log | cut -c1-3000
timeout 600 python tools/gpu_lane_stats.py 0.012 200 > gpurun_out/r2e_lane.log 2>&1; tail -7 gpurun_out/r2e_lane.log
