#!/bin/bash
# round 2, GPU call A: full GPU suite (incl. the new full-size oracle parity tests), default bench, legacy tile sort A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.json
timeout 900 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r2a_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2a_pytest.log
tail -30 gpurun_out/r2a_pytest.log
timeout 300 python bench.py --steps 30 --warmup 5 > gpurun_out/r2a_bench.log 2>&1
tail -2 gpurun_out/r2a_bench.log | cut -c1-3000
timeout 300 python bench.py --steps 30 --warmup 5 --opt tile_sort_mode=1 --no-cpu-baseline --train-steps 0 > gpurun_out/r2a_bench_legacy.log 2>&1
tail -1 gpurun_out/r2a_bench_legacy.log | cut -c1-1500
timeout 120 python tools/gpu_stats.py > gpurun_out/r2a_stats.log 2>&1
tail -2 gpurun_out/r2a_stats.log
