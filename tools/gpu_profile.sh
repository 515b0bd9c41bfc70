#!/bin/bash
# round-end measurement bundle: default bench line, rocprofv3 kernel stats, PMC passes (FETCH_SIZE / WRITE_SIZE separately)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
( time timeout 900 python bench.py ) > gpurun_out/p_bench_default.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_stats" -o r1 -- python "$R/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --densify-iters 0 --train-steps 15 --min-warm-seconds 0.2 > "$R/gpurun_out/p_prof_stats.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/prof_fetch" -o r1 -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --densify-iters 0 --train-steps 5 --min-warm-seconds 0.2 > "$R/gpurun_out/p_prof_fetch.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/prof_write" -o r1 -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --densify-iters 0 --train-steps 5 --min-warm-seconds 0.2 > "$R/gpurun_out/p_prof_write.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d "$R/gpurun_out/prof_sq" -o r1 -- python "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --densify-iters 0 --train-steps 4 --min-warm-seconds 0.2 > "$R/gpurun_out/p_prof_sq.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU --output-format csv -d "$R/gpurun_out/prof_sq2" -o r1 -- python "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --densify-iters 0 --train-steps 4 --min-warm-seconds 0.2 > "$R/gpurun_out/p_prof_sq2.log" 2>&1
cd "$R"
# steady-state per-kernel statistics (median / p10 / p90, warm-up dispatches dropped) from the per-dispatch trace, before it is deleted
python tools/kernel_trace_stats.py gpurun_out/prof_stats gpurun_out/kernel_stats_steady.csv 0.25
tail -4 gpurun_out/p_bench_default.log
ls gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq gpurun_out/prof_sq2
# keep the merged-back volume small: drop the per-dispatch traces of the PMC runs except the counter csv
find gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq gpurun_out/prof_sq2 -name "*kernel_trace*" -delete
