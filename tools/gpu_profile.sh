#!/bin/bash
# round-end measurement bundle: rocprofv3 kernel trace + PMC passes (FETCH_SIZE / WRITE_SIZE / SQ, each in its own run), summarised ON the
# box (the raw counter files exceed what gpurun merges back), then the default bench line -- taken last, so that its roofline.traffic /
# valu_* fields come from THIS run's counter passes (profiles/pmc_latest.json is replaced on the box before bench.py reads it)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
TAG="${ROUND_TAG:-r04}"
B="--no-cpu-baseline --no-other-configs --no-in-flight --no-pmc --densify-iters 0 --min-warm-seconds 0.2"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_stats" -o r1 -- python "$R/bench.py" --steps 30 --warmup 5 --train-steps 15 $B > "$R/gpurun_out/p_prof_stats.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/prof_fetch" -o r1 -- python "$R/bench.py" --steps 10 --warmup 2 --train-steps 5 $B > "$R/gpurun_out/p_prof_fetch.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/prof_write" -o r1 -- python "$R/bench.py" --steps 10 --warmup 2 --train-steps 5 $B > "$R/gpurun_out/p_prof_write.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d "$R/gpurun_out/prof_sq" -o r1 -- python "$R/bench.py" --steps 6 --warmup 2 --train-steps 4 $B > "$R/gpurun_out/p_prof_sq.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU --output-format csv -d "$R/gpurun_out/prof_sq2" -o r1 -- python "$R/bench.py" --steps 6 --warmup 2 --train-steps 4 $B > "$R/gpurun_out/p_prof_sq2.log" 2>&1
cd "$R"
# steady-state per-kernel statistics (median / p10 / p90, warm-up dispatches dropped) from the per-dispatch trace, before it is deleted
python tools/kernel_trace_stats.py gpurun_out/prof_stats gpurun_out/kernel_stats_steady.csv 0.25
python tools/collect_profiles.py "$TAG" "gpurun_out/profiles_$TAG"
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq gpurun_out/prof_sq2
cp "gpurun_out/profiles_$TAG/pmc_latest.json" profiles/pmc_latest.json
( time timeout 1200 python bench.py ) > gpurun_out/p_bench_default.log 2>&1
python - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
line = [ln for ln in open("gpurun_out/p_bench_default.log") if ln.startswith("{")][-1]
d = json.loads(line)
open(f"gpurun_out/profiles_{tag}/{tag}_bench_default.json", "w").write(line)
print("bench:", d["value"], "Mpix/s", d["ms_per_step"], "ms; train", d.get("train_iters_per_s"), "it/s; full loop", d.get("train_iters_per_s_full_loop_configs2"))
PY
tail -4 gpurun_out/p_bench_default.log | cut -c1-300
