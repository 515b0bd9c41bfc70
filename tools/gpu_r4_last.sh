#!/bin/bash
# round 4, last measurement call (17 GPU-minutes were left): steady-state kernel trace + FETCH_SIZE / WRITE_SIZE passes of the final code,
# the default bench line taken after them (so that its traffic fields come from this run's passes), then the GPU test suite.
# The SQ passes of tools/gpu_profile.sh are skipped (the committed r04_pmc_sq_* files are from the interim bundle; the kernels' instruction
# streams have not changed since).
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
TAG=r04
B="--no-cpu-baseline --no-other-configs --no-in-flight --no-full-loop --densify-iters 0 --min-warm-seconds 0.2"
T0=$SECONDS
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_stats" -o r1 -- python "$R/bench.py" --steps 30 --warmup 5 --train-steps 15 $B > "$R/gpurun_out/p_prof_stats.log" 2>&1
echo "kstats done at $((SECONDS-T0)) s"
timeout 180 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/prof_fetch" -o r1 -- python "$R/bench.py" --steps 10 --warmup 2 --train-steps 5 $B > "$R/gpurun_out/p_prof_fetch.log" 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/prof_write" -o r1 -- python "$R/bench.py" --steps 10 --warmup 2 --train-steps 5 $B > "$R/gpurun_out/p_prof_write.log" 2>&1
echo "pmc done at $((SECONDS-T0)) s"
cd "$R"
python tools/kernel_trace_stats.py gpurun_out/prof_stats gpurun_out/kernel_stats_steady.csv 0.25
python tools/collect_profiles.py "$TAG" "gpurun_out/profiles_$TAG"
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
# SQ_* fields: carried over from the committed interim bundle (no SQ passes in this call), marked as such
python - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
new_p = f"gpurun_out/profiles_{tag}/pmc_latest.json"
try:
    new = json.load(open(new_p)); old = json.load(open("profiles/pmc_latest.json"))
    if new.get("kernels"):
        for k, v in new["kernels"].items():
            for n, x in old.get("kernels", {}).get(k, {}).items():
                if n.startswith("SQ_") and n not in v:
                    v[n] = x
        new["sq_source"] = "SQ_* fields: rocprofv3 SQ passes of the interim round-4 bundle (commit 1bc1ee2; same instruction streams), FETCH / WRITE: this run"
        json.dump(new, open(new_p, "w"), indent=1)
        json.dump(new, open("profiles/pmc_latest.json", "w"), indent=1)
        print("pmc_latest.json: FETCH/WRITE of", len(new["kernels"]), "kernels refreshed")
except Exception as ex:
    print("pmc merge skipped:", repr(ex))
PY
( time timeout 400 python bench.py ) > gpurun_out/p_bench_default.log 2>&1
echo "bench done at $((SECONDS-T0)) s"
python - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
line = [ln for ln in open("gpurun_out/p_bench_default.log") if ln.startswith("{")][-1]
d = json.loads(line)
open(f"gpurun_out/profiles_{tag}/{tag}_bench_default.json", "w").write(line)
print("bench:", d["value"], "Mpix/s", d["ms_per_step"], "ms; train", d.get("train_iters_per_s"), "it/s; full loop", d.get("train_iters_per_s_full_loop_configs2"))
print("stages:", {k: v.get("ms") for k, v in d.get("stages", {}).items()})
PY
timeout 420 python -m pytest tests -m gpu -q -x --durations=6 > gpurun_out/last_pytest.log 2>&1; echo "pytest rc=$? at $((SECONDS-T0)) s"; tail -12 gpurun_out/last_pytest.log | cut -c1-300
cp gpurun_out/parity_report.json "gpurun_out/profiles_$TAG/${TAG}_parity_report.json" 2>/dev/null
du -sh gpurun_out
