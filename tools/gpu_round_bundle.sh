#!/bin/bash
# round-end measurement bundle: profiles + default bench line (incl. the configs[2] full-loop leg), the two extra configs[2] runs (growth,
# opt-in fused SH step), the per-rank shard model, the GPU test suite (its parity numbers land in gpurun_out/parity_report.json)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -n "$WANT_FAST_BOX" ]; then bash tools/gpu_box_probe.sh || exit 7; fi
if [ -z "$SKIP_AB" ] && [ -f gaussian-splatting_amd/lib_prev/libgsr_hip.so ]; then
  # same-box comparison with the round-3 library (tools/build_prev_lib.sh eb6bc02): interleaved forward-only runs
  bash tools/gpu_ab_prev.sh > gpurun_out/bundle_ab_prev.log 2>&1
  python - "${ROUND_TAG:-r04}" <<'PY'
import json, statistics, sys, os
tag = sys.argv[1]
def ms(lib):
    out = []
    for rep in (1, 2, 3):
        try:
            out.append(json.loads([l for l in open(f"gpurun_out/abp_{lib}_{rep}.log") if l.startswith("{")][-1])["ms_per_step"])
        except Exception:
            pass
    return out
cur, prev = ms("lib"), ms("lib_prev")
if cur and prev:
    os.makedirs(f"gpurun_out/profiles_{tag}", exist_ok=True)
    d = {"what": "python bench.py (forward only, 50 steps) with the current library and with the round-3 library (commit eb6bc02) on ONE box, runs interleaved",
         "round4_ms_per_frame": cur, "round3_ms_per_frame": prev,
         "round4_ms_per_frame_median": statistics.median(cur), "round3_ms_per_frame_median": statistics.median(prev)}
    json.dump(d, open(f"gpurun_out/profiles_{tag}/{tag}_ab_round3.json", "w"), indent=1)
    print("A/B round 3 -> round 4:", prev, "->", cur)
PY
fi
if [ -z "$SKIP_PROFILE" ]; then bash tools/gpu_profile.sh > gpurun_out/bundle_profile.log 2>&1; tail -6 gpurun_out/bundle_profile.log | cut -c1-400; fi
if [ -z "$SKIP_TRAIN" ]; then
timeout 1200 python tools/train_run.py --grad-threshold 0.00002 --tag _growth > gpurun_out/bundle_train_growth.log 2>&1
tail -1 gpurun_out/bundle_train_growth.log | cut -c1-300
timeout 900 python tools/train_run.py --fuse-sh-step --tag _fused_sh > gpurun_out/bundle_train_fused.log 2>&1
tail -1 gpurun_out/bundle_train_fused.log | cut -c1-300
timeout 900 python tools/train_run.py > gpurun_out/bundle_train_ref.log 2>&1
tail -1 gpurun_out/bundle_train_ref.log | cut -c1-300
fi
if [ -z "$SKIP_MODEL" ]; then timeout 900 python tools/gpu_shard_model.py > gpurun_out/bundle_shard_model.log 2>&1; tail -c 300 gpurun_out/bundle_shard_model.log; fi
if [ -z "$SKIP_TESTS" ]; then timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/bundle_pytest.log 2>&1; echo "pytest rc=$?"; tail -14 gpurun_out/bundle_pytest.log; fi
du -sh gpurun_out
