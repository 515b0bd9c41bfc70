#!/bin/bash
# Round 5, final measurement bundle on the merged sources: rocprofv3 kernel trace + FETCH / WRITE / SQ passes (summarised on the box), the default
# bench line taken after them (its traffic / valu_* fields then come from THIS box's counter passes), the three configs[2] training runs, the whole GPU
# test suite (parity report), the render fuzz with the conditioning-aware bar.   WANT_FAST_BOX=1: leave at once (exit 7) on a box of the slow kind.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export ROUND_TAG=r05 SKIP_AB=1 SKIP_MODEL=1 TMPDIR=/tmp
mkdir -p gpurun_out
if [ -n "$WANT_FAST_BOX" ]; then BOX_PROBE_LIMIT_MS="${BOX_PROBE_LIMIT_MS:-0.350}" bash tools/gpu_box_probe.sh || exit 7; fi
bash tools/gpu_round_bundle.sh
cp gpurun_out/parity_report.json gpurun_out/profiles_r05/r05_parity_report.json 2>/dev/null
for f in train_run_sparse.json train_run_sparse_growth.json train_run_sparse_fused_sh.json; do [ -f gpurun_out/$f ] && cp gpurun_out/$f gpurun_out/profiles_r05/r05_$f; done
( time timeout 420 python tools/gpu_fuzz_render.py 140 53 > gpurun_out/r5_final_fuzz_render.log 2>&1 ) 2>&1 | tail -3
cp gpurun_out/fuzz_render_53.json gpurun_out/profiles_r05/r05_fuzz_render.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/fuzz_render_53.json"))
print("render fuzz: frames", d["frames"], "failures", [(f["it"], f["kind"], f.get("error", "")[:80], f.get("fp64")) for f in d["failures"]], "adjudicated", d.get("adjudicated_by_fp64"))
PY
ls gpurun_out/profiles_r05
