#!/bin/bash
# Round 5: the measurement bundle once more at HEAD (after gsr_sparse_adam_step_multi), without the render fuzz; leaves at once on a slow box.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export ROUND_TAG=r05 SKIP_AB=1 SKIP_MODEL=1 TMPDIR=/tmp
mkdir -p gpurun_out
BOX_PROBE_LIMIT_MS="${BOX_PROBE_LIMIT_MS:-0.345}" bash tools/gpu_box_probe.sh || exit 7
bash tools/gpu_round_bundle.sh
cp gpurun_out/parity_report.json gpurun_out/profiles_r05/r05_parity_report.json 2>/dev/null
for f in train_run_sparse.json train_run_sparse_growth.json train_run_sparse_fused_sh.json; do [ -f gpurun_out/$f ] && cp gpurun_out/$f gpurun_out/profiles_r05/r05_$f; done
ls gpurun_out/profiles_r05
