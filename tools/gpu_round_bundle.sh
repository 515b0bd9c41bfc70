#!/bin/bash
# round-end measurement bundle: profiles + configs[2] training runs (reference schedule, growth, opt-in fused SH step) + GPU tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu_profile.sh > gpurun_out/bundle_profile.log 2>&1
tail -3 gpurun_out/bundle_profile.log | cut -c1-400
timeout 900 python tools/train_run.py > gpurun_out/bundle_train_ref.log 2>&1
tail -1 gpurun_out/bundle_train_ref.log | cut -c1-2500
timeout 1200 python tools/train_run.py --grad-threshold 0.00002 --tag _growth > gpurun_out/bundle_train_growth.log 2>&1
tail -1 gpurun_out/bundle_train_growth.log | cut -c1-2500
timeout 900 python tools/train_run.py --fuse-sh-step --tag _fused_sh > gpurun_out/bundle_train_fused.log 2>&1
tail -1 gpurun_out/bundle_train_fused.log | cut -c1-400
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/bundle_pytest.log 2>&1
echo "pytest rc=$?"; tail -1 gpurun_out/bundle_pytest.log
