#!/bin/bash
# kernel durations of the loss kernels for several launch shapes of the marching form (ssim_target_waves)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for t in "$@"; do
  echo "== $t"
  bash tools/gpu_loss_kstats.sh $t 2>&1 | grep -i "march\|ssim_.*kernel"
done
