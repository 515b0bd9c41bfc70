#!/bin/bash
# Round 5, ninth GPU call: SparseGaussianAdam.step as ONE launch for all parameter groups (gsr_sparse_adam_step_multi) -- its GPU tests, and what it is worth:
# the configs[2] loop (host-bound at P ~ 100-140 K) and the sparse-optimizer train legs of bench.py, current library + package.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_next_rows.py -q -m gpu -x 2>&1 | tail -3
for rep in 1 2; do timeout 300 python tools/train_run.py --iters 10000 --tag _multi 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train_run 10000 it:', d['value'], 'it/s', [w['iters_per_s'] for w in d['windows']])"; done
timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-in-flight --densify-iters 600 > gpurun_out/r5c9_bench.log 2>&1
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5c9_bench.log") if l.startswith("{")][-1])
print("bench:", d["value"], d["ms_per_step"], "train", d["train_iters_per_s"], "sparse", d["train_iters_per_s_sparse_adam"], "densify", d["train_iters_per_s_densify"], d["train_iters_per_s_sh_step_in_backward"])
PY
