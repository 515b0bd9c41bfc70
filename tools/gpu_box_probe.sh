#!/bin/bash
# Box lottery guard: the pool has two kinds of MI355X boxes (round-4 notebook: the forward of configs[1] runs at ~0.353 ms on one
# kind and ~0.385 ms on the other, every memory-bound kernel ~15-25 % slower).  Profiles that are to be compared with an earlier
# table want the same kind.  Exit 0 on a box of the fast kind, 7 otherwise (prints the probe's frame time either way).
# usage (first line of a bundle):  bash tools/gpu_box_probe.sh || exit 7
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
LIMIT="${BOX_PROBE_LIMIT_MS:-0.368}"
timeout 300 python bench.py --steps 40 --warmup 10 --train-steps 0 --no-cpu-baseline --no-full-loop --no-other-configs --no-in-flight --densify-iters 0 --min-warm-seconds 1.0 > gpurun_out/box_probe.json 2> gpurun_out/box_probe.err || { echo "probe failed"; tail -5 gpurun_out/box_probe.err; exit 1; }
python - "$LIMIT" <<'PY'
import json, sys
line = [l for l in open("gpurun_out/box_probe.json") if l.startswith("{")][-1]
ms = json.loads(line)["ms_per_step"]
lim = float(sys.argv[1])
print(f"box probe: forward {ms:.4f} ms per frame (limit {lim})")
sys.exit(0 if ms <= lim else 7)
PY
