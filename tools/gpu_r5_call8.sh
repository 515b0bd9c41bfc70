#!/bin/bash
# Round 5, eighth GPU call: where does the HOST spend an iteration of the configs[2] loop (P ~ 100-140 K: the GPU is done in ~0.6 ms, the loop runs at
# ~1.36 ms per iteration)?  cProfile of 4000 iterations of tools/train_run.py, top functions by own time and by cumulative time.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m cProfile -o gpurun_out/train_run.prof tools/train_run.py --iters 4000 --tag _prof > gpurun_out/r5c8_train_prof.log 2>&1
python - <<'PY'
import pstats, io
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats("gpurun_out/train_run.prof", stream=s).sort_stats(key).print_stats(45)
    txt = s.getvalue()
    open(f"gpurun_out/r5c8_host_profile_{key}.txt", "w").write(txt)
    print("\n".join(l[:170] for l in txt.splitlines()[:60]))
PY
tail -1 gpurun_out/r5c8_train_prof.log | cut -c1-300
