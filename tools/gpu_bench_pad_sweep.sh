#!/bin/bash
# Does a bench leg's rate depend on where the allocator puts its arrays?  The headline legs with an n-MiB allocation kept alive from the start of the process
# (GSR_BENCH_PAD_MB), for every library in $LIBS (default: lib), interleaved.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
for pad in ${PADS:-0 1 3 16 100 0}; do
  for lib in ${LIBS:-lib}; do
    GSR_BENCH_PAD_MB=$pad GSR_LIB="$PWD/gaussian-splatting_amd/$lib/libgsr_hip.so" timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-in-flight --no-pmc --densify-iters 0 > gpurun_out/pad_${lib}_$pad.log 2>&1
    python - "$lib" "$pad" "gpurun_out/pad_${lib}_$pad.log" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
print(f"{sys.argv[1]} pad {sys.argv[2]} MiB:", d["ms_per_step"], "ms/frame;", d["train_iters_per_s"], "it/s; sparse", d["train_iters_per_s_sparse_adam"], "; events", d["gpu_event_ms"]["train_ssim"]["median_ms"])
PY
  done
done
