#!/bin/bash
# Round 5, sixth GPU call: the two "needles" frames of the render fuzz (seed 53) whose gradients left the 2e-3 bar against the fp32 oracle -- the same
# frames through round 4's library (regression or conditioning?) and through the fp64 adjudication; the big bins fuzz; the shard model with
# back-to-back chain timing.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( GSR_LIB="$PWD/gaussian-splatting_amd/lib_prev/libgsr_hip.so" timeout 600 python tools/gpu_fuzz_render.py 140 53 > gpurun_out/r5c6_fuzz_prev.log 2>&1; cp gpurun_out/fuzz_render_53.json gpurun_out/r5c6_fuzz_render_prev.json ) &
sleep 1
( timeout 600 python tools/gpu_fuzz_render.py 140 5353 > gpurun_out/r5c6_fuzz_cur_other_seed.log 2>&1 ) &
wait
timeout 600 python tools/gpu_fuzz_render.py 140 53 > gpurun_out/r5c6_fuzz_cur.log 2>&1; cp gpurun_out/fuzz_render_53.json gpurun_out/r5c6_fuzz_render_cur.json
for f in gpurun_out/r5c6_fuzz_render_prev.json gpurun_out/r5c6_fuzz_render_cur.json gpurun_out/fuzz_render_5353.json; do
python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], "frames", d["frames"], "failures", [(f["it"], f["kind"], f.get("error", "")[:90]) for f in d["failures"]], "adjudicated", d.get("adjudicated_by_fp64"))
PY
done
FUZZ_BIG=1 timeout 400 python tools/gpu_fuzz_bins.py 24 52 > gpurun_out/r5c6_fuzz_big.log 2>&1; tail -2 gpurun_out/r5c6_fuzz_big.log | cut -c1-500; cp gpurun_out/fuzz_bins.json gpurun_out/r05_fuzz_bins_big.json
( time timeout 900 python tools/gpu_shard_model.py > gpurun_out/r5c6_shard.log 2>&1 ) 2>&1 | tail -3
python - <<'PY'
import json
d = json.load(open("gpurun_out/shard_model.json"))
for name, c in d.items():
    print(name, c["1"], {G: {k: c[G][k] for k in ("slowest_rank_gpu_ms", "slowest_rank_chain_back_to_back_ms", "speedup_pipelined", "speedup_back_to_back", "speedup_back_to_back_fixed_exchange")} for G in ("2", "4", "8") if G in c})
PY
