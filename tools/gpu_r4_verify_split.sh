#!/bin/bash
# the three-way build of preprocess.hip (fused forward / backward without the SLP vectoriser, backward at three waves per SIMD, its own grid cap):
# same-box A/B against the library of the commit before (lib_prev), then the whole GPU suite on the new library
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out gpurun_out/profiles_r04
export TMPDIR=/tmp
R="$PWD"
for rep in 1 2 3; do for lib in lib_prev lib; do
  GSR_LIB="$R/gaussian-splatting_amd/$lib/libgsr_hip.so" timeout 100 python bench.py --steps 30 --warmup 5 --train-steps 40 --no-other-configs --no-cpu-baseline --no-in-flight --no-full-loop --densify-iters 0 --min-warm-seconds 0.3 > gpurun_out/sp_${lib}_$rep.log 2>&1
done; done
python - <<'PY'
import json, statistics
def rows(lib):
    out = []
    for rep in (1, 2, 3):
        try:
            d = json.loads([l for l in open(f"gpurun_out/sp_{lib}_{rep}.log") if l.startswith("{")][-1])
            s = d["stage_ms"]
            out.append({"forward_ms_per_frame": d["ms_per_step"], "preprocess_ms": s.get("preprocess"), "preprocess_bwd_ms": s.get("preprocess_bwd"),
                        "train_iters_per_s": d.get("train_iters_per_s"), "train_iters_per_s_sparse_adam": d.get("train_iters_per_s_sparse_adam"),
                        "train_iters_per_s_depth_supervised": d.get("train_iters_per_s_depth_supervised")})
        except Exception as e:
            print(lib, rep, "failed", e); print(open(f"gpurun_out/sp_{lib}_{rep}.log").read()[-600:])
    return out
a, b = rows("lib_prev"), rows("lib")
for r in a: print("before", r)
for r in b: print("after ", r)
if a and b:
    med = lambda rs, k: statistics.median(r[k] for r in rs if r[k] is not None)
    out = {"what": "same-box interleaved A/B (3 x 2 runs of bench.py --steps 30 --train-steps 40, 1 M Gaussians @1080p): preprocess.hip as ONE object with default "
                   "flags (before) against the three-way build -- fused-SH forward and the backward without the SLP vectoriser, the backward at "
                   "amdgpu_waves_per_eu(3,3) (168 VGPRs, nothing spilled) with 1536 workgroups, split-SH forward unchanged (after)",
           "before": a, "after": b,
           "median": {k: {"before": med(a, k), "after": med(b, k)} for k in a[0]}}
    json.dump(out, open("gpurun_out/profiles_r04/r04_ab_preprocess_split_build.json", "w"), indent=1)
    print("median:", out["median"])
PY
echo "A/B done at $SECONDS s"
timeout 330 python -m pytest tests -m gpu -q -x > gpurun_out/split_pytest.log 2>&1; echo "pytest rc=$? at $SECONDS s"; tail -4 gpurun_out/split_pytest.log | cut -c1-200
cp gpurun_out/parity_report.json gpurun_out/profiles_r04/r04_parity_report.json 2>/dev/null
