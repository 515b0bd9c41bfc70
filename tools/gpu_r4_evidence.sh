#!/bin/bash
# round 4, evidence call on the final sources: the two SQ counter passes (instruction mix / wait states), the same-box A/B against the
# round-3 library (tools/build_prev_lib.sh eb6bc02 -> lib_prev), a bounded run of the whole-operator fuzz
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
TAG=r04
B="--no-cpu-baseline --no-other-configs --no-in-flight --no-full-loop --densify-iters 0 --min-warm-seconds 0.2"
# ---- experiment: blend backward at 7 waves per SIMD (lib_occ7 = same sources + -DGSR_BWD_OCC=amdgpu_waves_per_eu(7,7): 78 -> 72 VGPRs, three
# registers spilled OUTSIDE the walk loop); interleaved train-step A/B, then the backward parity tests on that library
if [ -f gaussian-splatting_amd/lib_occ7/libgsr_hip.so ]; then
  for rep in 1 2 3; do
    for lib in lib lib_occ7; do
      GSR_LIB="$R/gaussian-splatting_amd/$lib/libgsr_hip.so" timeout 120 python bench.py --steps 20 --warmup 5 --train-steps 40 --no-other-configs --no-cpu-baseline --no-in-flight --no-full-loop --densify-iters 0 --min-warm-seconds 0.3 > gpurun_out/occ_${lib}_$rep.log 2>&1
      python - "$lib" "$rep" "gpurun_out/occ_${lib}_$rep.log" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
    print(f"{sys.argv[1]:9s} rep {sys.argv[2]}: train {d.get('train_iters_per_s')} it/s, render_bwd {d['stage_ms'].get('render_bwd')} ms, depth-supervised {d.get('train_iters_per_s_depth_supervised')}")
except Exception as e:
    print("bench failed", e); print(open(sys.argv[3]).read()[-800:])
PY
    done
  done
  python - <<'PY'
import json, statistics, os
def get(lib):
    rows = []
    for rep in (1, 2, 3):
        try:
            d = json.loads([l for l in open(f"gpurun_out/occ_{lib}_{rep}.log") if l.startswith("{")][-1])
            rows.append({"train_iters_per_s": d.get("train_iters_per_s"), "render_bwd_ms": d["stage_ms"].get("render_bwd"),
                         "train_iters_per_s_depth_supervised": d.get("train_iters_per_s_depth_supervised")})
        except Exception:
            pass
    return rows
a, b = get("lib"), get("lib_occ7")
if a and b:
    os.makedirs("gpurun_out/profiles_r04", exist_ok=True)
    out = {"what": "same-box interleaved A/B of the blend backward's occupancy: default build (78 VGPRs, 6 waves per SIMD) against "
                   "-DGSR_BWD_OCC=__attribute__((amdgpu_waves_per_eu(7,7))) (72 VGPRs, 3 spilled outside the walk loop; LDS then caps a CU at 26 waves); "
                   "bench.py --train-steps 40, 1 M Gaussians @1080p",
           "default": a, "waves_per_eu_7": b,
           "render_bwd_ms_median": {"default": statistics.median(r["render_bwd_ms"] for r in a), "waves_per_eu_7": statistics.median(r["render_bwd_ms"] for r in b)},
           "train_iters_per_s_median": {"default": statistics.median(r["train_iters_per_s"] for r in a), "waves_per_eu_7": statistics.median(r["train_iters_per_s"] for r in b)}}
    json.dump(out, open("gpurun_out/profiles_r04/r04_ab_bwd_occupancy.json", "w"), indent=1)
    print("occupancy A/B:", out["render_bwd_ms_median"], out["train_iters_per_s_median"])
PY
  echo "occ A/B done at $SECONDS s"
  GSR_LIB="$R/gaussian-splatting_amd/lib_occ7/libgsr_hip.so" timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_next_rows.py -m gpu -q -x -k "backward or grad or train or reproducible" > gpurun_out/occ7_pytest.log 2>&1; echo "occ7 pytest rc=$? at $SECONDS s"; tail -3 gpurun_out/occ7_pytest.log | cut -c1-200
fi
cd /tmp
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d "$R/gpurun_out/prof_sq" -o r1 -- python "$R/bench.py" --steps 6 --warmup 2 --train-steps 4 $B > "$R/gpurun_out/p_prof_sq.log" 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU --output-format csv -d "$R/gpurun_out/prof_sq2" -o r1 -- python "$R/bench.py" --steps 6 --warmup 2 --train-steps 4 $B > "$R/gpurun_out/p_prof_sq2.log" 2>&1
cd "$R"
echo "sq passes done at $SECONDS s"
python - "$TAG" <<'PY'
# per-kernel means of the SQ counters -> gpurun_out/profiles_rNN/rNN_pmc_sq_per_kernel.csv + the SQ_* fields of pmc_latest.json
import collections, csv, glob, json, os, re, sys
tag = sys.argv[1]
out = f"gpurun_out/profiles_{tag}"
os.makedirs(out, exist_ok=True)
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)
sq = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in ("prof_sq", "prof_sq2"):
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = sq[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
NAMES = ['SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES',
         'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_LDS_BANK_CONFLICT']
ks = [k for k in sq if not (k.startswith("at::") or k.startswith("__amd"))]
ks.sort(key=lambda k: -sq[k].get('SQ_WAVE_CYCLES', [1, 0])[1] / max(1, sq[k].get('SQ_WAVE_CYCLES', [1, 0])[0]))
with open(f"{out}/{tag}_pmc_sq_per_kernel.csv", "w") as f:
    f.write("kernel," + ",".join(NAMES) + "\n")
    for k in ks:
        f.write('"' + k + '",' + ",".join(f"{sq[k][n][1] / sq[k][n][0]:.0f}" if n in sq[k] else "" for n in NAMES) + "\n")
p = json.load(open("profiles/pmc_latest.json"))
n_upd = 0
for k, v in p.get("kernels", {}).items():
    for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
        if k in sq and n in sq[k]:
            v[n] = int(sq[k][n][1] / sq[k][n][0]); n_upd += 1
p["sq_source"] = "SQ_* fields: rocprofv3 SQ passes of the final round-4 sources (tools/gpu_r4_evidence.sh); FETCH / WRITE: tools/gpu_r4_last.sh, same sources"
json.dump(p, open(f"{out}/pmc_latest.json", "w"), indent=1)
print("SQ kernels:", len(ks), "fields updated:", n_upd)
for k in ks[:4]:
    print(" ", k[:50], {n: int(sq[k][n][1] / sq[k][n][0]) for n in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY") if n in sq[k]})
PY
rm -rf gpurun_out/prof_sq gpurun_out/prof_sq2
if [ -f gaussian-splatting_amd/lib_prev/libgsr_hip.so ]; then
  export GSR_ALLOW_ABI_MISMATCH=1
  for rep in 1 2; do for lib in lib lib_prev; do
    GSR_LIB="$R/gaussian-splatting_amd/$lib/libgsr_hip.so" timeout 120 python bench.py --no-other-configs --no-cpu-baseline --no-in-flight --no-full-loop --densify-iters 0 --train-steps 0 --min-warm-seconds 0.5 > gpurun_out/abp_${lib}_$rep.log 2>&1
  done; done
  unset GSR_ALLOW_ABI_MISMATCH
  python - "$TAG" <<'PY'
import json, statistics, sys, os
tag = sys.argv[1]
def ms(lib):
    out = []
    for rep in (1, 2, 3):
        try:
            out.append(json.loads([l for l in open(f"gpurun_out/abp_{lib}_{rep}.log") if l.startswith("{")][-1])["ms_per_step"])
        except Exception:
            pass
    return out
cur, prev = ms("lib"), ms("lib_prev")
if cur and prev:
    os.makedirs(f"gpurun_out/profiles_{tag}", exist_ok=True)
    d = {"what": "python bench.py (forward only, 50 steps) with the current library and with the round-3 library (commit eb6bc02) on ONE box, runs interleaved",
         "round4_ms_per_frame": cur, "round3_ms_per_frame": prev,
         "round4_ms_per_frame_median": statistics.median(cur), "round3_ms_per_frame_median": statistics.median(prev)}
    json.dump(d, open(f"gpurun_out/profiles_{tag}/{tag}_ab_round3.json", "w"), indent=1)
    print("A/B round 3 -> round 4:", prev, "->", cur)
else:
    print("A/B failed"); print(open("gpurun_out/abp_lib_prev_1.log").read()[-1500:])
PY
fi
echo "ab done at $SECONDS s"
FUZZ_SECONDS=${FUZZ_SECONDS:-60} timeout 120 python tools/gpu_fuzz_render.py 400 41 > gpurun_out/fuzz_render.log 2>&1; tail -1 gpurun_out/fuzz_render.log | cut -c1-1500
cp gpurun_out/fuzz_render.json "gpurun_out/profiles_$TAG/${TAG}_fuzz_render.json" 2>/dev/null
echo "all done at $SECONDS s"
