#!/bin/bash
# Round 5, last GPU call: more randomised cross-checks on the final binaries (bins: 4 x 400 frames + 60 big; whole operator: 150 frames, another seed).
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY'
import json, subprocess, sys, os
tot = {"frames": 0, "oracle_checked": 0, "kinds": {}, "failures": [], "seeds": [], "max_P": 0, "max_R": 0, "seconds": 0.0}
for seed in (61, 62, 63, 64):
    subprocess.run([sys.executable, "tools/gpu_fuzz_bins.py", "400", str(seed)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
    d = json.load(open("gpurun_out/fuzz_bins.json"))
    tot["frames"] += d["frames"]; tot["oracle_checked"] += d["oracle_checked"]; tot["failures"] += d["failures"]; tot["seeds"].append(d["seed"])
    tot["max_P"] = max(tot["max_P"], d["max_P"]); tot["max_R"] = max(tot["max_R"], d["max_R"]); tot["seconds"] += d["seconds"]
    for k, v in d["kinds"].items():
        tot["kinds"][k] = tot["kinds"].get(k, 0) + v
json.dump(tot, open("gpurun_out/r05_fuzz_bins_1600.json", "w"))
print("bins fuzz:", {k: v for k, v in tot.items() if k != "failures"}, "failures:", len(tot["failures"]))
env = dict(os.environ, FUZZ_BIG="1")
subprocess.run([sys.executable, "tools/gpu_fuzz_bins.py", "60", "65"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=400, env=env)
d = json.load(open("gpurun_out/fuzz_bins.json")); json.dump(d, open("gpurun_out/r05_fuzz_bins_big60.json", "w"))
print("big bins fuzz:", {k: v for k, v in d.items() if k != "failures"}, "failures:", len(d["failures"]))
PY
timeout 420 python tools/gpu_fuzz_render.py 150 71 > gpurun_out/r5c10_render.log 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/fuzz_render_71.json"))
print("render fuzz seed 71: frames", d["frames"], "failures", [(f["it"], f["kind"], f.get("error", "")[:80], f.get("fp64")) for f in d["failures"]], "adjudicated", d.get("adjudicated_by_fp64"), "worst image", d["worst_image_err"], "worst grad", d["worst_grad_err"])
PY
