#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2k_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2k_pytest.log
tail -3 gpurun_out/r2k_pytest.log
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r2k_bench.log 2>&1
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r2k_bench.log") if l.startswith("{")][-1])
print("default:", d["value"], d["train_iters_per_s"], d["train_iters_per_s_sparse_adam"], d["train_iters_per_s_l1"], d["stage_ms"])
PY
BENCH_EXTRA="--train-steps 8" bash tools/gpu_kstats.sh 2>&1 | head -26
timeout 900 python tools/gpu_band_model.py > gpurun_out/r2k_band.log 2>&1
python - <<'PY'
import json
d=json.load(open("gpurun_out/band_model.json"))
for name,r in d.items():
    print(name, "R", r["R"], "1 GPU", r["1"]["frame_ms"], "sum", round(sum(r["1"]["stage_ms"].values()),4), r["1"]["stage_ms"])
    for G in ("2","4","8"):
        x=r[G]; print("  G",G,"slowest band",x["slowest_band_ms"],"sum-of-stages",x["slowest_band_sum_of_stages_ms"],"gather us",x["allgather_model_us"],"speedup overlapped",x["speedup_if_gather_overlapped"],"serial",x["speedup_if_gather_serial"], "band0", x["bands"][0]["stage_ms"])
PY
