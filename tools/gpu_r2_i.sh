#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2i_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2i_pytest.log
tail -4 gpurun_out/r2i_pytest.log
GSR_LIB=$PWD/gaussian-splatting_amd/lib_ab/libgsr_hip.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2i_pytest_ab.log 2>&1
echo "pytest_ab rc=$?" >> gpurun_out/r2i_pytest_ab.log
tail -3 gpurun_out/r2i_pytest_ab.log
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r2i_bench.log 2>&1
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r2i_bench.log") if l.startswith("{")][-1])
print("default:", d["value"], d["train_iters_per_s"], d["train_iters_per_s_sparse_adam"], d["train_iters_per_s_l1"], d["stage_ms"])
PY
bash tools/gpu_pmc_sq.sh > /dev/null 2>&1
python tools/pmc_sq_summary.py $(find gpurun_out/prof_sq gpurun_out/prof_sq2 -name "*counter_collection.csv") > gpurun_out/r2i_sq.csv 2>&1
python - <<'PY'
import csv
rows=list(csv.reader(open('gpurun_out/r2i_sq.csv')))
hdr=rows[0]
sel=['kernel','SQ_WAVES','SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_WAVE_CYCLES','SQ_BUSY_CYCLES','SQ_ACTIVE_INST_VALU','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_LDS_BANK_CONFLICT']
idx=[hdr.index(s) for s in sel]
print(' | '.join(s.replace('SQ_','') for s in sel))
for r in rows[1:16]:
    print(' | '.join((r[i][:34] if j==0 else (f"{float(r[i])/1e6:.2f}M" if r[i] else '')) for j,i in enumerate(idx)))
PY
