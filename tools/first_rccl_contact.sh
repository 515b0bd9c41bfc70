#!/bin/bash
# First contact with RCCL on a multi-GPU node (no round had one: SCALE_r01..r04 were skipped).  ONE call that yields a scaling curve and, if a
# collective misbehaves, says which one:
#   bash tools/first_rccl_contact.sh [N_MAX]        (default: every GPU torch sees, up to 8)
# 1. probe_collectives on 2 ranks (one tiny instance of all_reduce / all_gather / uneven all_gather / all_to_all_single / reduce_scatter, each in
#    its own try / except: diff_gaussian_rasterization/parallel.py) -> gpurun_out/rccl_probe.json
# 2. bench.py at N = 1, 2, 4, 8 (mode C, exact and fixed-capacity exchange; mode A = north_star's all-gather / reduce-scatter wording), 5 warm-up + 20 steps
#    each, a 180 s collective timeout -> gpurun_out/rccl_scale.jsonl (one bench line per run) + a speed-up table on stdout
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
NGPU=$(python -c 'import torch; print(torch.cuda.device_count())')
NMAX=${1:-$NGPU}; [ "$NMAX" -gt 8 ] && NMAX=8
echo "GPUs visible: $NGPU, running up to $NMAX"
if [ "$NGPU" -ge 2 ]; then
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 - <<'PY' 2>&1 | tail -5
import json, os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.getcwd(), "gaussian-splatting_amd"))
rank = int(os.environ["RANK"]); torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ["LOCAL_RANK"])))
from diff_gaussian_rasterization.parallel import probe_collectives
r = probe_collectives(device=torch.device("cuda", int(os.environ["LOCAL_RANK"])))
if rank == 0:
    json.dump(r, open("gpurun_out/rccl_probe.json", "w"), indent=1); print("probe:", r)
dist.barrier(); dist.destroy_process_group()
PY
fi
: > gpurun_out/rccl_scale.jsonl
for n in 1 2 4 8; do
  [ "$n" -le "$NMAX" ] || continue
  for cfg in "--mode C --exchange exact" "--mode C --exchange fixed" "--mode A"; do
    [ "$n" -eq 1 ] && [ "$cfg" != "--mode C --exchange exact" ] && continue
    if [ "$n" -eq 1 ]; then CMD="python bench.py"; else CMD="python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29520 + n)) bench.py"; fi
    GSR_BENCH_TIMEOUT_S=180 timeout 900 $CMD --gpus $n --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-full-loop --densify-iters 0 $cfg > gpurun_out/rccl_n${n}.log 2>&1
    L=$(grep '^{' gpurun_out/rccl_n${n}.log | tail -1)
    if [ -n "$L" ]; then echo "$L" >> gpurun_out/rccl_scale.jsonl; else echo "N=$n $cfg: no bench line"; tail -5 gpurun_out/rccl_n${n}.log; fi
  done
done
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/rccl_scale.jsonl") if l.strip()]
base = next((r for r in rows if r["n_gpus"] == 1), None)
for r in rows:
    c = r["config"]
    print(f'N={r["n_gpus"]} mode {c.get("mode")} exchange {(c.get("exchange") or {}).get("form")}: {r["value"]} Mpix/s, {r["ms_per_step"]} ms/frame, train {r.get("train_iters_per_s")} it/s'
          + (f', speed-up {r["value"] / base["value"]:.2f}x' if base else ""))
PY
