#!/bin/bash
# first GPU contact: smoke, parity tests, bench (both render variants), rocprofv3 kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== device"; python -c "import torch;print(torch.cuda.get_device_name(0), torch.version.hip)"; nproc; rocm-smi --showmemuse 2>/dev/null | head -8
  echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -15
} > gpurun_out/a_smoke.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu --tb=short -x 2>&1 | tail -80 > gpurun_out/a_pytest.log
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/a_bench_v0.log 2>&1
timeout 600 python bench.py --steps 30 --warmup 5 --variant 1 --no-cpu-baseline --train-steps 0 > gpurun_out/a_bench_v1.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_a" -o r1a -- python "$OLDPWD/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --train-steps 10 > "$OLDPWD/gpurun_out/a_prof.log" 2>&1 )
ls -R gpurun_out/prof_a | head -30 >> gpurun_out/a_prof.log
tail -5 gpurun_out/a_smoke.log; tail -30 gpurun_out/a_pytest.log; tail -3 gpurun_out/a_bench_v0.log; tail -3 gpurun_out/a_bench_v1.log
