cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r2z_pytest.log 2>&1
echo "pytest rc=$?"; tail -1 gpurun_out/r2z_pytest.log
bash tools/gpu_ab.sh "tile_sort_mode=0" 2>&1 | tail -2
