set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
python -m pytest tests/test_parity_gpu.py -x -q -k "pipelined or run_get" > gpurun_out/r4h/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4h/pytest.log

