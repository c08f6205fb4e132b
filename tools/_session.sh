set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_parity_gpu.py tests/test_golden_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -2
python tools/bench_launches.py 2>/dev/null | tail -1
bash tools/trace_kernels.sh cfg23 tools/bench_configs.py 2>&1 | head -9
