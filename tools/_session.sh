set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j
python -m pytest tests/test_parity_gpu.py tests/test_golden_gpu.py tests/test_fuzz_gpu.py -x -q > gpurun_out/r4j/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4j/pytest.log
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > /dev/null 2>&1
bash tools/trace_kernels.sh f1 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary | head -8
