set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4o
python -m pytest tests -m gpu -x -q > gpurun_out/r4o/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4o/pytest.log
python bench.py --no-secondary --no-cpu-baseline > /dev/null 2>&1
bash tools/profile_gpu.sh r04z > gpurun_out/r4o/profile.log 2>&1; echo "profile rc=$?"
bash tools/trace_kernels.sh r04 bench.py --steps 5 --warmup 1 --no-secondary --no-cpu-baseline --no-overlap-phase > gpurun_out/r4o/trace.log 2>&1
python bench.py > gpurun_out/r4o/bench.json 2> gpurun_out/r4o/bench.err; echo "bench rc=$?"
