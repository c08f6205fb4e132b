set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4m
python bench.py --no-secondary --no-cpu-baseline > /dev/null 2>&1
bash tools/profile_gpu.sh r04z > gpurun_out/r4m/profile.log 2>&1; echo "profile rc=$?"
bash tools/trace_kernels.sh r04 bench.py --steps 5 --warmup 1 --no-secondary --no-cpu-baseline --no-overlap-phase > gpurun_out/r4m/trace.log 2>&1
python bench.py > gpurun_out/r4m/bench.json 2> gpurun_out/r4m/bench.err; echo "bench rc=$?"
