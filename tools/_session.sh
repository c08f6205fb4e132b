set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4p
python bench.py --no-secondary --no-cpu-baseline > /dev/null 2>&1
bash tools/profile_gpu.sh r04z > gpurun_out/r4p/profile.log 2>&1; echo "profile rc=$?"
bash tools/trace_kernels.sh r04 bench.py --steps 5 --warmup 1 --no-secondary --no-cpu-baseline --no-overlap-phase > gpurun_out/r4p/trace.log 2>&1
bash tools/trace_kernels.sh cfg23 tools/bench_configs.py > gpurun_out/r4p/cfg23.log 2>&1
python bench.py > gpurun_out/r4p/bench.json 2> gpurun_out/r4p/bench.err; echo "bench rc=$?"
python tools/kernel_resources.py > gpurun_out/r4p/kernel_resources.txt 2>&1
