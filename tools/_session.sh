set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4i
python bench.py > gpurun_out/r4i/bench.json 2> gpurun_out/r4i/bench.err; echo "bench rc=$?"
bash tools/profile_gpu.sh r04z > gpurun_out/r4i/profile.log 2>&1; echo "profile rc=$?"
bash tools/trace_kernels.sh r04 bench.py --steps 5 --warmup 1 --no-secondary --no-cpu-baseline > gpurun_out/r4i/trace.log 2>&1
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r4i/bench_after_profile.json 2>/dev/null
python tools/bench_scales.py > gpurun_out/r4i/scales.json 2>/dev/null
python tools/bench_views.py > gpurun_out/r4i/views.txt 2>/dev/null
python tools/bench_modes_step.py > gpurun_out/r4i/modes_step.json 2>/dev/null
python tools/bench_launches.py > gpurun_out/r4i/launches.json 2>/dev/null
python tools/kernel_resources.py > gpurun_out/r4i/kernel_resources.txt 2>&1
echo done
