#!/bin/bash
# round artefacts, part B: rocprofv3 kernel trace + PMC passes for cfg 1 (fp16 texels), fp32 texels, 20 views; texture path counters
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
T=${1:-r02z}
bash tools/profile_gpu.sh ${T} > /dev/null 2>&1; head -12 gpurun_out/prof_${T}/summary.txt
bash tools/profile_gpu.sh ${T}_f32 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --float-images > /dev/null 2>&1; head -4 gpurun_out/prof_${T}_f32/summary.txt
bash tools/profile_gpu.sh ${T}_v20 tools/run_views.py 20 > /dev/null 2>&1; head -6 gpurun_out/prof_${T}_v20/summary.txt
bash tools/profile_ta.sh > gpurun_out/prof_${T}/ta.txt 2>&1; tail -8 gpurun_out/prof_${T}/ta.txt
