#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-shapes}; mkdir -p $O
tools/bench_variants.sh real r8 r32 real r8 r32 2>&1 | tee $O/variants.txt
BENCH_ARGS=--float-images tools/bench_variants.sh real f4 f16 real f4 f16 2>&1 | tee -a $O/variants.txt
