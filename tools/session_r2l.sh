#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r2l; mkdir -p $O
tools/bench_variants.sh idxen nopark park idxen nopark park 2>&1 | tee $O/variants.txt
BENCH_ARGS=--float-images tools/bench_variants.sh nopark park 2>&1 | tee -a $O/variants.txt
VIEWS=8,24 tools/bench_views_variants.sh nopark park 2>&1 | tee $O/views.txt
