#!/bin/bash
# A/B of measurement builds for both texture formats: tools/session_ab2.sh <tag> v1 v2 ...
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
T=$1; shift
O=gpurun_out/$T; mkdir -p $O
tools/bench_variants.sh "$@" "$@" 2>&1 | tee $O/variants.txt
BENCH_ARGS=--float-images tools/bench_variants.sh "$@" 2>&1 | tee -a $O/variants.txt
