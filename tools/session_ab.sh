#!/bin/bash
# A/B of measurement builds under build/: tools/session_ab.sh <tag> v1 v2 ...   (each is run twice, interleaved)
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
T=$1; shift
O=gpurun_out/$T; mkdir -p $O
tools/bench_variants.sh "$@" "$@" 2>&1 | tee $O/variants.txt
