#!/bin/bash
# A/B with a per-run timeout (for builds that add barriers): tools/session_ab_safe.sh <tag> v1 v2 ...
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
T=$1; shift
O=gpurun_out/$T; mkdir -p $O
for rep in 1 2; do for v in "$@"; do
  lib=build/libmpmvs_hip_$v.so; [ $v = real ] && lib=mp-mvs_amd/csrc/libmpmvs_hip.so
  MPMVS_HIP_LIB=$PWD/$lib timeout -k 10 150 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('== $v', d['value'], 'Mpix/s  k_update', d['roofline']['avg_launch_ms'], 'ms  frac', d['roofline']['frac'], ' gt', d['within_1pct_of_gt'])" || { echo "$v failed or timed out"; exit 1; }
done; done 2>&1 | tee $O/variants.txt
