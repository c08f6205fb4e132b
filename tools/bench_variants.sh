#!/bin/bash
# bench.py (cfg 1, k_update average) for measurement builds of the library under build/: tools/bench_variants.sh real rows8 ...
cd ${GRAFT_REPO_ROOT:?}
for v in "$@"; do
  lib=build/libmpmvs_hip_$v.so
  [ $v = real ] && lib=mp-mvs_amd/csrc/libmpmvs_hip.so
  MPMVS_HIP_LIB=$PWD/$lib python bench.py --steps 5 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('== $v', d['value'], 'Mpix/s  k_update', d['roofline']['avg_launch_ms'], 'ms  frac', d['roofline']['frac'], ' gt', d['within_1pct_of_gt'])"
done
