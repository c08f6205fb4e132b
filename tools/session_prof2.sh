#!/bin/bash
# PMC comparison of two builds of the library under build/
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
for v in "$@"; do
  MPMVS_HIP_LIB=$PWD/build/libmpmvs_hip_$v.so bash tools/profile_gpu.sh cmp_$v > /dev/null 2>&1
  echo "=== $v"; grep -A12 "k_update" gpurun_out/prof_cmp_$v/summary.txt | grep -v "k_init\|k_filter" | head -80
done
