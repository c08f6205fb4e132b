#!/bin/bash
# what would a third wave per SIMD buy?  NCC core alone (k_eval_ncc) with 4-byte LDS records (wrong results, same instruction mix):
# half3 = LDS halved (3 blocks per CU fit), half2 = same code with the LDS allocation of the real build (2 blocks per CU)
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-occ}; mkdir -p $O
for v in real half2 half3 half2 half3; do
  lib=build/libmpmvs_hip_$v.so; [ $v = real ] && lib=mp-mvs_amd/csrc/libmpmvs_hip.so
  echo "== $v" | tee -a $O/occ.txt
  MPMVS_HIP_LIB=$PWD/$lib FORMATS=u8 MAPPINGS=0 SCALES=0 python tools/bench_coop.py 2>/dev/null | grep -v "^{" | tee -a $O/occ.txt
done
