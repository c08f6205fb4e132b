#!/bin/bash
# What bounds the NCC core?  The probe of tools/bench_coop.py (one thread per pixel, u8 textures, window scale 0) with
# measurement builds of the library under build/ (made with -D flags): e.g. noload = no gathers at
# all (VALU + LDS only; computes garbage), addrmask = every gather of a wave in one cache line.
# usage: tools/bench_bound.sh real noload scalar ...
cd ${GRAFT_REPO_ROOT:?}
mkdir -p gpurun_out
for v in "$@"; do
  lib=build/libmpmvs_hip_$v.so
  [ $v = real ] && lib=mp-mvs_amd/csrc/libmpmvs_hip.so
  echo -n "== $v: "
  MPMVS_HIP_LIB=$PWD/$lib NH=4 FORMATS=${FORMATS:-u8} MAPPINGS=${MAPPINGS:-0} SCALES=0 python tools/bench_coop.py 2>/dev/null | grep -E "^(u8|f32) scale 0" | tr '\n' ';'
  echo
done
