#!/bin/bash
# Measurement builds of the HIP library under build/ (never shipped):  tools/build_variant.sh NAME [hipcc flags...]
#   tools/build_variant.sh wt -DPM_DBG_WAVETIME        ->  build/libmpmvs_hip_wt.so   (use with MPMVS_HIP_LIB / tools/bench_variants.sh)
cd "$(dirname "$0")/.." || exit 1
name=$1; shift
mkdir -p build
exec make -C mp-mvs_amd/csrc variant OUT=../../build/libmpmvs_hip_$name.so EXTRA="$*" >/dev/null
