#!/bin/bash
# tools/bench_views.py for measurement builds under build/: tools/bench_views_variants.sh real geomcall ...
cd ${GRAFT_REPO_ROOT:?}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  lib=build/libmpmvs_hip_$v.so
  [ $v = real ] && lib=mp-mvs_amd/csrc/libmpmvs_hip.so
  echo "== $v"
  MPMVS_HIP_LIB=$PWD/$lib python tools/bench_views.py 2>/dev/null | grep -v "^{"
done
