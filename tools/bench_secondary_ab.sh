#!/bin/bash
# full bench.py line (cfg 1 + secondary: cfg 2, cfg 3, fp32 textures, q8 twin, 20 views) for measurement builds under build/: one summary row each
cd ${GRAFT_REPO_ROOT:?}
for v in "$@"; do
  lib=build/libmpmvs_hip_$v.so
  [ $v = real ] && lib=mp-mvs_amd/csrc/libmpmvs_hip.so
  MPMVS_HIP_LIB=$PWD/$lib python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-overlap-phase 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['secondary']
print('== $v cfg1', d['value'], 'k_update', d['roofline']['avg_launch_ms'], '| cfg2', s['cfg2']['Mpix_per_s'], s['cfg2']['k_update_avg_ms_all_modes'], '| cfg3', s['cfg3']['Mpix_per_s'], s['cfg3']['k_update_avg_ms_all_modes'], 'mirror', s['cfg3']['via_process_problem_mirror_s'], '| fp32', s['cfg1_fp32_textures']['k_update_avg_ms'], '| 20 views', s['cfg1_20_views']['k_update_avg_ms'], s['cfg1_20_views']['ns_per_nominal_evaluation'])"
done
