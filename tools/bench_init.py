#!/usr/bin/env python3
"""k_init and the update passes from RANDOM planes against the converged ones (cfg 1, 1600x1200, 8 views): HIP-event time of
InitializeScore and of one-iteration chains started from random / from converged planes.  With the measurement builds
(MPMVS_HIP_LIB=build/libmpmvs_hip_noload.so: no gathers; ..._oneline.so: every gather of a wave in one cache line) it sizes what
the scatter of random planes costs the texture path."""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pm = importlib.import_module("mp-mvs_amd")
engine = importlib.import_module("mp-mvs_amd.engine")
cams, imgs, gts = bench.load_views(pm, 1600, 1200, bench.problem_centers(pm, 8), "p8")
imgs = [np.rint(im).astype(np.float32) for im in imgs]
dmin, dmax = pm.synth.kernel_depth_range(cams[0])
h = engine.create(0)
h.set_views(cams, imgs)
h.set_profiling(True)
p = pm.PatchMatchParams(num_images=9, depth_min=float(dmin), depth_max=float(dmax), max_scale=0, max_iterations=1)
out = {}
for rep in range(3):
    h.run(p, 10 + rep)                       # InitializeScore (random planes) + black + red from random planes, chained
    ms, cnt = h.kernel_times()
out["init_random_ms"] = round(ms[pm.KIND_INIT], 4)
out["first_iteration_from_random_planes_ms_per_pass"] = round((ms[pm.KIND_BLACK] + ms[pm.KIND_RED]) / 2, 4)
p3 = pm.PatchMatchParams(num_images=9, depth_min=float(dmin), depth_max=float(dmax), max_scale=0, max_iterations=3)
h.run(p3, 1)
planes, costs = h.get()
# a geometric-style re-start: InitializeScore branch C re-encodes the stored planes (converged), then one iteration
pg = pm.PatchMatchParams(num_images=9, depth_min=float(dmin), depth_max=float(dmax), max_scale=0, max_iterations=1, geom_consistency=True)
h.set_src_depths([g.astype(np.float32) for g in gts[1:]])
for rep in range(3):
    h.set_state(planes, costs)
    h.run(pg, 20 + rep)
    ms, cnt = h.kernel_times()
out["init_converged_ms"] = round(ms[pm.KIND_INIT], 4)
out["one_iteration_from_converged_planes_geometric_ms_per_pass"] = round((ms[pm.KIND_BLACK] + ms[pm.KIND_RED]) / 2, 4)
print(json.dumps(out))
