#!/usr/bin/env python3
"""One BlackPixelUpdate launch from the SAME converged state in the three modes (photometric / geometric / planar prior),
1600x1200, 8 views: isolates what a mode adds to the launch (tools/bench_scales.py times whole Run()s, whose states differ)."""
import importlib, os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
pm = importlib.import_module("mp-mvs_amd")
engine = importlib.import_module("mp-mvs_amd.engine")
W, H, V = 1600, 1200, 8
sc = pm.synth.make_problem_scene(W, H, n_src=V, quantize=True)
cams, imgs = sc.problem(0, list(range(1, V + 1)))
dmin, dmax = pm.synth.kernel_depth_range(cams[0])
h = engine.create(0)
h.set_views(cams, imgs)
h.set_profiling(True)
p = pm.PatchMatchParams(num_images=V + 1, depth_min=float(dmin), depth_max=float(dmax), max_scale=0)
h.run(p, 1)
planes, costs = h.get()
rng = np.random.default_rng(0)
h.set_src_depths([sc.views[i].gt_depth * (1 + 0.005 * rng.standard_normal((H, W))).astype(np.float32) for i in range(1, V + 1)])
prior = np.zeros((H, W, 4), np.float32)
prior[..., 2] = -1.0
prior[..., 3] = sc.views[0].gt_depth
h.set_prior(prior, (rng.uniform(size=(H, W)) < 0.6).astype(np.uint32))
out = {}
prev = h.kernel_times()[0][pm.KIND_BLACK]   # the per-kind times accumulate until the next Run()
for name, geom, pri in (("photometric", False, False), ("geometric", True, False), ("prior", False, True), ("photometric_again", False, False)):
    ms_all = []
    for rep in range(3):
        h.set_state(planes, costs)
        q = pm.PatchMatchParams(num_images=V + 1, depth_min=float(dmin), depth_max=float(dmax), max_scale=0)
        q.geom_consistency = True     # InitializeScore branch C (re-encode the stored state) in every mode
        h.step(q, 7, pm.KIND_INIT, 0, 0, 0)
        q.geom_consistency, q.planar_prior = geom, pri
        h.step(q, 7, pm.KIND_BLACK, 1, 0, 1)
        ms, cnt = h.kernel_times()
        ms_all.append(ms[pm.KIND_BLACK] - prev)
        prev = ms[pm.KIND_BLACK]
    out[name] = round(min(ms_all), 4)
print(json.dumps(out))
