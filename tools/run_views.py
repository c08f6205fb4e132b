#!/usr/bin/env python3
"""one Problem with V source views through a photometric, a geometric and a planar-prior Run() (800x600): the workload behind the
many-view rows of profiles/ (tools/profile_gpu.sh r02v20 tools/run_views.py 20)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import bench  # noqa: E402

pm = importlib.import_module("mp-mvs_amd")
engine = importlib.import_module("mp-mvs_amd.engine")
V = int(sys.argv[1]) if len(sys.argv) > 1 else 20
fmt = sys.argv[2] if len(sys.argv) > 2 else "u8"
W, H = 800, 600
cams, imgs, gts = bench.load_views(pm, W, H, bench.problem_centers(pm, 24), "p24")
cams, imgs = cams[:V + 1], imgs[:V + 1]
if fmt == "u8":
    imgs = [np.rint(im).astype(np.float32) for im in imgs]
dmin, dmax = pm.synth.kernel_depth_range(cams[0])
h = engine.create(0)
h.set_views(cams, imgs)
p = pm.PatchMatchParams(num_images=V + 1, depth_min=float(dmin), depth_max=float(dmax), max_scale=0)
h.run(p, 1)
rng = np.random.default_rng(0)
h.set_src_depths([gts[i] * (1 + 0.005 * rng.standard_normal((H, W))).astype(np.float32) for i in range(1, V + 1)])
p.geom_consistency, p.max_iterations = True, 2
h.run(p, 2)
prior = np.zeros((H, W, 4), np.float32)
prior[..., 2] = -1.0
prior[..., 3] = gts[0]
h.set_prior(prior, (rng.uniform(size=(H, W)) < 0.6).astype(np.uint32))
p.geom_consistency, p.planar_prior, p.max_iterations = False, True, 3
h.run(p, 3)
print("ok", V, fmt)
