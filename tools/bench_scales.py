#!/usr/bin/env python3
"""per-launch time of the update kernel at window scales 0/1/2 and in geometric mode (1600x1200, V=8)"""
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
if os.environ.get("MPMVS_FORCE_F32"):   # the fp32 texture format on the same 8-bit images
    h.set_texture_format(True)
h.set_views(cams, imgs)
h.set_profiling(True)
p = pm.PatchMatchParams(num_images=V + 1, depth_min=float(dmin), depth_max=float(dmax), max_scale=2)
h.run(p, 1)  # converged state
out = {}
for ms_ in (0, 1, 2):
    p.max_scale = ms_
    h.run(p, 3)
    ms, cnt = h.kernel_times()
    out[f"run_max_scale{ms_}"] = {"update_total_ms": round(ms[1] + ms[2], 3), "launches": cnt[1] + cnt[2], "init_ms": round(ms[0], 3)}
rng = np.random.default_rng(0)
h.set_src_depths([sc.views[i].gt_depth * (1 + 0.005 * rng.standard_normal((H, W))).astype(np.float32) for i in range(1, V + 1)])
p.geom_consistency, p.max_iterations = True, 2
h.run(p, 4)
ms, cnt = h.kernel_times()
out["geom_run"] = {"update_total_ms": round(ms[1] + ms[2], 3), "launches": cnt[1] + cnt[2], "init_ms": round(ms[0], 3)}
# planar-prior mode: prior planes from the true surface normal/depth (fronto-parallel approximation), mask on 60 % of the pixels
u, v = np.meshgrid(np.arange(W), np.arange(H))
gt = sc.views[0].gt_depth
cam = cams[0]
X = np.stack([gt * (u - cam.K[2]) / cam.K[0], gt * (v - cam.K[5]) / cam.K[4], gt], -1)
prior = np.zeros((H, W, 4), np.float32)
prior[..., 2] = -1.0
prior[..., 3] = X[..., 2]
mask = (rng.uniform(size=(H, W)) < 0.6).astype(np.uint32)
h.set_prior(prior, mask)
p.geom_consistency, p.planar_prior, p.max_iterations = False, True, 3
h.run(p, 5)
ms, cnt = h.kernel_times()
out["prior_run"] = {"update_total_ms": round(ms[1] + ms[2], 3), "launches": cnt[1] + cnt[2], "init_ms": round(ms[0], 3)}
print(json.dumps(out))
