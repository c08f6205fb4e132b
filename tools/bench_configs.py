#!/usr/bin/env python3
"""Times BASELINE.json configs[2] and configs[3] on one GPU (not the driver's bench
contract -- bench.py covers configs[1]; this feeds DESIGN.md section 6).

cfg 2: photometric multi-scale (max_scale 2) then one geometric-consistency Run
cfg 3: the shipped config.yaml schedule: photometric multi-scale -> geom Run + host
       planar prior + prior Run -> geom Run   (reference src/main.cpp:20-41)
Source depth maps = ground truth + 0.5 % noise, held fixed.
"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (one HIP runtime, see engine.load)

pm = importlib.import_module("mp-mvs_amd")
engine = importlib.import_module("mp-mvs_amd.engine")
hostlib = importlib.import_module("mp-mvs_amd.hostlib")
W, H, V = 1600, 1200, 8
PRIOR_SEED_OFFSET = 0x9E3779B97F4A7C15


def main():
    sc = pm.synth.make_problem_scene(W, H, n_src=V, quantize=True)
    cams, imgs = sc.problem(0, list(range(1, V + 1)))
    rng = np.random.default_rng(7)
    src_depths = [sc.views[i].gt_depth * (1.0 + 0.005 * rng.standard_normal((H, W))).astype(np.float32) for i in range(1, V + 1)]
    gt = sc.views[0].gt_depth
    dmin, dmax = pm.synth.kernel_depth_range(cams[0])
    h = engine.create(0)
    h.set_views(cams, imgs)
    h.set_src_depths(src_depths)
    out = {}
    for cfg, geom_iters, geom_pp in (("cfg2", 1, False), ("cfg3", 2, True)):
        t = {"gpu_run": 0.0, "host_prior": 0.0, "transfers": 0.0}
        t0 = time.perf_counter()

        def run(p, seed):
            a = time.perf_counter()
            h.run(p, seed)
            t["gpu_run"] += time.perf_counter() - a

        p = pm.PatchMatchParams(num_images=V + 1, depth_min=float(dmin), depth_max=float(dmax), max_scale=2)
        run(p, 1)
        for g in range(geom_iters):
            planar = geom_pp and g != geom_iters - 1
            p.geom_consistency, p.planar_prior, p.max_iterations, p.geomPlanarPrior = True, False, 2, planar
            run(p, 2 + g)
            if planar:
                a = time.perf_counter()
                planes, costs, gc = h.get(geom=True)
                t["transfers"] += time.perf_counter() - a
                a = time.perf_counter()
                prior, mask, ntri = hostlib.build_prior(cams[0], planes, costs, gc, True, p.depth_min, p.depth_max)
                t["host_prior"] += time.perf_counter() - a
                a = time.perf_counter()
                h.set_prior(prior, mask)
                t["transfers"] += time.perf_counter() - a
                p.geom_consistency, p.planar_prior, p.max_iterations = False, True, 3
                run(p, 2 + g + PRIOR_SEED_OFFSET)
                out[cfg + "_triangles"] = ntri
                out[cfg + "_mask_fraction"] = round(float((mask > 0).mean()), 4)
        a = time.perf_counter()
        planes, costs = h.get()
        t["transfers"] += time.perf_counter() - a
        wall = time.perf_counter() - t0
        rel = np.abs(planes[..., 3] - gt) / gt
        out[cfg] = {"wall_s": round(wall, 4), "Mpix_per_s_wall": round(W * H / wall / 1e6, 2),
                    "Mpix_per_s_gpu_runs_only": round(W * H / t["gpu_run"] / 1e6, 2),
                    "seconds": {k: round(v, 4) for k, v in t.items()},
                    "within_1pct_of_gt": round(float((rel < 0.01).mean()), 4), "mean_cost": round(float(costs.mean()), 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
