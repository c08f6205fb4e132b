#!/usr/bin/env python3
"""update-kernel time per NOMINAL (hypothesis, view) evaluation as a function of the number of source views, 800x600, for the
three Run() modes (photometric / geometric consistency / planar prior) and both texture formats.  Distinct source views on the
5x5 camera grid (bench.problem_centers), not repetitions of 8.  Prints one table and a JSON line."""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import bench  # noqa: E402

pm = importlib.import_module("mp-mvs_amd")
engine = importlib.import_module("mp-mvs_amd.engine")
W, H = 800, 600
VIEWS = [int(v) for v in os.environ.get("VIEWS", "8,12,16,20,24").split(",")]
FORMATS = os.environ.get("FORMATS", "u8,f32").split(",")
cams_all, imgs_all, gts_all = bench.load_views(pm, W, H, bench.problem_centers(pm, 24), "p24")
out = {}
rng = np.random.default_rng(0)
for fmt in FORMATS:
    for V in VIEWS:
        cams = cams_all[:V + 1]
        imgs = [np.rint(im).astype(np.float32) for im in imgs_all[:V + 1]] if fmt == "u8" else imgs_all[:V + 1]
        dmin, dmax = pm.synth.kernel_depth_range(cams[0])
        h = engine.create(0)
        h.set_views(cams, imgs)
        h.set_profiling(True)
        p = pm.PatchMatchParams(num_images=V + 1, depth_min=float(dmin), depth_max=float(dmax), max_scale=0)
        h.run(p, 1)
        row = {}

        def timed(tag, seed):
            h.run(p, seed)
            ms, cnt = h.kernel_times()
            upd = (ms[1] + ms[2]) / (cnt[1] + cnt[2])
            row[tag] = {"update_ms": round(upd, 4), "ns_per_eval": round(upd * 1e6 / (W * H / 2 * 14 * V), 5)}

        timed("photometric", 2)
        h.set_src_depths([gts_all[i] * (1 + 0.005 * rng.standard_normal((H, W))).astype(np.float32) for i in range(1, V + 1)])
        p.geom_consistency, p.max_iterations = True, 2
        timed("geometric", 3)
        u, v = np.meshgrid(np.arange(W), np.arange(H))
        prior = np.zeros((H, W, 4), np.float32)
        prior[..., 2] = -1.0
        prior[..., 3] = gts_all[0]
        h.set_prior(prior, (rng.uniform(size=(H, W)) < 0.6).astype(np.uint32))
        p.geom_consistency, p.planar_prior, p.max_iterations = False, True, 3
        timed("prior", 4)
        out[f"{fmt}_V{V}"] = row
        print(f"{fmt} V={V:2d}: " + "  ".join(f"{k} {r['update_ms']:.3f} ms = {r['ns_per_eval']:.4f} ns/eval" for k, r in row.items()), flush=True)
print(json.dumps(out))
