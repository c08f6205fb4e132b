#!/usr/bin/env python3
"""cfg-4-style throughput on ONE GPU: 8 Problems (3x3 camera grid minus... a 4x2 grid, each with its
nearest neighbours) through the shipped pass schedule, 1 vs 3 host worker threads."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
pm = importlib.import_module("mp-mvs_amd")
engine = importlib.import_module("mp-mvs_amd.engine")
sched = importlib.import_module("mp-mvs_amd.schedule")
W, H = 1600, 1200
sc, neigh = pm.synth.make_grid_scene(W, H, 4, 2, quantize=True)
cams = [v.cam for v in sc.views]
imgs = [v.image for v in sc.views]
out = {"problems": len(neigh), "size": [W, H], "src_views": len(neigh[0])}
for workers in (1, 3):
    s = sched.SceneScheduler(cams, imgs, neigh, lambda: engine.create(0), device_tensors=True, max_scale=2, workers=workers)
    t0 = time.perf_counter()
    s.run(geom_iterations=2, planar_prior=True, geom_planar_prior=True, seed=1)
    dt = time.perf_counter() - t0
    out[f"workers{workers}"] = {"seconds": round(dt, 3), "Mpix_per_s": round(len(neigh) * W * H / dt / 1e6, 2)}
    del s
print(json.dumps(out))
