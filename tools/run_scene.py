#!/usr/bin/env python3
"""End to end on one MI355X: a synthetic scene of nx x ny cameras through the reference's
pass schedule (photometric -> geometric passes with planar prior), then fusion and a
binary PLY -- the whole of reference src/main.cpp:6-55 minus file ingest.

    python tools/run_scene.py --size 800x600 --grid 4x2 --out /tmp/scene.ply
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402

pm = importlib.import_module("mp-mvs_amd")
engine = importlib.import_module("mp-mvs_amd.engine")
sched = importlib.import_module("mp-mvs_amd.schedule")
fusion = importlib.import_module("mp-mvs_amd.fusion")
hostlib = importlib.import_module("mp-mvs_amd.hostlib")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="800x600")
    ap.add_argument("--grid", default="4x2")
    ap.add_argument("--geom-iterations", type=int, default=2)
    ap.add_argument("--workers", type=int, default=3)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    W, H = map(int, a.size.split("x"))
    nx, ny = map(int, a.grid.split("x"))
    t0 = time.perf_counter()
    sc, neigh = pm.synth.make_grid_scene(W, H, nx, ny, quantize=True)
    cams = [v.cam for v in sc.views]
    imgs = [v.image for v in sc.views]
    t_scene = time.perf_counter() - t0
    t0 = time.perf_counter()
    s = sched.SceneScheduler(cams, imgs, neigh, lambda: engine.create(0), device_tensors=True, max_scale=2, workers=a.workers)
    t_upload = time.perf_counter() - t0
    t0 = time.perf_counter()
    res = s.run(geom_iterations=a.geom_iterations, planar_prior=True, geom_planar_prior=True, seed=2024)
    t_mvs = time.perf_counter() - t0
    depths = [res[i][0][..., 3] for i in range(len(cams))]
    normals = [np.ascontiguousarray(res[i][0][..., :3]) for i in range(len(cams))]
    t0 = time.perf_counter()
    cloud, valid, masks = fusion.fuse(cams, [True] * len(cams), depths, normals, imgs, neigh)
    t_fuse = time.perf_counter() - t0
    if a.out:
        hostlib.write_ply(a.out, cloud)
    err = np.abs(cloud[:, 2] - pm.synth.height_field(cloud[:, 0].astype(np.float64), cloud[:, 1].astype(np.float64)))
    acc = [float((np.abs(d - v.gt_depth) / v.gt_depth < 0.01).mean()) for d, v in zip(depths, sc.views)]
    print(json.dumps({"images": len(cams), "size": [W, H], "seconds": {"render_scene": round(t_scene, 2), "upload": round(t_upload, 3), "mvs_passes": round(t_mvs, 3), "fusion": round(t_fuse, 3)},
                      "Mpix_per_s_mvs": round(len(cams) * W * H / t_mvs / 1e6, 2), "depth_within_1pct_of_gt": round(float(np.mean(acc)), 4),
                      "fused_points": int(len(cloud)), "median_point_error": round(float(np.median(err)), 5), "ply": a.out}))


if __name__ == "__main__":
    main()
