#!/usr/bin/env python3
"""Folder flow (the reference's main(): JPEG images + cams + pair.txt in, depth maps and PLY out) on ONE GPU: 8 images of
1600x1200 with 7 source views each through the shipped schedule -- sequential (reference order) vs Jacobi with worker threads."""
import importlib, json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
pm = importlib.import_module("mp-mvs_amd")
hostlib = importlib.import_module("mp-mvs_amd.hostlib")
W, H = 1600, 1200
sc, neigh = pm.synth.make_grid_scene(W, H, 4, 2, quantize=True)
cams = [v.cam for v in sc.views]
imgs = [np.asarray(v.image).astype(np.uint8) for v in sc.views]
out = {"images": len(cams), "size": [W, H], "src_views": len(neigh[0])}
for mode, workers in (("sequential", 0), ("jacobi_w1", 1), ("jacobi_w3", 3), ("jacobi_w4", 4)):
    d = tempfile.mkdtemp(prefix="mpmvs_folder_")
    hostlib.write_dataset(d, cams, imgs, neigh, fmt="jpg", jpeg_options=dict(quality=95))
    t0 = time.perf_counter()
    if workers:
        hostlib.run_folder_jacobi(d, (0,), workers, geom_iterations=2, planar_prior=True, geom_planar_prior=True, max_scale=2, seed=1)
    else:
        hostlib.run_folder(d, 0, geom_iterations=2, planar_prior=True, geom_planar_prior=True, max_scale=2, seed=1)
    t1 = time.perf_counter()
    n = hostlib.fuse_folder(d)
    t2 = time.perf_counter()
    out[mode] = {"depth_maps_s": round(t1 - t0, 3), "fusion_s": round(t2 - t1, 3), "points": n, "Mpix_per_s": round(len(cams) * W * H / (t1 - t0) / 1e6, 2)}
    shutil.rmtree(d, ignore_errors=True)
print(json.dumps(out))
