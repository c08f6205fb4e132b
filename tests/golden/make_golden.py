#!/usr/bin/env python3
"""Generates tests/golden/pm_golden_v3.npz from the CPU oracle.

The reference holds no golden vectors for this path and cannot be run here
(DESIGN.md 3.7), so these are regression pins of the oracle's
own outputs on a small seeded scene: inputs (cameras, images, depth maps, prior)
and expected outputs of every kernel kind in all three modes.  Re-run only when
the canonical arithmetic of DESIGN.md section 3 changes on purpose:
    python tests/golden/make_golden.py
"""
import ctypes
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pm = importlib.import_module("mp-mvs_amd")
from oracle import binding as ob  # noqa: E402

W, H, V, SEED = 64, 48, 3, 20240309


def cam_bytes(cam):
    return np.frombuffer(ctypes.string_at(ctypes.addressof(cam), ctypes.sizeof(cam)), np.uint8).copy()


def main():
    out = {}
    for tag, quantize in (("f32", False), ("u8", True)):
        sc = pm.synth.make_problem_scene(W, H, n_src=V, spacing=0.5, rot_deg=2.0, quantize=quantize)
        cams, imgs = sc.problem(0, [1, 2, 3])
        dmin, dmax = pm.synth.kernel_depth_range(cams[0])
        out[f"{tag}_cams"] = np.stack([cam_bytes(c) for c in cams])
        out[f"{tag}_images"] = np.stack(imgs)
        h = ob.create()
        h.set_views(cams, imgs)
        prm = pm.PatchMatchParams(num_images=V + 1, depth_min=float(dmin), depth_max=float(dmax), max_scale=2)
        # T1: NCC of random planes at the three scales
        rng = np.random.default_rng(1)
        n = rng.normal(size=(H, W, 3))
        n[..., 2] = -np.abs(n[..., 2]) - 0.3
        n /= np.linalg.norm(n, axis=-1, keepdims=True)
        depth = rng.uniform(dmin, dmax, size=(H, W))
        u, v = np.meshgrid(np.arange(W), np.arange(H))
        cam = cams[0]
        X = np.stack([depth * (u - cam.K[2]) / cam.K[0], depth * (v - cam.K[5]) / cam.K[4], depth], -1)
        planes = np.concatenate([n, -(n * X).sum(-1)[..., None]], -1).astype(np.float32)
        out[f"{tag}_ncc_planes"] = planes
        for s in range(3):
            out[f"{tag}_ncc_scale{s}"] = h.eval_ncc(prm, planes, s)
        # T2: kernels one by one (photometric)
        h.step(prm, SEED, pm.KIND_INIT, 0, 2, 0)
        out[f"{tag}_init_planes"], out[f"{tag}_init_costs"] = h.get()
        out[f"{tag}_init_sel"] = h.get_selected_views()
        h.step(prm, SEED, pm.KIND_BLACK, 0, 2, 1)
        h.step(prm, SEED, pm.KIND_RED, 0, 2, 2)
        out[f"{tag}_it0_planes"], out[f"{tag}_it0_costs"] = h.get()
        out[f"{tag}_it0_sel"] = h.get_selected_views()
        # T3: whole runs, photometric -> geom -> prior on one context
        h.run(prm, SEED)
        out[f"{tag}_run_planes"], out[f"{tag}_run_costs"] = h.get()
        rng = np.random.default_rng(2)
        depths = np.stack([sc.views[i].gt_depth * (1.0 + 0.005 * rng.standard_normal((H, W))).astype(np.float32) for i in (1, 2, 3)])
        depths[:, ::9, ::7] = 0.0
        out[f"{tag}_src_depths"] = depths
        h.set_src_depths(list(depths))
        prm.geom_consistency = True
        prm.max_iterations = 2
        h.run(prm, SEED + 1)
        gp, gc, gg = h.get(geom=True)
        out[f"{tag}_geom_planes"], out[f"{tag}_geom_costs"], out[f"{tag}_geom_geom"] = gp, gc, gg
        prior = np.zeros((H, W, 4), np.float32)
        prior[..., 2] = -1.0
        prior[..., 0] = 0.05 * rng.standard_normal((H, W))
        nn = prior[..., :3].astype(np.float64)
        nn /= np.linalg.norm(nn, axis=-1, keepdims=True)
        gt = sc.views[0].gt_depth.astype(np.float64)
        Xg = np.stack([gt * (u - cam.K[2]) / cam.K[0], gt * (v - cam.K[5]) / cam.K[4], gt], -1)
        prior[..., :3] = nn
        prior[..., 3] = -(nn * Xg).sum(-1)
        mask = ((rng.uniform(size=(H, W)) < 0.6) * np.arange(1, H * W + 1).reshape(H, W)).astype(np.uint32)
        out[f"{tag}_prior"], out[f"{tag}_mask"] = prior, mask
        h.set_prior(prior, mask)
        prm.geom_consistency = False
        prm.planar_prior = True
        prm.max_iterations = 3
        h.run(prm, SEED + 2)
        out[f"{tag}_prior_planes"], out[f"{tag}_prior_costs"] = h.get()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pm_golden_v3.npz"), **out)
    print("wrote pm_golden_v3.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
