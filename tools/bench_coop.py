#!/usr/bin/env python3
"""Per-evaluation time of the NCC core alone (probe kernel k_eval_ncc through mpmvs_eval_ncc_multi): cfg-1 scene, NH hypotheses per pixel x 8 views, true-surface planes with small perturbations."""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (scene cache)


def main():
    pm = importlib.import_module("mp-mvs_amd")
    engine = importlib.import_module("mp-mvs_amd.engine")
    W, H, V = 1600, 1200, 8
    NH = int(os.environ.get("NH", "4"))
    res = {}
    fmts = os.environ.get("FORMATS", "u8,f32").split(",")
    maps = [int(m) for m in os.environ.get("MAPPINGS", "0").split(",")]
    scales = [int(m) for m in os.environ.get("SCALES", "0,2").split(",")]
    for quantize in [f == "u8" for f in fmts]:
        cams, imgs, gt = bench.load_scene(pm, W, H, V, quantize)
        dmin, dmax = pm.synth.kernel_depth_range(cams[0])
        prm = pm.PatchMatchParams(num_images=V + 1, depth_min=float(dmin), depth_max=float(dmax), max_scale=0)
        ctx = engine.create(0)
        ctx.set_views(cams, imgs)
        cam = cams[0]
        rng = np.random.default_rng(5)
        u, v = np.meshgrid(np.arange(W), np.arange(H))
        planes = []
        for h in range(NH):
            depth = gt.astype(np.float64) * rng.uniform(0.97, 1.03, gt.shape)
            n = np.zeros((H, W, 3))
            n[..., 2] = -1.0
            n[..., :2] = 0.2 * rng.normal(size=(H, W, 2))
            n /= np.linalg.norm(n, axis=-1, keepdims=True)
            X = np.stack([depth * (u - cam.K[2]) / cam.K[0], depth * (v - cam.K[5]) / cam.K[4], depth], -1)
            planes.append(np.concatenate([n, -(n * X).sum(-1)[..., None]], -1).astype(np.float32))
        planes = np.stack(planes)
        ref = None
        for scale in scales:
            for mapping in maps:
                best = 1e9
                for rep in range(3):
                    out, ms = ctx.eval_ncc_multi(prm, planes, scale, mapping)
                    best = min(best, ms)
                if ref is None or mapping == 0:
                    ref = out
                same = bool(np.array_equal(out, ref))
                evals = W * H * NH * V
                res[f"{'u8' if quantize else 'f32'}_scale{scale}_map{mapping}"] = {"ms": round(best, 4), "ps_per_eval": round(best * 1e9 / evals, 2), "same_bits": same}
                print(f"{'u8' if quantize else 'f32'} scale {scale} mapping {mapping}: {best:.3f} ms, {best * 1e9 / evals:.1f} ps/eval, same bits {same}", flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
