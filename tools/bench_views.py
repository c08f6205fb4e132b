#!/usr/bin/env python3
"""update-kernel throughput as a function of the number of source views (800x600)"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
pm = importlib.import_module("mp-mvs_amd")
engine = importlib.import_module("mp-mvs_amd.engine")
W, H = 800, 600
sc = pm.synth.make_problem_scene(W, H, n_src=8, quantize=True)
out = {}
for V in (4, 8, 9, 12, 16, 20, 32):
    ids = [1 + (i % 8) for i in range(V)]
    cams, imgs = sc.problem(0, ids)
    dmin, dmax = pm.synth.kernel_depth_range(cams[0])
    h = engine.create(0)
    h.set_views(cams, imgs)
    h.set_profiling(True)
    p = pm.PatchMatchParams(num_images=V + 1, depth_min=float(dmin), depth_max=float(dmax), max_scale=0)
    h.run(p, 1)
    h.run(p, 2)
    ms, cnt = h.kernel_times()
    upd = (ms[1] + ms[2]) / (cnt[1] + cnt[2])
    out[f"V{V}"] = {"update_ms": round(upd, 3), "ns_per_eval": round(upd * 1e6 / (W * H / 2 * 14 * V), 3)}
print(json.dumps(out))
