#!/usr/bin/env python3
"""how many source views carry weight after the Monte-Carlo view selection, per pixel and per 64-pixel wave patch,
as a function of V (distinct views on a 5x5 camera grid), and the update-kernel time (800x600)"""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
pm = importlib.import_module("mp-mvs_amd")
engine = importlib.import_module("mp-mvs_amd.engine")
W, H = 800, 600
sc, neigh = pm.synth.make_grid_scene(W, H, 5, 5, spacing=0.15, quantize=True)
ref = 12
order = sorted(range(25), key=lambda j: (np.linalg.norm(np.asarray(sc.views[j].C) - np.asarray(sc.views[ref].C)), j))[1:]
out = {}
for V in (8, 12, 16, 20, 24):
    ids = order[:V]
    cams = [sc.views[ref].cam] + [sc.views[j].cam for j in ids]
    imgs = [sc.views[ref].image] + [sc.views[j].image for j in ids]
    dmin, dmax = pm.synth.kernel_depth_range(cams[0])
    h = engine.create(0)
    h.set_views(cams, imgs)
    h.set_profiling(True)
    p = pm.PatchMatchParams(num_images=V + 1, depth_min=float(dmin), depth_max=float(dmax), max_scale=0)
    h.run(p, 1)
    h.run(p, 2)
    ms, cnt = h.kernel_times()
    upd = (ms[1] + ms[2]) / (cnt[1] + cnt[2])
    sel = h.get_selected_views()
    pc = np.array([bin(int(x)).count("1") for x in sel.ravel()]).reshape(sel.shape)
    un, mx = [], []
    for y in range(0, H - 16, 16):
        for x in range(0, W - 16, 16):
            blk = sel[y:y + 16, x:x + 16]     # one colour of a 16x16 area ~ the 64 pixels of a wave
            un.append(bin(int(np.bitwise_or.reduce(blk.ravel()))).count("1"))
            mx.append(int(pc[y:y + 16, x:x + 16].max()))
    out[f"V{V}"] = {"update_ms": round(upd, 3), "ns_per_eval_nominal": round(upd * 1e6 / (W * H / 2 * 14 * V), 3),
                    "selected_per_pixel": round(float(pc.mean()), 2), "selected_per_wave_union": round(float(np.mean(un)), 2), "selected_per_wave_max": round(float(np.mean(mx)), 2)}
print(json.dumps(out))
