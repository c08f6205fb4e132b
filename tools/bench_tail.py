#!/usr/bin/env python3
"""Wave-quantisation ("tail") effect of the update launch: per-pixel time at image sizes whose block count (W/32 x H/16, 512 resident
blocks on 256 CUs x 2) is an exact number of rounds vs a fractional one."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
pm = importlib.import_module("mp-mvs_amd")
engine = importlib.import_module("mp-mvs_amd.engine")
out = {}
for (W, H) in ((1600, 1200), (2048, 1024), (2048, 1152), (1024, 1024), (1024, 768)):
    sc = pm.synth.make_problem_scene(W, H, n_src=8, quantize=True)
    cams, imgs = sc.problem(0, list(range(1, 9)))
    dmin, dmax = pm.synth.kernel_depth_range(cams[0])
    h = engine.create(0)
    h.set_views(cams, imgs)
    h.set_profiling(True)
    p = pm.PatchMatchParams(num_images=9, depth_min=float(dmin), depth_max=float(dmax), max_scale=0)
    h.run(p, 1)
    h.run(p, 2)
    ms, cnt = h.kernel_times()
    upd = (ms[1] + ms[2]) / (cnt[1] + cnt[2])
    blocks = ((W + 31) // 32) * ((H + 15) // 16)
    out[f"{W}x{H}"] = {"update_ms": round(upd, 3), "blocks": blocks, "rounds": round(blocks / 512, 2), "ns_per_pixel": round(upd * 1e6 / (W * H / 2), 3)}
    del h
print(json.dumps(out))
