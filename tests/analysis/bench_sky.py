"""mpmvs_sky_bilateral at 1600x1200 (the reference's Pixel_bilateral_filter): kernel time, instruction-rate estimate, oracle time."""
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from test_sky_cpu import sky_scene

fusion = importlib.import_module("mp-mvs_amd.fusion")
W, H = 1600, 1200
img, coarse, _ = sky_scene(W, H)
fusion.sky_bilateral(img, coarse)
t0 = time.perf_counter()
out = fusion.sky_bilateral(img, coarse)
wall = time.perf_counter() - t0
ms = fusion.last_sky_kernel_ms()
taps = W * H * 37 * 37
rep = {"size": [W, H], "kernel_ms": round(ms, 3), "wall_ms": round(wall * 1e3, 1), "Gtaps_per_s": round(taps / ms / 1e6, 1)}
if "--oracle" in sys.argv:
    ob = importlib.import_module("oracle.binding")
    t0 = time.perf_counter()
    ref = ob.sky_bilateral(img[:300], coarse[:300])
    rep["oracle_s_per_frame"] = round((time.perf_counter() - t0) * H / 300, 2)
    rep["oracle_threads"] = ob.num_threads()
    rep["bit_exact_rows_0_281"] = bool(np.array_equal(ref[:282], out[:282]))
print(json.dumps(rep))
