#!/usr/bin/env python3
"""Stage times of the C++ ProcessProblem (mp-mvs_amd/host) for one 1600x1200 Problem with 8 source views through the shipped
config.yaml schedule (photometric -> geom + planar prior -> geom): run with MPMVS_HOST_TIMING=1 to get the per-stage lines."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MPMVS_HOST_TIMING", "1")
import torch  # noqa: F401
pm = importlib.import_module("mp-mvs_amd")
hostlib = importlib.import_module("mp-mvs_amd.hostlib")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (scene cache)

cams, imgs, gt = bench.load_scene(pm, 1600, 1200, 8, True)
rng = np.random.default_rng(7)
src_depths = [gt * (1.0 + 0.005 * rng.standard_normal(gt.shape)).astype(np.float32) for _ in range(8)]   # stand-ins of the right size
for rep in range(int(os.environ.get("REPS", "2"))):
    t0 = time.perf_counter()
    hostlib.run_pipeline(0, cams, imgs, 2, 2, True, True, 5, src_depths)
    print(f"run_pipeline wall {time.perf_counter() - t0:.3f} s", file=sys.stderr)
