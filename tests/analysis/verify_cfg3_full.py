#!/usr/bin/env python3
"""One-off parity check at BASELINE size: configs[3] (the shipped config.yaml schedule: photometric 3 scales -> geometric Run +
host planar prior + prior Run -> geometric Run) on one 1600x1200 Problem with 8 source views, C++ ProcessProblem mirror on the
HIP path against the same schedule driven on the CPU oracle.  Takes a few minutes of host time."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
pm = importlib.import_module("mp-mvs_amd")
hostlib = importlib.import_module("mp-mvs_amd.hostlib")
from oracle import binding as ob
import bench
from test_pipeline_gpu import oracle_pipeline

cams, imgs, gt = bench.load_scene(pm, 1600, 1200, 8, True)
rng = np.random.default_rng(7)
src_depths = [gt * (1.0 + 0.005 * rng.standard_normal(gt.shape)).astype(np.float32) for _ in range(8)]
t0 = time.perf_counter()
depth, normal, cost = hostlib.run_pipeline(0, cams, imgs, 2, 2, True, True, 4242, src_depths)
t_gpu = time.perf_counter() - t0
ob.set_num_threads(min(16, len(os.sched_getaffinity(0))))
t0 = time.perf_counter()
planes, costs = oracle_pipeline(pm, ob, hostlib, cams, imgs, src_depths, 2, 2, True, True, 4242)
t_cpu = time.perf_counter() - t0
same = bool(np.array_equal(depth, planes[..., 3]) and np.array_equal(normal, planes[..., :3]) and np.array_equal(cost, costs))
print(json.dumps({"config": "cfg3, 1600x1200, 8 src views", "hip_s": round(t_gpu, 3), "oracle_s": round(t_cpu, 1), "oracle_threads": ob.num_threads(),
                  "bit_identical": same, "within_1pct_of_gt": round(float((np.abs(depth - gt) / gt < 0.01).mean()), 4)}))
