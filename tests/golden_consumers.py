"""Replay of tests/golden/consumers_golden_v1.npz (fusion, sky filter, JPEG) against an implementation."""
import ctypes
import importlib
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "consumers_golden_v1.npz")


def load():
    return np.load(PATH)


def fusion_inputs(pm, z):
    cams = []
    for raw in z["fuse_cams"]:
        cam = pm.Camera()
        ctypes.memmove(ctypes.addressof(cam), raw.tobytes(), ctypes.sizeof(cam))
        cams.append(cam)
    n = len(cams)
    sky = [z["fuse_sky"][k] if z["fuse_sky_present"][k] else None for k in range(n)]
    return cams, list(z["fuse_depths"]), list(z["fuse_normals"]), list(z["fuse_colors"]), sky, [list(r) for r in z["fuse_neigh"]]


def replay_fusion(pm, fuse_fn, z):
    """fuse_fn(cams, estimate, depths, normals, colors, sources, use_dynamic=..., sky=...) -> (cloud, valid, masks)"""
    cams, depths, normals, cols, sky, neigh = fusion_inputs(pm, z)
    for tag, dyn, use_sky in (("dyn", True, False), ("static", False, False), ("sky", True, True)):
        cloud, valid, masks = fuse_fn(cams, [True] * len(cams), depths, normals, cols, neigh, use_dynamic=dyn, sky=sky if use_sky else None)
        assert np.array_equal(cloud, z[f"fuse_{tag}_cloud"]), f"fusion {tag}: points"
        assert np.array_equal(np.stack(valid), z[f"fuse_{tag}_valid"]) and np.array_equal(np.stack(masks), z[f"fuse_{tag}_masks"]), f"fusion {tag}: maps"
