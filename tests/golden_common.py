"""Replays the committed fixture tests/golden/pm_golden_v3.npz on a PatchMatch
handle (CPU oracle or HIP context) and checks every stored output bit for bit."""
import ctypes
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pm_golden_v3.npz")
SEED = 20240309


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def replay(pm, make_handle, tag):
    z = np.load(GOLDEN)
    cams = []
    for raw in z[f"{tag}_cams"]:
        cam = pm.Camera()
        ctypes.memmove(ctypes.addressof(cam), raw.tobytes(), ctypes.sizeof(cam))
        cams.append(cam)
    imgs = list(z[f"{tag}_images"])
    dmin, dmax = pm.synth.kernel_depth_range(cams[0])
    h = make_handle()
    h.set_views(cams, imgs)
    prm = pm.PatchMatchParams(num_images=len(cams), depth_min=float(dmin), depth_max=float(dmax), max_scale=2)
    for s in range(3):
        assert _same(h.eval_ncc(prm, z[f"{tag}_ncc_planes"], s), z[f"{tag}_ncc_scale{s}"]), f"{tag} ncc scale {s}"
    h.step(prm, SEED, pm.KIND_INIT, 0, 2, 0)
    p, c = h.get()
    assert _same(p, z[f"{tag}_init_planes"]) and _same(c, z[f"{tag}_init_costs"]) and _same(h.get_selected_views(), z[f"{tag}_init_sel"]), f"{tag} init"
    h.step(prm, SEED, pm.KIND_BLACK, 0, 2, 1)
    h.step(prm, SEED, pm.KIND_RED, 0, 2, 2)
    p, c = h.get()
    assert _same(p, z[f"{tag}_it0_planes"]) and _same(c, z[f"{tag}_it0_costs"]) and _same(h.get_selected_views(), z[f"{tag}_it0_sel"]), f"{tag} iteration 0"
    h.run(prm, SEED)
    p, c = h.get()
    assert _same(p, z[f"{tag}_run_planes"]) and _same(c, z[f"{tag}_run_costs"]), f"{tag} photometric run"
    h.set_src_depths(list(z[f"{tag}_src_depths"]))
    prm.geom_consistency = True
    prm.max_iterations = 2
    h.run(prm, SEED + 1)
    p, c, g = h.get(geom=True)
    assert _same(p, z[f"{tag}_geom_planes"]) and _same(c, z[f"{tag}_geom_costs"]) and _same(g, z[f"{tag}_geom_geom"]), f"{tag} geom run"
    h.set_prior(z[f"{tag}_prior"], z[f"{tag}_mask"])
    prm.geom_consistency = False
    prm.planar_prior = True
    prm.max_iterations = 3
    h.run(prm, SEED + 2)
    p, c = h.get()
    assert _same(p, z[f"{tag}_prior_planes"]) and _same(c, z[f"{tag}_prior_costs"]), f"{tag} prior run"
