"""Host planar-prior construction (SURVEY a-16, reference src/PatchMatch.cpp:532-604, :723-853) against INDEPENDENT fixtures:
tests/golden/prior_golden_v1.npz was produced by tests/golden/make_prior_golden.py from scipy.spatial.Delaunay (Qhull) and
brute-force numpy/python restatements, without importing this repository -- so the product is no longer compared with itself."""
import importlib
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pm = importlib.import_module("mp-mvs_amd")
hostlib = importlib.import_module("mp-mvs_amd.hostlib")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "prior_golden_v1.npz"))


def make_cam(K, w, h):
    cam = pm.Camera()
    for i in range(9):
        cam.K[i] = float(K[i])
        cam.R[i] = 1.0 if i in (0, 4, 8) else 0.0
    cam.width, cam.height = int(w), int(h)
    return cam


def tri_key(tris_xy):
    """order- and orientation-independent key of triangles given as [n][3][2] coordinates"""
    t = np.asarray(tris_xy, np.int64)
    code = t[..., 1] * 100000 + t[..., 0]
    return set(map(tuple, np.sort(code, axis=1)))


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("geomprior", [False, True])
def test_vertices_equal_bruteforce(gold, tag, geomprior):
    costs, geom = gold[f"vert_{tag}_costs"], gold[f"vert_{tag}_geom"]
    got = hostlib.triangulate_vertices(costs, geom, geomprior)
    want = gold[f"vert_{tag}_geomprior" if geomprior else f"vert_{tag}_plain"]
    assert np.array_equal(got, want)   # same vertices in the same (cell row-major) order


def test_delaunay_equals_qhull_on_general_position(gold):
    w, h = gold["del_u_size"]
    pts = gold["del_u_points"]
    got = hostlib.delaunay(int(w), int(h), pts)
    want = pts[gold["del_u_tris"]]
    assert len(got) == len(want)
    assert tri_key(got) == tri_key(want)


def test_delaunay_jittered_grid_interior_equals_qhull(gold):
    w, h = gold["del_j_size"]
    pts = gold["del_j_points"]
    got = hostlib.delaunay(int(w), int(h), pts)
    hull = set(map(tuple, pts[gold["del_j_on_hull"]].tolist()))
    interior = [t for t in got.tolist() if not any(tuple(p) in hull for p in t)]
    assert tri_key(interior) == tri_key(pts[gold["del_j_tris_interior"]])


def test_raster_last_triangle_wins(gold):
    w, h = gold["ras_size"]
    cam = make_cam(gold["ras_K"], w, h)
    planes = np.zeros((h, w, 4), np.float32)
    planes[..., 3] = gold["ras_depth"]
    # an unbounded depth range leaves the raw labels
    _, label, _ = hostlib.prior_from_triangles(cam, gold["ras_tri_pts"], planes, -3.0e38, 3.0e38)
    assert np.array_equal(label, gold["ras_label"])


def test_planes_equal_svd_null_vector(gold):
    w, h = gold["ras_size"]
    cam = make_cam(gold["ras_K"], w, h)
    planes = np.zeros((h, w, 4), np.float32)
    planes[..., 3] = gold["ras_depth"]
    _, _, pl = hostlib.prior_from_triangles(cam, gold["ras_tri_pts"], planes, -3.0e38, 3.0e38)
    want = gold["ras_planes"]
    assert pl.shape == want.shape
    assert np.abs(pl.astype(np.float64) - want).max() < 1e-5
    assert np.all(pl[:, 3] > 0)                                        # reference :746-752
    assert np.abs(np.linalg.norm(pl[:, :3].astype(np.float64), axis=1) - 1.0).max() < 1e-6


def test_mask_after_depth_range_test(gold):
    w, h = gold["ras_size"]
    cam = make_cam(gold["ras_K"], w, h)
    planes = np.zeros((h, w, 4), np.float32)
    planes[..., 3] = gold["ras_depth"]
    prior, mask, pl = hostlib.prior_from_triangles(cam, gold["ras_tri_pts"], planes, float(gold["ras_dmin"]), float(gold["ras_dmax"]))
    sure = ~gold["ras_ambiguous"]
    assert np.array_equal(mask[sure], gold["ras_mask"][sure])
    assert 0.5 < (mask > 0).mean() < 0.95                              # the range test removed something, not everything
    on = mask > 0
    assert np.array_equal(prior[on], pl[mask[on] - 1])                  # CudaPlanarPriorInitialization, reference :986-991
    assert not prior[~on].any()
