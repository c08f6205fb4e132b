"""Planar prior built on the device (mp-mvs_amd/csrc/pm_prior.hpp; reference src/PatchMatch.cpp:532-604, SURVEY a-16):
against the independent scipy / numpy fixtures of tests/golden/prior_golden_v1.npz, and bit for bit against the host
implementation (mp-mvs_amd/host/planar_prior.cpp) on the state of a real Run()."""
import os

import numpy as np
import pytest

from test_prior_golden_cpu import make_cam

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "prior_golden_v1.npz"))


def blank_context(pm, engine, cam, w, h):
    """a context of the given size (the images do not matter for the prior)"""
    img = np.zeros((h, w), np.float32)
    gpu = engine.create(0)
    gpu.set_views([cam, cam], [img, img])
    return gpu


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("geomprior", [False, True])
def test_device_vertices_equal_bruteforce(pm, engine, gold, tag, geomprior):
    costs, geom = gold[f"vert_{tag}_costs"], gold[f"vert_{tag}_geom"]
    h, w = costs.shape
    gpu = blank_context(pm, engine, make_cam([100, 0, w / 2, 0, 100, h / 2, 0, 0, 1], w, h), w, h)
    gpu.set_state(None, costs)
    gpu.set_geom_costs(geom)
    got = gpu.prior_vertices(geomprior)
    assert np.array_equal(got, gold[f"vert_{tag}_geomprior" if geomprior else f"vert_{tag}_plain"])


def test_device_raster_planes_mask_equal_fixtures(pm, engine, gold):
    w, h = (int(v) for v in gold["ras_size"])
    cam = make_cam(gold["ras_K"], w, h)
    gpu = blank_context(pm, engine, cam, w, h)
    planes = np.zeros((h, w, 4), np.float32)
    planes[..., 3] = gold["ras_depth"]
    gpu.set_state(planes, None)
    tri = gold["ras_tri_pts"]
    prm = pm.PatchMatchParams(num_images=2, depth_min=-3.0e38, depth_max=3.0e38)
    gpu.prior_from_triangles(prm, tri)
    prior, label = gpu.get_prior()
    assert np.array_equal(label, gold["ras_label"])                     # raw raster: the last triangle wins
    # per-triangle planes, read back through the pixels they own
    want = gold["ras_planes"]
    on = label > 0
    assert np.abs(prior[on].astype(np.float64) - want[label[on] - 1]).max() < 1e-5
    prm = pm.PatchMatchParams(num_images=2, depth_min=float(gold["ras_dmin"]), depth_max=float(gold["ras_dmax"]))
    gpu.prior_from_triangles(prm, tri)
    prior, mask = gpu.get_prior()
    sure = ~gold["ras_ambiguous"]
    assert np.array_equal(mask[sure], gold["ras_mask"][sure])
    assert not prior[mask == 0].any()


@pytest.mark.parametrize("geom_rule", [False, True])
def test_device_prior_equals_host_prior_bit_for_bit(pm, engine, hostlib, geom_rule):
    """state of a real Run() -> vertices, (host Delaunay), raster, planes, mask: device == host, every bit"""
    W, H, V = 163, 121, 4
    sc = pm.synth.make_problem_scene(W, H, n_src=V, spacing=0.4, quantize=True)
    cams, imgs = sc.problem(0, list(range(1, V + 1)))
    dmin, dmax = pm.synth.kernel_depth_range(cams[0])
    gpu = engine.create(0)
    gpu.set_views(cams, imgs)
    prm = pm.PatchMatchParams(num_images=V + 1, depth_min=float(dmin), depth_max=float(dmax), max_scale=1)
    gpu.run(prm, 4242)
    if geom_rule:
        rng = np.random.default_rng(1)
        gpu.set_src_depths([sc.views[i].gt_depth * (1 + 0.005 * rng.standard_normal((H, W))).astype(np.float32) for i in range(1, V + 1)])
        prm.geom_consistency, prm.max_iterations, prm.geomPlanarPrior = True, 2, True
        gpu.run(prm, 4243)
    planes, costs, geom = gpu.get(geom=True)
    # vertices
    v_dev = gpu.prior_vertices(geom_rule)
    v_host = hostlib.triangulate_vertices(costs, geom, geom_rule)
    assert len(v_dev) > 100 and np.array_equal(v_dev, v_host)
    # triangles on the host, the rest on the device
    tris = hostlib.delaunay(W, H, v_dev)
    gpu.prior_from_triangles(prm, tris)
    p_dev, m_dev = gpu.get_prior()
    p_host, m_host, n_host = hostlib.build_prior(cams[0], planes, costs, geom if geom_rule else None, geom_rule, prm.depth_min, prm.depth_max)
    assert n_host == len(tris)
    assert np.array_equal(m_dev, m_host)
    assert np.array_equal(p_dev.view(np.uint32), p_host.view(np.uint32))
    assert 0.2 < (m_dev > 0).mean() <= 1.0


def test_device_prior_rejects_vertices_outside_the_image(pm, engine):
    w, h = 40, 30
    gpu = blank_context(pm, engine, make_cam([50, 0, 20, 0, 50, 15, 0, 0, 1], w, h), w, h)
    prm = pm.PatchMatchParams(num_images=2, depth_min=1.0, depth_max=9.0)
    with pytest.raises(RuntimeError, match="outside the image"):
        gpu.prior_from_triangles(prm, np.array([[[1, 1], [5, 2], [40, 3]]], np.int32))
