"""BASELINE.json configs[2], [3] and [4] at their real size (1600x1200, 8 source views) inside the driver-run GPU suite.

The oracle needs a minute per configuration at this size: configs[3] -- the longest schedule, every mode -- is compared with it
in full (test_cfg3_full_size_equals_the_oracle, round 5; until then the manual script tests/analysis/verify_cfg3_full.py), the
other tests use what the domain offers independently of the size: the C++ ProcessProblem mirror on 8-bit textures
with the device-built prior against the Python-driven schedule on fp32 textures with the host-built prior (two drivers, two
texture formats, two prior implementations: one result, bit for bit), determinism for a seed, sensitivity to the seed, value
ranges, unit normals facing the camera, convergence to the analytic ground truth; and for configs[4] the device-resident
exchange with worker threads against the same schedule staged through host arrays."""
import importlib

import numpy as np
import pytest

from test_pipeline_gpu import oracle_pipeline

pytestmark = pytest.mark.gpu

W, H, V = 1600, 1200, 8


@pytest.fixture(scope="module")
def scene(pm):
    sc = pm.synth.make_problem_scene(W, H, n_src=V, quantize=True)
    cams, imgs = sc.problem(0, list(range(1, V + 1)))
    rng = np.random.default_rng(7)
    # source depth maps of a self-contained single Problem: ground truth + 0.5 % noise (SURVEY 8d cfg 2)
    src_depths = [sc.views[i].gt_depth * (1.0 + 0.005 * rng.standard_normal((H, W))).astype(np.float32) for i in range(1, V + 1)]
    return sc, cams, imgs, src_depths


class _ForcedF32:
    """an engine context that keeps fp32 texels although the images are 8-bit exact (the other texture format)"""

    def __init__(self, engine):
        self.engine = engine

    def create(self):
        h = self.engine.create(0)
        h.set_texture_format(True)
        return h


@pytest.mark.parametrize("name,geom_iterations,planar_prior,geom_pp", [("cfg2", 1, False, False), ("cfg3", 2, True, True)])
def test_cfg2_cfg3_full_size(pm, engine, hostlib, scene, name, geom_iterations, planar_prior, geom_pp):
    sc, cams, imgs, src_depths = scene
    seed = 20240309
    depth, normal, cost = hostlib.run_pipeline(0, cams, imgs, 2, geom_iterations, planar_prior, geom_pp, seed, src_depths)
    # the same schedule driven from Python on a context with the OTHER texture format and the host-built prior
    planes, costs = oracle_pipeline(pm, _ForcedF32(engine), hostlib, cams, imgs, src_depths, 2, geom_iterations, planar_prior, geom_pp, seed)
    assert np.array_equal(depth, planes[..., 3]) and np.array_equal(normal, planes[..., :3]) and np.array_equal(cost, costs), name
    # determinism and seed sensitivity
    d2, n2, c2 = hostlib.run_pipeline(0, cams, imgs, 2, geom_iterations, planar_prior, geom_pp, seed, src_depths)
    assert np.array_equal(d2, depth) and np.array_equal(n2, normal) and np.array_equal(c2, cost), "same seed, same bits"
    d3, _, _ = hostlib.run_pipeline(0, cams, imgs, 2, geom_iterations, planar_prior, geom_pp, seed + 1, src_depths)
    assert not np.array_equal(d3, depth)
    # ranges: the last Run() is geometric: cost = photometric [0, 2] + 0.2 * geometric [0, 3]
    dmin, dmax = pm.synth.kernel_depth_range(cams[0])
    assert np.isfinite(depth).all() and np.isfinite(normal).all() and np.isfinite(cost).all()
    assert depth.min() >= dmin * 0.999 and depth.max() <= dmax * 1.001
    assert cost.min() >= 0.0 and cost.max() <= 2.6 + 1e-5
    assert np.allclose(np.linalg.norm(normal.astype(np.float64), axis=-1), 1.0, atol=1e-4)
    v0 = sc.views[0]
    ncam = normal.astype(np.float64) @ v0.R.T          # world -> camera
    u, v = np.meshgrid(np.arange(W), np.arange(H))
    view = np.stack([(u - v0.K[0, 2]) / v0.K[0, 0], (v - v0.K[1, 2]) / v0.K[1, 1], np.ones_like(u, float)], -1)
    assert ((ncam * view).sum(-1) <= 1e-6).mean() > 0.999
    gt = v0.gt_depth
    assert (np.abs(depth - gt) / gt < 0.01).mean() > 0.97, name


def test_cfg3_full_size_equals_the_oracle(pm, oracle, hostlib, scene):
    """configs[3] at BASELINE size against the ORACLE (not a property): the shipped config.yaml schedule -- photometric 3 scales ->
    geometric Run + planar prior + prior Run -> geometric Run, 228 hypothesis evaluations per pixel -- on one 1600x1200 Problem with
    8 source views, through the C++ ProcessProblem mirror on the HIP path, against the same schedule driven on the CPU oracle:
    depth, normal and cost maps bit for bit (about a minute of oracle time on 16 host threads)."""
    sc, cams, imgs, src_depths = scene
    seed = 4242
    depth, normal, cost = hostlib.run_pipeline(0, cams, imgs, 2, 2, True, True, seed, src_depths)
    planes, costs = oracle_pipeline(pm, oracle, hostlib, cams, imgs, src_depths, 2, 2, True, True, seed)
    assert np.array_equal(depth, planes[..., 3]), f"{int((depth != planes[..., 3]).sum())} of {depth.size} depths differ"
    assert np.array_equal(normal, planes[..., :3]) and np.array_equal(cost, costs)


def test_cfg0_shape_on_the_hip_path(pm, oracle, engine):
    """BASELINE.json configs[0] -- 2-view 320x240 pair, 3 PatchMatch iterations, photometric NCC only, single scale -- is the CPU
    plumbing case by definition; the same shape on the HIP path equals the oracle bit for bit (planes, costs, selected views)."""
    w, h = 320, 240
    sc = pm.synth.make_problem_scene(w, h, n_src=1, quantize=True)
    cams, imgs = sc.problem(0, [1])
    dmin, dmax = pm.synth.kernel_depth_range(cams[0])
    p = pm.PatchMatchParams(num_images=2, depth_min=float(dmin), depth_max=float(dmax), max_scale=0, max_iterations=3)
    gpu, cpu = engine.create(0), oracle.create()
    for hd in (gpu, cpu):
        hd.set_views(cams, imgs)
        hd.run(p, 12345)
    (gp, gc), (cp, cc) = gpu.get(), cpu.get()
    assert np.array_equal(gp, cp, equal_nan=True) and np.array_equal(gc, cc, equal_nan=True)
    assert np.array_equal(gpu.get_selected_views(), cpu.get_selected_views())
    gt = sc.views[0].gt_depth
    assert (np.abs(gp[..., 3] - gt) / gt < 0.01).mean() > 0.3   # one source view, one window scale, 3 iterations: 44 % of the pixels are there


def test_cfg4_shipped_schedule_full_size_device_exchange(pm, engine):
    """configs[4]'s machinery at BASELINE size on one GPU: 16 Problems of 1600x1200 (4x4 camera grid, 8 nearest neighbours as
    sources) through the SHIPPED schedule -- photometric 3 scales -> geometric + planar prior -> geometric -- with the depth maps
    exchanged in HBM (export -> gathered buffer -> device-to-device copies on the contexts' own streams; state resident; prior
    built on the device; 4 worker threads) against the same schedule with every map staged through host arrays on one thread."""
    from concurrent.futures import ThreadPoolExecutor
    sched = importlib.import_module("mp-mvs_amd.schedule")
    G = 4
    centers = [((i - (G - 1) / 2.0) * 0.15, (j - (G - 1) / 2.0) * 0.15, 0.0) for j in range(G) for i in range(G)]
    with ThreadPoolExecutor(4) as pool:
        views = list(pool.map(lambda i: pm.synth.make_scene(W, H, centers, quantize=True, only={i}).views[i], range(G * G)))
    cams, imgs = [v.cam for v in views], [v.image for v in views]
    neigh = []
    for j in range(G):
        for i in range(G):
            cand = sorted(((ii - i) ** 2 + (jj - j) ** 2, jj * G + ii) for jj in range(G) for ii in range(G) if (ii, jj) != (i, j))
            neigh.append([c[1] for c in cand[:8]])
    kw = dict(geom_iterations=2, planar_prior=True, geom_planar_prior=True, seed=4711)
    dev = sched.SceneScheduler(cams, imgs, neigh, lambda: engine.create(0), device_tensors=True, max_scale=2, workers=4)
    rd = dev.run(**kw)
    dd = dev.depth_maps()
    del dev
    host = sched.SceneScheduler(cams, imgs, neigh, lambda: engine.create(0), device_tensors=False, max_scale=2, workers=1)
    rh = host.run(**kw)
    assert np.array_equal(dd, host.depth_maps())
    for i in range(G * G):
        assert np.array_equal(rd[i][0], rh[i][0]) and np.array_equal(rd[i][1], rh[i][1]) and np.array_equal(rd[i][2], rh[i][2]), f"problem {i}"
    acc = [float((np.abs(dd[i] - views[i].gt_depth) / views[i].gt_depth < 0.01).mean()) for i in range(G * G)]
    assert min(acc) > 0.95, acc
