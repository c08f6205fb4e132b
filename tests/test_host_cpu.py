"""CPU tests of the host layer: C-ABI surface (symbols only, no compute without
a GPU) and the planar-prior construction (vertices, Delaunay, plane fit, raster).

OpenCV (cv::Subdiv2D, cv::SVD) is absent here and on the GPU box, so these pieces
are "parity unpinned" against the reference; they are pinned by the
properties a Delaunay triangulation and a 3-point plane must satisfy.
"""
import ctypes
import importlib
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hostlib(pm):
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "mp-mvs_amd", "host", "libmpmvs_host.so")):
        g.build()
    return importlib.import_module("mp-mvs_amd.hostlib")


def test_c_abi_exports_every_declared_symbol(pm):
    """every function include/mpmvs.h declares must be exported by the HIP library
    (loading needs no GPU; nothing is called)"""
    header = open(os.path.join(ROOT, "include", "mpmvs.h")).read()
    declared = sorted(set(re.findall(r"\b(mpmvs_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 20
    engine = importlib.import_module("mp-mvs_amd.engine")
    lib, fns = engine.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mpmvs.h but not exported"
    for name in engine.ALL_SYMBOLS:
        assert name in declared, f"{name} bound in Python but not declared in include/mpmvs.h"
    # the opt-in twin with 8-bit texture fractions is the same ABI
    lib_q8, _ = engine.load_variant(engine.LIB_Q8_PATH)
    for name in declared:
        assert hasattr(lib_q8, name), f"{name} declared in include/mpmvs.h but not exported by libmpmvs_hip_q8.so"
    assert lib.mpmvs_texture_filter_bits() == 0 and lib_q8.mpmvs_texture_filter_bits() == 8   # a host function: no GPU needed


def test_abi_struct_layouts(pm):
    assert ctypes.sizeof(pm.Camera) == 112 and ctypes.sizeof(pm.PatchMatchParams) == 56
    assert pm.Camera.height.offset == 96 and pm.Camera.width.offset == 100  # height first (reference PatchMatch.h:42-43)
    assert pm.PatchMatchParams.geom_consistency.offset == 52
    p = pm.PatchMatchParams()
    assert (p.max_iterations, p.top_k, p.max_scale, p.sigma_spatial, p.sigma_color) == (3, 4, 2, 5.0, 3.0)


def test_host_library_symbols(hostlib):
    lib = hostlib.load()
    for s in hostlib.SYMBOLS:
        assert hasattr(lib, s)


def test_product_has_no_cpu_fallback(pm):
    """creating a context without a HIP device must fail loudly"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    engine = importlib.import_module("mp-mvs_amd.engine")
    with pytest.raises(RuntimeError, match="mpmvs_create failed"):
        engine.create(0)


# ---------------------------------------------------------------------------
def _incircle(a, b, c, d):
    m = np.array([[a[0] - d[0], a[1] - d[1], (a[0] - d[0]) ** 2 + (a[1] - d[1]) ** 2],
                  [b[0] - d[0], b[1] - d[1], (b[0] - d[0]) ** 2 + (b[1] - d[1]) ** 2],
                  [c[0] - d[0], c[1] - d[1], (c[0] - d[0]) ** 2 + (c[1] - d[1]) ** 2]], dtype=object)
    return (m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0])
            + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]))


def _orient(a, b, c):
    return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])


@pytest.mark.parametrize("n,seed", [(5, 0), (60, 1), (400, 2)])
def test_delaunay_properties(hostlib, n, seed):
    rng = np.random.default_rng(seed)
    pts = np.unique(rng.integers(0, 200, size=(n, 2)), axis=0).astype(np.int32)
    rng.shuffle(pts)
    tris = hostlib.delaunay(200, 200, pts)
    ptset = {tuple(p) for p in pts.tolist()}
    area2 = 0
    for t in tris.tolist():
        a, b, c = [tuple(v) for v in t]
        assert a in ptset and b in ptset and c in ptset
        o = _orient(a, b, c)
        assert o > 0, "counter-clockwise, non-degenerate"
        area2 += o
        for p in ptset:  # empty circumcircle (exact integer arithmetic)
            if p not in (a, b, c):
                assert _incircle(a, b, c, p) <= 0
    from scipy.spatial import ConvexHull
    hull = ConvexHull(pts.astype(float))
    assert area2 == int(round(2 * hull.volume)), "the triangles tile the convex hull"
    used = {tuple(v) for t in tris.tolist() for v in t}
    assert used == ptset


def test_delaunay_grid_points_and_duplicates(hostlib):
    """cocircular grid points and a duplicated point: still a valid triangulation"""
    xs, ys = np.meshgrid(np.arange(0, 50, 5), np.arange(0, 40, 5))
    pts = np.stack([xs.ravel(), ys.ravel()], -1).astype(np.int32)
    pts = np.concatenate([pts, pts[:3]])
    tris = hostlib.delaunay(64, 64, pts)
    area2 = sum(_orient(*[tuple(v) for v in t]) for t in tris.tolist())
    assert area2 == 2 * 45 * 35
    assert len(tris) == 2 * 9 * 7


def _check_triangulation(pts, tris):
    """vectorised validity check of a triangulation of integer points: counter-clockwise triangles over the given points,
    every interior edge locally Delaunay (exact int64), triangles tile the convex hull, triangle count = 2n - 2 - hull"""
    from scipy.spatial import ConvexHull
    pts = np.asarray(pts, np.int64)
    n = len(pts)
    W = int(pts[:, 0].max()) + 1
    lut = {int(y) * W + int(x): i for i, (x, y) in enumerate(pts.tolist())}
    t = np.asarray(tris, np.int64)
    ids = np.array([lut[int(k)] for k in (t[:, :, 1] * W + t[:, :, 0]).ravel()], np.int64).reshape(-1, 3)
    P = pts[ids]                                            # T x 3 x 2
    o = (P[:, 1, 0] - P[:, 0, 0]) * (P[:, 2, 1] - P[:, 0, 1]) - (P[:, 1, 1] - P[:, 0, 1]) * (P[:, 2, 0] - P[:, 0, 0])
    assert (o > 0).all(), "counter-clockwise, non-degenerate"
    hull = ConvexHull(pts.astype(float))
    assert int(o.sum()) == int(round(2 * hull.volume)), "the triangles tile the convex hull"
    on_hull = 0                                             # points ON the hull boundary (qhull lists corners only)
    hv = pts[hull.vertices]
    for a, b in zip(hv, np.roll(hv, -1, axis=0)):
        d = b - a
        cross = (pts[:, 0] - a[0]) * d[1] - (pts[:, 1] - a[1]) * d[0]
        dot = (pts[:, 0] - a[0]) * d[0] + (pts[:, 1] - a[1]) * d[1]
        on_hull += int(((cross == 0) & (dot >= 0) & (dot < d @ d)).sum())
    assert len(t) == 2 * n - 2 - on_hull
    assert len(np.unique(ids)) == n, "every point is used"
    # directed edges u->v with the opposite corner w; the twin v->u belongs to the neighbour
    u = ids[:, [0, 1, 2]].ravel()
    v = ids[:, [1, 2, 0]].ravel()
    w = ids[:, [2, 0, 1]].ravel()
    key, twin = u * n + v, v * n + u
    assert len(np.unique(key)) == len(key), "no directed edge twice (no overlapping triangles)"
    order = np.argsort(key)
    pos = np.searchsorted(key[order], twin)
    pos[pos >= len(key)] = 0
    has = key[order][pos] == twin
    w2 = w[order][pos]
    A, B, C, D = pts[u[has]], pts[v[has]], pts[w[has]], pts[w2[has]]
    ax, ay = A[:, 0] - D[:, 0], A[:, 1] - D[:, 1]
    bx, by = B[:, 0] - D[:, 0], B[:, 1] - D[:, 1]
    cx, cy = C[:, 0] - D[:, 0], C[:, 1] - D[:, 1]
    a2, b2, c2 = ax * ax + ay * ay, bx * bx + by * by, cx * cx + cy * cy
    det = ax * (by * c2 - b2 * cy) - ay * (bx * c2 - b2 * cx) + a2 * (bx * cy - by * cx)
    assert (det <= 0).all(), "an opposite corner lies strictly inside a circumcircle"
    assert int((~has).sum()) == on_hull, "exactly the hull edges have no twin"


def _cell_vertices(W, H, per_cell, seed):
    """the vertex pattern of the planar prior: up to per_cell distinct pixels in every 5x5 cell, cells in raster order"""
    rng = np.random.default_rng(seed)
    out = []
    for cy in range(H // 5):
        for cx in range(W // 5):
            for q in sorted(rng.choice(25, per_cell, replace=False).tolist()):
                out.append((cx * 5 + q % 5, cy * 5 + q // 5))
    return np.array(out, np.int32)


@pytest.mark.parametrize("W,H,per_cell", [(400, 300, 3), (480, 320, 1), (1280, 95, 2)])
def test_delaunay_large_sets_parallel_and_serial_agree(hostlib, monkeypatch, W, H, per_cell):
    """above 4096 points the subtrees run on their own threads: the result is a valid Delaunay triangulation of the grid-bound
    (hence heavily cocircular) vertex pattern and identical, triangle for triangle, for every number of threads"""
    pts = _cell_vertices(W, H, per_cell, seed=W + per_cell)
    assert len(pts) >= 4096
    ref = None
    for threads in ("1", "2", "5", "8"):
        monkeypatch.setenv("MPMVS_HOST_THREADS", threads)
        tris = np.array(hostlib.delaunay(W, H, pts))
        if ref is None:
            ref = tris
            _check_triangulation(pts, tris)
        else:
            assert np.array_equal(ref, tris), f"{threads} threads"


@pytest.mark.parametrize("kind", range(6))
def test_delaunay_point_distributions(hostlib, monkeypatch, kind):
    """uniform, near-vertical and near-horizontal bands, one cluster, a lattice-like pattern, heavy duplication: valid and
    independent of the number of threads"""
    rng = np.random.default_rng(100 + kind)
    n = 9000
    if kind == 0:
        pts = rng.integers(0, 2000, size=(n, 2))
    elif kind == 1:
        pts = np.stack([rng.integers(0, 40, n) * 50 + rng.integers(0, 3, n), rng.integers(0, 3000, n)], -1)
    elif kind == 2:
        pts = np.stack([rng.integers(0, 3000, n), rng.integers(0, 30, n) * 70 + rng.integers(0, 2, n)], -1)
    elif kind == 3:
        pts = (rng.normal(size=(n, 2)) * 200 + 1000).astype(np.int64)
    elif kind == 4:
        pts = np.stack([np.arange(n) % 997, (np.arange(n) * 7) % 1013], -1)
    else:
        pts = rng.integers(0, 90, size=(n, 2))
    pts = np.clip(pts, 0, 4095).astype(np.int32)
    uniq = np.unique(pts, axis=0)
    ref = None
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("MPMVS_HOST_THREADS", threads)
        tris = np.array(hostlib.delaunay(4096, 4096, pts))
        if ref is None:
            ref = tris
            _check_triangulation(uniq, tris)
        else:
            assert np.array_equal(ref, tris), f"{threads} threads"


def test_delaunay_full_lattice_parallel(hostlib, monkeypatch):
    """every quadruple of neighbours cocircular, every row and column collinear, 8 threads"""
    monkeypatch.setenv("MPMVS_HOST_THREADS", "8")
    xs, ys = np.meshgrid(np.arange(0, 300, 3), np.arange(0, 180, 3))
    pts = np.stack([xs.ravel(), ys.ravel()], -1).astype(np.int32)
    np.random.default_rng(0).shuffle(pts)
    _check_triangulation(pts, hostlib.delaunay(300, 180, pts))


def test_delaunay_degenerate_inputs(hostlib):
    assert len(hostlib.delaunay(10, 10, np.zeros((0, 2), np.int32))) == 0
    assert len(hostlib.delaunay(10, 10, np.array([[1, 1], [5, 2]], np.int32))) == 0
    line = np.stack([np.arange(40), 2 * np.arange(40)], -1).astype(np.int32)
    assert len(hostlib.delaunay(100, 100, line)) == 0                                  # all collinear: no triangle
    vertical = np.stack([np.full(30, 7), np.arange(30)], -1).astype(np.int32)
    assert len(hostlib.delaunay(100, 100, vertical)) == 0
    fan = np.concatenate([vertical, np.array([[20, 11]], np.int32)])                   # a collinear chain and one apex
    tris = hostlib.delaunay(100, 100, fan)
    assert len(tris) == 29
    _check_triangulation(fan, tris)
    two_lines = np.concatenate([vertical, vertical + np.array([1, 0], np.int32)])      # two adjacent collinear chains
    _check_triangulation(two_lines, hostlib.delaunay(100, 100, two_lines))
    dup = np.repeat(np.array([[3, 4], [9, 1], [6, 8]], np.int32), 5, axis=0)            # duplicates of three points
    assert len(hostlib.delaunay(100, 100, dup)) == 1
    rng = np.random.default_rng(3)
    for n in range(3, 40):                                                             # every small piece size of the recursion
        pts = np.unique(rng.integers(0, 12, size=(n, 2)), axis=0).astype(np.int32)
        if len(pts) < 3:
            continue
        tris = hostlib.delaunay(12, 12, pts)
        if len(tris):
            _check_triangulation(pts, tris)
    far = np.array([[0, 0], [9000, 10], [20, 8999], [8000, 8000], [4000, 4100], [4100, 4000], [100, 5000]], np.int32)
    _check_triangulation(far, hostlib.delaunay(9001, 9000, far))                       # 128-bit in-circle path


def test_triangulate_vertices(hostlib):
    costs = np.full((23, 31), 1.5, np.float32)
    costs[2, 3] = 0.05      # cell (0,0) -> vertex (3,2)
    costs[7, 12] = 0.09     # cell (row 5.., col 10..)
    costs[7, 13] = 0.02     # lower cost in the same cell wins
    costs[12, 22] = 0.11    # above the 0.1 threshold: no vertex
    costs[22, 30] = 0.01    # ragged last cell
    v = hostlib.triangulate_vertices(costs)
    assert sorted(map(tuple, v.tolist())) == [(3, 2), (13, 7), (30, 22)]
    # geometric variant: up to three vertices per cell, cost < 1, geom cost < 0.4, below max(0.85*sum/(r*c), 0.2)
    geom = np.full_like(costs, 0.1)
    geom[7, 13] = 0.5
    costs[:] = 1.5
    costs[7, 12], costs[7, 13], costs[8, 11], costs[8, 12] = 0.09, 0.02, 0.15, 0.19
    v = hostlib.triangulate_vertices(costs, geom, True)
    assert [tuple(p) for p in v.tolist()] == [(12, 7), (11, 8), (12, 8)]


def test_build_prior_recovers_a_plane(pm, hostlib):
    """planes/costs of a perfectly reconstructed slanted plane -> every prior plane
    equals it, the mask covers the triangulated interior"""
    W, H = 80, 60
    sc = pm.synth.make_scene(W, H, [(0, 0, 0)], rot_deg=0.0)
    cam = sc.views[0].cam
    n = np.array([0.2, -0.1, -1.0])
    n /= np.linalg.norm(n)
    d = 5.0  # n.X + d = 0
    u, v = np.meshgrid(np.arange(W), np.arange(H))
    ray = np.stack([(u - cam.K[2]) / cam.K[0], (v - cam.K[5]) / cam.K[4], np.ones_like(u, float)], -1)
    depth = -d / (ray @ n)
    planes = np.zeros((H, W, 4), np.float32)
    planes[..., :3] = n
    planes[..., 3] = depth
    costs = np.full((H, W), 0.05, np.float32)
    prior, mask, ntri = hostlib.build_prior(cam, planes, costs, None, False, 1.0, 20.0)
    assert ntri > 100
    inside = mask > 0
    assert inside[5:-6, 5:-6].mean() > 0.98
    assert np.allclose(prior[inside][:, :3], n, atol=2e-3)
    assert np.allclose(prior[inside][:, 3], d, rtol=2e-3)
    assert (prior[inside][:, 3] > 0).all()  # offset made positive (reference PatchMatch.cpp:746-752)
    # a depth range that excludes the plane clears the mask (reference :583-595)
    _, mask2, _ = hostlib.build_prior(cam, planes, costs, None, False, 10.0, 20.0)
    assert mask2.max() == 0


# ---------------------------------------------------------------------------
# file formats
# ---------------------------------------------------------------------------
def test_dmb_round_trip_and_layout(hostlib, tmp_path):
    rng = np.random.default_rng(0)
    d = rng.uniform(0, 9, (7, 11)).astype(np.float32)
    n = rng.normal(size=(7, 11, 3)).astype(np.float32)
    hostlib.write_dmb(tmp_path / "d.dmb", d)
    hostlib.write_dmb(tmp_path / "n.dmb", n)
    raw = open(tmp_path / "n.dmb", "rb").read()
    assert np.frombuffer(raw[:16], np.int32).tolist() == [1, 7, 11, 3]      # type, h, w, nb (reference utility.cpp:287-296)
    assert np.array_equal(np.frombuffer(raw[16:], np.float32).reshape(7, 11, 3), n)
    assert np.array_equal(hostlib.read_dmb(tmp_path / "d.dmb"), d) and np.array_equal(hostlib.read_dmb(tmp_path / "n.dmb"), n)
    open(tmp_path / "bad.dmb", "wb").write(np.array([2, 7, 11, 1], np.int32).tobytes() + d.tobytes())
    with pytest.raises(RuntimeError):
        hostlib.read_dmb(tmp_path / "bad.dmb")                                 # type != 1 is rejected (reference :213-216)


def test_camera_pair_and_image_files(pm, hostlib, tmp_path):
    sc, neigh = pm.synth.make_grid_scene(40, 30, 3, 2, spacing=0.5, quantize=True)
    cams = [v.cam for v in sc.views]
    imgs = [v.image for v in sc.views]
    scores = [[50.0 - k for k in range(len(s))] for s in neigh]
    scores[1][2] = 0.0                                                         # score <= 0 is dropped (reference PatchMatch.cpp:98-100)
    hostlib.write_dataset(str(tmp_path), cams, imgs, neigh, scores)
    for i, cam in enumerate(cams):
        got = hostlib.read_camera(tmp_path / "cams" / f"{i:08d}_cam.txt")
        assert list(got.K) == list(cam.K) and list(got.R) == list(cam.R) and list(got.t) == list(cam.t)
        assert np.allclose(list(got.C), list(cam.C), atol=1e-6)               # C = -R^T t recomputed in fp32 (reference :134-136)
        assert (got.depth_min, got.depth_max) == (cam.depth_min, cam.depth_max)
        assert np.array_equal(hostlib.read_pgm(tmp_path / "images" / f"{i:08d}.pgm"), imgs[i])
    lst = hostlib.sample_list(tmp_path, max_src=3)
    assert len(lst) == 6 and all(e for e, _, _ in lst)
    assert lst[0][2] == [0] + neigh[0][:3]                                     # at most `Max source images num` kept
    assert lst[1][2] == [1] + [s for k, s in enumerate(neigh[1][:3]) if k != 2]


def _test_picture(w, h, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 100 * np.sin(xx / 7.0) * np.cos(yy / 5.0), 127 + 80 * np.cos(xx / 13.0 + yy / 9.0), (xx * 3 + yy * 2) % 256], -1)
    return np.clip(img + rng.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("size", [(1, 1), (3, 2), (17, 9), (67, 53), (200, 152)])
@pytest.mark.parametrize("options", [dict(quality=90, subsampling=0), dict(quality=90, subsampling=1), dict(quality=75, subsampling=2),
                                     dict(quality=30, subsampling=2, progressive=True), dict(quality=95, subsampling=0, progressive=True, optimize=True),
                                     dict(quality=85, subsampling=2, restart_marker_blocks=3), dict(quality=85, subsampling="4:1:1"),
                                     dict(quality=80, grey=True), dict(quality=80, grey=True, progressive=True)])
def test_jpeg_decoder_equals_libjpeg(hostlib, size, options):
    """Image ingest: the own JPEG decoder returns exactly the pixels libjpeg(-turbo) -- i.e.
    cv::imread, reference src/PatchMatch.cpp:877 (GRAYSCALE = the luminance plane) and :324 (COLOR = B,G,R) --
    returns, for baseline, progressive, restart-interval and subsampled files.  PIL is the libjpeg front end."""
    import io
    Image = pytest.importorskip("PIL.Image")
    w, h = size
    options = dict(options)
    grey = options.pop("grey", False)
    pic = _test_picture(w, h)
    buf = io.BytesIO()
    Image.fromarray(pic[..., 0] if grey else pic).save(buf, "JPEG", **options)
    data = buf.getvalue()
    lum = Image.open(io.BytesIO(data))
    lum.draft("L", (w, h))                      # libjpeg out_color_space = JCS_GRAYSCALE, what OpenCV requests
    lum = np.asarray(lum)
    rgb = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    assert np.array_equal(hostlib.decode_jpeg(data, 1), lum)
    assert np.array_equal(hostlib.decode_jpeg(data, 3)[..., ::-1], rgb)


@pytest.mark.parametrize("orientation", range(1, 9))
def test_jpeg_exif_orientation_is_applied(hostlib, orientation):
    """cv::imread turns the image upright according to EXIF tag 0x0112 (flags without IMREAD_IGNORE_ORIENTATION, as the
    reference's calls); so does the decoder -- checked against PIL's exif_transpose for all 8 orientations"""
    import io
    Image = pytest.importorskip("PIL.Image")
    from PIL import ImageOps
    pic = _test_picture(53, 37)
    ex = Image.Exif()
    ex[0x0112] = orientation
    buf = io.BytesIO()
    Image.fromarray(pic).save(buf, "JPEG", quality=90, exif=ex)
    data = buf.getvalue()
    want = np.asarray(ImageOps.exif_transpose(Image.open(io.BytesIO(data))).convert("RGB"))
    assert np.array_equal(hostlib.decode_jpeg(data, 3)[..., ::-1], want)
    assert hostlib.decode_jpeg(data, 1).shape == want.shape[:2]


def test_jpeg_decoder_rejects_bad_input(hostlib):
    import io
    Image = pytest.importorskip("PIL.Image")
    for junk in (b"", b"\xff\xd8", b"P5\n1 1\n255\n\x00", b"\xff\xd8\xff\xd9"):
        with pytest.raises(RuntimeError):
            hostlib.decode_jpeg(junk, 1)
    buf = io.BytesIO()
    Image.fromarray(_test_picture(200, 150)).save(buf, "JPEG", quality=90)
    data = buf.getvalue()
    assert hostlib.decode_jpeg(data[: 2 * len(data) // 3], 1).shape == (150, 200)      # truncated data decodes (zero bits), as libjpeg does
    cmyk = io.BytesIO()
    Image.fromarray(np.zeros((8, 8, 4), np.uint8), "CMYK").save(cmyk, "JPEG")
    with pytest.raises(RuntimeError):
        hostlib.decode_jpeg(cmyk.getvalue(), 1)                                     # 4 components: unsupported, loudly


def test_image_files_by_content(hostlib, tmp_path):
    """readGrayImage / readColorImage pick the format by content: JPEG, PGM, PPM (PPM -> grey with OpenCV's BGR2GRAY weights)"""
    pic = _test_picture(33, 21)
    open(tmp_path / "a.ppm", "wb").write(b"P6\n# comment\n33 21\n255\n" + pic.tobytes())
    open(tmp_path / "a.pgm", "wb").write(b"P5 33 21 255\n" + pic[..., 1].tobytes())
    assert np.array_equal(hostlib.read_image(tmp_path / "a.ppm", 3), pic[..., ::-1])
    r, g, b = (pic[..., k].astype(np.int64) for k in range(3))
    assert np.array_equal(hostlib.read_image(tmp_path / "a.ppm", 1), ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8))
    assert np.array_equal(hostlib.read_image(tmp_path / "a.pgm", 1), pic[..., 1])
    assert np.array_equal(hostlib.read_image(tmp_path / "a.pgm", 3), pic[..., 1:2].repeat(3, -1))
    assert np.array_equal(hostlib.read_pgm(tmp_path / "a.pgm"), pic[..., 1].astype(np.float32))
    with pytest.raises(RuntimeError):
        hostlib.read_image(tmp_path / "missing.jpg", 1)


def test_resize_linear_geometry(hostlib):
    """INTER_LINEAR geometry (reference src/PatchMatch.cpp:915): a linear ramp stays the same
    ramp under resampling (sampled at (x+0.5)*scale-0.5), constants stay constant, 2x
    shrink of an even image averages 2x2 blocks"""
    x = np.arange(40, dtype=np.float32)[None, :].repeat(30, 0)
    out = hostlib.resize_linear(x, 20, 15)
    assert np.allclose(out, ((np.arange(20) + 0.5) * 2 - 0.5)[None, :], atol=1e-5)
    assert np.all(hostlib.resize_linear(np.full((30, 40), 7.5, np.float32), 13, 11) == 7.5)
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 255, (30, 40)).astype(np.float32)
    half = hostlib.resize_linear(img, 20, 15)
    assert np.allclose(half, img.reshape(15, 2, 20, 2).mean((1, 3)), atol=1e-3)
    assert np.array_equal(hostlib.resize_linear(img, 40, 30), img)


def test_ply_layout(hostlib, tmp_path):
    pts = np.array([[1, 2, 3, 0, 0, -1, 10, 20, 30], [np.inf, 0, 0, 0, 1, 0, 255, 128, 0]], np.float32)
    hostlib.write_ply(tmp_path / "m.ply", pts)
    raw = open(tmp_path / "m.ply", "rb").read()
    head, body = raw.split(b"end_header\n")
    assert b"format binary_little_endian 1.0" in head and b"element vertex 2" in head
    assert len(body) == 2 * 27
    assert np.frombuffer(body[:24], np.float32).tolist() == [1, 2, 3, 0, 0, -1] and list(body[24:27]) == [30, 20, 10]   # red green blue
    assert np.frombuffer(body[27:39], np.float32).tolist() == [0, 0, 0]                                               # non-finite -> origin


def test_pair_list_drops_ids_that_do_not_exist(hostlib, tmp_path):
    """pair.txt naming a source view beyond the last image (malformed): the id is dropped, nothing indexes out of bounds"""
    (tmp_path / "pair.txt").write_text("3\n0\n2 1 10.0 7 9.0\n1\n3 0 5.0 2 4.0 -3 8.0\n2\n1 1 0.0\n")
    got = hostlib.sample_list(str(tmp_path))
    assert got == [(True, 0, [0, 1]), (True, 1, [1, 0, 2]), (True, 2, [2])]


def test_resize_linear_equals_independent_fixture(hostlib):
    """ResizeLinear + the target-size rule of PatchMatchInit's "Adjust image scale" (reference src/PatchMatch.cpp:893-925)
    against tests/golden/resize_golden_v1.npz, which tests/golden/make_resize_golden.py computed without this repository
    (torch's bilinear interpolation, align_corners=False, no antialias = cv::resize INTER_LINEAR's geometry, in float64).
    Tolerance: 1e-5 of the 0..255 range (fp32 interpolation against float64 rounded once)."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "resize_golden_v1.npz"))
    for k in range(int(z["n"])):
        src, want, mx = z[f"src{k}"], z[f"dst{k}"], int(z[f"max{k}"])
        rows, cols = src.shape
        f = min(np.float32(mx) / np.float32(cols), np.float32(mx) / np.float32(rows))
        new_cols, new_rows = int(np.floor(np.float32(cols) * f + np.float32(0.5))), int(np.floor(np.float32(rows) * f + np.float32(0.5)))
        assert (new_rows, new_cols) == want.shape
        got = hostlib.resize_linear(src, new_cols, new_rows)
        err = np.abs(got.astype(np.float64) - want.astype(np.float64)).max()
        assert err <= 255.0 * 1e-5, f"case {k}: max |difference| {err}"


def test_unassigned_cell_points_are_pixel_zero(hostlib):
    """GetTriangulateVertices, geomPlanarPrior branch (reference src/PatchMatch.cpp:810-850): the cell's threshold is
    0.85 x cost sum / (r_bound x c_bound); costs of a geometric Run() reach 2.6 and the first cell divides by 25, so the
    threshold can exceed the initial 2.0 of minCosts -- the reference then pushes its never-assigned `points(3)`, i.e. pixel
    (0, 0), three times.  The mirror reproduces that (and its Delaunay triangulation drops the duplicates)."""
    costs = np.full((10, 10), 0.5, np.float32)
    geom = np.full((10, 10), 3.0, np.float32)          # nothing is geometrically consistent: no pixel qualifies anywhere
    costs[:5, :5] = 2.6                                # first cell: 0.85 * 65 / 25 = 2.21 > 2.0
    v = hostlib.triangulate_vertices(costs, geom, True)
    assert v.tolist() == [[0, 0], [0, 0], [0, 0]]
    geom[7, 8] = 0.0                                   # one consistent pixel in the last cell: threshold max(0.85 * 12.5 / 100, 0.2) = 0.2 < 0.5
    assert hostlib.triangulate_vertices(costs, geom, True).tolist() == [[0, 0], [0, 0], [0, 0]]
    costs[7, 8] = 0.1
    assert hostlib.triangulate_vertices(costs, geom, True).tolist() == [[0, 0], [0, 0], [0, 0], [8, 7]]


@pytest.mark.parametrize("leave", ["_exit", "exit"])
def test_delaunay_in_a_forked_child(hostlib, leave):
    """the persistent thread pool of the triangulation does not survive a fork: the child must notice and run on its own
    thread instead of waiting for workers it does not have -- and, when it leaves through exit(), the static destructors of the
    pools must not join threads that do not exist in the child"""
    rng = np.random.default_rng(0)
    pts = rng.integers(0, 2000, size=(9000, 2)).astype(np.int32)
    ref = np.array(hostlib.delaunay(2000, 2000, pts))           # creates the pool in this process
    pid = os.fork()
    if pid == 0:
        try:
            ok = np.array_equal(ref, np.array(hostlib.delaunay(2000, 2000, pts)))
        except BaseException:
            ok = False
        if leave == "exit" and ok:
            import ctypes
            ctypes.CDLL(None).exit(0)      # libc exit(): runs the static destructors of the loaded libraries
        os._exit(0 if ok else 3)
    deadline = 60.0
    import time
    t0 = time.time()
    while True:
        done, status = os.waitpid(pid, os.WNOHANG)
        if done:
            break
        if time.time() - t0 > deadline:
            os.kill(pid, 9)
            os.waitpid(pid, 0)
            raise AssertionError("the forked child hung in the triangulation")
        time.sleep(0.05)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0


@pytest.mark.parametrize("kind", ["uniform", "band"])
def test_delaunay_large_sets_agree_across_thread_counts(hostlib, monkeypatch, kind):
    """60-70 k points (the level-by-level driver with 32 - 64 subtrees): the same triangulation on 1, 8 and 3 threads -- also
    for a band whose cuts along y meet only a dozen distinct coordinate values"""
    rng = np.random.default_rng(7)
    if kind == "uniform":
        pts = np.stack([rng.integers(0, 1600, 70000), rng.integers(0, 1200, 70000)], -1).astype(np.int32)
    else:
        pts = np.stack([rng.integers(0, 4000, 60000), rng.integers(0, 12, 60000)], -1).astype(np.int32)
    ref = None
    for threads in ("1", "8", "3"):
        monkeypatch.setenv("MPMVS_HOST_THREADS", threads)
        tris = np.array(hostlib.delaunay(4096, 4096, pts))
        if ref is None:
            ref = tris
            uniq = np.unique(pts, axis=0)
            assert len(tris) <= 2 * len(uniq) and len(tris) >= len(uniq)
        else:
            assert np.array_equal(ref, tris), f"{threads} threads"


def test_sealed_map_notices_a_direct_write(hostlib):
    """The C++ mirror skips the upload of a source depth map / start state the resident context already holds (Image::stamp).  A
    writer that changes `data` directly and forgets the stamp must not make it skip a CHANGED map: a sealed image carries a
    fingerprint of its contents (4096 strided samples), and the skip requires it unchanged."""
    import ctypes as C
    lib = hostlib.load()
    lib.mpmvs_host_test_seal.restype = C.c_int
    lib.mpmvs_host_test_seal.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_long, C.c_float]
    rng = np.random.default_rng(0)
    a = rng.uniform(1, 9, (1200, 1600)).astype(np.float32)
    assert lib.mpmvs_host_test_seal(a.ctypes.data, 1200, 1600, 0, 0, 0.0) == 1                 # untouched: still sealed
    assert lib.mpmvs_host_test_seal(a.ctypes.data, 1200, 1600, 0, a.size, 3.0) == 0            # a whole new map
    assert lib.mpmvs_host_test_seal(a.ctypes.data, 1200, 1600, 600 * 1600, 40 * 1600, 3.0) == 0  # a band of rows
    assert lib.mpmvs_host_test_seal(a.ctypes.data, 1200, 1600, a.size - 1, 1, 3.0) == 0        # the last element


def test_bench_quotes_pmc_figures_only_for_this_build_and_this_run(tmp_path, monkeypatch):
    """bench.py's roofline.traffic / valu_busy_from_profile come from a committed rocprofv3 PMC summary -- but only while that summary
    carries the hash of the HIP library in use and its kernel-trace average of k_update lies within 3 % of the run's own; otherwise
    the fields are None and a reason is given (a stale profile must not be quoted silently)"""
    import bench
    lib = tmp_path / "libfake.so"
    lib.write_bytes(b"binary one")
    monkeypatch.setenv("MPMVS_HIP_LIB", str(lib))
    h = bench.kernel_build_sha256()
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r99z_pmc_summary_cfg1.txt").write_text(
        f"# kernel_build_sha256: {h}\n# command: python3 bench.py\n# git_head: abc1234\n== kernel trace (ms) ==\n"
        "k_update           calls  126  total   335.914  avg   2.6660  min   2.4573  max   3.6558   91.2%\n"
        "== pmc_fetch (per-dispatch average) ==\n  k_update  vgpr/agpr/sgpr/scratch/lds = ('116', '0', '112', '352', '0')\n      FETCH_SIZE   n=126 avg=400000\n"
        "== pmc_write (per-dispatch average) ==\n  k_update  vgpr = x\n      WRITE_SIZE   n=126 avg=300000\n"
        "== pmc_sq1 ==\n  k_update  vgpr = x\n      SQ_ACTIVE_INST_VALU  n=126 avg=1.0e+09\n== pmc_grbm ==\n  k_update  vgpr = x\n      GRBM_GUI_ACTIVE  n=126 avg=5.0e+07\n")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    ok = bench.profile_counters(2.70)
    assert ok["reason"] is None and ok["traffic"] == 700000 * 1024.0 and ok["traffic_if_fetch_doubled"] == 1100000 * 1024.0
    assert abs(ok["valu_busy"] - 1.0e9 * 4 / (1024 * 5.0e7 / 8)) < 1e-4 and "abc1234" in ok["source"]
    far = bench.profile_counters(3.00)                       # this run's launches are 12 % slower than the profiled ones
    assert far["traffic"] is None and far["valu_busy"] is None and "3 %" in far["reason"]
    lib.write_bytes(b"binary two")                           # another build of the kernels
    stale = bench.profile_counters(2.70)
    assert stale["traffic"] is None and "another kernel build" in stale["reason"]
