"""mpmvs_sky_bilateral on the MI355X against the oracle: bit exact (the output is a thresholded mask, so any
arithmetic difference would show as flipped pixels), including image borders and sizes that are not tile multiples."""
import importlib

import numpy as np
import pytest

from test_sky_cpu import sky_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("size", [(96, 72), (33, 17), (5, 3), (130, 49)])
def test_sky_bilateral_bit_exact(oracle, engine, size):
    fusion = importlib.import_module("mp-mvs_amd.fusion")
    img, coarse, _ = sky_scene(*size, seed=size[0])
    got = fusion.sky_bilateral(img, coarse)
    assert np.array_equal(got, oracle.sky_bilateral(img, coarse))
    rng = np.random.default_rng(0)
    noise_img = rng.integers(0, 256, img.shape, dtype=np.uint8)
    noise_mask = rng.random(coarse.shape).astype(np.float32)
    assert np.array_equal(fusion.sky_bilateral(noise_img, noise_mask), oracle.sky_bilateral(noise_img, noise_mask))
    with pytest.raises(ValueError):
        fusion.sky_bilateral(img, coarse[:, :-1])


def test_sky_masks_through_the_folder_flow(pm, oracle, engine, tmp_path):
    """GenerateSkyRegionMask's file flow (reference src/PatchMatch.cpp:4-57) minus the network, then RunFusion with sky_seg
    (:360-388): JPEG colour images in, coarse skymask.pgm per image in, skymask_refine.pgm out, sky pixels absent from the cloud"""
    hostlib = importlib.import_module("mp-mvs_amd.hostlib")
    fusion = importlib.import_module("mp-mvs_amd.fusion")
    from test_fusion_cpu import _scene
    sc, cams, depths, normals, grays, neigh = _scene(pm, size=(96, 72))
    cols = [np.stack([g, 255 - g, g // 2 + 20], -1).astype(np.uint8) for g in (np.asarray(x).astype(np.uint8) for x in grays)]   # R,G,B for the files
    hostlib.write_dataset(str(tmp_path), cams, cols, neigh, fmt="jpg", jpeg_options=dict(quality=95, subsampling=2))
    coarse = []
    for i in range(6):
        d = tmp_path / "MPMVS" / f"2333_{i:08d}"
        d.mkdir(parents=True)
        hostlib.write_dmb(d / "depths.dmb", depths[i])
        hostlib.write_dmb(d / "normals.dmb", normals[i])
        m = np.zeros(depths[i].shape, np.uint8)
        m[: 10 + i] = 255
        coarse.append(m)
        open(d / "skymask.pgm", "wb").write(b"P5\n%d %d\n255\n" % (m.shape[1], m.shape[0]) + m.tobytes())
    assert hostlib.refine_sky_masks(tmp_path) == 6
    bgr = [hostlib.read_image(tmp_path / "images" / f"{i:08d}.jpg", 3) for i in range(6)]
    sky = []
    for i in range(6):
        want = oracle.sky_bilateral(bgr[i], coarse[i].astype(np.float32) / np.float32(255))
        got = hostlib.read_image(tmp_path / "MPMVS" / f"2333_{i:08d}" / "skymask_refine.pgm", 1)
        assert np.array_equal(got, want.astype(np.uint8)) and 0 < (got > 0).mean() < 0.5
        sky.append(got)
    file_cams = []
    for i in range(6):
        c = hostlib.read_camera(tmp_path / "cams" / f"{i:08d}_cam.txt")
        c.height, c.width = depths[i].shape
        file_cams.append(c)
    n_sky = hostlib.fuse_folder(tmp_path, sky_seg=True)
    cloud, valid, _ = oracle.fuse(file_cams, [True] * 6, depths, normals, bgr, neigh, sky=sky)
    assert n_sky == len(cloud) and all(v[s > 0].sum() == 0 for v, s in zip(valid, sky))
    body = open(tmp_path / "MPMVS" / "MPMVS_model.ply", "rb").read().split(b"end_header\n", 1)[1]
    rec = np.frombuffer(body, np.uint8).reshape(n_sky, 27)
    assert np.array_equal(rec[:, :12].copy().view(np.float32), cloud[:, :3])
    assert np.array_equal(rec[:, 24:27], cloud[:, [8, 7, 6]].astype(np.int32).astype(np.uint8))     # PLY stores red, green, blue from B,G,R (reference :181-186)
    assert hostlib.fuse_folder(tmp_path, sky_seg=False) > n_sky


def test_sky_and_fuse_argument_errors(pm, engine):
    """bad arguments come back as error codes / exceptions, never as a fault"""
    import ctypes as C
    fusion = importlib.import_module("mp-mvs_amd.fusion")
    lib, _ = engine.load()
    lib.mpmvs_sky_bilateral.restype = C.c_int
    lib.mpmvs_sky_bilateral.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    img = np.zeros((4, 4, 3), np.uint8)
    m = np.zeros((4, 4), np.float32)
    o = np.zeros((4, 4), np.float32)
    assert lib.mpmvs_sky_bilateral(0, None, m.ctypes.data, o.ctypes.data, 4, 4) != 0
    assert lib.mpmvs_sky_bilateral(0, img.ctypes.data, m.ctypes.data, o.ctypes.data, 0, 4) != 0
    assert lib.mpmvs_sky_bilateral(99, img.ctypes.data, m.ctypes.data, o.ctypes.data, 4, 4) != 0
    lib.mpmvs_fuse.restype = C.c_int
    lib.mpmvs_fuse.argtypes = [C.c_int] + fusion.FUSE_ARGTYPES_TAIL
    assert lib.mpmvs_fuse(0, 0, None, None, None, None, None, 3, None, None, None, 1, None, None, None) != 0      # n = 0
    sc, neigh = pm.synth.make_grid_scene(32, 24, 2, 1, spacing=0.4, quantize=True)
    cams = [v.cam for v in sc.views]
    d = [v.gt_depth for v in sc.views]
    n = [np.zeros(x.shape + (3,), np.float32) for x in d]
    with pytest.raises(AssertionError):
        fusion.fuse(cams, [True, True], d, n, [np.zeros((24, 32, 2), np.uint8)] * 2, neigh)                         # 2 colour channels
