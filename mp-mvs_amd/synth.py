"""Seeded synthetic multi-view scenes: a textured height field seen
by a grid of pinhole cameras, rendered by per-pixel ray / height-field
intersection.  No files, no OpenCV.  Used by bench.py and the tests; images are
fp32 in [0, 255] like the reference's (reference src/PatchMatch.cpp:877-882).
"""
import os
from dataclasses import dataclass, field
from typing import List

import numpy as np

from ._abi import Camera, make_camera

SCENE_SEED = 20240309


def height_field(X, Y):
    return 5.0 + 0.6 * np.sin(0.9 * X) * np.cos(0.7 * Y) + 0.25 * (0.3 * X - 0.2 * Y)


def _hash_noise(ix, iy, seed):
    """integer lattice hash -> [-1, 1)"""
    h = (ix.astype(np.int64) * 73856093) ^ (iy.astype(np.int64) * 19349663) ^ (int(seed) * 83492791)
    h = h.astype(np.uint64) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return h.astype(np.float64) / 2147483648.0 - 1.0


def albedo(X, Y, seed, fs):
    """procedural texture; `fs` scales the spatial frequencies so that small
    test images see the same per-pixel frequency content as 1600x1200."""
    a = (127.0 + 60.0 * np.sin(7.0 * fs * X) * np.cos(5.0 * fs * Y)
         + 40.0 * np.sin(23.0 * fs * X + 1.3 * fs * Y)
         + 25.0 * np.cos(41.0 * fs * Y - 3.0 * fs * X)
         + 12.0 * _hash_noise(np.floor(64.0 * fs * X), np.floor(64.0 * fs * Y), seed))
    return np.clip(a, 0.0, 255.0)


@dataclass
class View:
    cam: Camera
    image: np.ndarray      # (H, W) float32
    gt_depth: np.ndarray   # (H, W) float32, depth along the camera z axis
    K: np.ndarray
    R: np.ndarray
    C: np.ndarray


@dataclass
class Scene:
    width: int
    height: int
    views: List[View] = field(default_factory=list)

    def problem(self, ref, srcs):
        """(cams, images) of one Problem: reference view first (reference
        src/PatchMatch.cpp:867-890 orders srcID the same way)."""
        ids = [ref] + list(srcs)
        return [self.views[i].cam for i in ids], [self.views[i].image for i in ids]


def _small_rotation(rng, max_deg):
    if max_deg <= 0:
        return np.eye(3)
    yaw, pitch, roll = np.deg2rad(rng.uniform(-max_deg, max_deg, 3))
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    return Rz @ Rx @ Ry


def _render_rows(width, y0, y1, K, R, Cc, seed, fs):
    u, v = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(y0, y1, dtype=np.float64))
    rc = np.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], np.ones_like(u)], -1)
    rw = rc @ R  # R^T applied to each ray (row vectors); a stack of per-row products, so a band gives the bits of the whole
    d = np.full(u.shape, 5.0)
    for _ in range(24):
        X = Cc[0] + d * rw[..., 0]
        Y = Cc[1] + d * rw[..., 1]
        d = (height_field(X, Y) - Cc[2]) / rw[..., 2]
    X = Cc[0] + d * rw[..., 0]
    Y = Cc[1] + d * rw[..., 1]
    img = albedo(X, Y, seed, fs)
    return img.astype(np.float32), d.astype(np.float32)


def render_view(width, height, K, R, Cc, seed, fs):
    # every pixel is independent: large images are rendered in row bands on a few threads (numpy releases the GIL inside its
    # loops); the pixels are the same bits as in one piece
    bands = min(8, len(os.sched_getaffinity(0)), height // 64)
    if width * height < 400 * 300 or bands < 2:
        return _render_rows(width, 0, height, K, R, Cc, seed, fs)
    from concurrent.futures import ThreadPoolExecutor
    cuts = [height * i // bands for i in range(bands + 1)]
    with ThreadPoolExecutor(max_workers=bands) as pool:
        parts = list(pool.map(lambda i: _render_rows(width, cuts[i], cuts[i + 1], K, R, Cc, seed, fs), range(bands)))
    return np.concatenate([p[0] for p in parts], 0), np.concatenate([p[1] for p in parts], 0)


def make_view(width, height, Cc, R, Kv, seed, depth_min, depth_max, quantize):
    """one view of make_scene: camera with rotation R at centre Cc, rendered with the fp32-rounded camera"""
    fs = width / 1600.0
    t = -R @ Cc
    cam = make_camera(Kv, R, t, height, width, depth_min, depth_max)
    # render with the fp32-rounded camera so images and Camera agree
    Kf = np.array(cam.K, np.float64).reshape(3, 3)
    Rf = np.array(cam.R, np.float64).reshape(3, 3)
    Cf = np.array(cam.C, np.float64)
    img, gt = render_view(width, height, Kf, Rf, Cf, seed, fs)
    if quantize:
        img = np.rint(img).astype(np.float32)
    return View(cam, img, gt, Kf, Rf, Cf)


def make_scene(width, height, centers, seed=SCENE_SEED, rot_deg=2.0, depth_min=3.0, depth_max=8.0, quantize=False, focal_jitter=0.0, only=None):
    """Cameras at `centers` (N x 3 world positions), fx = fy = 0.9 W, principal
    point at the image centre; camera 0..N-1 in the order given.  quantize=True
    rounds the images to integers 0..255 like an 8-bit camera image (what the
    reference feeds its textures: imread(GRAYSCALE) -> CV_32F, src/PatchMatch.cpp:877-882).
    only: render just these view indices (the others are None) -- every view still gets the rotation and
    intrinsics it has in the full scene (the random draws are made for all views in order)."""
    rng = np.random.default_rng(seed)
    K = np.array([[0.9 * width, 0, width / 2.0], [0, 0.9 * width, height / 2.0], [0, 0, 1.0]])
    sc = Scene(width, height)
    for i, Cc in enumerate(np.asarray(centers, np.float64)):
        R = _small_rotation(rng, rot_deg)
        Kv = K.copy()
        if focal_jitter > 0:   # per-view intrinsics: different focal lengths and principal points
            f = rng.uniform(1.0 - focal_jitter, 1.0 + focal_jitter, 2)
            Kv[0, 0] *= f[0]
            Kv[1, 1] *= f[1]
            Kv[0, 2] += rng.uniform(-0.05, 0.05) * width
            Kv[1, 2] += rng.uniform(-0.05, 0.05) * height
        sc.views.append(make_view(width, height, Cc, R, Kv, seed, depth_min, depth_max, quantize) if only is None or i in only else None)
    return sc


def scene_cameras(width, height, centers, seed=SCENE_SEED, rot_deg=2.0, depth_min=3.0, depth_max=8.0, focal_jitter=0.0):
    """the cameras make_scene gives the views at `centers` (same random draws, same order), without rendering anything"""
    rng = np.random.default_rng(seed)
    K = np.array([[0.9 * width, 0, width / 2.0], [0, 0.9 * width, height / 2.0], [0, 0, 1.0]])
    cams = []
    for Cc in np.asarray(centers, np.float64):
        R = _small_rotation(rng, rot_deg)
        Kv = K.copy()
        if focal_jitter > 0:
            f = rng.uniform(1.0 - focal_jitter, 1.0 + focal_jitter, 2)
            Kv[0, 0] *= f[0]
            Kv[1, 1] *= f[1]
            Kv[0, 2] += rng.uniform(-0.05, 0.05) * width
            Kv[1, 2] += rng.uniform(-0.05, 0.05) * height
        cams.append(make_camera(Kv, R, -R @ Cc, height, width, depth_min, depth_max))
    return cams


# source-view order around the centre of a 3x3 grid: nearest first
_RING = [(1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, -1), (1, -1), (-1, 1)]


def make_problem_scene(width, height, n_src=8, spacing=0.15, **kw):  # kw: seed, rot_deg, depth_min, depth_max, quantize
    """One Problem: reference at the origin + n_src (<= 8) neighbours of the 3x3 grid."""
    assert 1 <= n_src <= 8
    centers = [(0.0, 0.0, 0.0)] + [(spacing * dx, spacing * dy, 0.0) for dx, dy in _RING[:n_src]]
    return make_scene(width, height, centers, **kw)


def make_grid_scene(width, height, nx, ny, spacing=0.15, **kw):
    """nx x ny camera grid (cfg 4 uses 8 x 8); returns the scene and, per camera,
    the list of its (up to 8) nearest grid neighbours as source views."""
    centers, neigh = [], []
    for j in range(ny):
        for i in range(nx):
            centers.append(((i - (nx - 1) / 2.0) * spacing, (j - (ny - 1) / 2.0) * spacing, 0.0))
    for j in range(ny):
        for i in range(nx):
            cand = []
            for jj in range(ny):
                for ii in range(nx):
                    if (ii, jj) != (i, j):
                        cand.append(((ii - i) ** 2 + (jj - j) ** 2, jj * nx + ii))
            cand.sort()
            neigh.append([c[1] for c in cand[:8]])
    return make_scene(width, height, centers, **kw), neigh


def kernel_depth_range(cam):
    """reference src/PatchMatch.cpp:929-930"""
    return np.float32(cam.depth_min) * np.float32(0.6), np.float32(cam.depth_max) * np.float32(1.2)
