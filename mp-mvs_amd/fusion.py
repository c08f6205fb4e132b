"""Depth-map fusion front end: marshals a scene into the flat C
signature shared by `mpmvs_fuse` (HIP, include/mpmvs.h) and, in tests, the oracle's
`orc_fuse`, and compacts the per-pixel outputs into the point list the reference
builds (reference src/PatchMatch.cpp:369-495: image order, then raster order)."""
import ctypes as C

import numpy as np

from ._abi import Camera

_PP_F = C.POINTER(C.POINTER(C.c_float))
_PP_U8 = C.POINTER(C.POINTER(C.c_ubyte))
FUSE_ARGTYPES_TAIL = [C.c_int, C.POINTER(Camera), C.POINTER(C.c_int), _PP_F, _PP_F, _PP_F, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int,
                      _PP_U8, _PP_F, _PP_U8]


def call_fuse(fn, lead_args, cams, estimate, depths, normals, grays, sources, use_dynamic=True):
    """fn(*lead_args, n, cams, estimate, depths, normals, gray, src_off, src_ids, use_dynamic, valid, points9, masks)
    sources[k] = source-view ids of image k (without k itself).  Returns
    (points [M, 9] in image-then-raster order, valid list, masks list)."""
    n = len(cams)
    d = [np.ascontiguousarray(x, np.float32) for x in depths]
    nm = [np.ascontiguousarray(x, np.float32) for x in normals]
    g = [np.ascontiguousarray(x, np.float32) for x in grays]
    for k in range(n):
        assert d[k].shape == (cams[k].height, cams[k].width) and nm[k].shape == d[k].shape + (3,) and g[k].shape == d[k].shape
    ids, off = [], [0]
    for k in range(n):
        ids += [k] + list(sources[k])
        off.append(len(ids))
    valid = [np.zeros(x.shape, np.uint8) for x in d]
    pts = [np.zeros(x.shape + (9,), np.float32) for x in d]
    masks = [np.zeros(x.shape, np.uint8) for x in d]
    fp = lambda arrs: (C.POINTER(C.c_float) * n)(*[a.ctypes.data_as(C.POINTER(C.c_float)) for a in arrs])
    up = lambda arrs: (C.POINTER(C.c_ubyte) * n)(*[a.ctypes.data_as(C.POINTER(C.c_ubyte)) for a in arrs])
    rc = fn(*lead_args, n, (Camera * n)(*cams), (C.c_int * n)(*[1 if e else 0 for e in estimate]), fp(d), fp(nm), fp(g),
            (C.c_int * (n + 1))(*off), (C.c_int * len(ids))(*ids), 1 if use_dynamic else 0, up(valid), fp(pts), up(masks))
    if rc != 0:
        raise RuntimeError(f"fuse failed ({rc})")
    cloud = np.concatenate([p[v.astype(bool)] for p, v in zip(pts, valid)], 0) if n else np.zeros((0, 9), np.float32)
    return cloud, valid, masks


def fuse(cams, estimate, depths, normals, grays, sources, use_dynamic=True, device=0):
    """fusion on the MI355X (mpmvs_fuse)"""
    from . import engine
    lib, _ = engine.load()
    fn = lib.mpmvs_fuse
    fn.restype = C.c_int
    fn.argtypes = [C.c_int] + FUSE_ARGTYPES_TAIL
    return call_fuse(fn, (int(device),), cams, estimate, depths, normals, grays, sources, use_dynamic)


def last_kernel_ms():
    from . import engine
    lib, _ = engine.load()
    lib.mpmvs_fuse_kernel_ms.restype = C.c_float
    return float(lib.mpmvs_fuse_kernel_ms())
