"""ctypes binding of the C++ host layer (host/libmpmvs_host.so): the planar-prior
construction of ProcessProblem (reference src/PatchMatch.cpp:532-604) and the
single-Problem pass schedule (reference src/main.cpp:20-41)."""
import ctypes as C
import os

import numpy as np

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "host", "libmpmvs_host.so")
SYMBOLS = ["mpmvs_host_triangulate_vertices", "mpmvs_host_delaunay", "mpmvs_host_build_prior", "mpmvs_host_run_pipeline"]
_cache = {}


def load():
    if "lib" not in _cache:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"host library not built: {LIB_PATH} (run __graft_entry__.build())")
        lib = C.CDLL(LIB_PATH)
        P = C.c_void_p
        lib.mpmvs_host_triangulate_vertices.restype = C.c_int
        lib.mpmvs_host_triangulate_vertices.argtypes = [C.c_int, C.c_int, P, P, C.c_int, P, C.c_int]
        lib.mpmvs_host_delaunay.restype = C.c_int
        lib.mpmvs_host_delaunay.argtypes = [C.c_int, C.c_int, P, C.c_int, P, C.c_int]
        lib.mpmvs_host_build_prior.restype = C.c_int
        lib.mpmvs_host_build_prior.argtypes = [C.POINTER(_abi.Camera), C.c_int, C.c_int, P, P, P, C.c_int, C.c_float, C.c_float, P, P]
        lib.mpmvs_host_run_pipeline.restype = C.c_int
        lib.mpmvs_host_run_pipeline.argtypes = [C.c_int, C.c_int, C.POINTER(_abi.Camera), C.POINTER(C.POINTER(C.c_float)), C.c_int, C.c_int,
                                                C.c_int, C.c_int, C.c_uint64, C.POINTER(C.POINTER(C.c_float)), P, P, P]
        _cache["lib"] = lib
    return _cache["lib"]


def triangulate_vertices(costs, geom_costs=None, geom_planar_prior=False):
    lib = load()
    costs = np.ascontiguousarray(costs, np.float32)
    h, w = costs.shape
    g = np.ascontiguousarray(geom_costs, np.float32) if geom_costs is not None else None
    cap = 3 * ((h + 4) // 5) * ((w + 4) // 5) + 8
    out = np.empty((cap, 2), np.int32)
    n = lib.mpmvs_host_triangulate_vertices(w, h, costs.ctypes.data, g.ctypes.data if g is not None else None,
                                            1 if geom_planar_prior else 0, out.ctypes.data, cap)
    return out[:n].copy()


def delaunay(w, h, points):
    lib = load()
    pts = np.ascontiguousarray(points, np.int32).reshape(-1, 2)
    cap = 2 * len(pts) + 8
    out = np.empty((cap, 3, 2), np.int32)
    n = lib.mpmvs_host_delaunay(w, h, pts.ctypes.data, len(pts), out.ctypes.data, cap)
    assert n <= cap
    return out[:n].copy()


def build_prior(cam, planes, costs, geom_costs, geom_planar_prior, depth_min, depth_max):
    """(prior planes HxWx4, mask HxW u32, number of triangles)"""
    lib = load()
    planes = np.ascontiguousarray(planes, np.float32)
    costs = np.ascontiguousarray(costs, np.float32)
    h, w = costs.shape
    g = np.ascontiguousarray(geom_costs, np.float32) if geom_costs is not None else None
    prior = np.zeros((h, w, 4), np.float32)
    mask = np.zeros((h, w), np.uint32)
    n = lib.mpmvs_host_build_prior(C.byref(cam), w, h, planes.ctypes.data, costs.ctypes.data, g.ctypes.data if g is not None else None,
                                   1 if geom_planar_prior else 0, float(depth_min), float(depth_max), prior.ctypes.data, mask.ctypes.data)
    return prior, mask, n


def run_pipeline(device, cams, images, max_scale, geom_iterations, planar_prior, geom_planar_prior, seed, src_depths=None):
    """one Problem through the reference's pass schedule on the HIP path; returns depth, normal, cost"""
    lib = load()
    n = len(cams)
    imgs = [np.ascontiguousarray(im, np.float32) for im in images]
    cam_arr = (_abi.Camera * n)(*cams)
    ptrs = (C.POINTER(C.c_float) * n)(*[im.ctypes.data_as(C.POINTER(C.c_float)) for im in imgs])
    dptr = None
    if src_depths is not None:
        ds = [np.ascontiguousarray(d, np.float32) for d in src_depths]
        dptr = (C.POINTER(C.c_float) * (n - 1))(*[d.ctypes.data_as(C.POINTER(C.c_float)) for d in ds])
    h, w = imgs[0].shape
    depth = np.empty((h, w), np.float32)
    normal = np.empty((h, w, 3), np.float32)
    cost = np.empty((h, w), np.float32)
    rc = lib.mpmvs_host_run_pipeline(int(device), n, cam_arr, ptrs, int(max_scale), int(geom_iterations), 1 if planar_prior else 0,
                                     1 if geom_planar_prior else 0, int(seed), dptr, depth.ctypes.data, normal.ctypes.data, cost.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"mpmvs_host_run_pipeline failed ({rc})")
    return depth, normal, cost
