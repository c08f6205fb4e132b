"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by the product package.
"""
import ctypes as C
import importlib
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_abi = importlib.import_module("mp-mvs_amd._abi")

_cache = {}


def build(force=False):
    """make decides what is stale (both sources are prerequisites of liboracle.so)"""
    subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []) + ["liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


def lib():
    if "lib" not in _cache:
        build()
        l = C.CDLL(_LIB)
        fns = _abi.bind(l, "orc_")
        l.orc_create.restype = C.c_void_p
        l.orc_create.argtypes = []
        l.orc_eval_ncc_literal.restype = C.c_int
        l.orc_eval_ncc_literal.argtypes = [C.c_void_p, C.POINTER(_abi.PatchMatchParams), C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        l.orc_num_threads.restype = C.c_int
        l.orc_set_num_threads.argtypes = [C.c_int]
        _cache["lib"] = l
        _cache["fns"] = fns
    return _cache["lib"], _cache["fns"]


def create():
    l, fns = lib()
    return _abi.PatchMatchHandle(fns, l.orc_create())


def fns():
    return lib()[1]


def num_threads():
    return lib()[0].orc_num_threads()


def set_num_threads(n):
    lib()[0].orc_set_num_threads(int(n))


def set_literal_mode(handle, mode):
    """0: canonical arithmetic (default); 1: the WHOLE path of run()/step() in the reference's literal operation order with IEEE
    operations and libm; 2: additionally CUDA's 8-bit texture interpolation fractions; 3: a model of the reference's
    --use_fast_math build (approximate exp / sin / cos / reciprocal, contracted multiply-adds, 8-bit fractions).  Measurement only."""
    l, _ = lib()
    l.orc_set_literal_mode.restype = C.c_int
    l.orc_set_literal_mode.argtypes = [C.c_void_p, C.c_int]
    if l.orc_set_literal_mode(handle._ctx, int(mode)) != 0:
        raise RuntimeError("orc_set_literal_mode failed")


def set_texture_q8(handle, on=True):
    """canonical arithmetic with CUDA's 8-bit texture interpolation fractions: the checker of the opt-in build libmpmvs_hip_q8.so"""
    l, _ = lib()
    l.orc_set_texture_q8.restype = C.c_int
    l.orc_set_texture_q8.argtypes = [C.c_void_p, C.c_int]
    if l.orc_set_texture_q8(handle._ctx, 1 if on else 0) != 0:
        raise RuntimeError("orc_set_texture_q8 failed")


def eval_ncc_literal(handle, params, planes_cam, scale, quantize_fraction=False, mode=None):
    """NCC in the reference's literal operation order (see pm_oracle.cpp); measurement only.  mode 1: IEEE + libm; 2: with CUDA's
    8-bit texture fractions (= quantize_fraction); 3: the fast-math model of the reference's build"""
    if mode is not None:
        quantize_fraction = int(mode) - 1
    import numpy as np
    l, _ = lib()
    p = np.ascontiguousarray(planes_cam, np.float32)
    out = np.empty((params.num_images - 1, handle.H, handle.W), np.float32)
    rc = l.orc_eval_ncc_literal(handle._ctx, C.byref(params), p.ctypes.data, int(scale), int(quantize_fraction), out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"orc_eval_ncc_literal failed ({rc})")
    return out


def eval_geom_literal(handle, params, planes_cam):
    """geometric-consistency cost through the reference's literal chain of world coordinates (9 divisions per check); the
    canonical form (two composed projective maps, handle.eval_geom) is measured against it"""
    import numpy as np
    l, _ = lib()
    l.orc_eval_geom_literal.restype = C.c_int
    l.orc_eval_geom_literal.argtypes = [C.c_void_p, C.POINTER(_abi.PatchMatchParams), C.c_void_p, C.c_void_p]
    p = np.ascontiguousarray(planes_cam, np.float32)
    out = np.empty((params.num_images - 1, handle.H, handle.W), np.float32)
    rc = l.orc_eval_geom_literal(handle._ctx, C.byref(params), p.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"orc_eval_geom_literal failed ({rc})")
    return out


def fuse(cams, estimate, depths, normals, colors, sources, use_dynamic=True, sequential_literal=False, sky=None, reference_order=False):
    """oracle fusion: mode 0 = the snapshot formulation (the GPU's default), mode 1 = the reference's literal sequential
    order with libm (measurement only), mode 2 (reference_order) = the sequential order in the canonical arithmetic = what
    the GPU's MPMVS_FUSE_REFERENCE_ORDER mode computes"""
    fusion = importlib.import_module("mp-mvs_amd.fusion")
    l, _ = lib()
    fn = l.orc_fuse
    fn.restype = C.c_int
    fn.argtypes = [C.c_int] + fusion.FUSE_ARGTYPES_TAIL
    return fusion.call_fuse(fn, (1 if sequential_literal else (2 if reference_order else 0),), cams, estimate, depths, normals, colors, sources, use_dynamic, sky)


def sky_bilateral(bgr, mask, literal=False):
    """oracle of mpmvs_sky_bilateral (mode 0) or the reference's literal arithmetic (mode 1, measurement only)"""
    import numpy as np
    l, _ = lib()
    fn = l.orc_sky_bilateral
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    bgr = np.ascontiguousarray(bgr, np.uint8)
    mask = np.ascontiguousarray(mask, np.float32)
    assert bgr.shape == mask.shape + (3,)
    out = np.empty(mask.shape, np.float32)
    if fn(1 if literal else 0, bgr.ctypes.data, mask.ctypes.data, out.ctypes.data, mask.shape[0], mask.shape[1]) != 0:
        raise RuntimeError("orc_sky_bilateral failed")
    return out
