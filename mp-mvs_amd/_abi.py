"""ctypes view of the C ABI declared in include/mpmvs.h.

`Camera` and `PatchMatchParams` are byte-compatible with the reference PODs
(reference include/PatchMatch.h:35-67); the kernel kinds follow the launch
order of the reference's PatchMatchCUDA::Run() (reference src/PatchMatch.cu:1188-1254).
"""
import ctypes as C

import numpy as np

MAX_SRC_VIEWS = 32  # reference src/PatchMatch.cu:500 (cost_vector[32]) and the u32 view mask

KIND_INIT = 0          # InitializeScore        (reference .cu:536)
KIND_BLACK = 1         # BlackPixelUpdate       (reference .cu:1000)
KIND_RED = 2           # RedPixelUpdate         (reference .cu:1011)
KIND_DEPTH_NORMAL = 3  # GetDepthandNormal      (reference .cu:1021)
KIND_FILTER_BLACK = 4  # BlackPixelFilter       (reference .cu:1152)
KIND_FILTER_RED = 5    # RedPixelFilter         (reference .cu:1164)


class Camera(C.Structure):
    """reference include/PatchMatch.h:35-46 (112 bytes, height before width)."""
    _fields_ = [
        ("K", C.c_float * 9),
        ("R", C.c_float * 9),
        ("t", C.c_float * 3),
        ("C", C.c_float * 3),
        ("height", C.c_int),
        ("width", C.c_int),
        ("depth_min", C.c_float),
        ("depth_max", C.c_float),
    ]


class PatchMatchParams(C.Structure):
    """reference include/PatchMatch.h:48-67 (56 bytes), same defaults."""
    _fields_ = [
        ("max_iterations", C.c_int),
        ("nSizeHalfWindow", C.c_int),
        ("num_images", C.c_int),
        ("max_image_size", C.c_int),
        ("nSizeStep", C.c_int),
        ("sigma_spatial", C.c_float),
        ("sigma_color", C.c_float),
        ("top_k", C.c_int),
        ("depth_min", C.c_float),
        ("depth_max", C.c_float),
        ("max_scale", C.c_int),
        ("scaled_cols", C.c_float),
        ("scaled_rows", C.c_float),
        ("geom_consistency", C.c_bool),
        ("geomPlanarPrior", C.c_bool),
        ("planar_prior", C.c_bool),
    ]

    def __init__(self, **kw):
        super().__init__()
        self.max_iterations = 3
        self.nSizeHalfWindow = 5
        self.num_images = 5
        self.max_image_size = 3200
        self.nSizeStep = 2
        self.sigma_spatial = 5.0
        self.sigma_color = 3.0
        self.top_k = 4
        self.depth_min = 0.0
        self.depth_max = 1.0
        self.max_scale = 2
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


assert C.sizeof(Camera) == 112
assert C.sizeof(PatchMatchParams) == 56


def make_camera(K, R, t, height, width, depth_min, depth_max):
    """Camera from 3x3 K, 3x3 R, 3-vector t; C = -R^T t as in reference
    src/PatchMatch.cpp:134-136."""
    K = np.asarray(K, np.float64).reshape(3, 3)
    R = np.asarray(R, np.float64).reshape(3, 3)
    t = np.asarray(t, np.float64).reshape(3)
    cam = Camera()
    cam.K[:] = [float(np.float32(v)) for v in K.reshape(-1)]
    cam.R[:] = [float(np.float32(v)) for v in R.reshape(-1)]
    cam.t[:] = [float(np.float32(v)) for v in t]
    Cc = -R.T @ t
    cam.C[:] = [float(np.float32(v)) for v in Cc]
    cam.height = int(height)
    cam.width = int(width)
    cam.depth_min = float(depth_min)
    cam.depth_max = float(depth_max)
    return cam


# name -> (restype, argtypes); every entry point include/mpmvs.h declares.
_P = C.c_void_p
_FPP = C.POINTER(C.POINTER(C.c_float))
SIGNATURES = {
    "create": (_P, [C.c_int]),
    "destroy": (None, [_P]),
    "last_error": (C.c_char_p, [_P]),
    "set_views": (C.c_int, [_P, C.c_int, C.POINTER(Camera), _FPP, C.POINTER(C.c_size_t)]),
    "set_src_depths": (C.c_int, [_P, C.c_int, _FPP, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "set_state": (C.c_int, [_P, _P, _P]),
    "set_selected_views": (C.c_int, [_P, _P]),
    "set_prior": (C.c_int, [_P, _P, _P]),
    "run": (C.c_int, [_P, C.POINTER(PatchMatchParams), C.c_uint64]),
    "step": (C.c_int, [_P, C.POINTER(PatchMatchParams), C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_uint32]),
    "get": (C.c_int, [_P, _P, _P, _P]),
    "get_selected_views": (C.c_int, [_P, _P]),
    "eval_ncc": (C.c_int, [_P, C.POINTER(PatchMatchParams), _P, C.c_int, _P]),
    "eval_geom": (C.c_int, [_P, C.POINTER(PatchMatchParams), _P, _P]),
    "math": (C.c_int, [C.c_int, _P, _P, C.c_int]),
    "rng": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, _P]),
    "homography": (C.c_int, [_P, _P, C.c_int, _P]),
}


def bind(lib, prefix, names=None):
    """Attach restype/argtypes to `prefix + name` for every known entry point."""
    out = {}
    for name, (res, args) in SIGNATURES.items():
        if names is not None and name not in names:
            continue
        fn = getattr(lib, prefix + name)
        fn.restype = res
        fn.argtypes = args
        out[name] = fn
    return out


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class PatchMatchHandle:
    """Thin object wrapper over one C-ABI context (`<prefix>ctx*`).

    The same wrapper drives the HIP library (prefix `mpmvs_`) and, from tests
    only, the CPU oracle (prefix `orc_`): both export the same entry points.
    """

    def __init__(self, fns, ctx):
        self._f = fns
        self._ctx = ctx
        self.W = self.H = self.n_img = 0
        if not ctx:
            raise RuntimeError("context creation failed")

    def close(self):
        if self._ctx:
            self._f["destroy"](self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            msg = self._f["last_error"](self._ctx)
            raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    # -- uploads -----------------------------------------------------------
    def set_views(self, cams, images):
        n = len(cams)
        assert n == len(images)
        imgs = [_f32(im) for im in images]
        for cam, im in zip(cams, imgs):
            assert im.shape == (cam.height, cam.width), (im.shape, cam.height, cam.width)
        cam_arr = (Camera * n)(*cams)
        ptrs = (C.POINTER(C.c_float) * n)(*[im.ctypes.data_as(C.POINTER(C.c_float)) for im in imgs])
        pitches = (C.c_size_t * n)(*[im.strides[0] for im in imgs])
        self._chk(self._f["set_views"](self._ctx, n, cam_arr, ptrs, pitches), "set_views")
        self.n_img = n
        self.H, self.W = imgs[0].shape

    def set_src_depths(self, depths):
        n = len(depths)
        ds = [_f32(d) for d in depths]
        ptrs = (C.POINTER(C.c_float) * n)(*[d.ctypes.data_as(C.POINTER(C.c_float)) for d in ds])
        ws = (C.c_int * n)(*[d.shape[1] for d in ds])
        hs = (C.c_int * n)(*[d.shape[0] for d in ds])
        pitches = (C.c_size_t * n)(*[d.strides[0] for d in ds])
        self._chk(self._f["set_src_depths"](self._ctx, n, ptrs, ws, hs, pitches), "set_src_depths")

    def set_state(self, planes=None, costs=None):
        p = _f32(planes) if planes is not None else None
        c = _f32(costs) if costs is not None else None
        if p is not None:
            assert p.shape == (self.H, self.W, 4)
        if c is not None:
            assert c.shape == (self.H, self.W)
        self._chk(self._f["set_state"](self._ctx, p.ctypes.data if p is not None else None,
                                       c.ctypes.data if c is not None else None), "set_state")

    def set_selected_views(self, sel):
        s = np.ascontiguousarray(sel, dtype=np.uint32)
        assert s.shape == (self.H, self.W)
        self._chk(self._f["set_selected_views"](self._ctx, s.ctypes.data), "set_selected_views")

    def set_prior(self, prior_planes, mask):
        p = _f32(prior_planes)
        m = np.ascontiguousarray(mask, dtype=np.uint32)
        assert p.shape == (self.H, self.W, 4) and m.shape == (self.H, self.W)
        self._chk(self._f["set_prior"](self._ctx, p.ctypes.data, m.ctypes.data), "set_prior")

    # -- compute -----------------------------------------------------------
    def run(self, params, seed):
        self._chk(self._f["run"](self._ctx, C.byref(params), int(seed)), "run")

    def step(self, params, seed, kind, it=0, scale=0, launch=0):
        self._chk(self._f["step"](self._ctx, C.byref(params), int(seed), int(kind), int(it), int(scale), int(launch)), "step")

    def eval_ncc(self, params, planes_cam, scale):
        p = _f32(planes_cam)
        assert p.shape == (self.H, self.W, 4)
        out = np.empty((params.num_images - 1, self.H, self.W), np.float32)
        self._chk(self._f["eval_ncc"](self._ctx, C.byref(params), p.ctypes.data, int(scale), out.ctypes.data), "eval_ncc")
        return out

    def eval_geom(self, params, planes_cam):
        p = _f32(planes_cam)
        out = np.empty((params.num_images - 1, self.H, self.W), np.float32)
        self._chk(self._f["eval_geom"](self._ctx, C.byref(params), p.ctypes.data, out.ctypes.data), "eval_geom")
        return out

    def homography(self, plane, v):
        p = _f32(plane).reshape(4)
        out = np.empty(9, np.float32)
        self._chk(self._f["homography"](self._ctx, p.ctypes.data, int(v), out.ctypes.data), "homography")
        return out.reshape(3, 3)

    # -- downloads ---------------------------------------------------------
    def get(self, geom=False):
        planes = np.empty((self.H, self.W, 4), np.float32)
        costs = np.empty((self.H, self.W), np.float32)
        g = np.empty((self.H, self.W), np.float32) if geom else None
        self._chk(self._f["get"](self._ctx, planes.ctypes.data, costs.ctypes.data, g.ctypes.data if geom else None), "get")
        return (planes, costs, g) if geom else (planes, costs)

    def get_into(self, planes, costs, geom=None):
        """the same into caller-owned (e.g. pinned) float32 arrays of the right shape"""
        assert planes.shape == (self.H, self.W, 4) and costs.shape == (self.H, self.W) and planes.dtype == np.float32 and costs.dtype == np.float32
        self._chk(self._f["get"](self._ctx, planes.ctypes.data, costs.ctypes.data, geom.ctypes.data if geom is not None else None), "get")

    def get_selected_views(self):
        s = np.empty((self.H, self.W), np.uint32)
        self._chk(self._f["get_selected_views"](self._ctx, s.ctypes.data), "get_selected_views")
        return s


def math_probe(fns, fn_id, x):
    x = _f32(x).reshape(-1)
    out = np.empty_like(x)
    rc = fns["math"](int(fn_id), x.ctypes.data, out.ctypes.data, x.size)
    if rc != 0:
        raise RuntimeError(f"math probe {fn_id} failed ({rc})")
    return out


def rng_probe(fns, seed, pix, launch, n):
    out = np.empty(n, np.float32)
    rc = fns["rng"](int(seed), int(pix), int(launch), int(n), out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"rng probe failed ({rc})")
    return out
