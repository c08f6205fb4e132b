"""mp-mvs_amd: MI355X-native PatchMatch hot path of MP-MVS behind the reference's
PatchMatch.h interface.  The compute lives in
csrc/ (hand-written HIP for gfx950, C ABI in include/mpmvs.h); this package is
the thin Python host layer used by bench.py and the tests.
"""
from . import _abi, synth  # noqa: F401
from ._abi import (Camera, PatchMatchParams, make_camera, KIND_INIT, KIND_BLACK, KIND_RED,  # noqa: F401
                   KIND_DEPTH_NORMAL, KIND_FILTER_BLACK, KIND_FILTER_RED, MAX_SRC_VIEWS)
