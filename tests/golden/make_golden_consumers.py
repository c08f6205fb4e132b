#!/usr/bin/env python3
"""Generates tests/golden/consumers_golden_v1.npz: regression pins for the rows either side of the hot path.

  fusion / sky filter   the CPU oracle's outputs (oracle/fusion_oracle.cpp) on a small seeded scene -- like
                        pm_golden_v3.npz these pin the oracle's own arithmetic (the reference cannot be run here);
  JPEG                  three small JPEG files and the pixels libjpeg-turbo (through PIL) decodes from them -- these
                        ARE third-party answers: the decoder of mp-mvs_amd/host/jpeg_decode.cpp must reproduce them.

Re-run only when the canonical arithmetic changes on purpose:   python tests/golden/make_golden_consumers.py
"""
import ctypes
import importlib
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pm = importlib.import_module("mp-mvs_amd")
from oracle import binding as ob  # noqa: E402
from test_fusion_cpu import _colours_and_sky, _scene  # noqa: E402
from test_sky_cpu import sky_scene  # noqa: E402


def cam_bytes(cam):
    return np.frombuffer(ctypes.string_at(ctypes.addressof(cam), ctypes.sizeof(cam)), np.uint8).copy()


def main():
    out = {}
    sc, cams, depths, normals, grays, neigh = _scene(pm, size=(48, 36))
    cols, sky = _colours_and_sky(grays)
    out["fuse_cams"] = np.stack([cam_bytes(c) for c in cams])
    out["fuse_depths"] = np.stack(depths)
    out["fuse_normals"] = np.stack(normals)
    out["fuse_colors"] = np.stack(cols)
    out["fuse_sky"] = np.stack([np.zeros(depths[0].shape, np.uint8) if m is None else m for m in sky])
    out["fuse_sky_present"] = np.array([m is not None for m in sky])
    out["fuse_neigh"] = np.array(neigh, np.int32)
    for tag, dyn, use_sky in (("dyn", True, False), ("static", False, False), ("sky", True, True)):
        cloud, valid, masks = ob.fuse(cams, [True] * len(cams), depths, normals, cols, neigh, use_dynamic=dyn, sky=sky if use_sky else None)
        out[f"fuse_{tag}_cloud"], out[f"fuse_{tag}_valid"], out[f"fuse_{tag}_masks"] = cloud, np.stack(valid), np.stack(masks)
    img, coarse, _ = sky_scene(72, 54, seed=11)
    out["sky_img"], out["sky_coarse"], out["sky_out"] = img, coarse, ob.sky_bilateral(img, coarse)

    from PIL import Image
    rng = np.random.default_rng(4)
    yy, xx = np.mgrid[0:45, 0:61]
    pic = np.clip(np.stack([127 + 100 * np.sin(xx / 7.0) * np.cos(yy / 5.0), 127 + 80 * np.cos(xx / 13.0 + yy / 9.0), (xx * 3 + yy * 2) % 256], -1)
                  + rng.normal(0, 10, (45, 61, 3)), 0, 255).astype(np.uint8)
    for tag, grey, opts in (("base420", False, dict(quality=85, subsampling=2)), ("prog444", False, dict(quality=92, subsampling=0, progressive=True)),
                            ("grey_rst", True, dict(quality=75, restart_marker_blocks=2))):
        buf = io.BytesIO()
        Image.fromarray(pic[..., 1] if grey else pic).save(buf, "JPEG", **opts)
        data = buf.getvalue()
        lum = Image.open(io.BytesIO(data))
        lum.draft("L", lum.size)
        out[f"jpeg_{tag}_file"] = np.frombuffer(data, np.uint8)
        out[f"jpeg_{tag}_gray"] = np.asarray(lum)
        out[f"jpeg_{tag}_bgr"] = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))[..., ::-1].copy()
    path = os.path.join(ROOT, "tests", "golden", "consumers_golden_v1.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
