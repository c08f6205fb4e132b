#!/usr/bin/env python3
"""Generates tests/golden/resize_golden_v1.npz: bilinear down-scaling of fp32 images as PatchMatchInit's "Adjust image scale"
asks for it (reference src/PatchMatch.cpp:893-925: cv::resize(image, Size(new_cols, new_rows), 0, 0, INTER_LINEAR) on CV_32F,
target size = round(size * min(max / cols, max / rows))), computed WITHOUT this repository: torch.nn.functional.interpolate(
mode="bilinear", align_corners=False, antialias=False) has the half-pixel geometry of cv::resize INTER_LINEAR on float
images (sample position (x + 0.5) * src / dst - 0.5, edge texels replicated, no prefilter).  Independent second
implementation of the same published rule; tolerance in the replay test 1e-5 relative to the 0..255 range.

Run in the build container (needs torch, numpy): python tests/golden/make_resize_golden.py"""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def target_size(cols, rows, max_image_size):
    """reference src/PatchMatch.cpp:898-903, float arithmetic as written there"""
    fx = np.float32(max_image_size) / np.float32(cols)
    fy = np.float32(max_image_size) / np.float32(rows)
    f = min(fx, fy)
    # std::round: half away from zero
    new_cols = int(np.floor(np.float32(cols) * f + np.float32(0.5)))
    new_rows = int(np.floor(np.float32(rows) * f + np.float32(0.5)))
    return new_cols, new_rows


def main():
    rng = np.random.default_rng(20240309)
    out = {}
    # (cols, rows, max_image_size): landscape with a non-integer ratio, portrait, and the shipped limit on a 2:1 reduction
    cases = [(207, 154, 80), (126, 195, 100), (320, 240, 160), (101, 67, 100)]
    for k, (cols, rows, mx) in enumerate(cases):
        y, x = np.mgrid[0:rows, 0:cols].astype(np.float64)
        img = (127.0 + 60.0 * np.sin(0.11 * x) * np.cos(0.07 * y) + 40.0 * np.sin(0.31 * x + 0.013 * y) + 20.0 * rng.standard_normal((rows, cols)))
        img = np.clip(img, 0.0, 255.0).astype(np.float32)
        if k == 2:
            img = np.rint(img).astype(np.float32)   # an 8-bit image, as imread delivers it
        nc, nr = target_size(cols, rows, mx)
        t = torch.from_numpy(img)[None, None].double()   # float64 interpolation: the reference value, rounded once to fp32
        res = torch.nn.functional.interpolate(t, size=(nr, nc), mode="bilinear", align_corners=False, antialias=False)[0, 0]
        out[f"src{k}"] = img
        out[f"dst{k}"] = res.numpy().astype(np.float32)
        out[f"max{k}"] = np.int32(mx)
    out["n"] = np.int32(len(cases))
    np.savez_compressed(os.path.join(HERE, "resize_golden_v1.npz"), **out)
    for k in range(len(cases)):
        print(k, out[f"src{k}"].shape, "->", out[f"dst{k}"].shape)


if __name__ == "__main__":
    main()
