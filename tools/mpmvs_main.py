#!/usr/bin/env python3
"""The reference's main() flow (src/main.cpp:6-55) over a dataset folder, driven by the reference's own config.yaml keys
(src/utility.cpp:8-35; sample: config/config.yaml): photometric pass, geometric-consistency passes with planar prior,
optional sky-mask refinement, fusion -> <Output-folder>/MPMVS/MPMVS_model.ply.  A thin convenience wrapper around
mp-mvs_amd/host (libmpmvs_host.so); all work happens in the C++/HIP libraries.

  python tools/mpmvs_main.py --config config.yaml [--device 0] [--seed 12345]

Differences from the reference: the segmentation network is not run -- with `Sky segment: 1` the coarse masks are
expected at <Input-folder>/MPMVS/2333_<id>/skymask.{jpg,pgm} (where the reference writes them) -- and the
`Save ... as JPG` visualisation switches are ignored."""
import argparse
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def read_config(path):
    """OpenCV FileStorage YAML ('%YAML:1.0' header, keys with spaces) -> dict"""
    import yaml
    text = open(path).read()
    lines = [l for l in text.splitlines() if not l.startswith("%YAML") and l.strip() != "---"]
    return yaml.safe_load("\n".join(lines)) or {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--seed", type=int, default=int(time.time()))    # the reference seeds from clock64()
    ap.add_argument("--jacobi-workers", type=int, default=0,
                    help="0 (default): the reference's sequential, in-place pass order; N > 0: Jacobi order (every pass reads the previous "
                         "pass's maps) with N host threads, which overlaps the host work of one image with the kernels of others")
    ap.add_argument("--devices", default=None, help="comma-separated GPU indices for --jacobi-workers (default: --device)")
    a = ap.parse_args()
    cfg = read_config(a.config)
    folder = str(cfg["Input-folder"]).rstrip("/")
    out = str(cfg.get("Output-folder", folder)).rstrip("/")
    if os.path.realpath(out) != os.path.realpath(folder):
        raise SystemExit("Output-folder must equal Input-folder: the geometric passes read <Input-folder>/MPMVS (reference src/PatchMatch.cpp:941)")
    hostlib = importlib.import_module("mp-mvs_amd.hostlib")
    max_src = int(cfg.get("Max source images num", 20))
    max_size = int(cfg.get("Max image size", 3200))
    t0 = time.perf_counter()
    schedule = (int(cfg.get("Geometric consistency iterations", 2)), bool(cfg.get("Planer prior", 1)), bool(cfg.get("Geometric consistency planer prior", 1)))
    if a.jacobi_workers > 0:
        devices = tuple(int(d) for d in a.devices.split(",")) if a.devices else (a.device,)
        hostlib.run_folder_jacobi(folder, devices, a.jacobi_workers, max_src, *schedule, 2, a.seed, max_size)
    else:
        hostlib.run_folder(folder, a.device, max_src, *schedule, 2, a.seed, max_size)
    t1 = time.perf_counter()
    sky = bool(cfg.get("Sky segment", 0))
    n_masks = hostlib.refine_sky_masks(folder, a.device, max_src, max_size) if sky else 0
    t2 = time.perf_counter()
    n = hostlib.fuse_folder(folder, a.device, max_src, bool(cfg.get("Use dynamic_consistency to fuse", 1)), sky)
    t3 = time.perf_counter()
    print(json.dumps({"folder": folder, "seconds": {"depth_maps": round(t1 - t0, 3), "sky_masks": round(t2 - t1, 3), "fusion": round(t3 - t2, 3)},
                      "sky_masks": n_masks, "fused_points": n, "ply": os.path.join(folder, "MPMVS", "MPMVS_model.ply")}))


if __name__ == "__main__":
    main()
