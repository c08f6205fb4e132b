#!/usr/bin/env python3
"""Would a per-(pixel, candidate slot) cache of phase-A costs hit?  A cost is a pure function of (pixel, plane, view, window
scale): if the candidate a slot picks is the same neighbour as at the pixel's previous update AND that neighbour's plane has
not changed a bit since, the V evaluations of the slot could be reused exactly.  Measured on the CPU oracle, cfg-1 schedule.
usage: python tests/analysis/reuse_stats.py [W H] [iterations]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

DIRS = [
    [(-5, -6), (5, -6), (-6, -7), (6, -7), (-7, -8), (7, -8), (-8, -9), (8, -9), (-9, -10), (9, -10), (-10, -11), (10, -11)],
    [(-5, 6), (5, 6), (-6, 7), (6, 7), (-7, 8), (7, 8), (-8, 9), (8, 9), (-9, 10), (9, 10), (-10, 11), (10, 11)],
    [(-6, -5), (-6, 5), (-7, -6), (-7, 6), (-8, -7), (-8, 7), (-9, -8), (-9, 8), (-10, -9), (-10, 9), (-11, -10), (-11, 10)],
    [(6, -5), (6, 5), (7, -6), (7, 6), (8, -7), (8, 7), (9, -8), (9, 8), (10, -9), (10, 9), (11, -10), (11, 10)],
    [(0, -5), (0, -7), (0, -9), (0, -11), (0, -13), (0, -15), (0, -17), (0, -19), (0, -21), (0, -23)],
    [(0, 5), (0, 7), (0, 9), (0, 11), (0, 13), (0, 15), (0, 17), (0, 19), (0, 21), (0, 23)],
    [(-5, 0), (-7, 0), (-9, 0), (-11, 0), (-13, 0), (-15, 0), (-17, 0), (-19, 0), (-21, 0), (-23, 0)],
    [(5, 0), (7, 0), (9, 0), (11, 0), (13, 0), (15, 0), (17, 0), (19, 0), (21, 0), (23, 0)]]


def picks(costs):
    """per pixel and region the index of the lowest stored cost (first minimum in list order), -1 where the region is empty"""
    H, W = costs.shape
    yy, xx = np.mgrid[0:H, 0:W]
    out = np.full((8, H, W), -1, np.int64)
    for k, region in enumerate(DIRS):
        best = np.full((H, W), np.float32(3.402823466e+38))
        for dx, dy in region:
            nx, ny = xx + dx, yy + dy
            ok = (nx >= 0) & (ny >= 0) & (nx < W) & (ny < H)
            nc = np.where(ok, costs[np.clip(ny, 0, H - 1), np.clip(nx, 0, W - 1)], np.float32(np.inf))
            better = ok & (best > nc)
            best = np.where(better, nc, best)
            out[k] = np.where(better, ny * W + nx, out[k])
    return out


def main():
    argv = sys.argv[1:]
    W, H = (int(argv[0]), int(argv[1])) if len(argv) >= 2 else (400, 304)
    iters = int(argv[2]) if len(argv) >= 3 else 3
    V = 8
    pm = importlib.import_module("mp-mvs_amd")
    from oracle import binding as ob
    sc = pm.synth.make_problem_scene(W, H, n_src=V, quantize=True)
    cams, imgs = sc.problem(0, list(range(1, V + 1)))
    dmin, dmax = pm.synth.kernel_depth_range(cams[0])
    prm = pm.PatchMatchParams(num_images=V + 1, depth_min=float(dmin), depth_max=float(dmax), max_scale=0)
    o = ob.create()
    o.set_views(cams, imgs)
    o.step(prm, 12345, pm.KIND_INIT, 0, 0, 0)
    yy, xx = np.mgrid[0:H, 0:W]
    prev_pick = {0: None, 1: None}
    prev_plane = {0: None, 1: None}
    launch = 1
    for it in range(iters):
        for kind, parity in ((pm.KIND_BLACK, 0), (pm.KIND_RED, 1)):
            planes, costs = o.get()
            pk = picks(costs)
            flat = planes.reshape(-1, 4).view(np.uint32)
            cand_plane = flat[np.clip(pk, 0, None)]                       # [8][H][W][4]
            upd = ((xx + yy) & 1) == parity
            msg = ""
            if prev_pick[parity] is not None:
                same_q = (pk == prev_pick[parity]) & (pk >= 0)
                same_plane = (cand_plane == prev_plane[parity]).all(-1)
                hit = same_q & same_plane
                msg = (f"same neighbour {float(same_q[:, upd].mean()):.3f}, and its plane unchanged {float(hit[:, upd].mean()):.3f}"
                       f"  (pixels with all 8 slots hit {float(hit[:, upd].all(0).mean()):.3f})")
            prev_pick[parity], prev_plane[parity] = pk, cand_plane
            o.step(prm, 12345, kind, it, 0, launch)
            after, _ = o.get()
            changed = (after.view(np.uint32) != planes.view(np.uint32)).any(-1)
            print(f"iter {it} {'black' if parity == 0 else 'red  '}: planes changed by this launch {float(changed[upd].mean()):.3f}   cache: {msg}")
            launch += 1


if __name__ == "__main__":
    main()
