#!/usr/bin/env python3
"""How many refinement evaluations of k_update can no longer change the result?  (sizing of the early-rejection step)

Runs the cfg-1 schedule on the CPU oracle with its statistics hook (oracle/pm_oracle.cpp: orc_set_refinement_stats) and
aggregates, per update launch, the (view, candidate) steps of the refinement at WAVE level for the kernel's wave shape
(16 x 8 pixel patch of one colour = 8 lanes x 8 rows):

  now        steps executed today: every candidate for every view that has weight > 0 in some lane
  uniform    candidate loop stays wave-uniform, a step is skipped when no lane is still live for it
  compacted  every lane walks its own list of live candidates of the view; trips = longest list in the wave
  cross-lane the live (pixel, candidate) items of a view are dealt to the 64 lanes; trips = ceil(items / 64)

CPU only (test infrastructure); usage: python tests/analysis/prune_stats.py [W H]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    prior_mode = "--prior" in sys.argv
    argv = [a for a in sys.argv if a != "--prior"]
    W, H = (int(argv[1]), int(argv[2])) if len(argv) > 2 else (400, 304)
    V = 8
    pm = importlib.import_module("mp-mvs_amd")
    from oracle import binding as ob
    lib = ob.lib()[0]
    lib.orc_set_refinement_stats.argtypes = [C.c_void_p, C.c_void_p]
    sc = pm.synth.make_problem_scene(W, H, n_src=V, quantize=True)
    cams, imgs = sc.problem(0, list(range(1, V + 1)))
    dmin, dmax = pm.synth.kernel_depth_range(cams[0])
    prm = pm.PatchMatchParams(num_images=V + 1, depth_min=float(dmin), depth_max=float(dmax), max_scale=0)
    o = ob.create()
    o.set_views(cams, imgs)
    lib.orc_set_propagation_stats.argtypes = [C.c_void_p]
    bad3 = np.zeros((H, W, 32), np.uint8)
    death = np.full((H, W, 5), -128, np.int8)
    wmask = np.zeros((H, W), np.uint32)
    if prior_mode:
        # the planar-prior Run() of the shipped schedule: a photometric Run(), the prior triangulated from it, then the launches
        # of the prior Run() with the hook on
        hostlib = importlib.import_module("mp-mvs_amd.hostlib")
        o.run(prm, 12345)
        planes, costs = o.get()
        prior, mask, ntri = hostlib.build_prior(cams[0], planes, costs, None, False, prm.depth_min, prm.depth_max)
        print(f"planar prior: {ntri} triangles, {float((mask > 0).mean()):.3f} of the pixels masked")
        o.set_prior(prior, mask)
        prm.planar_prior = True
        seed0 = 777
    else:
        seed0 = 12345
    o.step(prm, seed0, pm.KIND_INIT, 0, 0, 0)
    launch = 1
    yy, xx = np.mgrid[0:H, 0:W]
    print(f"{W}x{H}, {V} views; per launch: executed refinement steps per wave (of {5 * V} nominal)")
    for it in range(3):
        for kind, parity in ((pm.KIND_BLACK, 0), (pm.KIND_RED, 1)):
            death[...] = -128
            lib.orc_set_refinement_stats(death.ctypes.data, wmask.ctypes.data)
            lib.orc_set_propagation_stats(bad3.ctypes.data)
            o.step(prm, seed0, kind, it, 0, launch)
            lib.orc_set_refinement_stats(None, None)
            lib.orc_set_propagation_stats(None)
            launch += 1
            upd = (death[..., 0] != -128)
            assert (upd == (((xx + yy) & 1) == parity))[: 2 * 16 * (((H // 2) + 15) // 16)].all()
            now = uni = comp = xl = 0.0
            a_now = a_xl = a_lane = 0.0
            nw = 0
            lane_live = []
            for y0 in range(0, H - 7, 8):
                for x0 in range(0, W - 15, 16):
                    m = upd[y0:y0 + 8, x0:x0 + 16]
                    d = death[y0:y0 + 8, x0:x0 + 16][m].astype(np.int32)       # [64][5]
                    wm = wmask[y0:y0 + 8, x0:x0 + 16][m]                        # [64]
                    if len(wm) != 64:
                        continue
                    nw += 1
                    b3 = bad3[y0:y0 + 8, x0:x0 + 16][m].astype(np.int32)        # [64][32]
                    fl = b3[:, 31]
                    for v in range(V):
                        a_now += 8
                        alive = np.stack([((fl >> j) & 1).astype(bool) & (b3[:, v] >= j) for j in range(8)], 1)   # [64][8]
                        a_xl += -(-int(alive.sum()) // 64)
                        a_lane += alive.sum() / 64.0
                    for v in range(V):
                        has_w = ((wm >> v) & 1).astype(bool)                     # lanes with weight on view v
                        if not has_w.any():
                            continue
                        now += 5
                        live = has_w[:, None] & (d >= v)                         # step (v, ci) still needed by the lane
                        uni += live.any(0).sum()
                        comp += live.sum(1).max()
                        xl += -(-int(live.sum()) // 64)
                        lane_live.append(live.sum() / 64.0)
            print(f"  iter {it} {'black' if parity == 0 else 'red  '}: now {now / nw:5.2f}  uniform-skip {uni / nw:5.2f}  lane-compacted {comp / nw:5.2f}  cross-lane-compacted {xl / nw:5.2f}"
                  f"   (mean live steps per LANE {np.sum(lane_live) / nw:5.2f})   dead-from-start {float((death[upd] == -1).mean()):.3f}"
                  f"   | propagation: now {a_now / nw:5.2f}  cross-lane-compacted {a_xl / nw:5.2f}  (mean per lane {a_lane / nw:5.2f})")


if __name__ == "__main__":
    main()
