#!/usr/bin/env python3
"""Slot occupancy of the update launches over time, from per-wave start / end stamps.

Needs a measurement build of the HIP library with -DPM_DBG_WAVETIME (pm_kernels.hpp, WaveTimer: every wave of k_update records
s_memrealtime at its start and end, HW_ID and XCC_ID):
    make -C mp-mvs_amd/csrc EXTRA=-DPM_DBG_WAVETIME ... -o build/libmpmvs_hip_wt.so      (tools/build_variants.sh wt)
    MPMVS_HIP_LIB=$PWD/build/libmpmvs_hip_wt.so python tools/wave_timeline.py [--out profiles/r05_wave_timeline.txt]

For every update launch of one cfg-1 Run() (1600x1200, 8 views, photometric, 3 iterations) it prints
  * the span of the launch (first wave start -> last wave end) and the share of the wave slots (CUs x 8) that held a wave,
  * where the empty slot-time went:
      ramp   = before every CU had its first block,
      tail   = after the last block of the launch had started (no work left to hand out),
      held   = slots of a block whose wave had ended while a sibling wave still ran (the block keeps its LDS until its slowest wave
               retires, so no new block can start there),
      gaps   = the rest: slots idle between the end of one block and the start of the next on the same CU,
  * the spread of wave durations, inside a block and over the launch,
  * an ASCII plot of resident waves over time (50 bins).
"""
import argparse
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TICK_NS = 10.0   # s_memrealtime: 100 MHz


def analyse(rec, waves_per_block, out):
    """rec: [n_waves][4] u64 (start, end, hw | xcc << 32, wave | block << 32) of ONE launch (rows of waves that never ran are zero)"""
    ran = rec[:, 0] != 0
    rec = rec[ran]
    start, end = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64)
    hw, xcc = (rec[:, 2] & 0xffffffff).astype(np.int64), (rec[:, 2] >> 32).astype(np.int64) & 0xf
    block = (rec[:, 3] >> 32).astype(np.int64)
    waited = ((rec[:, 3] >> 8) & 0xffffff).astype(np.float64)   # ticks between kernel entry and the start of the update itself
    t0, t1 = start.min(), end.max()
    span = float(t1 - t0)
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    n_cu = len(np.unique(cu_key))
    slots = n_cu * 8
    dur = (end - start).astype(np.float64)
    busy = dur.sum()
    occ = busy / (span * slots)
    # per block
    order = np.argsort(block, kind="stable")
    b_sorted = block[order]
    uniq, first = np.unique(b_sorted, return_index=True)
    b_start = np.minimum.reduceat(start[order], first)
    b_end = np.maximum.reduceat(end[order], first)
    b_cnt = np.diff(np.append(first, len(order)))
    b_of = np.searchsorted(uniq, block)
    held = (b_end[b_of] - end).astype(np.float64).sum()
    last_block_start = b_start.max()
    # resident waves over time -> ramp and tail losses
    ev_t = np.concatenate([start, end])
    ev_d = np.concatenate([np.ones_like(start), -np.ones_like(end)])
    o = np.argsort(ev_t, kind="stable")
    ev_t, ev_d = ev_t[o], ev_d[o]
    active = np.cumsum(ev_d)
    dt = np.diff(np.append(ev_t, t1)).astype(np.float64)
    empty = (slots - active) * dt
    # ramp: until every CU has had its first wave
    first_on_cu = {}
    for k, s in zip(cu_key, start):
        if k not in first_on_cu or s < first_on_cu[k]:
            first_on_cu[k] = s
    ramp_end = max(first_on_cu.values())
    ramp = empty[ev_t < ramp_end].sum()
    tail = empty[ev_t >= last_block_start].sum()
    # the part of `held` that falls into the tail is counted there already
    held_in_tail = np.clip(b_end[b_of] - np.maximum(end, last_block_start), 0, None).astype(np.float64).sum()
    held_mid = held - held_in_tail
    total_empty = span * slots - busy
    gaps = total_empty - ramp - tail - held_mid
    w = lambda v: f"{100.0 * v / (span * slots):5.2f} %"
    out.append(f"  span {span * TICK_NS / 1e6:.3f} ms, {len(rec)} waves in {len(uniq)} blocks ({waves_per_block} waves per block) on {n_cu} CUs = {slots} wave slots; "
               f"slot occupancy {100 * occ:.2f} %")
    out.append(f"  empty slot-time {w(total_empty)} of the launch = ramp {w(ramp)} + tail (after the last block started) {w(tail)} + "
               f"held by a sibling's LDS {w(held_mid)} + gaps between blocks {w(gaps)}")
    out.append(f"  tail lasts {(t1 - last_block_start) * TICK_NS / 1e3:.0f} us ({100.0 * (t1 - last_block_start) / span:.1f} % of the span); "
               f"ramp {(ramp_end - t0) * TICK_NS / 1e3:.1f} us")
    out.append(f"  wave duration: mean {dur.mean() * TICK_NS / 1e3:.1f} us, p5 {np.percentile(dur, 5) * TICK_NS / 1e3:.1f}, p50 {np.percentile(dur, 50) * TICK_NS / 1e3:.1f}, "
               f"p95 {np.percentile(dur, 95) * TICK_NS / 1e3:.1f}, max {dur.max() * TICK_NS / 1e3:.1f}")
    out.append(f"  kernel entry -> start of the update (ticket, wait for the neighbouring blocks of the pass before, barriers): mean {waited.mean() * TICK_NS / 1e3:.2f} us, "
               f"p50 {np.percentile(waited, 50) * TICK_NS / 1e3:.2f}, p95 {np.percentile(waited, 95) * TICK_NS / 1e3:.2f}, max {waited.max() * TICK_NS / 1e3:.1f}; "
               f"{100.0 * waited.sum() / (span * slots):.2f} % of the launch's slot-time")
    full = b_cnt == waves_per_block
    if waves_per_block > 1 and full.any():
        # slowest wave of a block against the mean of its waves
        sums = np.add.reduceat(dur[order], first)
        maxs = np.maximum.reduceat(dur[order], first)
        rel = (maxs[full] / (sums[full] / waves_per_block) - 1.0) * 100.0
        out.append(f"  inside a block: slowest wave over the block's mean wave: mean +{rel.mean():.1f} %, p50 +{np.percentile(rel, 50):.1f} %, p95 +{np.percentile(rel, 95):.1f} %")
    # block start-to-next-start gap on a CU: how long a freed half-CU waits for its next block
    bk_cu = cu_key[order][first]
    gaps_us = []
    for k in np.unique(bk_cu):
        m = bk_cu == k
        s, e = np.sort(b_start[m]), np.sort(b_end[m])
        # with two blocks resident per CU the i-th end is followed by the (i + 2)-th start
        res = max(1, 8 // waves_per_block)
        for i in range(len(e) - res):
            gaps_us.append((s[i + res] - e[i]) * TICK_NS / 1e3)
    if gaps_us:
        g = np.array(gaps_us)
        out.append(f"  end of a block -> start of the block that takes its place on the CU: p50 {np.percentile(g, 50):.1f} us, p95 {np.percentile(g, 95):.1f} us")
    # ASCII plot
    bins = 50
    edges = np.linspace(t0, t1, bins + 1)
    res = np.zeros(bins)
    for i in range(bins):
        lo, hi = edges[i], edges[i + 1]
        ov = np.clip(np.minimum(end, hi) - np.maximum(start, lo), 0, None).sum()
        res[i] = ov / (hi - lo)
    out.append("  resident waves over the launch (each column = 2 % of the span; rows = share of the slots):")
    for level in (1.0, 0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3, 0.2, 0.1):
        out.append(f"   {int(level * 100):3d} % |" + "".join("#" if r / slots >= level - 0.05 else " " for r in res) + "|")
    return occ, span * TICK_NS / 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--size", default="1600x1200")
    ap.add_argument("--threads", type=int, default=64, help="threads per update block of the library in use (PM_UPD_THREADS)")
    args = ap.parse_args()
    sys.argv = [sys.argv[0]]
    import bench
    pm = importlib.import_module("mp-mvs_amd")
    engine = importlib.import_module("mp-mvs_amd.engine")
    w, h = (int(v) for v in args.size.split("x"))
    cams, imgs, gts = bench.load_views(pm, w, h, bench.problem_centers(pm, 8), "p8")
    imgs = [np.rint(im).astype(np.float32) for im in imgs]
    dmin, dmax = pm.synth.kernel_depth_range(cams[0])
    prm = pm.PatchMatchParams(num_images=9, depth_min=float(dmin), depth_max=float(dmax), max_scale=0, max_iterations=3)
    lib, _ = engine.load()
    fn = lib.mpmvs_dbg_wavetime
    fn.restype = C.c_long
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    ctx = engine.create(0)
    ctx.set_views(cams, imgs)
    ctx.set_profiling(True)
    ctx.run(prm, 1)      # warm
    ctx.run(prm, 12345)
    ms, cnt = ctx.kernel_times()
    n = fn(ctx._ctx, None, 0)
    buf = np.zeros(n // 8, np.uint64)
    assert fn(ctx._ctx, buf.ctypes.data, n) == n
    wpb = args.threads // 64
    bw, bh = 16, 8 * wpb
    nblocks = ((w + bw - 1) // bw) * ((min(h, 2 * 16 * (((h // 2) + 15) // 16)) + bh - 1) // bh)
    waves = nblocks * wpb
    out = [f"# wave timeline of the update launches of one cfg-1 Run() ({w}x{h}, 8 views, photometric, 3 iterations); library {os.environ.get('MPMVS_HIP_LIB', 'default')}",
           f"# HIP-event averages of this run: k_update {(ms[1] + ms[2]) / max(cnt[1] + cnt[2], 1):.4f} ms per launch (the stamps add two scalar memory operations per wave)"]
    occs, recs = [], []
    for launch in range(1, 7):
        rec = buf[(launch & 15) * waves * 4:((launch & 15) + 1) * waves * 4].reshape(waves, 4)
        recs.append(rec[rec[:, 0] != 0])
        out.append(f"launch {launch} ({'black' if launch % 2 else 'red'}, iteration {(launch - 1) // 2}):")
        occ, span = analyse(rec, wpb, out)
        occs.append((occ, span))
    out.append(f"# mean slot occupancy over the six launches {100 * np.mean([o for o, _ in occs]):.2f} %, mean span {np.mean([s for _, s in occs]):.3f} ms")
    # the six passes as one piece of work (what counts when they are chained into one launch: MPMVS_CHAIN, k_update)
    allr = np.concatenate(recs)
    st, en = allr[:, 0].astype(np.int64), allr[:, 1].astype(np.int64)
    span = float(en.max() - st.min())
    busy = float((en - st).sum())
    firsts = [int(r[:, 0].astype(np.int64).min()) for r in recs]
    lasts = [int(r[:, 1].astype(np.int64).max()) for r in recs]
    overlap = [max(0, lasts[i] - firsts[i + 1]) * TICK_NS / 1e3 for i in range(5)]
    out.append(f"# all six passes together: first wave start -> last wave end {span * TICK_NS / 1e6:.3f} ms = {span * TICK_NS / 6e6:.3f} ms per pass, slot occupancy {100 * busy / (span * 2048):.2f} % of 2048 slots; "
               f"pass p + 1 starts before pass p ends by {', '.join(f'{o:.0f}' for o in overlap)} us")
    text = "\n".join(out)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
