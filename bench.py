#!/usr/bin/env python3
"""bench.py -- Mpix/s of estimated depth+normal for the MP-MVS PatchMatch hot path.

One "step" = one PatchMatchCUDA::Run() schedule (reference src/PatchMatch.cu:1188-1254)
over one reference-image Problem of BASELINE.json configs[1]: 1 reference + 8
source views, 1600x1200, single scale (max_scale = 0), photometric only, 3
red/black iterations, synthetic seeded scene.  Inputs are resident
in HBM before the timed region; results stay in HBM (the D2H copy of Run() is
timed separately and reported in `d2h_ms`, never in `value`).

Multi-GPU (--gpus N under torch.distributed.run): Problems are independent, one
per rank per step, no data-path collective in this configuration (weak scaling);
RCCL only provides the barrier.  value = N * W * H * steps / max-over-ranks time.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, V, ITERS = 1600, 1200, 8, 3
FLOP_PER_EVAL = 2144.0        # one (hypothesis, view) NCC evaluation = 36 taps
HYP_PER_UPDATE = 14           # 8 propagated + current + 5 refinement
PEAK_VALU_TFLOPS = 157.3      # MI355X_MICROARCH.md: peak FP32 vector
PEAK_HBM_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E peak


def load_scene(pm, w, h, v, quantize):
    """seeded synthetic scene, cached on local disk (rendering 9 x 1600x1200 takes ~30 s of numpy)"""
    cache_dir = os.path.join(os.environ.get("TMPDIR", "/tmp"), "mpmvs_scene_cache")
    path = os.path.join(cache_dir, f"scene_{w}x{h}_v{v}_{pm.synth.SCENE_SEED}_{'u8' if quantize else 'f32'}.npz")
    sc = None
    if os.path.exists(path):
        try:
            z = np.load(path)
            imgs = [z[f"img{i}"] for i in range(v + 1)]
            cams = []
            for i in range(v + 1):
                cam = pm.Camera()
                raw = z[f"cam{i}"].tobytes()
                import ctypes
                ctypes.memmove(ctypes.addressof(cam), raw, len(raw))
                cams.append(cam)
            return cams, imgs, z["gt0"]
        except Exception:
            sc = None
    sc = pm.synth.make_problem_scene(w, h, n_src=v, quantize=quantize)
    cams, imgs = sc.problem(0, list(range(1, v + 1)))
    try:
        os.makedirs(cache_dir, exist_ok=True)
        import ctypes
        kw = {f"img{i}": imgs[i] for i in range(v + 1)}
        kw.update({f"cam{i}": np.frombuffer(ctypes.string_at(ctypes.addressof(cams[i]), ctypes.sizeof(cams[i])), np.uint8) for i in range(v + 1)})
        kw["gt0"] = sc.views[0].gt_depth
        tmp = path + f".{os.getpid()}.tmp.npz"
        np.savez(tmp, **kw)
        os.replace(tmp, path)
    except Exception:
        pass
    return cams, imgs, sc.views[0].gt_depth


def measured_traffic_bytes():
    """HBM bytes per k_update launch (FETCH_SIZE + WRITE_SIZE, KiB at the L2's memory
    side) from the committed rocprofv3 PMC passes of this same command
    (profiles/, collected with tools/profile_gpu.sh: PMC cannot be read in-process)"""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    if not os.path.isdir(pdir):
        return None
    for name in sorted(os.listdir(pdir)):
        if not name.endswith(".txt") or "pmc_summary" not in name:
            continue
        vals, sect, kern = {}, None, None
        for line in open(os.path.join(pdir, name)):
            t = line.strip()
            if t.startswith("== pmc_"):
                sect = t
            elif t.startswith("k_"):
                kern = t.split()[0]
            elif kern == "k_update" and (t.startswith("FETCH_SIZE") or t.startswith("WRITE_SIZE")):
                vals[t.split()[0]] = float(t.split("avg=")[1])
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            best = (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0   # last file in name order = latest round
    return best


def cpu_baseline(pm, ctx, cams, imgs, prm, seed):
    """the oracle (our CPU port; the reference has no CPU path) on the host cores: the SAME workload on the same
    inputs (one full 1600x1200 Problem, ~13 s on 16 threads), and -- since both are then at hand -- a bit-for-bit comparison
    of its result with the HIP path's for the same seed"""
    from oracle import binding as ob
    # the GPU box gives one GPU a share of 16 host cores
    ncore = min(16, len(os.sched_getaffinity(0)))
    ob.set_num_threads(ncore)
    o = ob.create()
    o.set_views(cams, imgs)
    t0 = time.perf_counter()
    o.run(prm, seed)
    dt = time.perf_counter() - t0
    op, oc = o.get()
    ctx.run(prm, seed)             # untimed
    gp, gc = ctx.get()
    return {"value": round(W * H / dt / 1e6, 5), "unit": "Mpix/s", "cores": ob.num_threads(), "kind": "port",
            "sample": f"the whole workload: one {W}x{H} Problem, {V} src views, same inputs, seed and Run() schedule, OpenMP oracle, {dt:.1f} s",
            "hip_result_bit_identical": bool(np.array_equal(op, gp) and np.array_equal(oc, gc))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only to rehearse the multi-rank path on one GPU)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--float-images", action="store_true",
                    help="keep the rendered images as non-integer fp32 (rescaled-image case) instead of 8-bit camera-like images")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    dist = None
    dev_index = 0 if args.share_device else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")

    pm = importlib.import_module("mp-mvs_amd")
    engine = importlib.import_module("mp-mvs_amd.engine")

    quantize = not args.float_images
    cams, imgs, gt = load_scene(pm, W, H, V, quantize)
    dmin, dmax = pm.synth.kernel_depth_range(cams[0])
    prm = pm.PatchMatchParams(num_images=V + 1, depth_min=float(dmin), depth_max=float(dmax), max_scale=0, max_iterations=ITERS)
    ctx = engine.create(dev_index)
    ctx.set_views(cams, imgs)   # inputs resident in HBM from here on
    ctx.set_profiling(True)
    seed = 12345 + rank

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        ctx.run(prm, seed + 1000 * i)
    barrier()
    upd_ms, upd_n, all_ms = 0.0, 0, 0.0
    t0 = time.perf_counter()
    for i in range(args.steps):
        ctx.run(prm, seed + i)
        ms, cnt = ctx.kernel_times()
        upd_ms += ms[pm.KIND_BLACK] + ms[pm.KIND_RED]
        upd_n += cnt[pm.KIND_BLACK] + cnt[pm.KIND_RED]
        all_ms += sum(ms)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # sanity of the last result + the (untimed) D2H leg
    t1 = time.perf_counter()
    planes, costs = ctx.get()
    d2h_ms = (time.perf_counter() - t1) * 1e3
    rel = np.abs(planes[..., 3] - gt) / gt
    within = float((rel < 0.01).mean())

    if rank == 0:
        mpix = world * W * H * args.steps / dt / 1e6
        upd_avg_ms = upd_ms / max(upd_n, 1)
        flops_per_launch = (W * H / 2) * HYP_PER_UPDATE * V * FLOP_PER_EVAL
        hbm_bytes_per_launch = W * H * (4 * (V + 1) + 36)      # SURVEY.md 8d COMPULSORY_HBM_BYTES / L
        tflops = flops_per_launch / (upd_avg_ms * 1e-3) / 1e12
        gbps = hbm_bytes_per_launch / (upd_avg_ms * 1e-3) / 1e9
        out = {
            "metric": "Mpix/s depth+normal (fixed iters, 1600x1200, 8 src views)",
            "value": round(mpix, 3),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic, seeded height-field scene; images " + ("rounded to 8 bits like the reference's imread input" if quantize else "non-integer fp32") + f"; resident texture format {ctx.texture_format()}",
            "config": {"workload": "configs[1]: 1 ref + 8 src views, 1600x1200, single-scale, photometric only, 3 red/black iterations, one Problem per GPU per step",
                       "width": W, "height": H, "src_views": V, "max_scale": 0, "iterations": ITERS},
            "roofline": {
                "kernel": "k_update<photometric> (BlackPixelUpdate/RedPixelUpdate)",
                "bound": "valu_fp32",
                "achieved": round(tflops, 3),
                "peak": PEAK_VALU_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(tflops / PEAK_VALU_TFLOPS, 4),
                "traffic": measured_traffic_bytes(),
                "avg_launch_ms": round(upd_avg_ms, 4),
                "launches_timed": upd_n,
                "algorithmic_flop_per_launch": flops_per_launch,
                "hbm": {"achieved": round(gbps, 2), "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(gbps / PEAK_HBM_GBPS, 5),
                        "algorithmic_bytes_per_launch": hbm_bytes_per_launch},
            },
            "kernel_ms_per_step": round(all_ms / args.steps, 3),
            "d2h_ms": round(d2h_ms, 2),
            "within_1pct_of_gt": round(within, 4),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pm, ctx, cams, imgs, prm, seed)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
