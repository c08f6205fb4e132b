"""Problem-level multi-GPU schedule.

The reference processes its Problems (one per reference image) strictly one
after the other on device 0 and exchanges depth maps through depths.dmb files
(reference src/main.cpp:20-41, src/PatchMatch.cpp:620-633 -> :941-948).  Here:

  * Problems are dealt round-robin to the ranks (one process per GPU); each
    owned Problem keeps its context -- images, planes, costs -- resident in HBM
    across all passes (about 0.3 GB per 1600x1200 Problem);
  * within a pass Problems are independent, so there is no data-path collective;
  * between passes every Problem's new depth map must be visible to the
    Problems that list it as a source: ONE all-gather of H*W fp32 depth maps per
    pass (RCCL over xGMI when the tensors are on GPUs), which is also the barrier;
  * semantics are Jacobi at every world size -- pass k+1 reads only pass-k depth
    maps -- so 1/2/4/8-GPU results are bit-identical (the reference's in-place
    file order is Gauss-Seidel; deviation recorded in DESIGN.md section 7).
"""
import ctypes

import numpy as np

from . import hostlib
from ._abi import PatchMatchParams
from .synth import kernel_depth_range

PRIOR_SEED_OFFSET = 0x9E3779B97F4A7C15


def owned_problems(n_problems, rank, world):
    return [i for i in range(n_problems) if i % world == rank]


class SceneScheduler:
    def __init__(self, cams, images, sources, make_handle, rank=0, world=1, dist=None, device_tensors=False, max_scale=2, workers=1, iterations=None):
        """cams/images: all views of the scene; sources[i]: source-view ids of Problem i;
        make_handle(): a fresh PatchMatch handle (engine.create(local_rank) in production);
        workers: host threads per rank.  Within a pass the owned Problems are independent, so with
        workers > 1 the host half of one Problem (planar-prior construction, transfers) overlaps the
        kernels of others -- every context has its own stream and the C calls release the GIL.
        Results do not depend on the interleaving."""
        self.cams, self.images, self.sources = cams, images, sources
        self.n = len(sources)
        self.rank, self.world, self.dist = rank, world, dist
        self.max_scale = max_scale
        self.iterations = iterations   # None: the reference's counts (3, geometric 2); tests shorten the schedule with it
        self.workers = max(1, int(workers))
        self.owned = owned_problems(self.n, rank, world)
        self.per_rank = (self.n + world - 1) // world
        self.H, self.W = images[0].shape
        for im in images:
            assert im.shape == (self.H, self.W), "the all-gather assumes equally sized depth maps"
        self.device_tensors = device_tensors
        self.fetch_results = True   # device mode only: False leaves every pass's maps in HBM (bench.py --workload cfg4)
        self.force_collective = False   # tests: issue the all-gather also with a single rank (RCCL's one-rank path on one GPU)
        self.handles = {}
        for i in self.owned:
            h = make_handle()
            ids = [i] + list(sources[i])
            h.set_views([cams[j] for j in ids], [images[j] for j in ids])
            self.handles[i] = h
        self.results = {}   # problem -> (planes HxWx4 [world normal, depth], costs, geom or None)
        self.all_depths = None

    # -- one Run() (+ optional planar-prior re-run) of one Problem -----------
    def _params(self, i, geom, planar):
        cam = self.cams[i]
        dmin, dmax = kernel_depth_range(cam)
        p = PatchMatchParams(num_images=1 + len(self.sources[i]), depth_min=float(dmin), depth_max=float(dmax), max_scale=self.max_scale)
        p.geom_consistency = geom
        p.max_iterations = self.iterations or (2 if geom else 3)   # reference src/PatchMatch.cpp:655-665
        p.geomPlanarPrior = bool(geom and planar)
        return p

    def _process(self, i, geom, planar, seed):
        h = self.handles[i]
        p = self._params(i, geom, planar)
        if geom:
            self._attach_source_depths(i, h)
            if not self.device_tensors:                 # device mode: the context still holds its own previous result
                planes, costs, _ = self.results[i]
                h.set_state(planes, costs)
        h.run(p, seed)
        if planar:                                      # reference src/PatchMatch.cpp:532-607
            geom_pp = bool(p.geomPlanarPrior)
            if hasattr(h, "prior_vertices"):
                # HIP context: vertex selection, rasterisation, plane fit and range test run on the device; only the vertex
                # and triangle lists cross PCIe, the Delaunay triangulation is the host's (bit-identical to the host path)
                verts = h.prior_vertices(geom_pp)
                if len(verts) == 0:
                    raise RuntimeError("No Point to Triangulate!")
                h.prior_from_triangles(p, hostlib.delaunay(self.W, self.H, verts))
            else:
                planes, costs, g = h.get(geom=True)
                prior, mask, ntri = hostlib.build_prior(self.cams[i], planes, costs, g if geom_pp else None, geom_pp, p.depth_min, p.depth_max)
                if ntri <= 0:
                    raise RuntimeError("No Point to Triangulate!")
                h.set_prior(prior, mask)
            p.planar_prior = True
            p.geom_consistency = False
            p.max_iterations = self.iterations or 3
            h.run(p, (seed + PRIOR_SEED_OFFSET) & 0xFFFFFFFFFFFFFFFF)
        if self.device_tensors and not self.fetch_results:
            return None                                 # maps stay in HBM; fetch() brings them to the host on demand
        planes, costs, g = h.get(geom=True)
        return planes, costs, g

    # -- exchange --------------------------------------------------------------
    def _slot(self, j):
        """position of Problem j's depth map in the gathered buffer: rank r's k-th slot holds Problem r + k * world"""
        return (j % self.world) * self.per_rank + j // self.world

    def _attach_source_depths(self, i, h):
        srcs = self.sources[i]
        if self.device_tensors:
            # the gathered buffer is read in place, slot by slot (no re-ordering copy); the call copies the maps into the
            # context on the context's stream and returns when that copy is complete
            base = self.all_depths.data_ptr()
            stride = self.H * self.W * 4
            h.set_src_depths_device([base + self._slot(j) * stride for j in srcs], [self.W] * len(srcs), [self.H] * len(srcs))
        else:
            h.set_src_depths([self.all_depths[self._slot(j)] for j in srcs])

    def depth_maps(self):
        """the depth maps of the last pass in Problem order, on the host: [n][H][W]"""
        full = self.all_depths.cpu().numpy() if self.device_tensors else self.all_depths
        return np.ascontiguousarray(full[[self._slot(j) for j in range(self.n)]])

    def _exchange(self):
        """all-gather of the depth maps of this pass; doubles as the pass barrier.

        Stream ordering in device mode.  Three kinds of streams touch the buffers: torch's current stream (allocation, the
        padding fill), the contexts' own non-blocking streams (export kernel; later the device-to-device copies of
        set_src_depths_device) and RCCL's (ordered against torch's current stream by torch).  Non-blocking streams do not
        order against torch's, so every hand-over is made explicit:
          1. the padding fill on torch's stream is complete before the first export may write next to it;
          2. each export is complete when its call returns (mpmvs_export_depth_device synchronises the context's stream),
             so the collective, enqueued afterwards, sees every map;
          3. the collective is complete (torch.cuda.synchronize) before this function returns, i.e. before any context of
             the next pass copies its source maps out of the gathered buffer.
        The gathered buffer is used as it arrives (rank-major slots, _slot()): no re-ordering kernel is left in flight."""
        if self.device_tensors:
            import torch
            mine = torch.empty((self.per_rank, self.H, self.W), dtype=torch.float32, device="cuda")
            if len(self.owned) < self.per_rank:
                mine[len(self.owned):].zero_()            # padding slots of an uneven shard
            torch.cuda.current_stream().synchronize()     # (1)
            for k, i in enumerate(self.owned):
                self.handles[i].export_depth_device(mine[k].data_ptr())   # (2)
            if self.world > 1 or (self.force_collective and self.dist is not None):
                import time
                full = torch.empty((self.world * self.per_rank, self.H, self.W), dtype=torch.float32, device="cuda")
                timed = isinstance(getattr(self, "timing", None), list)
                if timed:
                    # a timed run separates the wait for the slowest rank (a barrier) from the transfer itself, so that the
                    # collective's time is a bandwidth figure and not this rank's lead over the others
                    torch.cuda.synchronize()
                    ta = time.perf_counter()
                    self.dist.barrier()
                    tb = time.perf_counter()
                self.dist.all_gather_into_tensor(full, mine)
                if timed:
                    torch.cuda.synchronize()
                    tc = time.perf_counter()
                    recv = (self.world - 1) * mine.numel() * 4
                    self.last_exchange = {"wait_for_slowest_rank_ms": round((tb - ta) * 1e3, 3), "all_gather_ms": round((tc - tb) * 1e3, 3),
                                          "bytes_sent_per_rank": mine.numel() * 4, "bytes_received_per_rank": recv,
                                          "received_GB_per_s": round(recv / max(tc - tb, 1e-9) / 1e9, 2)}
            else:
                full = mine
            torch.cuda.synchronize()                      # (3)
            self.all_depths = full
        else:
            mine = np.zeros((self.per_rank, self.H, self.W), np.float32)
            for k, i in enumerate(self.owned):
                mine[k] = self.results[i][0][..., 3]
            if self.world > 1:
                import time
                import torch
                t = torch.from_numpy(mine)
                outs = [torch.empty_like(t) for _ in range(self.world)]
                timed = isinstance(getattr(self, "timing", None), list)
                if timed:
                    ta = time.perf_counter()
                    self.dist.barrier()
                    tb = time.perf_counter()
                self.dist.all_gather(outs, t)
                if timed:
                    tc = time.perf_counter()
                    recv = (self.world - 1) * mine.size * 4
                    self.last_exchange = {"wait_for_slowest_rank_ms": round((tb - ta) * 1e3, 3), "all_gather_ms": round((tc - tb) * 1e3, 3),
                                          "bytes_sent_per_rank": mine.size * 4, "bytes_received_per_rank": recv,
                                          "received_GB_per_s": round(recv / max(tc - tb, 1e-9) / 1e9, 2), "staged_through_host": True}
                full = np.concatenate([o.numpy() for o in outs], 0)
            else:
                full = mine
            self.all_depths = full

    # -- the pass schedule of reference src/main.cpp:20-41 ----------------------
    def run(self, geom_iterations=2, planar_prior=True, geom_planar_prior=True, seed=12345):
        planar0 = (not geom_planar_prior) and planar_prior
        self._timed_pass("photometric", lambda i: self._process(i, False, planar0, seed + i))
        for g in range(geom_iterations):
            planar = bool(geom_planar_prior and g != geom_iterations - 1)
            # Jacobi: nothing of pass g is visible during pass g (results swapped afterwards)
            self._timed_pass("geometric" + (" + planar prior" if planar else ""),
                             lambda i, g=g, planar=planar: self._process(i, True, planar, seed + 100003 * (g + 1) + i))
        return self.results

    def _timed_pass(self, name, fn):
        """one pass over the owned Problems, then the exchange (= barrier); if `self.timing` is a list, the wall times of
        both halves on this rank are appended to it"""
        import time
        t0 = time.perf_counter()
        self.results = self._pass(fn)
        t1 = time.perf_counter()
        self.last_exchange = None
        self._exchange()
        t2 = time.perf_counter()
        if isinstance(getattr(self, "timing", None), list):
            rec = {"pass": name, "compute_ms": round((t1 - t0) * 1e3, 2), "exchange_ms": round((t2 - t1) * 1e3, 2)}
            if self.last_exchange:
                rec["collective"] = self.last_exchange
            self.timing.append(rec)

    def fetch(self):
        """results of the last pass on the host: problem -> (planes, costs, geom costs)"""
        return {i: self.handles[i].get(geom=True) for i in self.owned}

    def _pass(self, fn):
        if self.workers == 1 or len(self.owned) <= 1:
            return {i: fn(i) for i in self.owned}
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=self.workers) as pool:
            return dict(zip(self.owned, pool.map(fn, self.owned)))
