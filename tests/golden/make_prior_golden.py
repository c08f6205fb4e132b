#!/usr/bin/env python3
"""Independent golden vectors for the host planar-prior construction.

Generates tests/golden/prior_golden_v1.npz WITHOUT importing anything of this repository (neither mp-mvs_amd/ nor
oracle/): only numpy and scipy, so the fixtures pin the product against a second, independent implementation:

  vertices   brute-force numpy/python restatement of GetTriangulateVertices (reference src/PatchMatch.cpp:782-853), both
             modes, fp32 arithmetic step by step (incl. the `cost_sum / (r_bound * c_bound) * 0.85` quirk)
  delaunay   scipy.spatial.Delaunay (Qhull) on GENERAL-POSITION integer points (no four cocircular points among adjacent
             triangles, checked in exact integer arithmetic), where the Delaunay triangulation is unique: triangle SETS must
             be equal (cv::Subdiv2D, the reference's triangulator, src/PatchMatch.cpp:757-780, returns that same unique set)
  raster     python-loop restatement of the barycentric stepping (src/PatchMatch.cpp:554-570) with the reference's mixed
             float/double arithmetic, sequential "last triangle wins"
  planes     null vector of the 3x4 matrix [X 1] by numpy's SVD in float64 (src/PatchMatch.cpp:723-755 uses cv::SVD::solveZ),
             normalised so that d > 0; the product's closed form must agree to 1e-5
  mask       depth-range test of the prior depth (src/PatchMatch.cpp:583-595); pixels whose prior depth is within 1e-4
             (relative) of a bound are flagged ambiguous (the two plane solvers differ by rounding)

Run from the repository root:  python tests/golden/make_prior_golden.py
"""
import os

import numpy as np
from scipy.spatial import Delaunay

F = np.float32
HERE = os.path.dirname(os.path.abspath(__file__))


# ---------------------------------------------------------------------------------------------------------------
def vertices_ref(costs, geom, geom_planar_prior):
    """GetTriangulateVertices, reference src/PatchMatch.cpp:782-853"""
    h, w = costs.shape
    out = []
    for row in range(0, h, 5):
        for col in range(0, w, 5):
            c_bound, r_bound = min(w, col + 5), min(h, row + 5)
            if not geom_planar_prior:
                min_cost, best = F(2.0), None
                for r in range(row, r_bound):
                    for c in range(col, c_bound):
                        cost = costs[r, c]
                        if cost < F(2.0) and min_cost > cost:
                            best, min_cost = (c, r), cost
                if min_cost < F(0.1):
                    out.append(best)
            else:
                mc = [F(2.0), F(2.0), F(2.0)]
                pts = [(0, 0), (0, 0), (0, 0)]
                cost_sum = F(0.0)
                for r in range(row, r_bound):
                    for c in range(col, c_bound):
                        cost = costs[r, c]
                        cost_sum = F(cost_sum + cost)
                        if cost < F(1.0) and geom[r, c] < F(0.4) and cost < mc[2]:
                            mc[2], pts[2] = cost, (c, r)
                            for i in (1, 0):
                                if mc[i] <= mc[i + 1]:
                                    break
                                mc[i], mc[i + 1] = mc[i + 1], mc[i]
                                pts[i], pts[i + 1] = pts[i + 1], pts[i]
                # float / int -> float, * 0.85 (double) -> double, stored to float
                cost_sum = F(float(F(cost_sum / F(r_bound * c_bound))) * 0.85)
                thresh = max(cost_sum, F(0.2))
                for i in range(3):
                    if mc[i] < thresh:
                        out.append(pts[i])
                    else:
                        break
    return np.array(out, np.int32).reshape(-1, 2)


# ---------------------------------------------------------------------------------------------------------------
def incircle(a, b, c, d):
    """exact sign of the in-circle determinant for integer points (python ints)"""
    ax, ay, bx, by, cx, cy = a[0] - d[0], a[1] - d[1], b[0] - d[0], b[1] - d[1], c[0] - d[0], c[1] - d[1]
    return (ax * ax + ay * ay) * (bx * cy - by * cx) - (bx * bx + by * by) * (ax * cy - ay * cx) + (cx * cx + cy * cy) * (ax * by - ay * bx)


def general_position_points(rng, w, h, n):
    """n distinct integer points in [0,w) x [0,h) whose Delaunay triangulation is unique and non-degenerate"""
    while True:
        flat = rng.choice(w * h, size=n, replace=False)
        pts = np.stack([flat % w, flat // w], 1).astype(np.int64)
        tri = Delaunay(pts.astype(np.float64))
        if len(tri.coplanar):
            continue
        ok = True
        P = [tuple(int(v) for v in p) for p in pts]
        for t, (simp, nbrs) in enumerate(zip(tri.simplices, tri.neighbors)):
            a, b, c = (P[i] for i in simp)
            area2 = (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])
            if area2 == 0:
                ok = False
                break
            for k, nb in enumerate(nbrs):
                if nb < 0:
                    continue
                opp = [i for i in tri.simplices[nb] if i not in simp]
                if len(opp) != 1 or incircle(a, b, c, P[opp[0]]) == 0:
                    ok = False
                    break
            if not ok:
                break
        if not ok:
            continue
        # hull without three collinear consecutive points (a triangulator working from an enclosing
        # super-triangle and Qhull agree on such hulls only up to zero-area slivers)
        hull = tri.convex_hull
        on_hull = set(int(i) for e in hull for i in e)
        coll = False
        hl = sorted(on_hull)
        for i in hl:
            for j in hl:
                for k in hl:
                    if i < j < k:
                        a, b, c = P[i], P[j], P[k]
                        if (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]) == 0:
                            coll = True
        if coll:
            continue
        tris = np.sort(tri.simplices.astype(np.int32), axis=1)
        tris = tris[np.lexsort((tris[:, 2], tris[:, 1], tris[:, 0]))]
        return pts.astype(np.int32), tris


# ---------------------------------------------------------------------------------------------------------------
def raster_ref(w, h, tri_pts):
    """reference src/PatchMatch.cpp:554-570: label = index + 1 of the LAST triangle that steps on the pixel"""
    label = np.zeros((h, w), np.uint32)
    for idx, t in enumerate(tri_pts):
        (x1, y1), (x2, y2), (x3, y3) = [(int(p[0]), int(p[1])) for p in t]
        L01 = F(np.sqrt(float((x1 - x2) ** 2 + (y1 - y2) ** 2)))
        L02 = F(np.sqrt(float((x1 - x3) ** 2 + (y1 - y3) ** 2)))
        L12 = F(np.sqrt(float((x2 - x3) ** 2 + (y2 - y3) ** 2)))
        max_edge = max(L01, L02, L12)
        step = F(1.0 / float(max_edge))
        p = F(0.0)
        while float(p) < 1.0:
            q = F(0.0)
            while float(q) < 1.0 - float(p):
                r = (1.0 - float(p)) - float(q)
                x = int(float(F(F(p * F(x1)) + F(q * F(x2)))) + r * x3)
                y = int(float(F(F(p * F(y1)) + F(q * F(y2)))) + r * y3)
                label[y, x] = idx + 1
                q = F(q + step)
            p = F(p + step)
    return label


def plane_ref(K, tri, depth):
    """GetPriorPlaneParams (reference src/PatchMatch.cpp:723-755) with a float64 SVD null vector"""
    A = np.ones((3, 4), np.float64)
    for i, (x, y) in enumerate(tri):
        d = depth[y, x]
        A[i, 0] = float(F(F(d * F(F(x) - K[2])) / K[0]))   # Get3DPointonRefCam, fp32 (reference :200-209)
        A[i, 1] = float(F(F(d * F(F(y) - K[5])) / K[4]))
        A[i, 2] = float(d)
    _, s, vt = np.linalg.svd(A)
    n4 = vt[-1]
    norm = np.sqrt(n4[0] ** 2 + n4[1] ** 2 + n4[2] ** 2)
    if n4[3] < 0:
        norm = -norm
    return (n4 / norm), float(s[2] / s[0])


def depth_from_plane(K, pl, x, y):
    """GetDepthFromPlaneParam, reference src/PatchMatch.cpp:650-653 (float64 here; compared with tolerance)"""
    return -pl[3] * K[0] / ((x - K[2]) * pl[0] + (K[0] / K[4]) * (y - K[5]) * pl[1] + K[0] * pl[2])


# ---------------------------------------------------------------------------------------------------------------
def main():
    rng = np.random.default_rng(20240311)
    out = {}

    # -- vertices: sizes that are not multiples of the 5x5 cell, costs spanning every branch
    for tag, (h, w) in {"a": (47, 63), "b": (30, 41)}.items():
        costs = rng.uniform(0.0, 2.2, (h, w)).astype(F)
        costs[rng.uniform(size=(h, w)) < 0.35] *= F(0.04)           # plenty of cells below 0.1
        costs[rng.uniform(size=(h, w)) < 0.05] = F(2.0)             # invalid pixels
        geom = rng.uniform(0.0, 0.9, (h, w)).astype(F)
        # one cell with equal minima (first one wins) and one without any candidate
        costs[5:10, 5:10] = F(0.05)
        costs[10:15, 10:15] = F(1.7)
        out[f"vert_{tag}_costs"], out[f"vert_{tag}_geom"] = costs, geom
        out[f"vert_{tag}_plain"] = vertices_ref(costs, geom, False)
        out[f"vert_{tag}_geomprior"] = vertices_ref(costs, geom, True)

    # -- Delaunay on general-position point sets (uniform; and jittered 5x5 cells like real vertex sets)
    w, h = 160, 120
    pts, tris = general_position_points(rng, w, h, 220)
    out["del_u_size"], out["del_u_points"], out["del_u_tris"] = np.array([w, h], np.int32), pts, tris
    while True:
        cells = [(cx, cy) for cy in range(0, 90, 5) for cx in range(0, 120, 5) if rng.uniform() < 0.8]
        cand = np.array([(cx + rng.integers(0, 5), cy + rng.integers(0, 5)) for cx, cy in cells], np.int64)
        tri = Delaunay(cand.astype(np.float64))
        P = [tuple(int(v) for v in p) for p in cand]
        bad = len(tri.coplanar) > 0
        for simp, nbrs in zip(tri.simplices, tri.neighbors):
            a, b, c = (P[i] for i in simp)
            if (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]) == 0:
                bad = True
            for nb in nbrs:
                if nb >= 0:
                    opp = [i for i in tri.simplices[nb] if i not in simp]
                    if incircle(a, b, c, P[opp[0]]) == 0:
                        bad = True
        if not bad:
            break
    # hull triangles of a jittered grid may be slivers along collinear hull points: compare interior triangles only,
    # i.e. those none of whose vertices lies on the convex hull
    on_hull = np.zeros(len(cand), bool)
    on_hull[np.unique(tri.convex_hull)] = True
    t2 = np.sort(tri.simplices.astype(np.int32), axis=1)
    t2 = t2[~on_hull[t2].any(1)]
    t2 = t2[np.lexsort((t2[:, 2], t2[:, 1], t2[:, 0]))]
    out["del_j_size"], out["del_j_points"], out["del_j_tris_interior"] = np.array([120, 90], np.int32), cand.astype(np.int32), t2
    out["del_j_on_hull"] = on_hull

    # -- raster + planes + mask on the uniform set, triangles in a shuffled (but recorded) order so that
    #    "last triangle wins" is exercised on every shared edge pixel
    K = np.array([150.0, 0.0, 79.3, 0.0, 148.0, 60.7, 0.0, 0.0, 1.0], F)
    order = rng.permutation(len(tris))
    tri_pts = pts[tris[order]]                       # [n][3][2]
    yy, xx = np.mgrid[0:h, 0:w]
    depth = (5.0 + 0.8 * np.sin(0.05 * xx) * np.cos(0.04 * yy) + 0.004 * xx).astype(F)
    label = raster_ref(w, h, tri_pts)
    planes = np.zeros((len(tri_pts), 4), np.float64)
    cond = np.zeros(len(tri_pts), np.float64)
    for i, t in enumerate(tri_pts):
        planes[i], cond[i] = plane_ref(K, t, depth)
    dmin, dmax = F(4.4), F(5.9)
    mask = np.zeros((h, w), np.uint32)
    ambiguous = np.zeros((h, w), bool)
    for y in range(h):
        for x in range(w):
            lab = int(label[y, x])
            if lab:
                d = depth_from_plane(K.astype(np.float64), planes[lab - 1], x, y)
                if abs(d - float(dmin)) < 1e-4 * float(dmin) or abs(d - float(dmax)) < 1e-4 * float(dmax):
                    ambiguous[y, x] = True
                if float(dmin) <= d <= float(dmax):
                    mask[y, x] = lab
    out.update(ras_K=K, ras_size=np.array([w, h], np.int32), ras_tri_pts=tri_pts.astype(np.int32), ras_depth=depth, ras_label=label,
               ras_planes=planes, ras_plane_cond=cond, ras_dmin=dmin, ras_dmax=dmax, ras_mask=mask, ras_ambiguous=ambiguous)

    path = os.path.join(HERE, "prior_golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})
    print("label coverage", float((label > 0).mean()), "mask coverage", float((mask > 0).mean()), "ambiguous", int(ambiguous.sum()),
          "worst plane conditioning s3/s1", float(cond.min()))


if __name__ == "__main__":
    main()
