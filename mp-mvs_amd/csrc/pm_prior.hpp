// pm_prior.hpp -- the planar-prior block of ProcessProblem (reference src/PatchMatch.cpp:532-604) on
// the device, so that the maps a Run() left in HBM never cross PCIe between the two Run() calls of a Problem:
//   k_prior_cells + scan/scatter   GetTriangulateVertices (ref :782-853): per 5x5 cell the reliable pixels, compacted in cell
//                                  raster order -- only the vertex list (a few hundred KB) goes to the host;
//   [host]                         Delaunay triangulation of the vertices (mp-mvs_amd/host/planar_prior.cpp);
//   k_prior_raster                 triangle rasterisation by barycentric stepping (ref :554-570), "last triangle wins" as an
//                                  atomic max over labels, and the plane through the three back-projected vertices
//                                  (GetPriorPlaneParams, ref :723-755) -- one wave per 64 stepping rows of a triangle;
//   k_prior_finish                 depth-range test of the prior depth (ref :583-595) and the per-pixel prior planes / mask
//                                  CudaPlanarPriorInitialization uploads in the reference (ref :978-996).
// Arithmetic: the same fp32 / fp64 IEEE operations in the same order as the host implementation (planar_prior.cpp), so the
// two produce identical bits (tests/test_prior_gpu.py) and both are pinned by the independent fixtures of
// tests/golden/prior_golden_v1.npz.
#pragma once

#include "pm_device.hpp"

namespace pm {

constexpr int kPriorCell = 5;

// ---------------------------------------------------------------------------------------------------------------------
// vertices
// ---------------------------------------------------------------------------------------------------------------------
// one thread per cell: count (0..3) and the picked pixels, (y << 16) | x
__global__ __launch_bounds__(256) void k_prior_cells(const float* __restrict__ cost, const float* __restrict__ geom, int W, int H, int geom_rule,
                                                     int ncx, int ncells, int* __restrict__ cell_cnt, uint32_t* __restrict__ cell_pts) {
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= ncells) return;
    const int cy = cell / ncx, cx = cell - cy * ncx;
    const int x0 = cx * kPriorCell, y0 = cy * kPriorCell;
    const int x1 = min(W, x0 + kPriorCell), y1 = min(H, y0 + kPriorCell);
    uint32_t where[3] = {0u, 0u, 0u};
    int n = 0;
    if (!geom_rule) {
        // the pixel of lowest valid cost, kept if below 0.1; ties keep the first in raster order
        float lowest = 2.0f;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                const float c = cost[(size_t)y * W + x];
                if (c < 2.0f && c < lowest) {
                    lowest = c;
                    where[0] = ((uint32_t)y << 16) | (uint32_t)x;
                }
            }
        n = lowest < 0.1f ? 1 : 0;
    } else {
        // the three lowest costs among pixels with cost < 1 and geometric cost < 0.4, ascending, kept while below
        // max(0.2, 0.85 * cell cost sum / (x1 * y1))   (the reference's divisor: product of the END coordinates)
        float low[3] = {2.0f, 2.0f, 2.0f};
        float sum = 0.0f;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                const size_t i = (size_t)y * W + x;
                const float c = cost[i];
                sum += c;
                if (!(c < 1.0f && geom[i] < 0.4f && c < low[2])) continue;
                const uint32_t at = ((uint32_t)y << 16) | (uint32_t)x;
                if (low[1] > c) {
                    low[2] = low[1];
                    where[2] = where[1];
                    if (low[0] > c) {
                        low[1] = low[0];
                        where[1] = where[0];
                        low[0] = c;
                        where[0] = at;
                    } else {
                        low[1] = c;
                        where[1] = at;
                    }
                } else {
                    low[2] = c;
                    where[2] = at;
                }
            }
        const float scaled = (float)((double)(sum / (float)(y1 * x1)) * 0.85);
        const float limit = scaled > 0.2f ? scaled : 0.2f;
        while (n < 3 && low[n] < limit) ++n;
    }
    cell_cnt[cell] = n;
    cell_pts[cell * 3 + 0] = where[0];
    cell_pts[cell * 3 + 1] = where[1];
    cell_pts[cell * 3 + 2] = where[2];
}

// per-256-cell block sums
__global__ __launch_bounds__(256) void k_prior_block_sums(const int* __restrict__ cell_cnt, int ncells, int* __restrict__ block_sums) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    int v = i < ncells ? cell_cnt[i] : 0;
    __shared__ int part[256];
    part[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = part[0];
}

// block_sums[0 .. nb) -> exclusive prefix sums in place; block_sums[nb] = total (one block)
__global__ __launch_bounds__(256) void k_prior_scan(int* __restrict__ block_sums, int nb) {
    __shared__ int part[256];
    const int per = (nb + 255) / 256, lo = min((int)threadIdx.x * per, nb), hi = min(lo + per, nb);
    int s = 0;
    for (int k = lo; k < hi; ++k) s += block_sums[k];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int k = 0; k < 256; ++k) {
            const int c = part[k];
            part[k] = run;
            run += c;
        }
        block_sums[nb] = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int k = lo; k < hi; ++k) {
        const int c = block_sums[k];
        block_sums[k] = run;
        run += c;
    }
}

// vertices in cell raster order: out_xy[2 * k] = x, [2 * k + 1] = y
__global__ __launch_bounds__(256) void k_prior_scatter(const int* __restrict__ cell_cnt, const uint32_t* __restrict__ cell_pts, int ncells,
                                                       const int* __restrict__ block_offsets, int cap, int* __restrict__ out_xy) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int v = i < ncells ? cell_cnt[i] : 0;
    __shared__ int part[256];
    part[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int k = 0; k < 256; ++k) {
            const int c = part[k];
            part[k] = run;
            run += c;
        }
    }
    __syncthreads();
    const int base = block_offsets[blockIdx.x] + part[threadIdx.x];
    for (int k = 0; k < v; ++k) {
        if (base + k >= cap) return;
        const uint32_t p = cell_pts[i * 3 + k];
        out_xy[2 * (base + k)] = (int)(p & 0xffffu);
        out_xy[2 * (base + k) + 1] = (int)(p >> 16);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// raster + plane fit.  The reference steps barycentric coordinates p, q in increments of 1 / (longest edge)
// (ref :564-570), i.e. about L^2 / 2 samples for a triangle of longest edge L.  Inside the image L is a few pixels, but the
// slivers that close the convex hull along the image border reach L ~ image width: a handful of them hold most of the
// samples (measured on cfg 3: 60 % of all samples in triangles with L > 100) and nearly all of their samples fall on the
// same two pixel rows.  So the work is cut into TASKS of 64 consecutive p-rows of one triangle (host-built table, one
// wave per task, one row per lane: every lane re-derives its p by the reference's own fp32 accumulation), and a sample
// only issues its atomic max when the label it can see is still smaller -- labels only grow, so a stale value can only cause
// a redundant atomic, never a missed one.  Triangle k gets label k + 1 ("the last triangle wins" = the largest label).
// The task with row 0 also fits the plane.  tri = n x {x1 y1 x2 y2 x3 y3}; planes4 = the state of the last Run():
// (world normal, depth).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prior_raster(const int* __restrict__ tri, int n_tasks, const int* __restrict__ task_tri,
                                                      const int* __restrict__ task_row0, const ProblemDev* __restrict__ Pp,
                                                      const float4* __restrict__ planes4, uint32_t* __restrict__ label,
                                                      float4* __restrict__ tri_planes) {
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (task >= n_tasks) return;
    const ProblemDev& P = *Pp;
    const int W = P.W, H = P.H;
    const int t = task_tri[task], row0 = task_row0[task];
    const int x1 = tri[6 * t], y1 = tri[6 * t + 1], x2 = tri[6 * t + 2], y2 = tri[6 * t + 3], x3 = tri[6 * t + 4], y3 = tri[6 * t + 5];
    const float L01 = (float)__builtin_sqrt((double)((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2)));
    const float L02 = (float)__builtin_sqrt((double)((x1 - x3) * (x1 - x3) + (y1 - y3) * (y1 - y3)));
    const float L12 = (float)__builtin_sqrt((double)((x2 - x3) * (x2 - x3) + (y2 - y3) * (y2 - y3)));
    const float max_edge = fmaxf(L01, fmaxf(L02, L12));
    const float step = (float)(1.0 / (double)max_edge);
    const uint32_t lab = (uint32_t)t + 1u;
    const bool degenerate = !(step > 0.0f) || !(step < 3.0e38f);  // all three vertices equal: one sample (as the host path)
    // this lane's row: p after (row0 + lane) accumulations, exactly as the reference's loop produces it
    const int row = row0 + lane;
    float p = 0.0f;
    bool live = !degenerate || row == 0;
    for (int k = 0; k < row && live; ++k) {
        p += step;
        live = p < 1.0;
    }
    if (live && p < 1.0) {
        for (float q = 0.0f; q < 1.0 - p; q += step) {
            const int x = (int)((double)(p * (float)x1 + q * (float)x2) + (1.0 - p - q) * x3);
            const int y = (int)((double)(p * (float)y1 + q * (float)y2) + (1.0 - p - q) * y3);
            if (x >= 0 && y >= 0 && x < W && y < H) {
                uint32_t* cell = &label[(size_t)y * W + x];
                if (*cell < lab) atomicMax(cell, lab);
            }
            if (degenerate) break;
        }
    }
    if (row != 0) return;
    // plane through the three back-projected vertices (closed form of the reference's 3 x 4 null-space solve), double
    const float K0 = P.cam.K[0], K2 = P.cam.K[2], K4 = P.cam.K[4], K5 = P.cam.K[5];
    const float d1 = planes4[(size_t)y1 * W + x1].w, d2 = planes4[(size_t)y2 * W + x2].w, d3 = planes4[(size_t)y3 * W + x3].w;
    const double X1[3] = {(double)(d1 * ((float)x1 - K2) / K0), (double)(d1 * ((float)y1 - K5) / K4), (double)d1};
    const double X2[3] = {(double)(d2 * ((float)x2 - K2) / K0), (double)(d2 * ((float)y2 - K5) / K4), (double)d2};
    const double X3[3] = {(double)(d3 * ((float)x3 - K2) / K0), (double)(d3 * ((float)y3 - K5) / K4), (double)d3};
    const double u[3] = {X2[0] - X1[0], X2[1] - X1[1], X2[2] - X1[2]};
    const double v[3] = {X3[0] - X1[0], X3[1] - X1[1], X3[2] - X1[2]};
    double nn[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
    double norm = __builtin_sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
    if (!(norm > 1e-30)) {  // collinear points: the fronto-parallel plane
        nn[0] = 0.0;
        nn[1] = 0.0;
        nn[2] = -1.0;
        norm = 1.0;
    }
    const double d = -(nn[0] * X1[0] + nn[1] * X1[1] + nn[2] * X1[2]);
    if (d < 0) norm = -norm;  // ref :746-752: offset made positive
    tri_planes[t] = make_float4((float)(nn[0] / norm), (float)(nn[1] / norm), (float)(nn[2] / norm), (float)(d / norm));
}

// label -> (prior plane, mask): pixels whose prior depth leaves [depth_min, depth_max] lose their label (ref :583-595);
// `mask` holds the raster labels on entry
__global__ __launch_bounds__(256) void k_prior_finish(const ProblemDev* __restrict__ Pp, const float4* __restrict__ tri_planes, float depth_min,
                                                      float depth_max, uint32_t* __restrict__ mask, float4* __restrict__ prior) {
    const ProblemDev& P = *Pp;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.W * P.H) return;
    const int y = i / P.W, x = i - y * P.W;
    uint32_t lab = mask[i];
    float4 pl = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lab > 0) {
        pl = tri_planes[lab - 1];
        const float K0 = P.cam.K[0], K2 = P.cam.K[2], K4 = P.cam.K[4], K5 = P.cam.K[5];
        const float dd = -pl.w * K0 / (((float)x - K2) * pl.x + (K0 / K4) * ((float)y - K5) * pl.y + K0 * pl.z);  // ref :650-653
        if (!(dd <= depth_max && dd >= depth_min)) {
            lab = 0;
            pl = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    mask[i] = lab;
    prior[i] = pl;
}

}  // namespace pm
