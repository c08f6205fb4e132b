// pm_fusion.hpp -- depth-map fusion on the GPU: the consumer of the
// hot path's depth/normal maps (reference RunFusion, src/PatchMatch.cpp:287-504).
//
// The reference fuses sequentially: every accepted point masks the source pixels it
// used, and later pixels / images skip masked pixels, so its result depends on raster
// order (and on a used_list that is never reset between pixels).  That cannot run in
// parallel.  This kernel implements the "snapshot" formulation of DESIGN.md section 8:
// images are still fused in index order, but all pixels of one image read the masks as
// they were when the image started; the masks it produces become visible to the next
// image.  One thread = one pixel; HBM-bound (per pixel: own depth/normal/colour plus,
// per source view, one mask byte, one depth, one normal, one colour).
#pragma once

#include "pm_device.hpp"

namespace pm {

struct FuseView {
    CamDev cam;
    int w, h;
    const float* depth;
    const float* normal;  // 3 floats per pixel (world)
    const unsigned char* color;  // 8-bit, cch interleaved channels (1 = grey, 3 = B,G,R as cv::Vec3b)
    int cch;
    const unsigned char* sky;    // optional sky mask (> 0 = sky, reference :385-388) or null
    unsigned char* mask;         // snapshot read by the kernel (written only at a thread's own sky pixel)
    unsigned char* mask_next;    // marks written by the kernel
};

// reference src/PatchMatch.cpp:211-231
PM_DEV void point_on_world(const CamDev& cam, int x, int y, float depth, float& o0, float& o1, float& o2) {
    backproject(cam, (float)x, (float)y, depth, o0, o1, o2);
}
// reference src/PatchMatch.cpp:251-261
PM_DEV void project_depth(const CamDev& cam, float p0, float p1, float p2, float& u, float& v, float& depth) {
    const float t0 = ((cam.R[0] * p0 + cam.R[1] * p1) + cam.R[2] * p2) + cam.t[0];
    const float t1 = ((cam.R[3] * p0 + cam.R[4] * p1) + cam.R[5] * p2) + cam.t[1];
    const float t2 = ((cam.R[6] * p0 + cam.R[7] * p1) + cam.R[8] * p2) + cam.t[2];
    depth = (cam.K[6] * t0 + cam.K[7] * t1) + cam.K[8] * t2;
    u = ((cam.K[0] * t0 + cam.K[1] * t1) + cam.K[2] * t2) / depth;
    v = ((cam.K[3] * t0 + cam.K[4] * t1) + cam.K[5] * t2) / depth;
}
PM_DEV bool round_index(float v, int& out) {
    const float f = v + 0.5f;
    if (!(f > -1.0f && f < 1.0e8f)) return false;
    out = (int)f;
    return true;
}

constexpr int kMaxFuseNgb = kMaxViews + 1;

PM_DEV void load_color(const FuseView& V, size_t idx, float& c0, float& c1, float& c2) {
    if (V.cch == 3) {
        c0 = (float)V.color[idx * 3];
        c1 = (float)V.color[idx * 3 + 1];
        c2 = (float)V.color[idx * 3 + 2];
    } else {
        c0 = c1 = c2 = (float)V.color[idx];
    }
}

__global__ __launch_bounds__(256) void k_fuse(const FuseView* __restrict__ views, int i, const int* __restrict__ src_ids, int num_ngb,
                                              int use_dynamic, unsigned char* __restrict__ out_valid, float* __restrict__ out9) {
    const FuseView& R = views[i];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), r = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (c >= R.w || r >= R.h) return;
    const size_t pix = (size_t)r * R.w + c;
    if (R.mask[pix] == 1) return;
    if (R.sky && R.sky[pix] > 0) {  // only this thread ever reads mask[pix] of the image being fused
        R.mask[pix] = 1;
        R.mask_next[pix] = 1;
        return;
    }
    const float ref_depth = R.depth[pix];
    if (ref_depth <= 0.0f) return;
    const float n0 = R.normal[pix * 3], n1 = R.normal[pix * 3 + 1], n2 = R.normal[pix * 3 + 2];
    float X0, X1, X2;
    point_on_world(R.cam, c, r, ref_depth, X0, X1, X2);
    float sp0 = X0, sp1 = X1, sp2 = X2, sn0 = n0, sn1 = n1, sn2 = n2;
    float sc0, sc1, sc2;
    load_color(R, pix, sc0, sc1, sc2);
    int num = 0;
    float dyn = 0.0f;
    int used[kMaxFuseNgb];
    for (int j = 0; j < num_ngb; ++j) used[j] = -1;
    for (int j = 1; j < num_ngb; ++j) {
        if (j == num_ngb - 1 && num == 0) break;  // ref :402-403
        const int s = src_ids[j];
        const FuseView& S = views[s];
        float u, v, pd;
        project_depth(S.cam, X0, X1, X2, u, v, pd);
        int sr, sc;
        if (!round_index(v, sr) || !round_index(u, sc)) continue;
        if (!(sc >= 0 && sc < S.w && sr >= 0 && sr < S.h)) continue;
        const size_t sidx = (size_t)sr * S.w + sc;
        if (S.mask[sidx] == 1) continue;
        const float sd = S.depth[sidx];
        if (sd <= 0.0f) continue;
        const float m0 = S.normal[sidx * 3], m1 = S.normal[sidx * 3 + 1], m2 = S.normal[sidx * 3 + 2];
        float T0, T1, T2;
        point_on_world(S.cam, sc, sr, sd, T0, T1, T2);
        float bu, bv, bd;
        project_depth(R.cam, T0, T1, T2, bu, bv, bd);
        const float dc = (float)c - bu, dr = (float)r - bv;
        const float err = __builtin_sqrtf(dc * dc + dr * dr);
        if (!(err < 2.0f)) continue;
        const float rel = __builtin_fabsf(bd - ref_depth) / ref_depth;
        if (!(rel < 0.01f)) continue;
        const float dot = (n0 * m0 + n1 * m1) + n2 * m2;
        float angle = d_acos(dot);
        if (angle != angle) angle = 0.0f;  // ref :233-242
        if (angle < 0.174533f) {
            used[j] = (int)sidx;
            sp0 += T0;
            sp1 += T1;
            sp2 += T2;
            sn0 += m0;
            sn1 += m1;
            sn2 += m2;
            float g0, g1, g2;
            load_color(S, sidx, g0, g1, g2);
            sc0 += g0;
            sc1 += g1;
            sc2 += g2;
            const float idx = (err + 200.0f * rel) + angle * 10.0f;
            dyn += d_exp(-idx);
            num++;
        }
    }
    const bool ok = use_dynamic ? (num >= 1 && dyn > 0.3f * (float)num) : (num >= 2);
    if (!ok) return;
    const float d = (float)num + 1.0f;
    float* o = out9 + pix * 9;
    o[0] = sp0 / d;
    o[1] = sp1 / d;
    o[2] = sp2 / d;
    o[3] = sn0 / d;
    o[4] = sn1 / d;
    o[5] = sn2 / d;
    o[6] = sc0 / d;
    o[7] = sc1 / d;
    o[8] = sc2 / d;
    out_valid[pix] = 1;
    for (int j = 1; j < num_ngb; ++j)
        if (used[j] != -1) views[src_ids[j]].mask_next[used[j]] = 1;  // idempotent
}

// ---------------------------------------------------------------------------
// Compaction of one image's fused points into the reference's PointCloud order (raster order inside the image; images are
// appended in index order) as PLY vertex records: x y z nx ny nz (float32) red green blue (uint8), 27 bytes, exactly what
// StoreColorPlyFileBinaryPointCloud writes (ref src/PatchMatch.cpp:145-198).  Three small passes, deterministic:
// per-256-pixel block counts -> exclusive scan (one block) -> scatter with in-block ranks.
// ---------------------------------------------------------------------------
constexpr int kPlyRecord = 27;

__global__ __launch_bounds__(256) void k_fuse_count(const unsigned char* __restrict__ valid, int n, int* __restrict__ block_counts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int v = (i < n && valid[i]) ? 1 : 0;
    __shared__ int wave_sum[4];
    const unsigned long long b = __ballot(v);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = (wave_sum[0] + wave_sum[1]) + (wave_sum[2] + wave_sum[3]);
}

// block_counts[0 .. nb) -> exclusive prefix sums in place; block_counts[nb] receives the image's point count
__global__ __launch_bounds__(256) void k_fuse_scan(int* __restrict__ block_counts, int nb) {
    __shared__ int part[256];
    const int per = (nb + 255) / 256, lo = min(threadIdx.x * per, nb), hi = min(lo + per, nb);
    int s = 0;
    for (int k = lo; k < hi; ++k) s += block_counts[k];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int k = 0; k < 256; ++k) {
            const int c = part[k];
            part[k] = run;
            run += c;
        }
        block_counts[nb] = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int k = lo; k < hi; ++k) {
        const int c = block_counts[k];
        block_counts[k] = run;
        run += c;
    }
}

__global__ __launch_bounds__(256) void k_fuse_scatter(const unsigned char* __restrict__ valid, const float* __restrict__ pts9, int n,
                                                      const int* __restrict__ block_offsets, const long long* __restrict__ base,
                                                      unsigned char* __restrict__ records) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int v = (i < n && valid[i]) ? 1 : 0;
    __shared__ int wave_sum[4];
    const unsigned long long b = __ballot(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wave_sum[wv] = __popcll(b);
    __syncthreads();
    if (!v) return;
    int rank = __popcll(b & ((1ull << lane) - 1ull));
    for (int k = 0; k < wv; ++k) rank += wave_sum[k];
    const long long slot = *base + block_offsets[blockIdx.x] + rank;
    const float* p = pts9 + (size_t)i * 9;
    float rec[6] = {p[0], p[1], p[2], p[3], p[4], p[5]};
    const float big = 3.402823466e+38f;  // FLT_MAX
    const bool finite = (rec[0] < big && rec[0] > -big) && (rec[1] < big && rec[1] > -big) && (rec[2] < big && rec[2] >= -big);  // ref :176-178
    if (!finite) rec[0] = rec[1] = rec[2] = 0.0f;
    unsigned char* o = records + slot * kPlyRecord;
    const unsigned char* src = (const unsigned char*)rec;
    for (int k = 0; k < 24; ++k) o[k] = src[k];
    o[24] = (unsigned char)(int)p[8];  // red   <- colour[2] (the image is B,G,R; ref :181-186)
    o[25] = (unsigned char)(int)p[7];  // green
    o[26] = (unsigned char)(int)p[6];  // blue
}

// *base += count of the image just scattered (block_counts[nb] from k_fuse_scan)
__global__ void k_fuse_advance(long long* __restrict__ base, const int* __restrict__ image_count) { *base += *image_count; }

}  // namespace pm
