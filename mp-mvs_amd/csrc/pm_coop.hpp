// pm_coop.hpp -- cooperative mapping of the PatchMatch evaluations onto wave64: a GROUP of G adjacent lanes owns one
// reference pixel and the (hypothesis, source view) evaluations of that pixel are dealt to the G lanes as a flat list of
// pairs.  Compared with one thread per pixel (pm_kernels.hpp: k_update) this
//   * shares the pixel's 36 bilateral weight records in LDS between G lanes (288 B per pixel instead of per thread),
//     so LDS no longer pins the kernel at 2 waves per SIMD;
//   * keeps the per-pixel cost matrix (8 candidates x V views) in LDS instead of per-thread scratch;
//   * skips dead evaluations at the granularity of a LANE instead of a wave: zero-weight views after the view
//     sampling (about half of them with the shipped 20-view configuration), candidates without a valid neighbour at the
//     image border;
//   * every lane samples its own source view: all textures live in one allocation behind one wave-uniform buffer
//     resource (LaneTex), the view's constants come from an LDS copy of the per-view table.
// The arithmetic of every evaluation and the order of every reduction are those of pm_device.hpp / the oracle: results
// are bit-identical to the one-thread-per-pixel kernels.
#pragma once

#include "pm_kernels.hpp"

namespace pm {

// per-view constants as the cooperative kernels read them from LDS (5 x 16 bytes)
struct ViewLds {
    float A[9];
    float b[3];
    float wf, hf, wm1, hm1;
    int pitch;
    int base;
    int pad0, pad1;
};
static_assert(sizeof(ViewLds) == 80, "ViewLds is read as 5 float4");

PM_DEV void stage_views(const ProblemDev& P, ViewLds* vl, bool u8) {
    for (int v = threadIdx.x; v < P.V; v += kBlockThreads) {
        const ViewDev& s = P.views[v];
        ViewLds o;
#pragma unroll
        for (int i = 0; i < 9; ++i) o.A[i] = s.A[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) o.b[i] = s.b[i];
        o.wf = s.wf;
        o.hf = s.hf;
        o.wm1 = s.wm1;
        o.hm1 = s.hm1;
        o.pitch = u8 ? s.pitch8 : s.pitch;
        o.base = (int)s.tex_base;
        o.pad0 = o.pad1 = 0;
        vl[v] = o;
    }
}

PM_DEV __amdgpu_buffer_rsrc_t make_tex_all(const ProblemDev& P) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.tex_all), (short)0, (int)P.tex_all_bytes, 0x00020000);
}

// one (hypothesis, view) evaluation by one lane: m = plane_to_m of the hypothesis, view constants from LDS
template <bool U8, int LWSTRIDE>
PM_DEV float ncc_pair(__amdgpu_buffer_rsrc_t rsrc, const ViewLds& vw, const RefWin& rw, int px, int py, int step, int radius, float m0, float m1,
                      float m2) {
    const float4* q = reinterpret_cast<const float4*>(&vw);
    const float4 c0 = q[0], c1 = q[1], c2 = q[2], c3 = q[3], c4 = q[4];  // A0..3 | A4..7 | A8 b0 b1 b2 | wf hf wm1 hm1 | pitch base
    const float H0 = __builtin_fmaf(-c2.y, m0, c0.x);
    const float H1 = __builtin_fmaf(-c2.y, m1, c0.y);
    const float H2 = __builtin_fmaf(-c2.y, m2, c0.z);
    const float H3 = __builtin_fmaf(-c2.z, m0, c0.w);
    const float H4 = __builtin_fmaf(-c2.z, m1, c1.x);
    const float H5 = __builtin_fmaf(-c2.z, m2, c1.y);
    const float H6 = __builtin_fmaf(-c2.w, m0, c1.z);
    const float H7 = __builtin_fmaf(-c2.w, m1, c1.w);
    const float H8 = __builtin_fmaf(-c2.w, m2, c2.x);
    LaneTex tex;
    tex.rsrc = rsrc;
    tex.pitch = __float_as_int(c4.x);
    tex.base = __float_as_int(c4.y);
    tex.wm1 = c3.z;
    tex.hm1 = c3.w;
    return ncc_core<U8, LWSTRIDE>(tex, c3.x, c3.y, H0, H1, H2, H3, H4, H5, H6, H7, H8, rw, px, py, step, radius);
}

// ---------------------------------------------------------------------------------------------------------------------
// Reference-window weights of one pixel, computed by its G lanes together: lane g takes the window columns a = g, g + G
// (< 6), writes their records and column sums; every lane then adds the six column sums in the reference's order
// (ref .cu:365-395), so all G lanes hold identical (inv_w, mean_r, var_r).  lw = &records[pixel], record stride = PIX.
// colsum = 18 floats of LDS per pixel ([column][w, wr, wrr]).  Lanes of one group sit in one wave: the only
// synchronisation needed between the writes and the reads is the wave-level one below.
// ---------------------------------------------------------------------------------------------------------------------
PM_DEV void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int G, int PIX>
PM_DEV void ref_window_coop(float4* lw, float* colsum, int g, const float* ctr, int tpitch, int step, int radius, float two_ss, float two_sc,
                            RefWin& rw) {
    const float rc = ctr[0];
    for (int a = g; a < 6; a += G) {
        float pw = 0.0f, pwr = 0.0f, pwrr = 0.0f;
        float wv[6], wrv[6];
        const int dx = a * step - radius;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const int dy = b * step - radius;
            const float r = ctr[dy * tpitch + dx];
            const float sd = __builtin_sqrtf((float)dx * (float)dx + (float)dy * (float)dy);
            const float e = (-sd) / two_ss - __builtin_fabsf(r - rc) / two_sc;
            const float w = d_exp(e);
            const float wr = w * r;
            wv[b] = w;
            wrv[b] = wr;
            pw += w;
            pwr += wr;
            pwrr = __builtin_fmaf(wr, r, pwrr);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) lw[(a * 3 + j) * PIX] = make_float4(wv[2 * j], wv[2 * j + 1], wrv[2 * j], wrv[2 * j + 1]);
        colsum[a * 3 + 0] = pw;
        colsum[a * 3 + 1] = pwr;
        colsum[a * 3 + 2] = pwrr;
    }
    wave_lds_sync();
    float sw = 0.0f, swr = 0.0f, swrr = 0.0f;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        sw += colsum[a * 3 + 0];
        swr += colsum[a * 3 + 1];
        swrr += colsum[a * 3 + 2];
    }
    rw.lw = lw;
    rw.inv_w = 1.0f / sw;
    rw.mean_r = swr * rw.inv_w;
    const float mrr = swrr * rw.inv_w;
    rw.var_r = __builtin_fmaf(-rw.mean_r, rw.mean_r, mrr);
}

// LDS of the cooperative kernels (floats): [18 records x PIX float4][colsum 18 x PIX][views V x 20][tile or cost matrix]
template <int G>
constexpr int kCoopPix = kBlockThreads / G;
template <int G>
constexpr int kCoopFixedFloats = 18 * 4 * kCoopPix<G> + 18 * kCoopPix<G> + kMaxViews * 20;

// ---------------------------------------------------------------------------------------------------------------------
// Probe: ComputeBilateralNCC of nh planes per pixel against every view, cooperative mapping; out [nh][V][H][W].
// Block = 8 x (32 / G) ... dense patch of kCoopPix<G> pixels: 8 wide, kCoopPix/8 high; a wave holds 64/G of them.
// ---------------------------------------------------------------------------------------------------------------------
template <bool U8, int G, int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_eval_ncc_coop(const ProblemDev* __restrict__ Pp, const float4* __restrict__ planes, int nh,
                                                                      float* __restrict__ out, LaunchArgs a) {
    constexpr int PIX = kCoopPix<G>;
    constexpr int BW = 8, BH = PIX / 8;
    const ProblemDev& P = *Pp;
    float4* rec = reinterpret_cast<float4*>(pm_lds);
    float* colsum_all = pm_lds + 18 * 4 * PIX;
    ViewLds* vl = reinterpret_cast<ViewLds*>(pm_lds + 18 * 4 * PIX + 18 * PIX);
    float* tile = pm_lds + kCoopFixedFloats<G>;

    const int pix = threadIdx.x / G, g = threadIdx.x % G;
    const int x0 = blockIdx.x * BW, y0 = blockIdx.y * BH;
    const int x = x0 + (pix % BW), y = y0 + (pix / BW);
    const bool valid = x < P.W && y < P.H;
    const int step = 2 << a.scale, radius = 5 * step / 2;
    stage_views(P, vl, U8);
    int tpitch;
    const float* ctr = ref_center(P, tile, x, y, x0, y0, BW, BH, radius, a.scale, tpitch);
    if (!use_ref_tile(a.scale, BW, BH)) __syncthreads();  // the staged view table
    if (!valid) return;
    const int idx = y * P.W + x;
    RefWin rw;
    ref_window_coop<G, PIX>(rec + pix, colsum_all + pix * 18, g, ctr, tpitch, step, radius, a.two_ss, a.two_sc, rw);
    const __amdgpu_buffer_rsrc_t rsrc = make_tex_all(P);
    const long wh = (long)P.W * P.H;
    const int V = P.V, npairs = nh * V;
    int h = 0, v = g;
    while (v >= V) {
        v -= V;
        ++h;
    }
    for (int k = g; k < npairs; k += G) {
        float m0, m1, m2;
        plane_to_m(P, planes[h * wh + idx], m0, m1, m2);
        out[(long)k * wh + idx] = ncc_pair<U8, PIX>(rsrc, vl[v], rw, x, y, step, radius, m0, m1, m2);
        v += G;
        while (v >= V) {
            v -= V;
            ++h;
        }
    }
}

}  // namespace pm
