// pm_sky.hpp -- joint-bilateral refinement of a sky-probability mask:
// the reference's only other device kernel, Pixel_bilateral_filter
// (SkySegment/src/SkyRegionDetect.cu:3-34).  Per pixel: a 37 x 37 window, weight
// exp(-|offset| / 72 - |colour difference| / 8), weighted mean of the coarse mask,
// threshold 0.6 -> 255 / 0.  The segmentation network that produces the coarse mask is
// outside the scope (external runtime); the filter and the mask's use in fusion are here.
//
// Arithmetic (canonical): IEEE sqrt and division, own exp; the
// reference's tap order (x offset outer, y offset inner) is kept, out-of-image taps
// contribute an exact +0 (the reference skips them), `prob += w * mask` is one fma (what
// nvcc's default contraction makes of SkyRegionDetect.cu:29-30).
//
// Mapping: a 256-thread block produces a 32 x 16 tile, two pixels per thread (rows ty and
// ty + 8, sharing the spatial term and doubling ILP); the 68 x 52 texel neighbourhood
// (colour packed in one dword + a validity byte, mask as fp32: 8 B per texel, 28 KB) and the
// 37 x 37 spatial table (5.5 KB) live in LDS.  VALU bound: 1369 taps x ~36 instructions.
#pragma once

#include "pm_device.hpp"

namespace pm {

constexpr int kSkyHalf = 18, kSkyWin = 2 * kSkyHalf + 1;
constexpr int kSkyTW = 32, kSkyTH = 16;
constexpr int kSkyLW = kSkyTW + 2 * kSkyHalf, kSkyLH = kSkyTH + 2 * kSkyHalf;

PM_DEV void sky_tap(uint2 t, float cb, float cg, float cr, float sp, float& wsum, float& prob) {
    const float db = ubyte_to_float<0>(t.x) - cb, dg = ubyte_to_float<1>(t.x) - cg, dr = ubyte_to_float<2>(t.x) - cr;
    const float d2 = (db * db + dg * dg) + dr * dr;  // integers below 2^24: exact in any order
    const float dc = d_sqrt_normal(d2);  // 0 or an integer below 2^18
    const float e = __builtin_fmaf(dc, -0.125f, sp);  // -distance/72 - dis_color/8; the product is exact
    float w = d_exp_inrange(e);  // e in [-56, 0]
    w = (t.x >> 24) ? w : 0.0f;
    wsum += w;
    prob = __builtin_fmaf(w, __uint_as_float(t.y), prob);
}

__global__ __launch_bounds__(256) void k_sky_bilateral(const unsigned char* __restrict__ bgr, const float* __restrict__ mask, float* __restrict__ out,
                                                       int height, int width) {
    __shared__ uint2 tile[kSkyLH][kSkyLW];
    __shared__ float spatial[kSkyWin * kSkyWin];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int x0 = blockIdx.x * kSkyTW - kSkyHalf, y0 = blockIdx.y * kSkyTH - kSkyHalf;
    for (int k = threadIdx.x; k < kSkyLH * kSkyLW; k += 256) {
        const int ly = k / kSkyLW, lx = k - ly * kSkyLW;
        const int gx = x0 + lx, gy = y0 + ly;
        uint2 v = make_uint2(0u, 0u);
        if (gx >= 0 && gx < width && gy >= 0 && gy < height) {
            const size_t g = (size_t)gy * width + gx;
            v.x = (unsigned)bgr[3 * g] | ((unsigned)bgr[3 * g + 1] << 8) | ((unsigned)bgr[3 * g + 2] << 16) | 0xFF000000u;
            v.y = __float_as_uint(mask[g]);
        }
        tile[ly][lx] = v;
    }
    for (int k = threadIdx.x; k < kSkyWin * kSkyWin; k += 256) {
        const int i = k / kSkyWin - kSkyHalf, j = k % kSkyWin - kSkyHalf;
        spatial[k] = -(__builtin_sqrtf((float)(i * i + j * j)) / 72.0f);
    }
    __syncthreads();
    const uint2 c0 = tile[ty + kSkyHalf][tx + kSkyHalf], c1 = tile[ty + 8 + kSkyHalf][tx + kSkyHalf];
    const float b0 = ubyte_to_float<0>(c0.x), g0 = ubyte_to_float<1>(c0.x), r0 = ubyte_to_float<2>(c0.x);
    const float b1 = ubyte_to_float<0>(c1.x), g1 = ubyte_to_float<1>(c1.x), r1 = ubyte_to_float<2>(c1.x);
    float ws0 = 0.0f, pr0 = 0.0f, ws1 = 0.0f, pr1 = 0.0f;
    for (int i = 0; i < kSkyWin; ++i) {
#pragma unroll 4
        for (int j = 0; j < kSkyWin; ++j) {
            const float sp = spatial[i * kSkyWin + j];
            sky_tap(tile[ty + j][tx + i], b0, g0, r0, sp, ws0, pr0);
            sky_tap(tile[ty + 8 + j][tx + i], b1, g1, r1, sp, ws1, pr1);
        }
    }
    const int gx = blockIdx.x * kSkyTW + tx, gy = blockIdx.y * kSkyTH + ty;
    if (gx < width) {
        if (gy < height) out[(size_t)gy * width + gx] = (double)(pr0 / ws0) > 0.6 ? 255.0f : 0.0f;
        if (gy + 8 < height) out[(size_t)(gy + 8) * width + gx] = (double)(pr1 / ws1) > 0.6 ? 255.0f : 0.0f;
    }
}

}  // namespace pm
