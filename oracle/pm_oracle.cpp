// oracle/pm_oracle.cpp
//
// TEST INFRASTRUCTURE ONLY.  CPU restatement of the MP-MVS PatchMatch hot path
// (reference: /root/reference/src/PatchMatch.cu, cited per function below as
// "ref .cu:LINES").  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this library; the product (mp-mvs_amd/) never does.
//
// PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
//  and cannot be built in this image (nvcc, CUDA runtime,
// cuRAND and OpenCV are absent; writing stand-ins for them is not allowed), so
// nothing pins this restatement to reference outputs.  It follows the
// reference's algorithm, including its bug-level quirks, under
// the "canonical arithmetic" of DESIGN.md section 3: IEEE fp32 with explicit
// fmaf, no libm transcendentals (own exp/sin/cos/acos/reciprocal, specified
// to the coefficient), and a counter-based Philox4x32-10 generator in place
// of cuRAND XORWOW.  The HIP kernels implement the same specification
// independently, so HIP-vs-oracle comparisons are bit-exact.
//
// Written as plain scalar C++ (one pixel at a time, OpenMP over rows of one
// checkerboard colour); it shares no source with the product.

#include <cfloat>
#include <cmath>
#include <limits>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#if defined(_OPENMP)
#include <omp.h>
#endif

namespace {

// ---------------------------------------------------------------------------
// PODs, byte-compatible with ref include/PatchMatch.h:35-67
// ---------------------------------------------------------------------------
struct Camera {
    float K[9];
    float R[9];
    float t[3];
    float C[3];
    int height;
    int width;
    float depth_min;
    float depth_max;
};
static_assert(sizeof(Camera) == 112, "Camera must match the reference layout");

struct Params {
    int max_iterations;
    int nSizeHalfWindow;
    int num_images;
    int max_image_size;
    int nSizeStep;
    float sigma_spatial;
    float sigma_color;
    int top_k;
    float depth_min;
    float depth_max;
    int max_scale;
    float scaled_cols;
    float scaled_rows;
    bool geom_consistency;
    bool geomPlanarPrior;
    bool planar_prior;
};
static_assert(sizeof(Params) == 56, "PatchMatchParams must match the reference layout");

struct F4 {
    float x, y, z, w;
};

constexpr int kMaxViews = 32;  // ref .cu:500 (cost_vector[32]), bitmask width

// ---------------------------------------------------------------------------
// Canonical deterministic math.  Every operation is a
// correctly rounded IEEE-754 binary32 operation (+ - * / sqrt fma rint floor)
// or an integer operation, so CPU and GPU produce identical bits.
// ---------------------------------------------------------------------------
inline uint32_t f2u(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
inline float u2f(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// reciprocal as the GPU's d_rcp returns it (hardware seed + two Newton steps; tools/verify_rcp.hip checks this rule against
// the device for all 2^32 inputs): the correctly rounded quotient where z and 1 / z are normal, a signed zero where 1 / z
// would be denormal, NaN for zero, denormal, infinite and NaN z.
inline float det_rcp(float z) {
    const float az = std::fabs(z);
    if (!(az >= 1.17549435e-38f) || az > 3.40282347e+38f) return std::numeric_limits<float>::quiet_NaN();
    if (az > 0x1p+126f) return std::copysign(0.0f, z);
    return 1.0f / z;
}

// fractional part as the GPU's V_FRACT_F32 computes it: x - floor(x), kept below 1
// (a tiny negative x would otherwise round to 1.0)
inline float det_fract(float x) {
    const float f = x - floorf(x);
    return (f < 0.99999994f) ? f : 0.99999994f;
}

// exp: Cody-Waite reduction by ln2 (hi/lo), degree-5 polynomial (Cephes expf
// coefficients), exponent add through the integer representation.
// x < -80 -> 0, x > 80 -> +inf, NaN -> NaN.
inline float det_exp(float x) {
    if (x < -80.0f) return 0.0f;
    if (x > 80.0f) return u2f(0x7f800000u);
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float y = fmaf(p, r * r, r) + 1.0f;
    if (!(x == x)) return y;  // NaN propagates through y
    const int32_t ni = (int32_t)n;
    return u2f(f2u(y) + ((uint32_t)ni << 23));
}

// sin / cos core polynomials (Cephes sinf/cosf), valid for |a| <= pi/4; the
// algorithm only calls them with |a| < 0.1 (ref .cu:464-473, perturbations).
inline float det_sin(float a) {
    const float z = a * a;
    float p = -1.9515295891e-4f;
    p = fmaf(p, z, 8.3321608736e-3f);
    p = fmaf(p, z, -1.6666654611e-1f);
    return fmaf(p * z, a, a);
}
inline float det_cos(float a) {
    const float z = a * a;
    float p = 2.443315711809948e-5f;
    p = fmaf(p, z, -1.388731625493765e-3f);
    p = fmaf(p, z, 4.166664568298827e-2f);
    return fmaf(p * z, z, fmaf(-0.5f, z, 1.0f));
}

// asin for |x| <= 0.5 (Cephes asinf polynomial)
inline float det_asin_core(float x) {
    const float z = x * x;
    float p = 4.2163199048e-2f;
    p = fmaf(p, z, 2.4181311049e-2f);
    p = fmaf(p, z, 4.5470025998e-2f);
    p = fmaf(p, z, 7.4953002686e-2f);
    p = fmaf(p, z, 1.6666752422e-1f);
    return fmaf(p * z, x, x);
}
// acos; |x| > 1 or NaN -> NaN (ref .cu:704,942,953 rely on NaN making the
// following comparison false).
inline float det_acos(float x) {
    if (!(x >= -1.0f && x <= 1.0f)) return u2f(0x7fc00000u);
    if (x > 0.5f) {
        const float s = sqrtf(0.5f * (1.0f - x));
        return 2.0f * det_asin_core(s);
    }
    if (x < -0.5f) {
        const float s = sqrtf(0.5f * (1.0f + x));
        return 3.14159265358979323846f - 2.0f * det_asin_core(s);
    }
    return 1.57079632679489661923f - det_asin_core(x);
}

// ---------------------------------------------------------------------------
// Math of the MEASUREMENT modes (Ctx::literal_mode).  Mode 0 is the canonical arithmetic above -- what the
// HIP kernels compute and every parity test compares.  The other modes exist to measure how far mode 0 is from the reference:
//   1  the reference's formulas as written, IEEE operations in its operation order, libm expf / sinf / cosf / acosf;
//   2  = 1 with CUDA's texture filter (8-bit interpolation fractions);
//   3  a model of the reference BINARY, which is built with nvcc --use_fast_math (ref CMakeLists.txt:18): __expf =
//      ex2.approx(x * log2 e), __sinf / __cosf (absolute error 2^-21.4), every division a multiplication by rcp.approx, fused
//      multiply-adds where nvcc contracts, 8-bit texture fractions.  The approximate instructions are modelled by correctly
//      rounded functions of the ROUNDED intermediate (their error bounds are 1-2 ulp; which ulp the hardware picks is not
//      published): mode 3 is "a second build of the reference's formulas", the control that shows how many pixels the
//      reference's own arithmetic variants flip against each other (tests/test_literal_gpu.py);
//   4  = 3 without the 8-bit texture fractions: the arithmetic of the build alone.
// ---------------------------------------------------------------------------
inline float m_div(int lm, float a, float b) { return lm >= 3 ? a * (1.0f / b) : a / b; }
inline float m_exp(int lm, float x) {
    if (lm == 0) return det_exp(x);
    if (lm >= 3) return (float)exp2((double)(x * 1.44269504f));
    return expf(x);
}
inline float m_sin(int lm, float a) {
    if (lm == 0) return det_sin(a);
    if (lm >= 3) return rintf((float)sin((double)a) * 4194304.0f) / 4194304.0f;
    return sinf(a);
}
inline float m_cos(int lm, float a) {
    if (lm == 0) return det_cos(a);
    if (lm >= 3) return rintf((float)cos((double)a) * 4194304.0f) / 4194304.0f;
    return cosf(a);
}
inline float m_acos(int lm, float x) { return lm == 0 ? det_acos(x) : acosf(x); }

// Philox4x32-10 (Salmon et al. 2011), counter = (pixel, launch, block, tag).
struct Rng {
    uint32_t key0, key1, pix, launch, k;
    uint32_t buf[4];
};
inline void philox_block(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                         uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
inline Rng rng_make(uint64_t seed, uint32_t pix, uint32_t launch) {
    Rng g;
    g.key0 = (uint32_t)seed;
    g.key1 = (uint32_t)(seed >> 32);
    g.pix = pix;
    g.launch = launch;
    g.k = 0;
    g.buf[0] = g.buf[1] = g.buf[2] = g.buf[3] = 0;
    return g;
}
// uniform in (0,1], exactly representable: ((x >> 8) + 1) * 2^-24  (stands in
// for curand_uniform, ref .cu:200 etc.)
inline float rng_uniform(Rng& g) {
    if ((g.k & 3u) == 0u) philox_block(g.pix, g.launch, g.k >> 2, 0x4D504D56u, g.key0, g.key1, g.buf);
    const uint32_t x = g.buf[g.k & 3u];
    g.k++;
    return (float)((x >> 8) + 1u) * 5.9604644775390625e-8f;
}

// ---------------------------------------------------------------------------
// Context
// ---------------------------------------------------------------------------
struct Image {
    std::vector<float> px;
    int w = 0, h = 0;
    // clamp-addressed texel (CUDA clamp addressing; ref .cpp:1014-1018, SURVEY a-2)
    inline float at(int x, int y) const {
        x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
        y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
        return px[(size_t)y * w + x];
    }
};

// per-source-view constants: H = A - b * m^T  (algebraic
// refactoring of ref .cu:228-279)
struct ViewConst {
    float A[9];
    float b[3];
    float wf, hf;  // (float)width, (float)height of the source
    // geometric consistency as two composed projective maps: reference pixel at depth z -> source pixel
    // ~ z * Gf (x, y, 1)^T + gf; source pixel at depth d -> reference pixel ~ d * Gb (u, v, 1)^T + gb
    float Gf[9], gf[3], Gb[9], gb[3];
};

struct Ctx {
    int n_img = 0;
    std::vector<Camera> cams;
    std::vector<Image> imgs;
    std::vector<Image> depths;  // per source view (index v-1)
    std::vector<ViewConst> vc;  // per source view
    float ifx = 0, ify = 0, cxfx = 0, cyfy = 0;
    int W = 0, H = 0;
    std::vector<F4> planes;
    std::vector<float> costs;
    std::vector<float> geom;
    std::vector<uint32_t> sel;
    std::vector<F4> prior;
    std::vector<uint32_t> mask;
    bool have_prior = false;
    int literal_mode = 0;  // 0 canonical; 1-3 measurement modes of the WHOLE path (see m_div / m_exp above)
    bool tex_q8 = false;   // canonical arithmetic with 8-bit interpolation fractions: the checker of libmpmvs_hip_q8.so
    std::string err;
};

// G = K_b (R_b R_a^T) Kinv'_a, g = K_b (R_b C_a + t_b): a pixel of camera a at depth z seen in camera b (ref .cu:582-615:
// BackProjectPoint2W applies Kinv' -- no skew --, R_a^T and C_a; ProjectPoint applies R_b, t_b and the full K_b).  Double, in
// this fixed order, rounded once.
void geom_maps(const Camera& a, const Camera& b, float G[9], float g[3]) {
    const double fx = a.K[0], fy = a.K[4], cx = a.K[2], cy = a.K[5];
    double Rba[9], M[9], tb[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Rba[i * 3 + j] = ((double)b.R[i * 3] * (double)a.R[j * 3] + (double)b.R[i * 3 + 1] * (double)a.R[j * 3 + 1]) +
                             (double)b.R[i * 3 + 2] * (double)a.R[j * 3 + 2];
    for (int i = 0; i < 3; ++i) {
        M[i * 3 + 0] = Rba[i * 3 + 0] / fx;
        M[i * 3 + 1] = Rba[i * 3 + 1] / fy;
        M[i * 3 + 2] = (Rba[i * 3 + 2] - (Rba[i * 3 + 0] * cx) / fx) - (Rba[i * 3 + 1] * cy) / fy;
        tb[i] = (((double)b.R[i * 3] * (double)a.C[0] + (double)b.R[i * 3 + 1] * (double)a.C[1]) + (double)b.R[i * 3 + 2] * (double)a.C[2]) + (double)b.t[i];
    }
    for (int i = 0; i < 3; ++i) {
        const double k0 = b.K[i * 3], k1 = b.K[i * 3 + 1], k2 = b.K[i * 3 + 2];
        for (int j = 0; j < 3; ++j) G[i * 3 + j] = (float)((k0 * M[0 + j] + k1 * M[3 + j]) + k2 * M[6 + j]);
        g[i] = (float)((k0 * tb[0] + k1 * tb[1]) + k2 * tb[2]);
    }
}

void precompute_views(Ctx& c) {
    const Camera& r = c.cams[0];
    const double fx = r.K[0], fy = r.K[4], cx = r.K[2], cy = r.K[5];
    c.ifx = (float)(1.0 / fx);
    c.ify = (float)(1.0 / fy);
    c.cxfx = (float)(cx / fx);
    c.cyfy = (float)(cy / fy);
    c.vc.assign(c.n_img > 1 ? c.n_img - 1 : 0, ViewConst());
    for (int v = 1; v < c.n_img; ++v) {
        const Camera& s = c.cams[v];
        double Rrel[9], trel[3], M[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                Rrel[i * 3 + j] = ((double)s.R[i * 3 + 0] * (double)r.R[j * 3 + 0] + (double)s.R[i * 3 + 1] * (double)r.R[j * 3 + 1]) +
                                  (double)s.R[i * 3 + 2] * (double)r.R[j * 3 + 2];
        double Crel[3];
        for (int k = 0; k < 3; ++k) Crel[k] = (double)r.C[k] - (double)s.C[k];
        for (int i = 0; i < 3; ++i)
            trel[i] = ((double)s.R[i * 3 + 0] * Crel[0] + (double)s.R[i * 3 + 1] * Crel[1]) + (double)s.R[i * 3 + 2] * Crel[2];
        for (int i = 0; i < 3; ++i) {
            M[i * 3 + 0] = Rrel[i * 3 + 0] / fx;
            M[i * 3 + 1] = Rrel[i * 3 + 1] / fy;
            M[i * 3 + 2] = (Rrel[i * 3 + 2] - (Rrel[i * 3 + 0] * cx) / fx) - (Rrel[i * 3 + 1] * cy) / fy;
        }
        ViewConst& o = c.vc[v - 1];
        const double k0 = s.K[0], k2 = s.K[2], k4 = s.K[4], k5 = s.K[5], k8 = s.K[8];
        for (int j = 0; j < 3; ++j) {
            o.A[0 + j] = (float)(k0 * M[0 + j] + k2 * M[6 + j]);
            o.A[3 + j] = (float)(k4 * M[3 + j] + k5 * M[6 + j]);
            o.A[6 + j] = (float)(k8 * M[6 + j]);
        }
        o.b[0] = (float)(k0 * trel[0] + k2 * trel[2]);
        o.b[1] = (float)(k4 * trel[1] + k5 * trel[2]);
        o.b[2] = (float)(k8 * trel[2]);
        o.wf = (float)s.width;
        o.hf = (float)s.height;
        geom_maps(r, s, o.Gf, o.gf);
        geom_maps(s, r, o.Gb, o.gb);
    }
}

// ---------------------------------------------------------------------------
// Geometry helpers
// ---------------------------------------------------------------------------
// ref .cu:84-87
inline float depth_from_plane(const Camera& cam, const F4& pl, int px, int py, int lm = 0) {
    const float den = ((float)px - cam.K[2]) * pl.x + (m_div(lm, cam.K[0], cam.K[4]) * ((float)py - cam.K[5])) * pl.y + cam.K[0] * pl.z;
    return m_div(lm, -pl.w * cam.K[0], den);
}
// ref .cu:163-176
inline float plane_offset(const Camera& cam, int px, int py, float depth, const F4& n, int lm = 0) {
    const float X0 = m_div(lm, depth * ((float)px - cam.K[2]), cam.K[0]);
    const float X1 = m_div(lm, depth * ((float)py - cam.K[5]), cam.K[4]);
    const float X2 = depth;
    return -((n.x * X0 + n.y * X1) + n.z * X2);
}
// ref .cu:179-186
inline void view_dir(const Camera& cam, int px, int py, float v[3], int lm = 0) {
    v[0] = m_div(lm, (float)px - cam.K[2], cam.K[0]);
    v[1] = m_div(lm, (float)py - cam.K[5], cam.K[4]);
    v[2] = 1.0f;
}
// ref .cu:188-195 (rsqrtf -> 1/sqrt, both correctly rounded)
inline void normalize3(F4& n) {
    const float ns = (n.x * n.x + n.y * n.y) + n.z * n.z;
    const float inv = 1.0f / sqrtf(ns);
    n.x *= inv;
    n.y *= inv;
    n.z *= inv;
}
// ref .cu:197-219
inline F4 random_normal(const Camera& cam, int px, int py, Rng& g, int lm = 0) {
    float q1, q2, s;
    do {
        q1 = 2.0f * rng_uniform(g) - 1.0f;
        q2 = 2.0f * rng_uniform(g) - 1.0f;
        s = q1 * q1 + q2 * q2;
    } while (s >= 1.0f);
    const float sq = sqrtf(1.0f - s);
    F4 n;
    n.x = (2.0f * q1) * sq;
    n.y = (2.0f * q2) * sq;
    n.z = 1.0f - 2.0f * s;
    n.w = 0.0f;
    float vd[3];
    view_dir(cam, px, py, vd, lm);
    const float dp = (n.x * vd[0] + n.y * vd[1]) + n.z * vd[2];
    if (dp > 0.0f) {
        n.x = -n.x;
        n.y = -n.y;
        n.z = -n.z;
    }
    normalize3(n);
    return n;
}
// ref .cu:460-495
inline F4 perturbed_normal(const Camera& cam, int px, int py, const F4& normal, Rng& g, float perturbation, int lm = 0) {
    float vd[3];
    view_dir(cam, px, py, vd, lm);
    const float a1 = (rng_uniform(g) - 0.5f) * perturbation;
    const float a2 = (rng_uniform(g) - 0.5f) * perturbation;
    const float a3 = (rng_uniform(g) - 0.5f) * perturbation;
    const float s1 = m_sin(lm, a1), s2 = m_sin(lm, a2), s3 = m_sin(lm, a3);
    const float c1 = m_cos(lm, a1), c2 = m_cos(lm, a2), c3 = m_cos(lm, a3);
    float R[9];
    R[0] = c2 * c3;
    R[1] = (c3 * s1) * s2 - c1 * s3;
    R[2] = s1 * s3 + (c1 * c3) * s2;
    R[3] = c2 * s3;
    R[4] = c1 * c3 + (s1 * s2) * s3;
    R[5] = (c1 * s2) * s3 - c3 * s1;
    R[6] = -s2;
    R[7] = c2 * s1;
    R[8] = c1 * c2;
    F4 np;
    np.x = (R[0] * normal.x + R[1] * normal.y) + R[2] * normal.z;
    np.y = (R[3] * normal.x + R[4] * normal.y) + R[5] * normal.z;
    np.z = (R[6] * normal.x + R[7] * normal.y) + R[8] * normal.z;
    np.w = normal.w;
    const float dp = (np.x * vd[0] + np.y * vd[1]) + np.z * vd[2];
    if (dp >= 0.0f) return normal;
    normalize3(np);
    return np;
}

// ---------------------------------------------------------------------------
// Reference-window statistics: the 36 bilateral weights and the weighted
// reference moments depend only on the reference image, so they are computed
// once per pixel and scale (ref .cu:318-323, :363-395 reference-image terms).
// ---------------------------------------------------------------------------
// Measurement hook: with Ctx::literal_mode != 0 every NCC evaluation of the schedule is computed by literal_ncc
// (below) instead of the canonical formulation, to compare end-to-end statistics (tests/test_oracle_cpu.py).  The
// literal form needs the plane, scale and parameters, which the canonical call sites pass on through m / RefWin:
// plane_to_m and ref_window (always called right before the evaluations they serve) leave them here.
static thread_local F4 tl_plane;
static thread_local int tl_scale = 0;
static thread_local const Params* tl_prm = nullptr;
float literal_ncc(const Ctx& c, const Params& prm, int px, int py, const F4& pl, int v, int scale, int lm);

struct RefWin {
    float w[36];
    float wr[36];
    float inv_w;
    float mean_r;
    float var_r;
    int dx[6];  // tap offsets along x (outer loop of the reference) == along y
};

inline void ref_window(const Ctx& c, const Params& prm, int px, int py, int scale, RefWin& rw) {
    tl_scale = scale;
    tl_prm = &prm;
    const Image& ref = c.imgs[0];
    const int step = 2 << scale;          // ref .cu:342-345
    const int radius = 5 * step / 2;      // ref .cu:346
    const float two_ss = (2.0f * prm.sigma_spatial) * prm.sigma_spatial;
    const float two_sc = (2.0f * prm.sigma_color) * prm.sigma_color;
    const float rc = ref.at(px, py);
    float sw = 0.0f, swr = 0.0f, swrr = 0.0f;
    for (int a = 0; a < 6; ++a) rw.dx[a] = -radius + a * step;
    for (int a = 0; a < 6; ++a) {
        float pw = 0.0f, pwr = 0.0f, pwrr = 0.0f;
        for (int b = 0; b < 6; ++b) {
            const int dx = rw.dx[a], dy = rw.dx[b];
            const float r = ref.at(px + dx, py + dy);
            const float sd = sqrtf((float)dx * (float)dx + (float)dy * (float)dy);
            const float e = (-sd) / two_ss - fabsf(r - rc) / two_sc;
            const float w = det_exp(e);
            const float wr = w * r;
            rw.w[a * 6 + b] = w;
            rw.wr[a * 6 + b] = wr;
            pw += w;
            pwr += wr;
            pwrr = fmaf(wr, r, pwrr);
        }
        sw += pw;
        swr += pwr;
        swrr += pwrr;
    }
    rw.inv_w = 1.0f / sw;
    rw.mean_r = swr * rw.inv_w;
    const float mrr = swrr * rw.inv_w;
    rw.var_r = fmaf(-rw.mean_r, rw.mean_r, mrr);
}

// software bilinear, texel centres at integer coordinates, clamp addressing
// (CUDA tex2D(t, x+0.5, y+0.5) with linear filtering; ref .cu:377, SURVEY a-2)
inline float bilinear(const Image& im, float sx, float sy, bool q8 = false) {
    const float wm1 = (float)(im.w - 1), hm1 = (float)(im.h - 1);
    float cx = (sx >= -1.0f) ? sx : -1.0f;
    cx = (cx <= wm1) ? cx : wm1;
    float cy = (sy >= -1.0f) ? sy : -1.0f;
    cy = (cy <= hm1) ? cy : hm1;
    const float fx = floorf(cx), fy = floorf(cy);
    float ax = det_fract(cx), ay = det_fract(cy);
    if (q8) {  // CUDA's texture unit: 8 bits of fraction (pm_device.hpp tex_fraction; the product by 256 is exact, one rounding in the sum)
        ax = floorf(ax * 256.0f + 0.5f) * 0.00390625f;
        ay = floorf(ay * 256.0f + 0.5f) * 0.00390625f;
    }
    const int ix = (int)fx, iy = (int)fy;
    const float t00 = im.at(ix, iy), t10 = im.at(ix + 1, iy);
    const float t01 = im.at(ix, iy + 1), t11 = im.at(ix + 1, iy + 1);
    // the bilinear polynomial t00 + ax dx + ay (dy + ax dxy) in three fmas
    const float dx = t10 - t00, dy = t01 - t00, dxy = (t11 - t01) - dx;
    const float top = fmaf(ax, dx, t00);
    const float ver = fmaf(ax, dxy, dy);
    return fmaf(ay, ver, top);
}

// plane -> m = (n^T K_r^-1) / d   (per hypothesis, shared by all views)
inline void plane_to_m(const Ctx& c, const F4& pl, float m[3]) {
    tl_plane = pl;
    const float inv_d = 1.0f / pl.w;
    m[0] = (pl.x * c.ifx) * inv_d;
    m[1] = (pl.y * c.ify) * inv_d;
    m[2] = fmaf(-pl.y, c.cyfy, fmaf(-pl.x, c.cxfx, pl.z)) * inv_d;
}

// ref .cu:325-414 ComputeBilateralNCC for one (plane, source view)
inline float ncc_cost(const Ctx& c, const RefWin& rw, int px, int py, const float m[3], int v /*0-based source*/) {
    if (c.literal_mode) return literal_ncc(c, *tl_prm, px, py, tl_plane, v, tl_scale, c.literal_mode);
    const ViewConst& vc = c.vc[v];
    const Image& src = c.imgs[v + 1];
    float Hm[9];
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) Hm[r * 3 + k] = fmaf(-vc.b[r], m[k], vc.A[r * 3 + k]);
    const float fpx = (float)px, fpy = (float)py;
    {
        const float X = fmaf(Hm[1], fpy, fmaf(Hm[0], fpx, Hm[2]));
        const float Y = fmaf(Hm[4], fpy, fmaf(Hm[3], fpx, Hm[5]));
        const float Z = fmaf(Hm[7], fpy, fmaf(Hm[6], fpx, Hm[8]));
        const float rz = det_rcp(Z);
        const float cx = X * rz, cy = Y * rz;
        if (!(cx >= 0.0f && cx < vc.wf && cy >= 0.0f && cy < vc.hf)) return 2.0f;  // ref .cu:351-353
    }
    float racc = 0.0f;
    // the even taps (b = 0, 2, 4) and the odd taps of all six columns accumulate separately (the GPU keeps them in the two
    // halves of packed fp32 registers) and are added once, at the end of the window
    float E1 = 0.0f, E2 = 0.0f, E3 = 0.0f, O1 = 0.0f, O2 = 0.0f, O3 = 0.0f;
    for (int a = 0; a < 6; ++a) {
        const float tx = (float)(px + rw.dx[a]);
        const float Cx = fmaf(Hm[0], tx, Hm[2]);
        const float Cy = fmaf(Hm[3], tx, Hm[5]);
        const float Cz = fmaf(Hm[6], tx, Hm[8]);
        float X[6], Y[6], Z[6], I[6];
        for (int b = 0; b < 6; ++b) {
            const float ty = (float)(py + rw.dx[b]);
            X[b] = fmaf(Hm[1], ty, Cx);
            Y[b] = fmaf(Hm[4], ty, Cy);
            Z[b] = fmaf(Hm[7], ty, Cz);
        }
        // the six perspective divides of a window column share ONE reciprocal
        // (DESIGN.md 3.3): 1/Z_i = rcp(prod Z) * prod_{j != i} Z_j, via pair products
        {
            const float q0 = Z[0] * Z[1], q1 = Z[2] * Z[3], q2 = Z[4] * Z[5];
            const float t = q0 * q1, u = q1 * q2, v = q0 * q2;
            // det_rcp maps an infinite / NaN / zero / denormal product to NaN: the sum of the six
            // column reciprocals is finite exactly when every column had a usable one (checked after the loop)
            const float r = det_rcp(t * q2);
            racc += r;
            const float iq0 = r * u, iq1 = r * v, iq2 = r * t;
            I[0] = iq0 * Z[1];
            I[1] = iq0 * Z[0];
            I[2] = iq1 * Z[3];
            I[3] = iq1 * Z[2];
            I[4] = iq2 * Z[5];
            I[5] = iq2 * Z[4];
        }
        for (int b = 0; b < 6; ++b) {
            const float sv = bilinear(src, X[b] * I[b], Y[b] * I[b], c.tex_q8);
            const float w = rw.w[a * 6 + b];
            const float ws = w * sv;
            if ((b & 1) == 0) {
                E1 = fmaf(w, sv, E1);
                E2 = fmaf(ws, sv, E2);
                E3 = fmaf(rw.wr[a * 6 + b], sv, E3);
            } else {
                O1 = fmaf(w, sv, O1);
                O2 = fmaf(ws, sv, O2);
                O3 = fmaf(rw.wr[a * 6 + b], sv, O3);
            }
        }
    }
    const float T1 = E1 + O1, T2 = E2 + O2, T3 = E3 + O3;
    // near-degenerate planes (the reference divides per tap and gets garbage coordinates there): sentinel cost
    if (!std::isfinite(racc)) return 2.0f;
    const float ms = T1 * rw.inv_w, mss = T2 * rw.inv_w, mrs = T3 * rw.inv_w;
    const float var_s = fmaf(-ms, ms, mss);
    if (rw.var_r < 1e-5f || var_s < 1e-5f) return 2.0f;  // ref .cu:406-408
    const float cov = fmaf(-rw.mean_r, ms, mrs);
    const float den = sqrtf(rw.var_r * var_s);
    float cost = 1.0f - cov / den;
    cost = (cost < 2.0f) ? cost : 2.0f;  // ref .cu:412 (NaN -> 2)
    cost = (cost > 0.0f) ? cost : 0.0f;
    return cost;
}

// ---------------------------------------------------------------------------
// Geometric consistency (ref .cu:582-640)
// ---------------------------------------------------------------------------
inline void backproject(const Camera& cam, float x, float y, float depth, float P[3], int lm = 0) {
    const float X0 = m_div(lm, depth * (x - cam.K[2]), cam.K[0]);
    const float X1 = m_div(lm, depth * (y - cam.K[5]), cam.K[4]);
    const float X2 = depth;
    const float t0 = (cam.R[0] * X0 + cam.R[3] * X1) + cam.R[6] * X2;
    const float t1 = (cam.R[1] * X0 + cam.R[4] * X1) + cam.R[7] * X2;
    const float t2 = (cam.R[2] * X0 + cam.R[5] * X1) + cam.R[8] * X2;
    P[0] = t0 + cam.C[0];
    P[1] = t1 + cam.C[1];
    P[2] = t2 + cam.C[2];
}
inline void project(const Camera& cam, const float P[3], float& u, float& v, int lm = 0) {
    const float t0 = ((cam.R[0] * P[0] + cam.R[1] * P[1]) + cam.R[2] * P[2]) + cam.t[0];
    const float t1 = ((cam.R[3] * P[0] + cam.R[4] * P[1]) + cam.R[5] * P[2]) + cam.t[1];
    const float t2 = ((cam.R[6] * P[0] + cam.R[7] * P[1]) + cam.R[8] * P[2]) + cam.t[2];
    const float d = (cam.K[6] * t0 + cam.K[7] * t1) + cam.K[8] * t2;
    u = m_div(lm, (cam.K[0] * t0 + cam.K[1] * t1) + cam.K[2] * t2, d);
    v = m_div(lm, (cam.K[3] * t0 + cam.K[4] * t1) + cam.K[5] * t2, d);
}
// the reference's own chain through world coordinates (ref .cu:617-640), IEEE operations in the reference's order: the LITERAL
// form, used when Ctx::literal_mode != 0 and by the probe that measures the distance of the canonical form below from it
inline float geom_cost_literal(const Ctx& c, int v /*0-based source*/, const F4& pl, int px, int py) {
    const Camera& rc = c.cams[0];
    const Camera& sc = c.cams[v + 1];
    const Image& dm = c.depths[v];
    const int lm = c.literal_mode >= 3 ? 3 : 1;  // the probe (orc_eval_geom_literal) calls it in mode 0 as well: IEEE then
    const float depth = depth_from_plane(rc, pl, px, py, lm);
    float Pw[3];
    backproject(rc, (float)px, (float)py, depth, Pw, lm);
    float su, sv;
    project(sc, Pw, su, sv, lm);
    // nearest texel, truncation toward zero, clamp addressing (ref .cu:626)
    float qx = (su >= 0.0f) ? su : 0.0f;
    qx = (qx <= (float)(dm.w - 1)) ? qx : (float)(dm.w - 1);
    float qy = (sv >= 0.0f) ? sv : 0.0f;
    qy = (qy <= (float)(dm.h - 1)) ? qy : (float)(dm.h - 1);
    const float sd = dm.px[(size_t)(int)qy * dm.w + (int)qx];
    if (sd == 0.0f) return 3.0f;
    float Ps[3];
    backproject(sc, su, sv, sd, Ps, lm);
    float bu, bv;
    project(rc, Ps, bu, bv, lm);
    const float dc = (float)px - bu, dr = (float)py - bv;
    const float e = sqrtf(dc * dc + dr * dr);
    return (e < 3.0f) ? e : 3.0f;
}
// Canonical form: each direction is one projective map with host-composed constants, two shared reciprocals,
// no division; what the HIP kernel computes (pm_device.hpp geom_cost_view_body), bit for bit.
inline float geom_cost(const Ctx& c, int v /*0-based source*/, const F4& pl, int px, int py) {
    if (c.literal_mode) return geom_cost_literal(c, v, pl, px, py);
    const ViewConst& vc = c.vc[v];
    const Image& dm = c.depths[v];
    const float z = depth_from_plane(c.cams[0], pl, px, py);
    const float fx = (float)px, fy = (float)py;
    const float q0 = fmaf(vc.Gf[1], fy, fmaf(vc.Gf[0], fx, vc.Gf[2]));
    const float q1 = fmaf(vc.Gf[4], fy, fmaf(vc.Gf[3], fx, vc.Gf[5]));
    const float q2 = fmaf(vc.Gf[7], fy, fmaf(vc.Gf[6], fx, vc.Gf[8]));
    const float h0 = fmaf(z, q0, vc.gf[0]), h1 = fmaf(z, q1, vc.gf[1]), h2 = fmaf(z, q2, vc.gf[2]);
    const float rh = det_rcp(h2);
    const float su = h0 * rh, sv = h1 * rh;
    float qx = (su >= 0.0f) ? su : 0.0f;
    qx = (qx <= (float)(dm.w - 1)) ? qx : (float)(dm.w - 1);
    float qy = (sv >= 0.0f) ? sv : 0.0f;
    qy = (qy <= (float)(dm.h - 1)) ? qy : (float)(dm.h - 1);
    const float sd = dm.px[(size_t)(int)qy * dm.w + (int)qx];
    if (sd == 0.0f) return 3.0f;
    const float p0 = fmaf(vc.Gb[1], sv, fmaf(vc.Gb[0], su, vc.Gb[2]));
    const float p1 = fmaf(vc.Gb[4], sv, fmaf(vc.Gb[3], su, vc.Gb[5]));
    const float p2 = fmaf(vc.Gb[7], sv, fmaf(vc.Gb[6], su, vc.Gb[8]));
    const float k0 = fmaf(sd, p0, vc.gb[0]), k1 = fmaf(sd, p1, vc.gb[1]), k2 = fmaf(sd, p2, vc.gb[2]);
    const float rk = det_rcp(k2);
    const float dc = fx - k0 * rk, dr = fy - k1 * rk;
    const float e = sqrtf(fmaf(dr, dr, dc * dc));
    return (e < 3.0f) ? e : 3.0f;
}

// ---------------------------------------------------------------------------
// Initial multi-view cost + selected views (ref .cu:497-534)
// ---------------------------------------------------------------------------
inline float initial_cost(const Ctx& c, const Params& prm, const RefWin& rw, int px, int py, const F4& pl, uint32_t& sel) {
    const int V = prm.num_images - 1;
    float cv[kMaxViews], sorted[kMaxViews];
    float m[3];
    plane_to_m(c, pl, m);
    int valid = 0;
    for (int v = 0; v < V; ++v) {
        cv[v] = sorted[v] = ncc_cost(c, rw, px, py, m, v);
        if (cv[v] < 2.0f) valid++;
    }
    for (int i = 1; i < V; ++i) {  // insertion sort, ref .cu:14-23
        const float tmp = sorted[i];
        int j = i;
        for (; j >= 1 && tmp < sorted[j - 1]; --j) sorted[j] = sorted[j - 1];
        sorted[j] = tmp;
    }
    sel = 0;
    const int top_k = valid < prm.top_k ? valid : prm.top_k;
    if (top_k > 0) {
        float cost = 0.0f;
        for (int i = 0; i < top_k; ++i) cost += sorted[i];
        const float thr = sorted[top_k - 1];
        for (int v = 0; v < V; ++v)
            if (cv[v] <= thr) sel |= (1u << v);
        return m_div(c.literal_mode, cost, (float)top_k);
    }
    return 2.0f;
}

// ref .cu:536-573
void init_pixel(Ctx& c, const Params& prm, uint64_t seed, uint32_t launch, int scale, int px, int py) {
    const int idx = py * c.W + px;
    const Camera& cam = c.cams[0];
    const int lm = c.literal_mode;  // 0: canonical arithmetic; 1-3: measurement modes (m_exp, m_div ...)
    Rng g = rng_make(seed, (uint32_t)idx, launch);
    RefWin rw;
    ref_window(c, prm, px, py, scale, rw);
    F4 pl;
    if (!prm.geom_consistency && !prm.planar_prior) {
        pl = random_normal(cam, px, py, g, lm);
        const float depth = rng_uniform(g) * (prm.depth_max - prm.depth_min) + prm.depth_min;
        pl.w = plane_offset(cam, px, py, depth, pl, lm);
    } else if (prm.planar_prior && c.mask[idx] > 0 && c.costs[idx] >= 0.1f) {
        const float perturbation = 0.02f;
        const F4 pp = c.prior[idx];
        float dpert = pp.w;
        const float dmin_p = (1.0f - 3.0f * perturbation) * dpert;
        const float dmax_p = (1.0f + 3.0f * perturbation) * dpert;
        dpert = rng_uniform(g) * (dmax_p - dmin_p) + dmin_p;
        pl = perturbed_normal(cam, px, py, pp, g, 0.18849556f /* 3*0.02*pi */, lm);
        pl.w = dpert;
    } else {
        const F4 st = c.planes[idx];  // (world normal, depth) from the previous run
        pl.x = (cam.R[0] * st.x + cam.R[1] * st.y) + cam.R[2] * st.z;
        pl.y = (cam.R[3] * st.x + cam.R[4] * st.y) + cam.R[5] * st.z;
        pl.z = (cam.R[6] * st.x + cam.R[7] * st.y) + cam.R[8] * st.z;
        pl.w = plane_offset(cam, px, py, st.w, pl, lm);
    }
    c.planes[idx] = pl;
    uint32_t sel;
    c.costs[idx] = initial_cost(c, prm, rw, px, py, pl, sel);
    c.sel[idx] = sel;
}

// ---------------------------------------------------------------------------
// Checkerboard propagation + refinement (ref .cu:724-998, :642-722)
// ---------------------------------------------------------------------------
// sampling regions, ref .cu:769-779
static const int8_t kDirs[8][12][2] = {
    {{-5, -6}, {5, -6}, {-6, -7}, {6, -7}, {-7, -8}, {7, -8}, {-8, -9}, {8, -9}, {-9, -10}, {9, -10}, {-10, -11}, {10, -11}},
    {{-5, 6}, {5, 6}, {-6, 7}, {6, 7}, {-7, 8}, {7, 8}, {-8, 9}, {8, 9}, {-9, 10}, {9, 10}, {-10, 11}, {10, 11}},
    {{-6, -5}, {-6, 5}, {-7, -6}, {-7, 6}, {-8, -7}, {-8, 7}, {-9, -8}, {-9, 8}, {-10, -9}, {-10, 9}, {-11, -10}, {-11, 10}},
    {{6, -5}, {6, 5}, {7, -6}, {7, 6}, {8, -7}, {8, 7}, {9, -8}, {9, 8}, {10, -9}, {10, 9}, {11, -10}, {11, 10}},
    {{0, -5}, {0, -7}, {0, -9}, {0, -11}, {0, -13}, {0, -15}, {0, -17}, {0, -19}, {0, -21}, {0, -23}, {0, 0}, {0, 0}},
    {{0, 5}, {0, 7}, {0, 9}, {0, 11}, {0, 13}, {0, 15}, {0, 17}, {0, 19}, {0, 21}, {0, 23}, {0, 0}, {0, 0}},
    {{-5, 0}, {-7, 0}, {-9, 0}, {-11, 0}, {-13, 0}, {-15, 0}, {-17, 0}, {-19, 0}, {-21, 0}, {-23, 0}, {0, 0}, {0, 0}},
    {{5, 0}, {7, 0}, {9, 0}, {11, 0}, {13, 0}, {15, 0}, {17, 0}, {19, 0}, {21, 0}, {23, 0}, {0, 0}, {0, 0}}};
static const int kNumDirs[8] = {12, 12, 12, 12, 10, 10, 10, 10};

inline float prior_term(int lm, float depth_diff, float angle_cos, float two_ds2, float two_as2) {
    const float ad = m_acos(lm, angle_cos);
    return 0.5f + m_exp(lm, m_div(lm, -depth_diff * depth_diff, two_ds2)) * m_exp(lm, m_div(lm, -ad * ad, two_as2));
}

// Statistics hook for sizing kernel optimisations (tests/analysis/prune_stats.py): when set, every update_pixel records, per
// refinement candidate, after how many views (ascending view index) its running weighted sum has provably lost against the
// cost the refinement started from -- i.e. which evaluations of the refinement nothing reads.  The oracle itself always
// evaluates everything (ref .cu:681).  g_stat_death[idx * 5 + i]: -1 = depth out of range (dead from the start), v = the
// candidate is decided after the evaluation of view v, V = never; g_stat_wmask[idx]: views with weight > 0.
static int8_t* g_stat_death = nullptr;
static uint32_t* g_stat_wmask = nullptr;
// phase A: g_stat_bad3[idx * 32 + v] = the candidate slot (0..7) at which view v's count of costs > 1.2 reaches 3 -- from then
// on the view's sampling probability is 0 whatever the remaining candidates cost (ref .cu:847-853) -- or 8; [idx * 32 + 31] =
// the bitmask of flagged (existing) candidates
static uint8_t* g_stat_bad3 = nullptr;

void update_pixel(Ctx& c, const Params& prm, uint64_t seed, uint32_t launch, int iter, int scale, int px, int py) {
    const int W = c.W, Hh = c.H;
    const int idx = py * W + px;
    const int V = prm.num_images - 1;
    const Camera& cam = c.cams[0];
    const bool geom = prm.geom_consistency, prior = prm.planar_prior;
    const int lm = c.literal_mode;  // 0: canonical arithmetic; 1-3: measurement modes (m_exp, m_div ...)
    Rng g = rng_make(seed, (uint32_t)idx, launch);
    RefWin rw;
    ref_window(c, prm, px, py, scale, rw);

    // -- 8 sampling regions: lowest stored cost per region (ref .cu:798-819)
    bool flag[8];
    int pos[8];
    F4 cand[8];
    float cost_arr[8][kMaxViews];
    for (int k = 0; k < 8; ++k) {
        float best = FLT_MAX;
        int bpos = -1;
        for (int d = 0; d < kNumDirs[k]; ++d) {
            const int nx = px + kDirs[k][d][0], ny = py + kDirs[k][d][1];
            if (!(nx >= 0 && ny >= 0 && nx < W && ny < Hh)) continue;
            const int nidx = ny * W + nx;
            const float nc = c.costs[nidx];
            if (best > nc) {
                best = nc;
                bpos = nidx;
            }
        }
        flag[k] = best < FLT_MAX;
        pos[k] = bpos;
        if (flag[k]) {
            cand[k] = c.planes[bpos];
            float m[3];
            plane_to_m(c, cand[k], m);
            for (int v = 0; v < V; ++v) cost_arr[k][v] = ncc_cost(c, rw, px, py, m, v);
        } else {
            // ref .cu:795: `cost_array[8][32] = {2.0f}` sets only [0][0]; the
            // rest is zero and still enters the good/bad counts (quirk a-9 i)
            for (int v = 0; v < V; ++v) cost_arr[k][v] = (k == 0 && v == 0) ? 2.0f : 0.0f;
            cand[k] = F4{0, 0, 0, 0};
        }
    }

    if (g_stat_bad3) {
        for (int v = 0; v < V; ++v) {
            int nb = 0, at = 8;
            for (int j = 0; j < 8; ++j)
                if (cost_arr[j][v] > 1.2f && ++nb == 3 && at == 8) at = j;
            g_stat_bad3[(size_t)idx * 32 + v] = (uint8_t)at;
        }
        uint8_t fm = 0;
        for (int j = 0; j < 8; ++j) fm |= flag[j] ? (1u << j) : 0;
        g_stat_bad3[(size_t)idx * 32 + 31] = fm;
    }
    // -- view weights (ref .cu:821-867)
    float view_w[kMaxViews];
    {
        float vprior[kMaxViews];
        for (int v = 0; v < V; ++v) vprior[v] = 0.0f;
        const int nbr[4] = {idx - W, idx + W, idx - 1, idx + 1};
        for (int i = 0; i < 4; ++i)
            if (flag[i]) {
                const uint32_t s = c.sel[nbr[i]];
                for (int v = 0; v < V; ++v) vprior[v] += ((s >> v) & 1u) ? 0.9f : 0.1f;
            }
        float probs[kMaxViews];
        const float thr = (float)(0.8 * (double)m_exp(lm, m_div(lm, (float)(iter * iter), -90.0f)));  // ref .cu:832: the product is formed in double
        for (int v = 0; v < V; ++v) {
            float count = 0.0f, tmpw = 0.0f;
            int count_false = 0;
            for (int j = 0; j < 8; ++j) {
                const float cj = cost_arr[j][v];
                if (cj < thr) {
                    tmpw += m_exp(lm, m_div(lm, cj * cj, -0.18f));
                    count += 1.0f;
                }
                if (cj > 1.2f) count_false++;
            }
            if (count > 2.0f && count_false < 3)
                probs[v] = m_div(lm, vprior[v] * tmpw, count);
            else if (count_false < 3)
                probs[v] = vprior[v] * m_exp(lm, m_div(lm, thr * thr, -0.32f));
            else
                probs[v] = 0.0f;
        }
        // ref .cu:42-56 (0 * inf = NaN when all probabilities vanish, quirk a-9 v)
        float psum = 0.0f;
        for (int v = 0; v < V; ++v) psum += probs[v];
        const float inv = m_div(lm, 1.0f, psum);
        float cum = 0.0f;
        for (int v = 0; v < V; ++v) {
            cum += probs[v] * inv;
            probs[v] = cum;
        }
        probs[V - 1] = 1.0f;
        // all 32 entries are zeroed as in ref .cu:821: the refinement's geometric
        // term reads view_w[candidate 0..4], beyond V when V < 5 (quirk a-10 iii)
        for (int v = 0; v < kMaxViews; ++v) view_w[v] = 0.0f;
        for (int s = 0; s < 15; ++s) {
            const float rp = rng_uniform(g) - FLT_EPSILON;
            for (int v = 0; v < V; ++v)
                if (probs[v] > rp) {
                    view_w[v] += 1.0f;
                    break;
                }
        }
    }
    uint32_t temp_sel = 0;
    float weight_norm = 0.0f;
    for (int v = 0; v < V; ++v)
        if (view_w[v] > 0.0f) {
            temp_sel |= (1u << v);
            weight_norm += view_w[v];
        }

    // -- weighted candidate costs (ref .cu:880-899)
    float final_costs[8];
    for (int i = 0; i < 8; ++i) {
        float fc = 0.0f;
        for (int v = 0; v < V; ++v) {
            if (view_w[v] > 0.0f) {
                if (geom) {
                    if (flag[i])
                        fc += view_w[v] * (cost_arr[i][v] + 0.2f * geom_cost(c, v, cand[i], px, py));
                    else
                        fc += view_w[v] * (cost_arr[i][v] + 0.1f * 3.0f);
                } else {
                    fc += view_w[v] * cost_arr[i][v];
                }
            }
        }
        final_costs[i] = m_div(lm, fc, weight_norm);
    }
    int min_idx = 0;
    {
        float mc = final_costs[0];
        for (int i = 1; i < 8; ++i)
            if (final_costs[i] <= mc) {
                mc = final_costs[i];
                min_idx = i;
            }
    }

    // -- current plane under the new weights (ref .cu:900-921)
    const F4 cur = c.planes[idx];
    float cost_now = 0.0f, geom_now = 0.0f;
    {
        float m[3];
        plane_to_m(c, cur, m);
        for (int v = 0; v < V; ++v) {
            const float cv = ncc_cost(c, rw, px, py, m, v);
            if (geom) {
                const float gt = 0.2f * geom_cost(c, v, cur, px, py);
                cost_now += view_w[v] * (cv + gt);
                geom_now += view_w[v] * gt;
            } else {
                cost_now += view_w[v] * cv;
            }
        }
    }
    cost_now = m_div(lm, cost_now, weight_norm);
    if (geom) {
        geom_now = m_div(lm, geom_now, weight_norm);
        c.geom[idx] = geom_now;
    }
    c.costs[idx] = cost_now;
    float depth_now = depth_from_plane(cam, cur, px, py, lm);
    float restricted_cost = 0.0f;
    F4 plane_now = cur;

    const float depth_sigma = m_div(lm, prm.depth_max - prm.depth_min, 64.0f);
    const float two_ds2 = (2.0f * depth_sigma) * depth_sigma;
    const float angle_sigma = 0.08726646f;  // pi * 5/180
    const float two_as2 = (2.0f * angle_sigma) * angle_sigma;
    const float beta = 0.18f;

    // -- planar-prior assisted acceptance (ref .cu:924-978)
    if (prior && !geom) {
        const F4 pp = c.prior[idx];
        const float depth_prior = depth_from_plane(cam, pp, px, py, lm);
        if (c.mask[idx] > 0) {
            float rfc[8];
            for (int i = 0; i < 8; ++i) {
                rfc[i] = 0.0f;
                if (flag[i]) {
                    const float di = depth_from_plane(cam, cand[i], px, py, lm);
                    const float ac = (pp.x * cand[i].x + pp.y * cand[i].y) + pp.z * cand[i].z;
                    const float pr = prior_term(lm, di - depth_prior, ac, two_ds2, two_as2);
                    rfc[i] = m_exp(lm, m_div(lm, -final_costs[i] * final_costs[i], beta)) * pr;
                }
            }
            int max_idx = 0;
            {
                float mc = rfc[0];
                for (int i = 1; i < 8; ++i)
                    if (rfc[i] >= mc) {
                        mc = rfc[i];
                        max_idx = i;
                    }
            }
            const float ac = (pp.x * cur.x + pp.y * cur.y) + pp.z * cur.z;
            const float pr = prior_term(lm, depth_now - depth_prior, ac, two_ds2, two_as2);
            const float rc_now = m_exp(lm, m_div(lm, -cost_now * cost_now, beta)) * pr;
            if (flag[max_idx]) {
                const float db = depth_from_plane(cam, cand[max_idx], px, py, lm);
                if (db >= prm.depth_min && db <= prm.depth_max && rfc[max_idx] > rc_now) {
                    // ref .cu:950 re-declares depth_now inside this block, so the
                    // assignment at :961 hits the shadow: the depth handed to the
                    // refinement stays that of the OLD plane (quirk)
                    plane_now = cand[max_idx];
                    c.costs[idx] = final_costs[max_idx];  // cost_now deliberately NOT updated (quirk a-9 iv)
                    restricted_cost = rfc[max_idx];
                    c.sel[idx] = temp_sel;
                }
            }
        } else if (flag[min_idx]) {
            const float db = depth_from_plane(cam, cand[min_idx], px, py, lm);
            if (db >= prm.depth_min && db <= prm.depth_max && final_costs[min_idx] < cost_now) {
                depth_now = db;
                plane_now = cand[min_idx];
                c.costs[idx] = final_costs[min_idx];
            }
        }
    }
    // -- plain acceptance (ref .cu:981-991)
    if (!prior && flag[min_idx]) {
        const float db = depth_from_plane(cam, cand[min_idx], px, py, lm);
        if (db >= prm.depth_min && db <= prm.depth_max && final_costs[min_idx] < cost_now) {
            depth_now = db;
            plane_now = cand[min_idx];
            cost_now = final_costs[min_idx];
            c.sel[idx] = temp_sel;
        }
    }

    // -- refinement (ref .cu:642-722)
    {
        const float perturbation = 0.02f;
        const bool masked = prior && c.mask[idx] > 0;
        F4 pp = F4{0, 0, 0, 0};
        float depth_prior = 0.0f;
        float depth_rand;
        F4 n_rand;
        if (masked) {
            pp = c.prior[idx];
            depth_prior = depth_from_plane(cam, pp, px, py, lm);
            // ref .cu:658-659: drawn, then always overwritten (missing else, quirk a-10 i)
            depth_rand = (rng_uniform(g) * 6.0f) * depth_sigma + (depth_prior - 3.0f * depth_sigma);
            n_rand = perturbed_normal(cam, px, py, pp, g, angle_sigma, lm);
        }
        depth_rand = rng_uniform(g) * (prm.depth_max - prm.depth_min) + prm.depth_min;
        n_rand = random_normal(cam, px, py, g, lm);

        const float dmin_p = (1.0f - perturbation) * depth_now;
        const float dmax_p = (1.0f + perturbation) * depth_now;
        const float depth_pert = rng_uniform(g) * (dmax_p - dmin_p) + dmin_p;  // loop never repeats (quirk a-10 ii)
        const F4 n_pert = perturbed_normal(cam, px, py, plane_now, g, 0.06283185f /* 0.02*pi */, lm);

        const float cost_now_at_refinement_start = cost_now;
        if (g_stat_wmask) {
            uint32_t m = 0;
            for (int v = 0; v < V; ++v)
                if (view_w[v] > 0.0f) m |= 1u << v;
            g_stat_wmask[idx] = m;
        }
        const float depths5[5] = {depth_rand, depth_now, depth_rand, depth_now, depth_pert};
        const F4 normals5[5] = {plane_now, n_rand, n_rand, n_pert, plane_now};
        // statistics hook, masked prior pixels with restricted_cost == 0: every in-range candidate with a valid prior term is
        // accepted in turn (the test does not depend on the cost), so only the last of them leaves a trace
        int stat_last_masked = -2;  // -2: the rule does not apply
        if (g_stat_death && masked && restricted_cost == 0.0f) {
            stat_last_masked = -1;
            for (int i = 0; i < 5; ++i) {
                F4 tp = normals5[i];
                tp.w = plane_offset(cam, px, py, depths5[i], tp, lm);
                const float dbs = depth_from_plane(cam, tp, px, py, lm);
                const float ac = (pp.x * tp.x + pp.y * tp.y) + pp.z * tp.z;
                const float pr = prior_term(lm, depths5[i] - depth_prior, ac, two_ds2, two_as2);
                if (dbs >= prm.depth_min && dbs <= prm.depth_max && pr > 0.0f) stat_last_masked = i;
            }
        }
        for (int i = 0; i < 5; ++i) {
            F4 tp = normals5[i];
            tp.w = plane_offset(cam, px, py, depths5[i], tp, lm);
            float m[3];
            plane_to_m(c, tp, m);
            float cv[kMaxViews];
            for (int v = 0; v < V; ++v) cv[v] = ncc_cost(c, rw, px, py, m, v);
            float tc = 0.0f, tg = 0.0f;
            int stat_death = V;
            if (g_stat_death && masked && restricted_cost > 0.0f) {  // before any view: the prior term alone may already be too small
                const float ac0 = (pp.x * tp.x + pp.y * tp.y) + pp.z * tp.z;
                const float pr0 = prior_term(lm, depths5[i] - depth_prior, ac0, two_ds2, two_as2);
                if (!(pr0 * 1.000001f > restricted_cost)) stat_death = -1;
            }
            const float stat_cost_start = cost_now_at_refinement_start;
            for (int v = 0; v < V; ++v) {
                if (view_w[v] > 0.0f) {
                    if (geom) {
                        const float gt = 0.2f * geom_cost(c, v, tp, px, py);
                        tc += view_w[v] * (cv[v] + gt);
                        tg += view_w[i] * gt;  // candidate index, not view index (quirk a-10 iii)
                    } else {
                        tc += view_w[v] * cv[v];
                    }
                    if (stat_death == V && !masked && m_div(lm, tc, weight_norm) >= stat_cost_start) stat_death = v;
                }
            }
            if (g_stat_death) {
                const float dbs = depth_from_plane(cam, tp, px, py, lm);
                int8_t dv = (int8_t)((dbs >= prm.depth_min && dbs <= prm.depth_max) ? stat_death : -1);
                if (stat_death == -1) dv = -1;
                if (stat_last_masked != -2) dv = (int8_t)(i == stat_last_masked ? V : -1);
                g_stat_death[(size_t)idx * 5 + i] = dv;
            }
            tc = m_div(lm, tc, weight_norm);
            if (geom) tg = m_div(lm, tg, weight_norm);
            const float db = depth_from_plane(cam, tp, px, py, lm);
            if (masked) {
                const float ac = (pp.x * tp.x + pp.y * tp.y) + pp.z * tp.z;
                const float pr = prior_term(lm, depths5[i] - depth_prior, ac, two_ds2, two_as2);
                const float rtc = m_exp(lm, m_div(lm, -tc * tc, beta)) * pr;
                if (db >= prm.depth_min && db <= prm.depth_max && rtc > restricted_cost) {
                    plane_now = tp;  // restricted_cost is never raised (quirk a-10 iv)
                    cost_now = tc;
                }
            } else if (db >= prm.depth_min && db <= prm.depth_max && tc < cost_now) {
                plane_now = tp;
                cost_now = tc;
                geom_now = tg;
            }
        }
    }
    c.costs[idx] = cost_now;
    c.planes[idx] = plane_now;
    if (geom) c.geom[idx] = geom_now;
}

// ref .cu:1021-1034
void depth_normal_pixel(Ctx& c, int px, int py) {
    const int idx = py * c.W + px;
    const Camera& cam = c.cams[0];
    F4 pl = c.planes[idx];
    pl.w = depth_from_plane(cam, pl, px, py, c.literal_mode);
    F4 o;
    o.x = (cam.R[0] * pl.x + cam.R[3] * pl.y) + cam.R[6] * pl.z;
    o.y = (cam.R[1] * pl.x + cam.R[4] * pl.y) + cam.R[7] * pl.z;
    o.z = (cam.R[2] * pl.x + cam.R[5] * pl.y) + cam.R[8] * pl.z;
    o.w = pl.w;
    c.planes[idx] = o;
}

// ref .cu:1036-1150
void filter_pixel(Ctx& c, int px, int py) {
    const int W = c.W, Hh = c.H;
    const int ctr = py * W + px;
    float f[21];
    int n = 0;
    f[n++] = c.planes[ctr].w;
    if (c.costs[ctr] < 0.001f) return;
    auto D = [&](int off) { return c.planes[ctr + off].w; };
    if (py > 0) f[n++] = D(-W);
    if (py > 2) f[n++] = D(-3 * W);
    if (py > 4) f[n++] = D(-5 * W);
    if (py < Hh - 1) f[n++] = D(W);
    if (py < Hh - 3) f[n++] = D(3 * W);
    if (py < Hh - 5) f[n++] = D(5 * W);
    if (px > 0) f[n++] = D(-1);
    if (px > 2) f[n++] = D(-3);
    if (px > 4) f[n++] = D(-5);
    if (px < W - 1) f[n++] = D(1);
    if (px < W - 3) f[n++] = D(3);
    if (px < W - 5) f[n++] = D(5);
    if (py > 0 && px < W - 2) f[n++] = D(-W + 2);
    if (py < Hh - 1 && px < W - 2) f[n++] = D(W + 2);
    if (py > 0 && px > 1) f[n++] = D(-W - 2);
    if (py < Hh - 1 && px > 1) f[n++] = D(W - 2);
    if (px > 0 && py > 2) f[n++] = D(-1 - 2 * W);
    if (px < W - 1 && py > 2) f[n++] = D(1 - 2 * W);
    if (px > 0 && py < Hh - 2) f[n++] = D(-1 + 2 * W);
    if (px < W - 1 && py < Hh - 2) f[n++] = D(1 + 2 * W);
    for (int i = 1; i < n; ++i) {
        const float tmp = f[i];
        int j = i;
        for (; j >= 1 && tmp < f[j - 1]; --j) f[j] = f[j - 1];
        f[j] = tmp;
    }
    const int mid = n / 2;
    c.planes[ctr].w = (n % 2 == 0) ? (f[mid - 1] + f[mid]) / 2.0f : f[mid];
}

// ---------------------------------------------------------------------------
// Launch geometry (ref .cu:1192-1196, :1000-1019): the checkerboard grid
// covers rows y < 2*16*ceil((H/2)/16) only (quirk a-9 ii).
// ---------------------------------------------------------------------------
inline int checker_ylimit(int H) { return 2 * 16 * (((H / 2) + 15) / 16); }

enum Kind { kInit = 0, kBlack = 1, kRed = 2, kDepthNormal = 3, kFilterBlack = 4, kFilterRed = 5 };

int check_ready(Ctx& c, const Params& p) {
    if (c.n_img < 2) { c.err = "set_views not called (need >= 2 views)"; return -1; }
    if (p.num_images != c.n_img) { c.err = "params.num_images != number of views"; return -2; }
    if (p.num_images - 1 > kMaxViews) { c.err = "too many source views"; return -3; }
    if (p.geom_consistency && (int)c.depths.size() != c.n_img - 1) { c.err = "geom_consistency needs source depth maps"; return -4; }
    if (p.planar_prior && !c.have_prior) { c.err = "planar_prior needs set_prior"; return -5; }
    // never reached by the reference (ref src/PatchMatch.cpp:535 clears geom_consistency before the prior Run())
    if (p.geom_consistency && p.planar_prior) { c.err = "geom_consistency and planar_prior are mutually exclusive"; return -7; }
    return 0;
}

int step(Ctx& c, const Params& prm, uint64_t seed, int kind, int iter, int scale, uint32_t launch) {
    const int rc = check_ready(c, prm);
    if (rc) return rc;
    const int W = c.W, H = c.H;
    const int ylim = checker_ylimit(H);
    // tasks of one row segment (kTaskW pixels): with whole rows as tasks a 1200-row image keeps only a few tasks per thread
    // on a many-core host and the slowest thread decides (the pixels are independent within a launch: any partition gives
    // the same result)
    constexpr int kTaskW = 128;
    const int ntx = (W + kTaskW - 1) / kTaskW;
    if (kind == kInit || kind == kDepthNormal) {
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
        for (int y = 0; y < H; ++y)
            for (int tx = 0; tx < ntx; ++tx)
                for (int x = tx * kTaskW; x < W && x < (tx + 1) * kTaskW; ++x) {
                    if (kind == kInit)
                        init_pixel(c, prm, seed, launch, scale, x, y);
                    else
                        depth_normal_pixel(c, x, y);
                }
        return 0;
    }
    const int parity = (kind == kBlack || kind == kFilterBlack) ? 0 : 1;
    const bool upd = (kind == kBlack || kind == kRed);
    if (!upd && kind != kFilterBlack && kind != kFilterRed) { c.err = "bad kernel kind"; return -6; }
    const int rows = H < ylim ? H : ylim;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int y = 0; y < rows; ++y)
        for (int tx = 0; tx < ntx; ++tx)
            for (int x = tx * kTaskW + ((y + parity) & 1); x < W && x < (tx + 1) * kTaskW; x += 2) {  // kTaskW is even
                if (upd)
                    update_pixel(c, prm, seed, launch, iter, scale, x, y);
                else
                    filter_pixel(c, x, y);
            }
    return 0;
}

// ref .cu:1188-1254 schedule
int run(Ctx& c, const Params& prm, uint64_t seed) {
    uint32_t launch = 0;
    int rc = step(c, prm, seed, kInit, 0, prm.max_scale, launch++);
    if (rc) return rc;
    if (prm.geom_consistency || prm.planar_prior) {
        for (int i = 0; i < prm.max_iterations; ++i) {
            if ((rc = step(c, prm, seed, kBlack, i, 0, launch++))) return rc;
            if ((rc = step(c, prm, seed, kRed, i, 0, launch++))) return rc;
        }
    } else {
        for (int s = prm.max_scale; s >= 0; --s)
            for (int i = 0; i < prm.max_iterations; ++i) {
                if ((rc = step(c, prm, seed, kBlack, i, s, launch++))) return rc;
                if ((rc = step(c, prm, seed, kRed, i, s, launch++))) return rc;
            }
    }
    if ((rc = step(c, prm, seed, kDepthNormal, 0, 0, launch++))) return rc;
    if ((rc = step(c, prm, seed, kFilterBlack, 0, 0, launch++))) return rc;
    if ((rc = step(c, prm, seed, kFilterRed, 0, 0, launch++))) return rc;
    return 0;
}

// ---------------------------------------------------------------------------
// Literal-arithmetic NCC: the reference's formulas in the reference's operation
// order -- homography assembled per evaluation with its divisions (ref .cu:228-279),
// one perspective division per tap (:281-288), libm expf/sqrtf for the bilateral
// weight (:318-323), weighted sums accumulated row by row (:365-395), no hoisting
// of the reference-window terms -- and, with quantize_fraction != 0, CUDA's texture
// filtering with 8-bit interpolation fractions.  It exists only to MEASURE how far
// the canonical arithmetic of DESIGN.md section 3 is from a literal transcription
// (tests/test_oracle_cpu.py::test_canonical_vs_literal_arithmetic); nothing else uses it.
// ---------------------------------------------------------------------------
float literal_tex(const Image& im, float x, float y, int quantize) {
    // tex2D(t, x + 0.5, y + 0.5), linear filter, clamp addressing
    const float xb = x, yb = y;
    const float fx = floorf(xb), fy = floorf(yb);
    float ax = xb - fx, ay = yb - fy;
    if (quantize) {
        ax = floorf(ax * 256.0f + 0.5f) / 256.0f;
        ay = floorf(ay * 256.0f + 0.5f) / 256.0f;
    }
    const int ix = (int)fx, iy = (int)fy;
    const float t00 = im.at(ix, iy), t10 = im.at(ix + 1, iy), t01 = im.at(ix, iy + 1), t11 = im.at(ix + 1, iy + 1);
    return (1.0f - ax) * (1.0f - ay) * t00 + ax * (1.0f - ay) * t10 + (1.0f - ax) * ay * t01 + ax * ay * t11;
}

// lm: 1 IEEE + libm; 2 the same with 8-bit texture fractions; 3 the fast-math model (m_div / m_exp, fused multiply-adds where
// nvcc contracts a * b + c, 8-bit fractions); 4 = 3 without the 8-bit fractions
float literal_ncc(const Ctx& c, const Params& prm, int px, int py, const F4& pl, int v, int scale, int lm) {
    const Camera& rc = c.cams[0];
    const Camera& sc = c.cams[v + 1];
    const bool fast = lm >= 3;
    const int quantize = lm == 2 || lm == 3;
    auto mad = [&](float a, float b, float acc) { return fast ? fmaf(a, b, acc) : acc + a * b; };
    // R_rel = R_s R_r^T, t_rel = R_s (C_r - C_s)
    float Rr[9], tr[3], Cd[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rr[i * 3 + j] = mad(sc.R[i * 3 + 2], rc.R[j * 3 + 2], mad(sc.R[i * 3 + 1], rc.R[j * 3 + 1], sc.R[i * 3] * rc.R[j * 3]));
    for (int k = 0; k < 3; ++k) Cd[k] = rc.C[k] - sc.C[k];
    for (int i = 0; i < 3; ++i) tr[i] = mad(sc.R[i * 3 + 2], Cd[2], mad(sc.R[i * 3 + 1], Cd[1], sc.R[i * 3] * Cd[0]));
    float Hm[9], T[9];
    const float n3[3] = {pl.x, pl.y, pl.z};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Hm[i * 3 + j] = Rr[i * 3 + j] - m_div(lm, tr[i] * n3[j], pl.w);
    for (int i = 0; i < 3; ++i) {
        T[i * 3 + 0] = m_div(lm, Hm[i * 3 + 0], rc.K[0]);
        T[i * 3 + 1] = m_div(lm, Hm[i * 3 + 1], rc.K[4]);
        T[i * 3 + 2] = -m_div(lm, Hm[i * 3 + 0] * rc.K[2], rc.K[0]) - m_div(lm, Hm[i * 3 + 1] * rc.K[5], rc.K[4]) + Hm[i * 3 + 2];
    }
    for (int j = 0; j < 3; ++j) {
        Hm[0 + j] = mad(sc.K[2], T[6 + j], sc.K[0] * T[0 + j]);
        Hm[3 + j] = mad(sc.K[5], T[6 + j], sc.K[4] * T[3 + j]);
        Hm[6 + j] = sc.K[8] * T[6 + j];
    }
    auto warp = [&](int x, int y, float& u, float& w) {
        const float fx = (float)x, fy = (float)y;
        const float a = mad(Hm[1], fy, Hm[0] * fx) + Hm[2], b = mad(Hm[4], fy, Hm[3] * fx) + Hm[5], z = mad(Hm[7], fy, Hm[6] * fx) + Hm[8];
        u = m_div(lm, a, z);
        w = m_div(lm, b, z);
    };
    float cu, cv;
    warp(px, py, cu, cv);
    if (cu >= sc.width || cu < 0.0f || cv >= sc.height || cv < 0.0f) return 2.0f;
    int step = 2;
    for (int i = 0; i < scale; ++i) step *= 2;
    const int radius = 5 * step / 2;
    const Image& ref = c.imgs[0];
    const Image& src = c.imgs[v + 1];
    const float centre = ref.at(px, py);
    float s_r = 0, s_rr = 0, s_s = 0, s_ss = 0, s_rs = 0, s_w = 0;
    for (int i = -radius; i < radius + 1; i += step) {
        float r_r = 0, r_rr = 0, r_s = 0, r_ss = 0, r_rs = 0, r_w = 0;
        for (int j = -radius; j < radius + 1; j += step) {
            const float rp = ref.at(px + i, py + j);
            float u, w2;
            warp(px + i, py + j, u, w2);
            const float sp = literal_tex(src, u, w2, quantize);
            const float sd = sqrtf((float)i * (float)i + (float)j * (float)j);
            const float wt = m_exp(lm, -m_div(lm, sd, 2.0f * prm.sigma_spatial * prm.sigma_spatial) - m_div(lm, fabsf(rp - centre), 2.0f * prm.sigma_color * prm.sigma_color));
            r_r = mad(wt, rp, r_r);
            r_rr = mad(wt * rp, rp, r_rr);
            r_s = mad(wt, sp, r_s);
            r_ss = mad(wt * sp, sp, r_ss);
            r_rs = mad(wt * rp, sp, r_rs);
            r_w += wt;
        }
        s_r += r_r; s_rr += r_rr; s_s += r_s; s_ss += r_ss; s_rs += r_rs; s_w += r_w;
    }
    const float inv = m_div(lm, 1.0f, s_w);
    s_r *= inv; s_rr *= inv; s_s *= inv; s_ss *= inv; s_rs *= inv;
    const float var_r = fast ? fmaf(-s_r, s_r, s_rr) : s_rr - s_r * s_r, var_s = fast ? fmaf(-s_s, s_s, s_ss) : s_ss - s_s * s_s;
    if (var_r < 1e-5f || var_s < 1e-5f) return 2.0f;
    const float cov = fast ? fmaf(-s_r, s_s, s_rs) : s_rs - s_r * s_s;
    return std::fmax(0.0f, std::fmin(2.0f, 1.0f - m_div(lm, cov, sqrtf(var_r * var_s))));
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI for ctypes (mirrors include/mpmvs.h one to one, prefix orc_)
// ---------------------------------------------------------------------------
extern "C" {

struct orc_ctx {
    Ctx c;
};

orc_ctx* orc_create(void) { return new orc_ctx(); }
void orc_destroy(orc_ctx* h) { delete h; }
const char* orc_last_error(const orc_ctx* h) { return h ? h->c.err.c_str() : "null ctx"; }

int orc_set_views(orc_ctx* h, int n, const void* cams, const float* const* images, const size_t* pitch_bytes) {
    Ctx& c = h->c;
    if (n < 2 || n - 1 > kMaxViews) { c.err = "need 2..33 views"; return -1; }
    c.n_img = n;
    c.cams.assign((const Camera*)cams, (const Camera*)cams + n);
    c.imgs.assign(n, Image());
    for (int i = 0; i < n; ++i) {
        Image& im = c.imgs[i];
        im.w = c.cams[i].width;
        im.h = c.cams[i].height;
        if (im.w <= 0 || im.h <= 0) { c.err = "bad image size"; return -2; }
        im.px.resize((size_t)im.w * im.h);
        const size_t pitch = pitch_bytes ? pitch_bytes[i] : (size_t)im.w * 4;
        for (int y = 0; y < im.h; ++y) std::memcpy(&im.px[(size_t)y * im.w], (const char*)images[i] + (size_t)y * pitch, (size_t)im.w * 4);
    }
    c.W = c.cams[0].width;
    c.H = c.cams[0].height;
    const size_t wh = (size_t)c.W * c.H;
    c.planes.assign(wh, F4{0, 0, 0, 0});
    c.costs.assign(wh, 0.0f);
    c.geom.assign(wh, 0.0f);
    c.sel.assign(wh, 0u);
    c.prior.assign(wh, F4{0, 0, 0, 0});
    c.mask.assign(wh, 0u);
    c.have_prior = false;
    c.depths.clear();
    precompute_views(c);
    return 0;
}

int orc_set_src_depths(orc_ctx* h, int n_src, const float* const* depths, const int* widths, const int* heights, const size_t* pitch_bytes) {
    Ctx& c = h->c;
    if (n_src != c.n_img - 1) { c.err = "n_src must equal number of source views"; return -1; }
    c.depths.assign(n_src, Image());
    for (int i = 0; i < n_src; ++i) {
        Image& im = c.depths[i];
        im.w = widths[i];
        im.h = heights[i];
        im.px.resize((size_t)im.w * im.h);
        const size_t pitch = pitch_bytes ? pitch_bytes[i] : (size_t)im.w * 4;
        for (int y = 0; y < im.h; ++y) std::memcpy(&im.px[(size_t)y * im.w], (const char*)depths[i] + (size_t)y * pitch, (size_t)im.w * 4);
    }
    return 0;
}

int orc_set_state(orc_ctx* h, const float* planes4, const float* costs) {
    Ctx& c = h->c;
    const size_t wh = (size_t)c.W * c.H;
    if (!wh) { c.err = "set_views first"; return -1; }
    if (planes4) std::memcpy(c.planes.data(), planes4, wh * 16);
    if (costs) std::memcpy(c.costs.data(), costs, wh * 4);
    return 0;
}
int orc_set_selected_views(orc_ctx* h, const uint32_t* sel) {
    Ctx& c = h->c;
    std::memcpy(c.sel.data(), sel, (size_t)c.W * c.H * 4);
    return 0;
}
int orc_set_prior(orc_ctx* h, const float* prior4, const uint32_t* mask) {
    Ctx& c = h->c;
    const size_t wh = (size_t)c.W * c.H;
    if (!wh) { c.err = "set_views first"; return -1; }
    std::memcpy(c.prior.data(), prior4, wh * 16);
    std::memcpy(c.mask.data(), mask, wh * 4);
    c.have_prior = true;
    return 0;
}
int orc_run(orc_ctx* h, const void* params, uint64_t seed) { return run(h->c, *(const Params*)params, seed); }
int orc_step(orc_ctx* h, const void* params, uint64_t seed, int kind, int iter, int scale, uint32_t launch) {
    return step(h->c, *(const Params*)params, seed, kind, iter, scale, launch);
}
int orc_get(orc_ctx* h, float* planes4, float* costs, float* geom) {
    Ctx& c = h->c;
    const size_t wh = (size_t)c.W * c.H;
    if (planes4) std::memcpy(planes4, c.planes.data(), wh * 16);
    if (costs) std::memcpy(costs, c.costs.data(), wh * 4);
    if (geom) std::memcpy(geom, c.geom.data(), wh * 4);
    return 0;
}
int orc_get_selected_views(orc_ctx* h, uint32_t* sel) {
    std::memcpy(sel, h->c.sel.data(), (size_t)h->c.W * h->c.H * 4);
    return 0;
}

// T1 probe: ComputeBilateralNCC of per-pixel camera-frame planes against
// every source view; out is [V][H][W].
int orc_eval_ncc(orc_ctx* h, const void* params, const float* planes_cam4, int scale, float* out) {
    Ctx& c = h->c;
    const Params& prm = *(const Params*)params;
    const int rc = check_ready(c, prm);
    if (rc) return rc;
    const int V = prm.num_images - 1;
    const size_t wh = (size_t)c.W * c.H;
#pragma omp parallel for schedule(dynamic, 2)
    for (int y = 0; y < c.H; ++y)
        for (int x = 0; x < c.W; ++x) {
            RefWin rw;
            ref_window(c, prm, x, y, scale, rw);
            const F4 pl = ((const F4*)planes_cam4)[(size_t)y * c.W + x];
            float m[3];
            plane_to_m(c, pl, m);
            for (int v = 0; v < V; ++v) out[(size_t)v * wh + (size_t)y * c.W + x] = ncc_cost(c, rw, x, y, m, v);
        }
    return 0;
}

// T1 probe: geometric-consistency cost of per-pixel camera-frame planes; out [V][H][W]
int orc_eval_geom(orc_ctx* h, const void* params, const float* planes_cam4, float* out) {
    Ctx& c = h->c;
    const Params& prm = *(const Params*)params;
    if ((int)c.depths.size() != c.n_img - 1) { c.err = "need source depth maps"; return -4; }
    const int V = prm.num_images - 1;
    const size_t wh = (size_t)c.W * c.H;
#pragma omp parallel for schedule(dynamic, 2)
    for (int y = 0; y < c.H; ++y)
        for (int x = 0; x < c.W; ++x) {
            const F4 pl = ((const F4*)planes_cam4)[(size_t)y * c.W + x];
            for (int v = 0; v < V; ++v) out[(size_t)v * wh + (size_t)y * c.W + x] = geom_cost(c, v, pl, x, y);
        }
    return 0;
}

// the same with the reference's literal chain through world coordinates (geom_cost_literal): measures the canonical form
int orc_eval_geom_literal(orc_ctx* h, const void* params, const float* planes_cam4, float* out) {
    Ctx& c = h->c;
    const Params& prm = *(const Params*)params;
    if ((int)c.depths.size() != c.n_img - 1) { c.err = "need source depth maps"; return -4; }
    const int V = prm.num_images - 1;
    const size_t wh = (size_t)c.W * c.H;
#pragma omp parallel for schedule(dynamic, 2)
    for (int y = 0; y < c.H; ++y)
        for (int x = 0; x < c.W; ++x) {
            const F4 pl = ((const F4*)planes_cam4)[(size_t)y * c.W + x];
            for (int v = 0; v < V; ++v) out[(size_t)v * wh + (size_t)y * c.W + x] = geom_cost_literal(c, v, pl, x, y);
        }
    return 0;
}

// canonical arithmetic with CUDA's 8-bit interpolation fractions: what libmpmvs_hip_q8.so computes
int orc_set_texture_q8(orc_ctx* h, int on) {
    if (!h) return -1;
    h->c.tex_q8 = on != 0;
    return 0;
}

int orc_set_literal_mode(orc_ctx* h, int mode) {
    if (!h || mode < 0 || mode > 4) return -1;
    h->c.literal_mode = mode;
    return 0;
}

// quantize_fraction: 0 = mode 1 (IEEE + libm), 1 = mode 2 (+ 8-bit texture fractions), 2 = mode 3 (the fast-math model)
int orc_eval_ncc_literal(orc_ctx* h, const void* params, const float* planes_cam4, int scale, int quantize_fraction, float* out) {
    Ctx& c = h->c;
    const Params& prm = *(const Params*)params;
    const int rc = check_ready(c, prm);
    if (rc) return rc;
    const int V = prm.num_images - 1;
    const size_t wh = (size_t)c.W * c.H;
#pragma omp parallel for schedule(dynamic, 2)
    for (int y = 0; y < c.H; ++y)
        for (int x = 0; x < c.W; ++x) {
            const F4 pl = ((const F4*)planes_cam4)[(size_t)y * c.W + x];
            for (int v = 0; v < V; ++v) out[(size_t)v * wh + (size_t)y * c.W + x] = literal_ncc(c, prm, x, y, pl, v, scale, 1 + quantize_fraction);
        }
    return 0;
}

// math probes: fn 0 rcp, 1 exp, 2 sin, 3 cos, 4 acos, 5 fract
int orc_math(int fn, const float* in, float* out, int n) {
    for (int i = 0; i < n; ++i) {
        const float x = in[i];
        float y;
        switch (fn) {
            case 0: y = det_rcp(x); break;
            case 1: y = det_exp(x); break;
            case 2: y = det_sin(x); break;
            case 3: y = det_cos(x); break;
            case 4: y = det_acos(x); break;
            case 5: y = det_fract(x); break;
            case 6: y = det_exp(x); break;  // the kernels' branch-free exp (d_exp_select) must equal the canonical exp
            default: return -1;
        }
        out[i] = y;
    }
    return 0;
}
// RNG probe: the first n uniforms of stream (seed, pix, launch)
int orc_rng(uint64_t seed, uint32_t pix, uint32_t launch, int n, float* out) {
    Rng g = rng_make(seed, pix, launch);
    for (int i = 0; i < n; ++i) out[i] = rng_uniform(g);
    return 0;
}
// homography probe: H (9 floats) for a camera-frame plane and 0-based source view
int orc_homography(orc_ctx* h, const float* plane4, int v, float* H9) {
    Ctx& c = h->c;
    if (v < 0 || v >= (int)c.vc.size()) return -1;
    F4 pl{plane4[0], plane4[1], plane4[2], plane4[3]};
    float m[3];
    plane_to_m(c, pl, m);
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) H9[r * 3 + k] = fmaf(-c.vc[v].b[r], m[k], c.vc[v].A[r * 3 + k]);
    return 0;
}
// statistics hook (see g_stat_death): buffers of H*W*5 int8 and H*W uint32, or NULL to switch it off
void orc_set_refinement_stats(int8_t* death5, uint32_t* wmask) {
    g_stat_death = death5;
    g_stat_wmask = wmask;
}
// phase-A statistics (see g_stat_bad3): a buffer of H*W*32 uint8, or NULL
void orc_set_propagation_stats(uint8_t* bad3) { g_stat_bad3 = bad3; }
int orc_num_threads(void) {
#if defined(_OPENMP)
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_num_threads(int n) {
#if defined(_OPENMP)
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
}  // extern "C"
