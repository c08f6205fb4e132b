// oracle/fusion_oracle.cpp
//
// TEST INFRASTRUCTURE ONLY (see pm_oracle.cpp).  CPU restatements of the depth-map
// fusion that consumes the hot path's outputs (reference RunFusion,
// src/PatchMatch.cpp:287-504):
//
//  mode 0  "snapshot" formulation -- the one the GPU implements.
//          Images are fused in index order as in the reference, but all pixels of one
//          image see the masks as they were when that image started, and the masks it
//          produces become visible to the next image.  Every pixel is then independent,
//          which is what makes the step data parallel and deterministic.  Canonical
//          arithmetic of DESIGN.md section 3 (own acos/exp), so HIP == oracle bit for bit.
//
//  mode 1  literal sequential restatement of the reference: masks updated pixel by
//          pixel in raster order, the used_list that is never reset between pixels
//          (ref :382,:416,:470-495), libm acosf/exp.  Only used to MEASURE how far the
//          snapshot formulation is from the reference's order-dependent result.
//
//  mode 2  the same sequential order (in-place masks, persistent used_list) in the canonical arithmetic: what the GPU's
//          MPMVS_FUSE_REFERENCE_ORDER mode must reproduce bit for bit (it computes it as a parallel fixpoint).
//
// PARITY UNPINNED against reference outputs, like the rest of the oracle.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct Camera {
    float K[9], R[9], t[3], C[3];
    int height, width;
    float depth_min, depth_max;
};

inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// canonical exp / acos: identical specification to pm_oracle.cpp
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
    if (!(x == x)) return y;
    return u2f(f2u(y) + ((uint32_t)(int32_t)n << 23));
}
inline float det_asin_core(float x) {
    const float z = x * x;
    float p = 4.2163199048e-2f;
    p = fmaf(p, z, 2.4181311049e-2f);
    p = fmaf(p, z, 4.5470025998e-2f);
    p = fmaf(p, z, 7.4953002686e-2f);
    p = fmaf(p, z, 1.6666752422e-1f);
    return fmaf(p * z, x, x);
}
inline float det_acos(float x) {
    if (!(x >= -1.0f && x <= 1.0f)) return u2f(0x7fc00000u);
    if (x > 0.5f) return 2.0f * det_asin_core(sqrtf(0.5f * (1.0f - x)));
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * det_asin_core(sqrtf(0.5f * (1.0f + x)));
    return 1.57079632679489661923f - det_asin_core(x);
}

// ref src/PatchMatch.cpp:211-231 (Get3DPointonWorld)
inline void point_on_world(const Camera& cam, int x, int y, float depth, float P[3]) {
    const float X0 = (depth * ((float)x - cam.K[2])) / cam.K[0];
    const float X1 = (depth * ((float)y - cam.K[5])) / cam.K[4];
    const float X2 = depth;
    const float t0 = (cam.R[0] * X0 + cam.R[3] * X1) + cam.R[6] * X2;
    const float t1 = (cam.R[1] * X0 + cam.R[4] * X1) + cam.R[7] * X2;
    const float t2 = (cam.R[2] * X0 + cam.R[5] * X1) + cam.R[8] * X2;
    P[0] = t0 + cam.C[0];
    P[1] = t1 + cam.C[1];
    P[2] = t2 + cam.C[2];
}
// ref src/PatchMatch.cpp:251-261 (ProjectonCamera)
inline void project(const Camera& cam, const float P[3], float& u, float& v, float& depth) {
    const float t0 = ((cam.R[0] * P[0] + cam.R[1] * P[1]) + cam.R[2] * P[2]) + cam.t[0];
    const float t1 = ((cam.R[3] * P[0] + cam.R[4] * P[1]) + cam.R[5] * P[2]) + cam.t[1];
    const float t2 = ((cam.R[6] * P[0] + cam.R[7] * P[1]) + cam.R[8] * P[2]) + cam.t[2];
    depth = (cam.K[6] * t0 + cam.K[7] * t1) + cam.K[8] * t2;
    u = ((cam.K[0] * t0 + cam.K[1] * t1) + cam.K[2] * t2) / depth;
    v = ((cam.K[3] * t0 + cam.K[4] * t1) + cam.K[5] * t2) / depth;
}
// int(v + 0.5f) with the out-of-range cases made explicit (NaN / huge -> rejected)
inline bool round_index(float v, int& out) {
    const float f = v + 0.5f;
    if (!(f > -1.0f && f < 1.0e8f)) return false;
    out = (int)f;
    return true;
}

struct Views {
    int n;
    const Camera* cams;
    const int* estimate;
    const float* const* depths;
    const float* const* normals;
    const unsigned char* const* colors;
    int cch;
    const int* src_off;
    const int* src_ids;
};

inline void load_color(const Views& V, int k, size_t idx, float c[3]) {
    if (V.cch == 3) {
        for (int q = 0; q < 3; ++q) c[q] = (float)V.colors[k][idx * 3 + q];
    } else {
        c[0] = c[1] = c[2] = (float)V.colors[k][idx];
    }
}

// one pixel of image i against the masks `M`; returns true and fills out9 / used[]
// (source pixel index per neighbour slot or -1) when a point is produced
template <bool LITERAL>
inline bool fuse_pixel(const Views& V, int i, int r, int c, unsigned char* const* M, int use_dynamic, float out9[9], std::vector<int>& used) {
    const Camera& rc = V.cams[i];
    const int cols = rc.width;
    const float ref_depth = V.depths[i][(size_t)r * cols + c];
    if (ref_depth <= 0.0f) return false;
    const float* rn = &V.normals[i][((size_t)r * cols + c) * 3];
    float PX[3];
    point_on_world(rc, c, r, ref_depth, PX);
    float sp[3] = {PX[0], PX[1], PX[2]}, sn[3] = {rn[0], rn[1], rn[2]};
    float scol[3];
    load_color(V, i, (size_t)r * cols + c, scol);
    int num = 0;
    float dyn = 0.0f;
    const int b = V.src_off[i], num_ngb = V.src_off[i + 1] - b;
    for (int j = 1; j < num_ngb; ++j) {
        if (j == num_ngb - 1 && num == 0) break;  // ref :402-403
        const int s = V.src_ids[b + j];
        const Camera& scam = V.cams[s];
        float u, v, pd;
        project(scam, PX, u, v, pd);
        int sr, scx;
        if (!round_index(v, sr) || !round_index(u, scx)) continue;
        if (!(scx >= 0 && scx < scam.width && sr >= 0 && sr < scam.height)) continue;
        const size_t sidx = (size_t)sr * scam.width + scx;
        if (M[s][sidx] == 1) continue;
        const float sd = V.depths[s][sidx];
        if (sd <= 0.0f) continue;
        const float* snrm = &V.normals[s][sidx * 3];
        float TX[3];
        point_on_world(scam, scx, sr, sd, TX);
        float bu, bv, bd;
        project(rc, TX, bu, bv, bd);
        const float dc = (float)c - bu, dr = (float)r - bv;
        const float err = LITERAL ? (float)std::sqrt(std::pow((double)dc, 2) + std::pow((double)dr, 2)) : sqrtf(dc * dc + dr * dr);
        if (!(err < 2.0f)) continue;
        const float rel = fabsf(bd - ref_depth) / ref_depth;
        if (!(rel < 0.01f)) continue;
        const float dot = (rn[0] * snrm[0] + rn[1] * snrm[1]) + rn[2] * snrm[2];
        float angle = LITERAL ? acosf(dot) : det_acos(dot);
        if (angle != angle) angle = 0.0f;  // ref :233-242 (GetAngle)
        if (angle < 0.174533f) {
            used[j] = (int)sidx;
            sp[0] += TX[0];
            sp[1] += TX[1];
            sp[2] += TX[2];
            sn[0] += snrm[0];
            sn[1] += snrm[1];
            sn[2] += snrm[2];
            float gs[3];
            load_color(V, s, sidx, gs);
            scol[0] += gs[0];
            scol[1] += gs[1];
            scol[2] += gs[2];
            const float idx = (err + 200.0f * rel) + angle * 10.0f;
            dyn += LITERAL ? (float)std::exp(-(double)idx) : det_exp(-idx);
            num++;
        }
    }
    const bool ok = use_dynamic ? (num >= 1 && dyn > 0.3f * (float)num) : (num >= 2);
    if (!ok) return false;
    const float d = (float)num + 1.0f;
    for (int k = 0; k < 3; ++k) {
        out9[k] = sp[k] / d;
        out9[3 + k] = sn[k] / d;
        out9[6 + k] = scol[k] / d;
    }
    return true;
}

}  // namespace

extern "C" int orc_fuse(int mode, int n, const void* cams_, const int* estimate, const float* const* depths, const float* const* normals,
                        const unsigned char* const* colors, int color_channels, const unsigned char* const* sky, const int* src_off,
                        const int* src_ids, int use_dynamic, unsigned char* const* out_valid, float* const* out_points9,
                        unsigned char* const* masks) {
    const Camera* cams = (const Camera*)cams_;
    Views V{n, cams, estimate, depths, normals, colors, color_channels, src_off, src_ids};
    for (int i = 0; i < n; ++i) {
        const size_t wh = (size_t)cams[i].width * cams[i].height;
        std::memset(masks[i], 0, wh);
        std::memset(out_valid[i], 0, wh);
    }
    int max_ngb = 1;
    for (int i = 0; i < n; ++i) max_ngb = std::max(max_ngb, src_off[i + 1] - src_off[i]);
    if (mode == 1 || mode == 2) {
        // sequential: masks updated in place, used_list persists across pixels of an image (mode 1: libm, mode 2: canonical math)
        for (int i = 0; i < n; ++i) {
            if (!estimate[i]) continue;
            const int rows = cams[i].height, cols = cams[i].width;
            const int b = src_off[i], num_ngb = src_off[i + 1] - b;
            std::vector<int> used(max_ngb, -1);
            for (int r = 0; r < rows; ++r)
                for (int c = 0; c < cols; ++c) {
                    if (masks[i][(size_t)r * cols + c] == 1) continue;
                    if (sky && sky[i] && sky[i][(size_t)r * cols + c] > 0) {  // ref :385-388
                        masks[i][(size_t)r * cols + c] = 1;
                        continue;
                    }
                    float o[9];
                    if (mode == 1 ? fuse_pixel<true>(V, i, r, c, masks, use_dynamic, o, used) : fuse_pixel<false>(V, i, r, c, masks, use_dynamic, o, used)) {
                        out_valid[i][(size_t)r * cols + c] = 1;
                        std::memcpy(&out_points9[i][((size_t)r * cols + c) * 9], o, sizeof(o));
                        for (int j = 1; j < num_ngb; ++j)
                            if (used[j] != -1) masks[src_ids[b + j]][used[j]] = 1;
                    }
                }
        }
        return 0;
    }
    // snapshot: all pixels of image i read the masks as of the start of image i
    std::vector<std::vector<unsigned char>> next(n);
    for (int i = 0; i < n; ++i) {
        if (!estimate[i]) continue;
        const int rows = cams[i].height, cols = cams[i].width;
        const int b = src_off[i], num_ngb = src_off[i + 1] - b;
        for (int j = 1; j < num_ngb; ++j) {
            const int s = src_ids[b + j];
            next[s].assign(masks[s], masks[s] + (size_t)cams[s].width * cams[s].height);
        }
#pragma omp parallel
        {
            std::vector<int> used(max_ngb, -1);
#pragma omp for schedule(dynamic, 4)
            for (int r = 0; r < rows; ++r)
                for (int c = 0; c < cols; ++c) {
                    if (masks[i][(size_t)r * cols + c] == 1) continue;
                    if (sky && sky[i] && sky[i][(size_t)r * cols + c] > 0) {  // own pixel: read by this iteration only
                        masks[i][(size_t)r * cols + c] = 1;
                        continue;
                    }
                    for (int j = 0; j < num_ngb; ++j) used[j] = -1;
                    float o[9];
                    if (fuse_pixel<false>(V, i, r, c, masks, use_dynamic, o, used)) {
                        out_valid[i][(size_t)r * cols + c] = 1;
                        std::memcpy(&out_points9[i][((size_t)r * cols + c) * 9], o, sizeof(o));
                        for (int j = 1; j < num_ngb; ++j)
                            if (used[j] != -1) next[src_ids[b + j]][used[j]] = 1;  // idempotent store of 1: race free in effect
                    }
                }
        }
        for (int j = 1; j < num_ngb; ++j) {
            const int s = src_ids[b + j];
            std::memcpy(masks[s], next[s].data(), next[s].size());
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------
// sky-mask joint-bilateral filter (reference SkySegment/src/SkyRegionDetect.cu:3-34;
// SURVEY.md row f-4).  mode 0: canonical arithmetic, what the HIP kernel computes
// (mp-mvs_amd/csrc/pm_sky.hpp); mode 1: literal -- libm expf, separate multiply and add,
// out-of-image taps skipped -- to measure the deviation.
// ---------------------------------------------------------------------------------------
extern "C" int orc_sky_bilateral(int mode, const unsigned char* bgr, const float* mask, float* out, int height, int width) {
    const int half = 18;
    std::vector<float> spatial((2 * half + 1) * (2 * half + 1));
    for (int i = -half; i <= half; ++i)
        for (int j = -half; j <= half; ++j) spatial[(i + half) * (2 * half + 1) + (j + half)] = -(sqrtf((float)(i * i + j * j)) / 72.0f);
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            const size_t idx = (size_t)y * width + x;
            const float cb = (float)bgr[3 * idx], cg = (float)bgr[3 * idx + 1], cr = (float)bgr[3 * idx + 2];
            float wsum = 0.0f, prob = 0.0f;
            for (int i = -half; i <= half; ++i)
                for (int j = -half; j <= half; ++j) {
                    const int nx = x + i, ny = y + j;
                    if (nx < 0 || nx >= width || ny < 0 || ny >= height) continue;
                    const size_t k = (size_t)ny * width + nx;
                    const float db = (float)bgr[3 * k] - cb, dg = (float)bgr[3 * k + 1] - cg, dr = (float)bgr[3 * k + 2] - cr;
                    const float dc = sqrtf((db * db + dg * dg) + dr * dr);
                    const float sp = spatial[(i + half) * (2 * half + 1) + (j + half)];
                    if (mode == 1) {
                        const float w = expf(sp - dc / 8.0f);
                        wsum += w;
                        const float p = w * mask[k];
                        prob += p;
                    } else {
                        const float w = det_exp(fmaf(dc, -0.125f, sp));
                        wsum += w;
                        prob = fmaf(w, mask[k], prob);
                    }
                }
            out[idx] = (double)(prob / wsum) > 0.6 ? 255.0f : 0.0f;
        }
    return 0;
}
