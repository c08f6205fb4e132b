// tools/verify_rcp.hip -- exhaustive check of the canonical reciprocal d_rcp (hardware seed v_rcp_f32 + two Newton steps in
// fma arithmetic) against the correctly rounded IEEE quotient 1.0f / x, over all 2^32 bit patterns of x.
//   hipcc -O2 --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off -o build/verify_rcp tools/verify_rcp.hip && build/verify_rcp
// Checks the rule the CPU oracle implements (oracle/pm_oracle.cpp det_rcp):
//   |x| in [2^-126, 2^126]  (x and 1/x normal)   the IEEE quotient 1.0f / x, bit for bit
//   2^126 < |x| < inf       (1/x denormal)        a zero with the sign of x (v_rcp_f32 flushes its denormal result)
//   zero, denormal, inf, NaN                      not finite (NaN or an infinity: the evaluation is discarded either way)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

__device__ __forceinline__ float seq_rcp(float z) {
    float r = __builtin_amdgcn_rcpf(z);
    r = __builtin_fmaf(r, __builtin_fmaf(-z, r, 1.0f), r);
    r = __builtin_fmaf(r, __builtin_fmaf(-z, r, 1.0f), r);
    return r;
}

struct Stats {
    unsigned long long normal_total, normal_bad, special_total, special_finite;
    uint32_t bad_example[8], special_example[16];
    float special_value[16];
    unsigned n_bad, n_special;
};

__global__ void k_check(uint32_t base, Stats* st) {
    const uint32_t bits = base + blockIdx.x * blockDim.x + threadIdx.x;
    const float x = __uint_as_float(bits);
    const float a = seq_rcp(x);
    const float b = 1.0f / x;
    const float ax = fabsf(x);
    const bool x_normal = ax >= 1.17549435e-38f && ax <= 3.40282347e+38f;
    const bool q_normal = x_normal && fabsf(b) >= 1.17549435e-38f;
    if (q_normal) {
        atomicAdd(&st->normal_total, 1ULL);
        if (__float_as_uint(a) != __float_as_uint(b)) {
            atomicAdd(&st->normal_bad, 1ULL);
            const unsigned k = atomicAdd(&st->n_bad, 1u);
            if (k < 8) st->bad_example[k] = bits;
        }
    } else {
        atomicAdd(&st->special_total, 1ULL);
        const bool finite = fabsf(a) <= 3.40282347e+38f;  // false for NaN
        const bool huge = x_normal;                          // x normal, 1/x denormal
        const bool ok = huge ? (__float_as_uint(a) == (bits & 0x80000000u)) : !finite;
        if (!ok) {
            atomicAdd(&st->special_finite, 1ULL);
            const unsigned k = atomicAdd(&st->n_special, 1u);
            if (k < 16) {
                st->special_example[k] = bits;
                st->special_value[k] = a;
            }
        }
    }
}

int main() {
    Stats* d;
    hipMalloc(&d, sizeof(Stats));
    hipMemset(d, 0, sizeof(Stats));
    for (uint64_t base = 0; base < (1ULL << 32); base += (1ULL << 24)) hipLaunchKernelGGL(k_check, dim3(1 << 16), dim3(256), 0, 0, (uint32_t)base, d);
    Stats h;
    hipMemcpy(&h, d, sizeof(Stats), hipMemcpyDeviceToHost);
    std::printf("x normal, 1/x normal: %llu inputs, %llu differ from the IEEE quotient\n", h.normal_total, h.normal_bad);
    for (unsigned i = 0; i < h.n_bad && i < 8; ++i) {
        float x;
        std::memcpy(&x, &h.bad_example[i], 4);
        std::printf("   x = %a (0x%08x)\n", x, h.bad_example[i]);
    }
    std::printf("other inputs (1/x denormal: signed zero expected; zero, denormal, inf, NaN: non-finite expected): %llu, %llu break the rule\n",
                h.special_total, h.special_finite);
    for (unsigned i = 0; i < h.n_special && i < 16; ++i) {
        float x;
        std::memcpy(&x, &h.special_example[i], 4);
        std::printf("   x = %a (0x%08x) -> %a\n", x, h.special_example[i], h.special_value[i]);
    }
    return (h.normal_bad || h.special_finite) ? 1 : 0;
}
