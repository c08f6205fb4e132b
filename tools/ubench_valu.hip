// tools/ubench_valu.hip -- issue-rate microbenchmark for the instruction classes of the NCC tap loop on gfx950.
// Measurement tool only (never linked into the product).  For each instruction class it runs a long stream of
// independent instructions at 1, 2 and 4 waves per SIMD (one workgroup per CU, pinned by a 96 KB LDS allocation) and
// prints cycles per wave-instruction per SIMD, from s_memtime (shader-clock ticks) around the loop.
//   hipcc -O3 --offload-arch=gfx950 -o gpurun_out/ubench_valu tools/ubench_valu.hip && gpurun_out/ubench_valu
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHK(x)                                                                                      \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                            \
            std::exit(1);                                                                           \
        }                                                                                           \
    } while (0)

constexpr int kIters = 2000;  // loop trips; 32 instructions each
extern __shared__ float lds_pin[];

// 8 independent destination registers, 4 instructions on each per trip
#define REP8(I)                                                                                                        \
    I("%0") I("%1") I("%2") I("%3") I("%4") I("%5") I("%6") I("%7")
#define BODY(I) REP8(I) REP8(I) REP8(I) REP8(I)

#define KERNEL(NAME, INSTR)                                                                                            \
    __global__ __launch_bounds__(1024) void NAME(unsigned long long* out, float seed) {                                \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,   \
              a7 = a0 + 7;                                                                                             \
        float b = seed * 0.5f, c = seed * 0.25f;                                                                       \
        lds_pin[threadIdx.x] = seed;                                                                                   \
        __syncthreads();                                                                                               \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                    \
        for (int i = 0; i < kIters; ++i) {                                                                             \
            asm volatile(BODY(INSTR)                                                                                   \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)              \
                         : "v"(b), "v"(c));                                                                            \
        }                                                                                                              \
        asm volatile("s_nop 0" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));              \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                    \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                              \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = 0;                                           \
    }

// 64-bit (register pair) destinations for the packed-fp32 classes
#define KERNEL2(NAME, INSTR)                                                                                           \
    __global__ __launch_bounds__(1024) void NAME(unsigned long long* out, float seed) {                                \
        typedef float f2 __attribute__((ext_vector_type(2)));                                                          \
        f2 a0 = {seed + threadIdx.x, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,  \
           a6 = a0 + 6.f, a7 = a0 + 7.f;                                                                               \
        f2 b = {seed * 0.5f, seed}, c = {seed * 0.25f, seed};                                                          \
        lds_pin[threadIdx.x] = seed;                                                                                   \
        __syncthreads();                                                                                               \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                    \
        for (int i = 0; i < kIters; ++i) {                                                                             \
            asm volatile(BODY(INSTR)                                                                                   \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)              \
                         : "v"(b), "v"(c));                                                                            \
        }                                                                                                              \
        asm volatile("s_nop 0" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));              \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                    \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                              \
        const f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                            \
        if (s.x + s.y == 12345.678f) out[0] = 0;                                                                       \
    }

#define I_FMA(d) "v_fma_f32 " d ", " d ", %8, %9\n"
#define I_MUL(d) "v_mul_f32 " d ", " d ", %8\n"
#define I_ADD(d) "v_add_f32 " d ", " d ", %8\n"
#define I_MED3(d) "v_med3_f32 " d ", " d ", %8, %9\n"
#define I_FRACT(d) "v_fract_f32 " d ", " d "\n"
#define I_CVTFLR(d) "v_cvt_flr_i32_f32 " d ", " d "\n"
#define I_MAD24(d) "v_mad_u32_u24 " d ", " d ", %8, %9\n"
#define I_LSHLADD(d) "v_lshl_add_u32 " d ", " d ", 2, %8\n"
#define I_PERM(d) "v_perm_b32 " d ", " d ", %8, %9\n"
#define I_AND(d) "v_and_b32 " d ", " d ", %8\n"
#define I_FMAMIX(d) "v_fma_mix_f32 " d ", " d ", %8, %9 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n"
#define I_PKADDH(d) "v_pk_add_f16 " d ", " d ", %8\n"
#define I_RCP(d) "v_rcp_f32 " d ", " d "\n"
#define I_SQRT(d) "v_sqrt_f32 " d ", " d "\n"
#define I_CVTUB(d) "v_cvt_f32_ubyte1 " d ", " d "\n"
#define I_MOV(d) "v_mov_b32 " d ", %8\n"
#define I_FLOOR(d) "v_floor_f32 " d ", " d "\n"
#define I_TRUNC(d) "v_trunc_f32 " d ", " d "\n"
#define I_RNDNE(d) "v_rndne_f32 " d ", " d "\n"
#define I_CVTI32(d) "v_cvt_i32_f32 " d ", " d "\n"
#define I_CVTU32(d) "v_cvt_u32_f32 " d ", " d "\n"
#define I_CVTF32I(d) "v_cvt_f32_i32 " d ", " d "\n"
#define I_CVTF32U(d) "v_cvt_f32_u32 " d ", " d "\n"
#define I_MAX(d) "v_max_f32 " d ", " d ", %8\n"
#define I_MIN(d) "v_min_f32 " d ", " d ", %8\n"
#define I_SUB(d) "v_sub_f32 " d ", " d ", %8\n"
#define I_LSHL(d) "v_lshlrev_b32 " d ", 2, " d "\n"
#define I_ADDU(d) "v_add_u32 " d ", " d ", %8\n"
#define I_SUBU(d) "v_sub_u32 " d ", " d ", %8\n"
#define I_MULU24(d) "v_mul_u32_u24 " d ", " d ", %8\n"
#define I_MULLO(d) "v_mul_lo_u32 " d ", " d ", %8\n"
#define I_OR(d) "v_or_b32 " d ", " d ", %8\n"
#define I_BFE(d) "v_bfe_u32 " d ", " d ", 8, 8\n"
#define I_CNDMASK(d) "v_cndmask_b32 " d ", " d ", %8, vcc\n"
#define I_FMAC(d) "v_fmac_f32 " d ", %8, %9\n"
#define I_MAX3(d) "v_max3_f32 " d ", " d ", %8, %9\n"
#define I_ADD3(d) "v_add3_u32 " d ", " d ", %8, %9\n"
#define I_LSHLOR(d) "v_lshl_or_b32 " d ", " d ", 2, %8\n"
#define I_MOVDPP(d) "v_mov_b32_dpp " d ", " d " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_ADDDPP(d) "v_add_f32_dpp " d ", " d ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_CVTPKRTZ(d) "v_cvt_pkrtz_f16_f32 " d ", " d ", %8\n"
#define I_MADMIX(d) "v_fma_mixlo_f16 " d ", " d ", %8, %9\n"
#define I_PKMULH(d) "v_pk_mul_f16 " d ", " d ", %8\n"
#define I_PKFMAH(d) "v_pk_fma_f16 " d ", " d ", %8, %9\n"
#define I_EXP(d) "v_exp_f32 " d ", " d "\n"
#define I_MULLEG(d) "v_mul_legacy_f32 " d ", " d ", %8\n"
#define I_CVTFLRPAIR(d) "v_cvt_flr_i32_f32 " d ", " d "\nv_fma_f32 " d ", " d ", %8, %9\n"
#define I_PKFMA(d) "v_pk_fma_f32 " d ", " d ", %8, %9\n"
#define I_PKMUL(d) "v_pk_mul_f32 " d ", " d ", %8\n"
#define I_PKADD(d) "v_pk_add_f32 " d ", " d ", %8\n"

KERNEL(k_fma, I_FMA)
KERNEL(k_mul, I_MUL)
KERNEL(k_add, I_ADD)
KERNEL(k_med3, I_MED3)
KERNEL(k_fract, I_FRACT)
KERNEL(k_cvtflr, I_CVTFLR)
KERNEL(k_mad24, I_MAD24)
KERNEL(k_lshladd, I_LSHLADD)
KERNEL(k_perm, I_PERM)
KERNEL(k_and, I_AND)
KERNEL(k_fmamix, I_FMAMIX)
KERNEL(k_pkaddh, I_PKADDH)
KERNEL(k_rcp, I_RCP)
KERNEL(k_sqrt, I_SQRT)
KERNEL(k_cvtub, I_CVTUB)
KERNEL(k_mov, I_MOV)
KERNEL(k_floor, I_FLOOR)
KERNEL(k_trunc, I_TRUNC)
KERNEL(k_rndne, I_RNDNE)
KERNEL(k_cvti32, I_CVTI32)
KERNEL(k_cvtu32, I_CVTU32)
KERNEL(k_cvtf32i, I_CVTF32I)
KERNEL(k_cvtf32u, I_CVTF32U)
KERNEL(k_max, I_MAX)
KERNEL(k_min, I_MIN)
KERNEL(k_sub, I_SUB)
KERNEL(k_lshl, I_LSHL)
KERNEL(k_addu, I_ADDU)
KERNEL(k_subu, I_SUBU)
KERNEL(k_mulu24, I_MULU24)
KERNEL(k_mullo, I_MULLO)
KERNEL(k_or, I_OR)
KERNEL(k_bfe, I_BFE)
KERNEL(k_cndmask, I_CNDMASK)
KERNEL(k_fmac, I_FMAC)
KERNEL(k_max3, I_MAX3)
KERNEL(k_add3, I_ADD3)
KERNEL(k_lshlor, I_LSHLOR)
KERNEL(k_movdpp, I_MOVDPP)
KERNEL(k_adddpp, I_ADDDPP)
KERNEL(k_cvtpkrtz, I_CVTPKRTZ)
KERNEL(k_madmix, I_MADMIX)
KERNEL(k_pkmulh, I_PKMULH)
KERNEL(k_pkfmah, I_PKFMAH)
KERNEL(k_exp, I_EXP)
KERNEL(k_mulleg, I_MULLEG)
KERNEL(k_flrfma, I_CVTFLRPAIR)
KERNEL2(k_pkfma, I_PKFMA)
KERNEL2(k_pkmul, I_PKMUL)
KERNEL2(k_pkadd, I_PKADD)

typedef void (*kern_t)(unsigned long long*, float);
struct Entry {
    const char* name;
    kern_t fn;
};

int main() {
    const Entry tab[] = {{"v_fma_f32", k_fma},         {"v_mul_f32", k_mul},         {"v_add_f32", k_add},
                         {"v_med3_f32", k_med3},       {"v_fract_f32", k_fract},     {"v_cvt_flr_i32_f32", k_cvtflr},
                         {"v_mad_u32_u24", k_mad24},   {"v_lshl_add_u32", k_lshladd}, {"v_perm_b32", k_perm},
                         {"v_and_b32", k_and},         {"v_fma_mix_f32", k_fmamix},  {"v_pk_add_f16", k_pkaddh},
                         {"v_rcp_f32", k_rcp},         {"v_sqrt_f32", k_sqrt},       {"v_cvt_f32_ubyte1", k_cvtub},
                         {"v_mov_b32", k_mov},
                         {"v_floor_f32", k_floor}, {"v_trunc_f32", k_trunc}, {"v_rndne_f32", k_rndne}, {"v_cvt_i32_f32", k_cvti32}, {"v_cvt_u32_f32", k_cvtu32},
                         {"v_cvt_f32_i32", k_cvtf32i}, {"v_cvt_f32_u32", k_cvtf32u}, {"v_max_f32", k_max}, {"v_min_f32", k_min}, {"v_sub_f32", k_sub},
                         {"v_lshlrev_b32", k_lshl}, {"v_add_u32", k_addu}, {"v_sub_u32", k_subu}, {"v_mul_u32_u24", k_mulu24}, {"v_mul_lo_u32", k_mullo},
                         {"v_or_b32", k_or}, {"v_bfe_u32", k_bfe}, {"v_cndmask_b32", k_cndmask}, {"v_fmac_f32", k_fmac}, 
                         {"v_max3_f32", k_max3}, {"v_add3_u32", k_add3}, {"v_lshl_or_b32", k_lshlor}, {"v_mov_b32_dpp quad", k_movdpp},
                         {"v_add_f32_dpp quad", k_adddpp}, {"v_cvt_pkrtz_f16_f32", k_cvtpkrtz}, {"v_fma_mixlo_f16", k_madmix}, {"v_pk_mul_f16", k_pkmulh},
                         {"v_pk_fma_f16", k_pkfmah}, {"v_exp_f32", k_exp}, {"v_mul_legacy_f32", k_mulleg}, {"cvt_flr+fma pair (x2)", k_flrfma},
                         {"v_pk_fma_f32", k_pkfma},    {"v_pk_mul_f32", k_pkmul},
                         {"v_pk_add_f32", k_pkadd}};
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned long long* d_out = nullptr;
    CHK(hipMalloc(&d_out, sizeof(unsigned long long) * 16 * cus));
    std::vector<unsigned long long> h(16 * cus);
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    std::printf("%d CUs; cycles per wave-instruction per SIMD (ticks of s_memtime = 100 MHz reference are scaled by the event clock)\n", cus);
    std::printf("%-20s %28s %28s %28s\n", "instruction", "1 wave/SIMD: cyc/inst  GHz", "2 waves/SIMD", "4 waves/SIMD");
    for (const Entry& en : tab) {
        std::printf("%-20s", en.name);
        for (int wps : {1, 2, 4}) {
            const int threads = 256 * wps;
            const size_t lds = 96 * 1024;  // one workgroup per CU
            CHK(hipFuncSetAttribute((const void*)en.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(en.fn, dim3(cus), dim3(threads), lds, 0, d_out, 1.0f);  // warm-up
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(en.fn, dim3(cus), dim3(threads), lds, 0, d_out, 1.0f);
            CHK(hipEventRecord(e1, 0));
            CHK(hipDeviceSynchronize());
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            CHK(hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * 16 * cus, hipMemcpyDeviceToHost));
            double ticks = 0;
            int n = 0;
            for (int b = 0; b < cus; ++b)
                for (int w = 0; w < 4 * wps; ++w) {
                    ticks += (double)h[b * 16 + w];
                    ++n;
                }
            ticks /= n;  // s_memtime ticks per wave for kIters*32 instructions
            const double inst_per_simd = (double)kIters * 32 * wps;
            // wall-clock view: kernel time * 2.4 GHz / instructions per SIMD (upper bound: includes launch + prologue)
            const double cyc_wall = ms * 1e-3 * 2.4e9 / inst_per_simd;
            const double cyc_tick = ticks / ((double)kIters * 32) / wps;  // ticks per instruction per SIMD if ticks were shader cycles
            std::printf("   tick %6.2f  wall@2.4GHz %6.2f", cyc_tick, cyc_wall);
        }
        std::printf("\n");
    }
    return 0;
}
