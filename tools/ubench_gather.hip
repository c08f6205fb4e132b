// tools/ubench_gather.hip -- cost of one wave-wide gather through the vector L1 (TA / TCP / TD) on gfx950 as a function of
// the ADDRESS PATTERN of the 64 lanes and of the load width.  Measurement tool only.  Every CU runs 8 waves that issue
// long streams of independent buffer loads from a small L1-resident table (so nothing but the L1 pipeline is timed);
// prints cycles per wave-load per CU.
//   hipcc -O3 --offload-arch=gfx950 -o build/ubench_gather tools/ubench_gather.hip && build/ubench_gather
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                                      \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                            \
            std::exit(1);                                                                           \
        }                                                                                           \
    } while (0)

constexpr int kIters = 512;   // loop trips, 8 loads each
constexpr int kTableBytes = 16 * 1024;

// offs: per-lane byte offsets of the pattern (64 entries); every load adds a small rotating displacement so that the loads
// are distinct instructions with distinct addresses inside the same resident table
template <int WIDTH>
__global__ __launch_bounds__(512) void k_gather(const char* table, const int* offs, float* sink, unsigned long long* ticks) {
    const int lane = threadIdx.x & 63;
    const int base = offs[lane];
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(table), (short)0, kTableBytes, 0x00020000);
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int off = (base + ((i * 5 + k * 3) & 15) * 1024) & (kTableBytes - 16);
            if constexpr (WIDTH == 4) {
                acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
            } else if constexpr (WIDTH == 8) {
                typedef unsigned u2 __attribute__((ext_vector_type(2)));
                const u2 v = __builtin_bit_cast(u2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0));
                acc += __builtin_bit_cast(float, v.x ^ v.y);
            } else {
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                const u4 v = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
                acc += __builtin_bit_cast(float, v.x ^ v.y ^ v.z ^ v.w);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) ticks[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    if (acc == 12345.678f) sink[0] = acc;
}

struct Pattern {
    const char* name;
    int offs[64];
};

int main() {
    std::vector<Pattern> pats;
    auto add = [&](const char* name, auto fn) {
        Pattern p;
        p.name = name;
        for (int l = 0; l < 64; ++l) p.offs[l] = fn(l);
        pats.push_back(p);
    };
    add("all lanes one address", [](int) { return 0; });
    add("contiguous 4 B (256 B)", [](int l) { return l * 4; });
    add("contiguous 8 B (512 B)", [](int l) { return l * 8; });
    add("contiguous 16 B (1 KB)", [](int l) { return l * 16; });
    add("stride 16 B, 8-B texels 2 px apart", [](int l) { return l * 16; });
    add("4 rows x 16 lanes, 8 B stride 16", [](int l) { return (l / 16) * 2048 + (l % 16) * 16 + 24; });
    add("8 rows x 8 lanes, 8 B stride 16", [](int l) { return (l / 8) * 1024 + (l % 8) * 16 + 24; });
    add("16 rows x 4 lanes, 8 B stride 16", [](int l) { return (l / 4) * 512 + (l % 4) * 16 + 24; });
    add("16 rows x 4 lanes, 4 B stride 8", [](int l) { return (l / 4) * 512 + (l % 4) * 8 + 12; });
    add("16 rows x 4 lanes, 16 B stride 32", [](int l) { return (l / 4) * 512 + (l % 4) * 32 + 48; });
    add("8 rows x 8 lanes, 16 B stride 32", [](int l) { return (l / 8) * 1024 + (l % 8) * 32 + 48; });
    add("64 lanes, 64 distinct 128-B lines", [](int l) { return l * 128; });
    add("64 lanes, 64 distinct 64-B chunks", [](int l) { return l * 64; });
    add("pairs share a 64-B chunk", [](int l) { return (l / 2) * 64 + (l % 2) * 8; });
    add("quads share a 64-B chunk", [](int l) { return (l / 4) * 64 + (l % 4) * 8; });
    add("8 lanes share a 64-B chunk", [](int l) { return (l / 8) * 64 + (l % 8) * 8; });
    add("16 lanes share a 128-B line", [](int l) { return (l / 16) * 128 + (l % 16) * 8; });
    add("quads straddle two 64-B chunks", [](int l) { return (l / 4) * 128 + 40 + (l % 4) * 16; });

    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    char* d_table;
    int* d_offs;
    float* d_sink;
    unsigned long long* d_ticks;
    CHK(hipMalloc(&d_table, kTableBytes));
    CHK(hipMemset(d_table, 1, kTableBytes));
    CHK(hipMalloc(&d_offs, 64 * 4));
    CHK(hipMalloc(&d_sink, 4));
    CHK(hipMalloc(&d_ticks, 8 * cus * 8));
    std::vector<unsigned long long> h(8 * cus);
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    std::printf("%d CUs, 8 waves per CU, %d loads per wave; cycles per wave-load per CU (wall clock at 2.4 GHz | s_memtime ticks)\n", cus, kIters * 8);
    std::printf("%-40s %22s %22s %22s\n", "pattern", "dword", "dwordx2", "dwordx4");
    for (const Pattern& p : pats) {
        CHK(hipMemcpy(d_offs, p.offs, 64 * 4, hipMemcpyHostToDevice));
        std::printf("%-40s", p.name);
        for (int width : {4, 8, 16}) {
            auto launch = [&] {
                if (width == 4) hipLaunchKernelGGL(k_gather<4>, dim3(cus), dim3(512), 0, 0, d_table, d_offs, d_sink, d_ticks);
                if (width == 8) hipLaunchKernelGGL(k_gather<8>, dim3(cus), dim3(512), 0, 0, d_table, d_offs, d_sink, d_ticks);
                if (width == 16) hipLaunchKernelGGL(k_gather<16>, dim3(cus), dim3(512), 0, 0, d_table, d_offs, d_sink, d_ticks);
            };
            launch();
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0, 0));
            launch();
            CHK(hipEventRecord(e1, 0));
            CHK(hipDeviceSynchronize());
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            CHK(hipMemcpy(h.data(), d_ticks, 8 * cus * 8, hipMemcpyDeviceToHost));
            double ticks = 0;
            for (int i = 0; i < 8 * cus; ++i) ticks += (double)h[i];
            ticks /= 8 * cus;
            const double loads_per_cu = 8.0 * kIters * 8;
            std::printf("   %8.1f | %8.1f", ms * 1e-3 * 2.4e9 / loads_per_cu, ticks / (kIters * 8) / 8.0);
        }
        std::printf("\n");
    }
    return 0;
}
