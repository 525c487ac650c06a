// Microbenchmark round 3 (dev tool): the IIR-batch access pattern.  A [channels][n] fp32 matrix, one wave per 64
// channels walking time in tiles of SEG samples per row: how much HBM bandwidth does the row-segment length cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
__device__ inline float4 ld(const float* p) { typedef float v4 __attribute__((ext_vector_type(4))); v4 v = __builtin_nontemporal_load((const v4*)p); return make_float4(v.x,v.y,v.z,v.w); }
__device__ inline void st(float* p, float4 a) { typedef float v4 __attribute__((ext_vector_type(4))); v4 v = {a.x,a.y,a.z,a.w}; __builtin_nontemporal_store(v, (v4*)p); }

template <int SEG>
__global__ __launch_bounds__(64) void copy_rows(const float* __restrict__ x, float* __restrict__ y, size_t n, size_t chunk) {
    extern __shared__ float dummy[];
    const int lane = threadIdx.x;
    const size_t group = (size_t)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    constexpr int LPR = SEG / 4 < 64 ? SEG / 4 : 64;   // lanes per row per instruction
    constexpr int RPI = 64 / LPR;                       // rows per instruction
    constexpr int IPR = SEG / 4 / LPR;                  // instructions per row (SEG > 256)
    constexpr int ITERS = 64 / RPI * IPR;               // instructions per tile
    const size_t ch0 = group * 64;
    if (lane == 0 && n == 1) dummy[0] = 0;
    for (size_t t0 = 0; t0 < n; t0 += SEG) {
        for (int base = 0; base < ITERS; base += 16) {
            float4 r[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int it = base + k;
                const size_t row = ch0 + (it / IPR) * RPI + lane / LPR;
                const size_t col = t0 + (size_t)(it % IPR) * 256 + (lane % LPR) * 4;
                r[k] = ld(x + row * n + col);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int it = base + k;
                const size_t row = ch0 + (it / IPR) * RPI + lane / LPR;
                const size_t col = t0 + (size_t)(it % IPR) * 256 + (lane % LPR) * 4;
                st(y + row * n + col, r[k]);
            }
        }
    }
}
int main() {
    const size_t channels = 262144, n = 4096, bytes = channels * n * 4;
    void *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, int lds, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        (void)hipEventRecord(e0); for (int i = 0; i < 10; ++i) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-12s lds %6d  %.4f ms  %.3f TB/s\n", name, lds, ms, 2.0 * bytes / ms / 1e9);
    };
    const float* A = (const float*)a; float* B = (float*)b;
    const size_t groups = channels / 64, chunk = groups / 8;
    for (int lds : {0, 17000, 34000}) {
        run("seg 64", lds, [&]{ hipLaunchKernelGGL((copy_rows<64>), dim3(groups), dim3(64), lds, 0, A, B, n, chunk); });
        run("seg 128", lds, [&]{ hipLaunchKernelGGL((copy_rows<128>), dim3(groups), dim3(64), lds, 0, A, B, n, chunk); });
        run("seg 256", lds, [&]{ hipLaunchKernelGGL((copy_rows<256>), dim3(groups), dim3(64), lds, 0, A, B, n, chunk); });
        run("seg 512", lds, [&]{ hipLaunchKernelGGL((copy_rows<512>), dim3(groups), dim3(64), lds, 0, A, B, n, chunk); });
        run("seg 1024", lds, [&]{ hipLaunchKernelGGL((copy_rows<1024>), dim3(groups), dim3(64), lds, 0, A, B, n, chunk); });
    }
    return 0;
}
