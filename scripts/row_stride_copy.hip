// row_stride_copy.hip -- dev tool: the batch IIR's access pattern (one wave per 64 rows of a [rows][n] fp32 matrix, walking
// time in 512-byte row segments, non-temporal 16-byte accesses) as a plain copy, over row lengths n: is a power-of-two row
// stride special for the memory system?   hipcc -O3 --offload-arch=gfx950 -o /tmp/rsc scripts/row_stride_copy.hip && /tmp/rsc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float v4 __attribute__((ext_vector_type(4)));

// MAP 0: wave g owns rows 64 g .. 64 g + 63.  MAP 1: rows g + groups * r (interleaved).  SEG samples per row and tile.
template <int SEG, int MAP>
__global__ __launch_bounds__(64) void copy_rows(const float* __restrict__ x, float* __restrict__ y, size_t n, size_t chunk, size_t groups) {
    extern __shared__ float dummy[];
    const int lane = threadIdx.x;
    const size_t group = (size_t)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (group >= groups) return;
    constexpr int LPR = SEG / 4, RPI = 64 / LPR, ITERS = 64 / RPI;
    if (lane == 0 && n == 1) dummy[0] = 0;
    for (size_t t0 = 0; t0 < n; t0 += SEG) {
        for (int base = 0; base < ITERS; base += 16) {
            v4 r[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int rr = (base + k) * RPI + lane / LPR;
                const size_t row = MAP ? group + groups * rr : group * 64 + rr;
                r[k] = __builtin_nontemporal_load((const v4*)(x + row * n + t0 + (lane % LPR) * 4));
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int rr = (base + k) * RPI + lane / LPR;
                const size_t row = MAP ? group + groups * rr : group * 64 + rr;
                __builtin_nontemporal_store(r[k], (v4*)(y + row * n + t0 + (lane % LPR) * 4));
            }
        }
    }
}

// As the batch IIR walks: the whole next tile (32 loads) is requested, `delay` rounds of s_sleep stand in for the recursion over
// the current tile, then the current tile's 32 stores go out.
template <int MAP>
__global__ __launch_bounds__(64) void copy_rows_pipelined(const float* __restrict__ x, float* __restrict__ y, size_t n, size_t chunk,
                                                          size_t groups, int delay) {
    extern __shared__ float dummy[];
    const int lane = threadIdx.x;
    const size_t group = MAP ? blockIdx.x : (size_t)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);  // MAP 1: launch order
    if (group >= groups) return;
    if (lane == 0 && n == 1) dummy[0] = 0;
    v4 cur[32], nxt[32];
    const size_t row0 = group * 64 + lane / 32;
    const float* px = x + row0 * n + (lane % 32) * 4;
    float* py = y + row0 * n + (lane % 32) * 4;
#pragma unroll
    for (int k = 0; k < 32; ++k) nxt[k] = __builtin_nontemporal_load((const v4*)(px + (size_t)(2 * k) * n));
    for (size_t t0 = 0; t0 < n; t0 += 128) {
#pragma unroll
        for (int k = 0; k < 32; ++k) cur[k] = nxt[k];
        if (t0 + 128 < n) {
#pragma unroll
            for (int k = 0; k < 32; ++k) nxt[k] = __builtin_nontemporal_load((const v4*)(px + (size_t)(2 * k) * n + t0 + 128));
        }
        for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(16);
#pragma unroll
        for (int k = 0; k < 32; ++k) __builtin_nontemporal_store(cur[k], (v4*)(py + (size_t)(2 * k) * n + t0));
    }
}

// ROWS rows per wave (one lane per row would use only ROWS of the 64 lanes), SEG samples per row and tile: fewer, longer streams
template <int SEG, int ROWS>
__global__ __launch_bounds__(64) void copy_rows_few(const float* __restrict__ x, float* __restrict__ y, size_t n, size_t chunk, size_t groups) {
    extern __shared__ float dummy[];
    const int lane = threadIdx.x;
    const size_t group = (size_t)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (group >= groups) return;
    constexpr int LPR = SEG / 4 < 64 ? SEG / 4 : 64, RPI = 64 / LPR, IPR = SEG / 4 / LPR, ITERS = ROWS / RPI * IPR;
    if (lane == 0 && n == 1) dummy[0] = 0;
    for (size_t t0 = 0; t0 < n; t0 += SEG) {
        for (int base = 0; base < ITERS; base += 16) {
            v4 r[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int it = base + k;
                const size_t row = group * ROWS + (it / IPR) * RPI + lane / LPR;
                r[k] = __builtin_nontemporal_load((const v4*)(x + row * n + t0 + (size_t)(it % IPR) * 256 + (lane % LPR) * 4));
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int it = base + k;
                const size_t row = group * ROWS + (it / IPR) * RPI + lane / LPR;
                __builtin_nontemporal_store(r[k], (v4*)(y + row * n + t0 + (size_t)(it % IPR) * 256 + (lane % LPR) * 4));
            }
        }
    }
}

int main(int argc, char** argv) {
    const size_t total = (size_t)1 << 28;
    void *a, *b; CK(hipMalloc(&a, total * 4 + (1 << 20))); CK(hipMalloc(&b, total * 4 + (1 << 20))); CK(hipMemset(a, 1, total * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int lds = argc > 1 ? atoi(argv[1]) : 30000;  // unused LDS: two one-wave workgroups per CU, as the batch IIR runs
    const size_t lens[] = {3840, 4096, 4224, 5120, 9984, 16384};
    for (size_t n : lens) {
        const size_t rows = total / n / 64 * 64, groups = rows / 64, chunk = (groups + 7) / 8;
        const double bytes = 2.0 * rows * n * 4;
        auto run = [&](const char* name, auto launch) {
            for (int i = 0; i < 30; ++i) launch();
            (void)hipEventRecord(e0); for (int i = 0; i < 10; ++i) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            printf("n %6zu rows %7zu %-28s %.4f ms  %.3f TB/s\n", n, rows, name, ms, bytes / ms / 1e9);
            return 0;
        };
        const float* A = (const float*)a; float* B = (float*)b;
        run("512 B segments, 64 adjacent rows", [&]{ hipLaunchKernelGGL((copy_rows<128, 0>), dim3(8 * chunk), dim3(64), lds, 0, A, B, n, chunk, groups); });
        run("512 B segments, interleaved rows", [&]{ hipLaunchKernelGGL((copy_rows<128, 1>), dim3(8 * chunk), dim3(64), lds, 0, A, B, n, chunk, groups); });
        run("1 KB segments, 64 adjacent rows", [&]{ hipLaunchKernelGGL((copy_rows<256, 0>), dim3(8 * chunk), dim3(64), lds, 0, A, B, n, chunk, groups); });
        for (int delay : {0, 2, 4})
            run(delay == 0 ? "pipelined tiles, no delay" : (delay == 2 ? "pipelined tiles, 2 x s_sleep 16" : "pipelined tiles, 4 x s_sleep 16"),
                [&]{ hipLaunchKernelGGL((copy_rows_pipelined<0>), dim3(8 * chunk), dim3(64), lds, 0, A, B, n, chunk, groups, delay); });
        run("pipelined tiles, launch-order groups", [&]{ hipLaunchKernelGGL((copy_rows_pipelined<1>), dim3(8 * chunk), dim3(64), lds, 0, A, B, n, chunk, groups, 2); });
        {
            const size_t g32 = rows / 32, c32 = (g32 + 7) / 8, g16 = rows / 16, c16 = (g16 + 7) / 8;
            run("32 rows x 1 KB per wave", [&]{ hipLaunchKernelGGL((copy_rows_few<256, 32>), dim3(8 * c32), dim3(64), lds, 0, A, B, n, c32, g32); });
            run("16 rows x 2 KB per wave", [&]{ hipLaunchKernelGGL((copy_rows_few<512, 16>), dim3(8 * c16), dim3(64), lds, 0, A, B, n, c16, g16); });
        }
        CK(hipGetLastError());
    }
    return 0;
}
