// Microbenchmark (dev tool): is there a read+write streaming pattern on this MI355X that beats the 5.45-5.66 TB/s the
// kernels' skeletons reach?  MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy.
//   hipcc -O3 --offload-arch=gfx950 -o scripts/copy_patterns4 scripts/copy_patterns4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
typedef float f4 __attribute__((ext_vector_type(4)));

// V float4 per thread, all loads first; tile = 256 * V float4, XCD-contiguous tile map, NT on/off
template <int V, bool NT, bool XCD>
__global__ __launch_bounds__(256) void copy4(const f4 *in, f4 *out, unsigned tiles) {
    const unsigned t = XCD ? (blockIdx.x & 7) * (tiles >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const f4 *s = in + (size_t)t * 256 * V + threadIdx.x;
    f4 *d = out + (size_t)t * 256 * V + threadIdx.x;
    f4 r[V];
#pragma unroll
    for (int m = 0; m < V; ++m) r[m] = NT ? __builtin_nontemporal_load(s + 256 * m) : s[256 * m];
#pragma unroll
    for (int m = 0; m < V; ++m) { if (NT) __builtin_nontemporal_store(r[m], d + 256 * m); else d[256 * m] = r[m]; }
}
// persistent grid-stride, W workgroups per CU
template <int V, bool NT>
__global__ __launch_bounds__(256) void copy4_persist(const f4 *in, f4 *out, unsigned tiles) {
    const unsigned per_xcd = gridDim.x >> 3, chunk = tiles >> 3;
    for (unsigned i = blockIdx.x >> 3; i < chunk; i += per_xcd) {
        const unsigned t = (blockIdx.x & 7) * chunk + i;
        const f4 *s = in + (size_t)t * 256 * V + threadIdx.x;
        f4 *d = out + (size_t)t * 256 * V + threadIdx.x;
        f4 r[V];
#pragma unroll
        for (int m = 0; m < V; ++m) r[m] = NT ? __builtin_nontemporal_load(s + 256 * m) : s[256 * m];
#pragma unroll
        for (int m = 0; m < V; ++m) { if (NT) __builtin_nontemporal_store(r[m], d + 256 * m); else d[256 * m] = r[m]; }
    }
}
// each thread owns a contiguous 16 * V bytes?  no: each WAVE owns a contiguous 1 KB * V run (lane-contiguous float4 per row)
template <int V, bool NT>
__global__ __launch_bounds__(256) void copy4_waverun(const f4 *in, f4 *out, unsigned tiles) {
    const unsigned t = (blockIdx.x & 7) * (tiles >> 3) + (blockIdx.x >> 3);
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const f4 *s = in + (size_t)t * 256 * V + wave * 64 * V + lane;
    f4 *d = out + (size_t)t * 256 * V + wave * 64 * V + lane;
    f4 r[V];
#pragma unroll
    for (int m = 0; m < V; ++m) r[m] = NT ? __builtin_nontemporal_load(s + 64 * m) : s[64 * m];
#pragma unroll
    for (int m = 0; m < V; ++m) { if (NT) __builtin_nontemporal_store(r[m], d + 64 * m); else d[64 * m] = r[m]; }
}

// THREADS per workgroup as a parameter: separates "bytes per thread" from "bytes per workgroup"
template <int THREADS, int V>
__global__ __launch_bounds__(THREADS) void copy4_t(const f4 *in, f4 *out, unsigned tiles) {
    const unsigned t = (blockIdx.x & 7) * (tiles >> 3) + (blockIdx.x >> 3);
    const f4 *s = in + (size_t)t * THREADS * V + threadIdx.x;
    f4 *d = out + (size_t)t * THREADS * V + threadIdx.x;
    f4 r[V];
#pragma unroll
    for (int m = 0; m < V; ++m) r[m] = __builtin_nontemporal_load(s + THREADS * m);
#pragma unroll
    for (int m = 0; m < V; ++m) __builtin_nontemporal_store(r[m], d + THREADS * m);
}
// loads through LDS with a barrier before the stores (what every transform kernel does)
template <int V>
__global__ __launch_bounds__(256) void copy4_lds(const f4 *in, f4 *out, unsigned tiles) {
    __shared__ f4 img[256 * V];
    const unsigned t = (blockIdx.x & 7) * (tiles >> 3) + (blockIdx.x >> 3);
    const f4 *s = in + (size_t)t * 256 * V + threadIdx.x;
    f4 *d = out + (size_t)t * 256 * V + threadIdx.x;
    f4 r[V];
#pragma unroll
    for (int m = 0; m < V; ++m) r[m] = __builtin_nontemporal_load(s + 256 * m);
#pragma unroll
    for (int m = 0; m < V; ++m) img[threadIdx.x + 256 * m] = r[m];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < V; ++m) r[m] = img[(threadIdx.x + 256 * m) ^ 1];
#pragma unroll
    for (int m = 0; m < V; ++m) __builtin_nontemporal_store(r[m], d + 256 * m);
}

// 8 float4 per thread, loaded in batches of B with a wait after each batch (fewer bytes in flight per wave), then stored
template <int B>
__global__ __launch_bounds__(256) void copy4_batched(const f4 *in, f4 *out, unsigned tiles) {
    constexpr int V = 8;
    const unsigned t = (blockIdx.x & 7) * (tiles >> 3) + (blockIdx.x >> 3);
    const f4 *s = in + (size_t)t * 256 * V + threadIdx.x;
    f4 *d = out + (size_t)t * 256 * V + threadIdx.x;
    f4 r[V];
#pragma unroll
    for (int m0 = 0; m0 < V; m0 += B) {
#pragma unroll
        for (int m = m0; m < m0 + B; ++m) r[m] = __builtin_nontemporal_load(s + 256 * m);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int m = 0; m < V; ++m) __builtin_nontemporal_store(r[m], d + 256 * m);
}

int main() {
    const size_t bytes = (size_t)2 << 30;  // 2 GiB in, 2 GiB out
    f4 *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t n4 = bytes / 16;
    auto time = [&](const char *name, auto launch) -> int {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
        printf("%-40s %.4f ms  %.3f TB/s (read + write)\n", name, ms, 2.0 * bytes / ms / 1e9);
        return 0;
    };
#define RUN(name, K, T) time(name, [&] { hipLaunchKernelGGL(K, dim3(T), dim3(256), 0, 0, in, out, (unsigned)(T)); })
    time("warm-up", [&] { hipLaunchKernelGGL((copy4<4, true, true>), dim3(n4 / 1024), dim3(256), 0, 0, in, out, (unsigned)(n4 / 1024)); });
    RUN("f4 x1  nt xcd", (copy4<1, true, true>), n4 / 256);
    RUN("f4 x2  nt xcd", (copy4<2, true, true>), n4 / 512);
    RUN("f4 x4  nt xcd", (copy4<4, true, true>), n4 / 1024);
    RUN("f4 x8  nt xcd", (copy4<8, true, true>), n4 / 2048);
    RUN("f4 x16 nt xcd", (copy4<16, true, true>), n4 / 4096);
    RUN("f4 x4  plain xcd", (copy4<4, false, true>), n4 / 1024);
    RUN("f4 x4  nt linear", (copy4<4, true, false>), n4 / 1024);
    RUN("f4 x4  plain linear", (copy4<4, false, false>), n4 / 1024);
    RUN("f4 x8  nt waverun", (copy4_waverun<8, true>), n4 / 2048);
    RUN("f4 x16 nt waverun", (copy4_waverun<16, true>), n4 / 4096);
#define RUNT(name, TH, V) time(name, [&] { hipLaunchKernelGGL((copy4_t<TH, V>), dim3(n4 / (TH * V)), dim3(TH), 0, 0, in, out, (unsigned)(n4 / (TH * V))); })
    RUNT("64 thr x 8   (8 KB/wg, 128 B/thr)", 64, 8);
    RUNT("64 thr x 1   (1 KB/wg, 16 B/thr)", 64, 1);
    RUNT("128 thr x 2  (4 KB/wg, 32 B/thr)", 128, 2);
    RUNT("512 thr x 1  (8 KB/wg, 16 B/thr)", 512, 1);
    RUNT("1024 thr x 1 (16 KB/wg, 16 B/thr)", 1024, 1);
    RUNT("1024 thr x 2 (32 KB/wg, 32 B/thr)", 1024, 2);
    RUNT("512 thr x 4  (32 KB/wg, 64 B/thr)", 512, 4);
    RUNT("256 thr x 8  (32 KB/wg, 128 B/thr)", 256, 8);
    RUN("lds x8 (32 KB/wg, barrier)", (copy4_lds<8>), n4 / 2048);
    RUN("lds x2 (8 KB/wg, barrier)", (copy4_lds<2>), n4 / 512);
    // the same 32 KB-per-workgroup copy with its occupancy limited by a dynamic LDS allocation it never touches
    for (int kb : {20, 40, 54, 80, 160}) {
        char nm[64]; snprintf(nm, 64, "256 thr x 8, %d KB LDS (%d wg/cu)", kb, 160 / kb);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(copy4<8, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        time(nm, [&] { hipLaunchKernelGGL((copy4<8, true, true>), dim3(n4 / 2048), dim3(256), kb * 1024, 0, in, out, (unsigned)(n4 / 2048)); });
    }
    for (int kb : {20, 40, 80, 160}) {
        char nm[64]; snprintf(nm, 64, "256 thr x 16, %d KB LDS (%d wg/cu)", kb, 160 / kb);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(copy4<16, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        time(nm, [&] { hipLaunchKernelGGL((copy4<16, true, true>), dim3(n4 / 4096), dim3(256), kb * 1024, 0, in, out, (unsigned)(n4 / 4096)); });
    }
    for (int kb : {10, 20, 40, 80}) {
        char nm[64]; snprintf(nm, 64, "256 thr x 1, %d KB LDS (%d wg/cu)", kb, 160 / kb);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(copy4<1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        time(nm, [&] { hipLaunchKernelGGL((copy4<1, true, true>), dim3(n4 / 256), dim3(256), kb * 1024, 0, in, out, (unsigned)(n4 / 256)); });
    }
    for (int kb : {20, 40}) {
        char nm[64];
        snprintf(nm, 64, "x8 in batches of 4, %d wg/cu", 160 / kb);
        time(nm, [&] { hipLaunchKernelGGL((copy4_batched<4>), dim3(n4 / 2048), dim3(256), kb * 1024, 0, in, out, (unsigned)(n4 / 2048)); });
        snprintf(nm, 64, "x8 in batches of 2, %d wg/cu", 160 / kb);
        time(nm, [&] { hipLaunchKernelGGL((copy4_batched<2>), dim3(n4 / 2048), dim3(256), kb * 1024, 0, in, out, (unsigned)(n4 / 2048)); });
        snprintf(nm, 64, "x8 in batches of 1, %d wg/cu", 160 / kb);
        time(nm, [&] { hipLaunchKernelGGL((copy4_batched<1>), dim3(n4 / 2048), dim3(256), kb * 1024, 0, in, out, (unsigned)(n4 / 2048)); });
    }
    for (int w : {2, 4, 8}) {
        char nm[64]; snprintf(nm, 64, "f4 x4 nt persistent %d wg/cu", w);
        time(nm, [&] { hipLaunchKernelGGL((copy4_persist<4, true>), dim3(256 * w), dim3(256), 0, 0, in, out, (unsigned)(n4 / 1024)); });
        snprintf(nm, 64, "f4 x8 nt persistent %d wg/cu", w);
        time(nm, [&] { hipLaunchKernelGGL((copy4_persist<8, true>), dim3(256 * w), dim3(256), 0, 0, in, out, (unsigned)(n4 / 2048)); });
    }
    time("hipMemcpyAsync D2D", [&] { (void)hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0); });
    // smaller footprint: 256 MiB in + 256 MiB out (fits the 256 MB of infinity cache?)
    {
        const size_t small = (size_t)256 << 20, m4 = small / 16;
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((copy4<4, true, true>), dim3(m4 / 1024), dim3(256), 0, 0, in, out, (unsigned)(m4 / 1024));
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((copy4<4, true, true>), dim3(m4 / 1024), dim3(256), 0, 0, in, out, (unsigned)(m4 / 1024));
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 50;
        printf("%-40s %.4f ms  %.3f TB/s (read + write)\n", "f4 x4 nt xcd, 256 MiB + 256 MiB", ms, 2.0 * small / ms / 1e9);
    }
    return 0;
}
