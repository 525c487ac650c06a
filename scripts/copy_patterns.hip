// Microbenchmark (dev tool, not product): HBM ceilings of the access patterns the FFT kernel could use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

// pattern 1: as fft_lds_kernel: tile of 4096 float2 per WG, thread t moves elements t + 256 m (8 B each)
__global__ __launch_bounds__(256) void copy_f2(const float2* in, float2* out) {
    const float2* s = in + (size_t)blockIdx.x * 4096; float2* d = out + (size_t)blockIdx.x * 4096;
    float2 r[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) r[m] = s[threadIdx.x + 256 * m];
#pragma unroll
    for (int m = 0; m < 16; ++m) d[threadIdx.x + 256 * m] = r[m];
}
// pattern 2: 16-byte accesses, lane pairs: even lane rows m<8, odd lane rows m>=8 (two 512 B runs per wave instr)
__global__ __launch_bounds__(256) void copy_f4_pair(const float4* in, float4* out) {
    const float4* s = in + (size_t)blockIdx.x * 2048; float4* d = out + (size_t)blockIdx.x * 2048;
    const int t = threadIdx.x, p = t >> 1, odd = t & 1;
    float4 r[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) r[m] = s[p + 128 * (m + 8 * odd)];
#pragma unroll
    for (int m = 0; m < 8; ++m) d[p + 128 * (m + 8 * odd)] = r[m];
}
// pattern 3: 16-byte accesses, thread t moves float4 t + 256 m (1 KiB contiguous per wave instr)
__global__ __launch_bounds__(256) void copy_f4(const float4* in, float4* out) {
    const float4* s = in + (size_t)blockIdx.x * 2048; float4* d = out + (size_t)blockIdx.x * 2048;
    float4 r[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) r[m] = s[threadIdx.x + 256 * m];
#pragma unroll
    for (int m = 0; m < 8; ++m) d[threadIdx.x + 256 * m] = r[m];
}
// pattern 4: grid-stride float4 copy, 2048 blocks
__global__ __launch_bounds__(256) void copy_gs(const float4* in, float4* out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
// pattern 5: nontemporal float4
__global__ __launch_bounds__(256) void copy_f4_nt(const float4* in, float4* out) {
    const float4* s = in + (size_t)blockIdx.x * 2048; float4* d = out + (size_t)blockIdx.x * 2048;
    float4 r[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) { const float* q = (const float*)(s + threadIdx.x + 256 * m);
        r[m] = make_float4(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q+1), __builtin_nontemporal_load(q+2), __builtin_nontemporal_load(q+3)); }
#pragma unroll
    for (int m = 0; m < 8; ++m) { float* q = (float*)(d + threadIdx.x + 256 * m);
        __builtin_nontemporal_store(r[m].x, q); __builtin_nontemporal_store(r[m].y, q+1); __builtin_nontemporal_store(r[m].z, q+2); __builtin_nontemporal_store(r[m].w, q+3); }
}
int main() {
    const size_t tiles = 65536, bytes = tiles * 4096 * 8;
    void *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-14s %.4f ms  %.3f TB/s\n", name, ms, 2.0 * bytes / ms / 1e9);
    };
    for (int rep = 0; rep < 2; ++rep) {
    run("f2 strided", [&]{ hipLaunchKernelGGL(copy_f2, dim3(tiles), dim3(256), 0, 0, (const float2*)a, (float2*)b); });
    run("f4 pair", [&]{ hipLaunchKernelGGL(copy_f4_pair, dim3(tiles), dim3(256), 0, 0, (const float4*)a, (float4*)b); });
    run("f4 tile", [&]{ hipLaunchKernelGGL(copy_f4, dim3(tiles), dim3(256), 0, 0, (const float4*)a, (float4*)b); });
    run("f4 gridstride", [&]{ hipLaunchKernelGGL(copy_gs, dim3(2048), dim3(256), 0, 0, (const float4*)a, (float4*)b, bytes / 16); });
    run("f4 nt", [&]{ hipLaunchKernelGGL(copy_f4_nt, dim3(tiles), dim3(256), 0, 0, (const float4*)a, (float4*)b); });
    }
    return 0;
}
