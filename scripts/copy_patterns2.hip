// Microbenchmark round 2 (dev tool): looking for the best read+write streaming pattern on MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
template <typename T> __device__ inline T ntload(const T* p);
template <> __device__ inline float2 ntload(const float2* p) { const float* q=(const float*)p; return make_float2(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q+1)); }
__device__ inline void ntstore(float2 v, float2* p) { float* q=(float*)p; __builtin_nontemporal_store(v.x,q); __builtin_nontemporal_store(v.y,q+1); }

template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_f2(const float2* in, float2* out) {
    const float2* s = in + (size_t)blockIdx.x * 4096; float2* d = out + (size_t)blockIdx.x * 4096;
    float2 r[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) r[m] = NTL ? ntload(s + threadIdx.x + 256 * m) : s[threadIdx.x + 256 * m];
#pragma unroll
    for (int m = 0; m < 16; ++m) { if (NTS) ntstore(r[m], d + threadIdx.x + 256 * m); else d[threadIdx.x + 256 * m] = r[m]; }
}
// persistent: grid = CUs*k, each WG loops over tiles
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_f2_persist(const float2* in, float2* out, unsigned tiles) {
    for (unsigned t = blockIdx.x; t < tiles; t += gridDim.x) {
        const float2* s = in + (size_t)t * 4096; float2* d = out + (size_t)t * 4096;
        float2 r[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) r[m] = NTL ? ntload(s + threadIdx.x + 256 * m) : s[threadIdx.x + 256 * m];
#pragma unroll
        for (int m = 0; m < 16; ++m) { if (NTS) ntstore(r[m], d + threadIdx.x + 256 * m); else d[threadIdx.x + 256 * m] = r[m]; }
    }
}
// XCD-contiguous mapping: block b -> tile (b % 8) * (tiles/8) + b / 8, so each XCD streams one contiguous eighth
__global__ __launch_bounds__(256) void copy_f2_xcd(const float2* in, float2* out, unsigned tiles) {
    const unsigned t = (blockIdx.x & 7) * (tiles >> 3) + (blockIdx.x >> 3);
    const float2* s = in + (size_t)t * 4096; float2* d = out + (size_t)t * 4096;
    float2 r[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) r[m] = s[threadIdx.x + 256 * m];
#pragma unroll
    for (int m = 0; m < 16; ++m) d[threadIdx.x + 256 * m] = r[m];
}
__global__ __launch_bounds__(256) void read_f2(const float2* in, float2* out) {
    const float2* s = in + (size_t)blockIdx.x * 4096;
    float2 acc = make_float2(0,0);
#pragma unroll
    for (int m = 0; m < 16; ++m) { float2 v = s[threadIdx.x + 256 * m]; acc.x += v.x; acc.y += v.y; }
    if (acc.x == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void write_f2(float2* out) {
    float2* d = out + (size_t)blockIdx.x * 4096;
#pragma unroll
    for (int m = 0; m < 16; ++m) d[threadIdx.x + 256 * m] = make_float2(1.f, 2.f);
}
int main() {
    const size_t tiles = 65536, bytes = tiles * 4096 * 8;
    void *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, double nbytes, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        (void)hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-22s %.4f ms  %.3f TB/s\n", name, ms, nbytes / ms / 1e9);
    };
    const float2* A = (const float2*)a; float2* B = (float2*)b;
    run("f2", 2.0*bytes, [&]{ hipLaunchKernelGGL((copy_f2<false,false>), dim3(tiles), dim3(256), 0, 0, A, B); });
    run("f2 ntl", 2.0*bytes, [&]{ hipLaunchKernelGGL((copy_f2<true,false>), dim3(tiles), dim3(256), 0, 0, A, B); });
    run("f2 nts", 2.0*bytes, [&]{ hipLaunchKernelGGL((copy_f2<false,true>), dim3(tiles), dim3(256), 0, 0, A, B); });
    run("f2 ntl+nts", 2.0*bytes, [&]{ hipLaunchKernelGGL((copy_f2<true,true>), dim3(tiles), dim3(256), 0, 0, A, B); });
    for (int k : {2, 4, 8}) { char nm[64]; snprintf(nm, 64, "f2 persist x%d", k);
        run(nm, 2.0*bytes, [&]{ hipLaunchKernelGGL((copy_f2_persist<false,false>), dim3(256*k), dim3(256), 0, 0, A, B, (unsigned)tiles); }); }
    run("f2 persist x4 nt", 2.0*bytes, [&]{ hipLaunchKernelGGL((copy_f2_persist<true,true>), dim3(1024), dim3(256), 0, 0, A, B, (unsigned)tiles); });
    run("f2 xcd-contig", 2.0*bytes, [&]{ hipLaunchKernelGGL(copy_f2_xcd, dim3(tiles), dim3(256), 0, 0, A, B, (unsigned)tiles); });
    run("read only", 1.0*bytes, [&]{ hipLaunchKernelGGL(read_f2, dim3(tiles), dim3(256), 0, 0, A, B); });
    run("write only", 1.0*bytes, [&]{ hipLaunchKernelGGL(write_f2, dim3(tiles), dim3(256), 0, 0, B); });
    run("in-place f2", 2.0*bytes, [&]{ hipLaunchKernelGGL((copy_f2<false,false>), dim3(tiles), dim3(256), 0, 0, (const float2*)b, B); });
    return 0;
}
