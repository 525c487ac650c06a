// Microbenchmark (dev tool): does the access width of the stores / loads matter at equal bytes per thread?
//   hipcc -O3 --offload-arch=gfx950 -o scripts/copy_patterns5 scripts/copy_patterns5.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// 64 B per thread, tile of 16 KB per workgroup: loads of LW floats, stores of SW floats (through registers)
template <int LW, int SW>
__global__ __launch_bounds__(256) void copyw(const float *in, float *out, unsigned tiles) {
    const unsigned t = (blockIdx.x & 7) * (tiles >> 3) + (blockIdx.x >> 3);
    const float *s = in + (size_t)t * 4096;
    float *d = out + (size_t)t * 4096;
    float r[16];
    if constexpr (LW == 4) {
#pragma unroll
        for (int m = 0; m < 4; ++m) { f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(s) + threadIdx.x + 256 * m); r[4*m]=v.x; r[4*m+1]=v.y; r[4*m+2]=v.z; r[4*m+3]=v.w; }
    } else {
#pragma unroll
        for (int m = 0; m < 16; ++m) r[m] = __builtin_nontemporal_load(s + threadIdx.x + 256 * m);
    }
    if constexpr (SW == 4) {
#pragma unroll
        for (int m = 0; m < 4; ++m) { f4 v = {r[4*m], r[4*m+1], r[4*m+2], r[4*m+3]}; __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(d) + threadIdx.x + 256 * m); }
    } else if constexpr (SW == 2) {
#pragma unroll
        for (int m = 0; m < 8; ++m) { f2 v = {r[2*m], r[2*m+1]}; __builtin_nontemporal_store(v, reinterpret_cast<f2 *>(d) + threadIdx.x + 256 * m); }
    } else {
#pragma unroll
        for (int m = 0; m < 16; ++m) __builtin_nontemporal_store(r[m], d + threadIdx.x + 256 * m);
    }
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    float *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned tiles = bytes / 16384;
    auto time = [&](const char *name, auto launch) -> int {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
        printf("%-44s %.4f ms  %.3f TB/s (read + write)\n", name, ms, 2.0 * bytes / ms / 1e9);
        return 0;
    };
#define RUN(name, K) time(name, [&] { hipLaunchKernelGGL(K, dim3(tiles), dim3(256), 0, 0, in, out, tiles); })
    RUN("warm-up", (copyw<4, 4>));
    RUN("16-byte loads, 16-byte stores", (copyw<4, 4>));
    RUN("16-byte loads,  8-byte stores", (copyw<4, 2>));
    RUN("16-byte loads,  4-byte stores", (copyw<4, 1>));
    RUN(" 4-byte loads, 16-byte stores", (copyw<1, 4>));
    RUN(" 4-byte loads,  4-byte stores", (copyw<1, 1>));
    return 0;
}
