// dispatch_rate.hip -- how fast can the chip START workgroups?  (dev tool, round 3)
// The IIR scan launches one one-wave workgroup per 4096-sample tile: 65,536+ workgroups of 64 threads with 17.9 KB of LDS and
// 96 registers for a 0.38-0.43 ms kernel.  This times kernels with the same launch shape and NO work, at 1, 2 and 4 waves
// per workgroup (same number of waves in all), to see how much of that time is the dispatcher's.
//   hipcc -O3 --offload-arch=gfx950 -o scripts/dispatch_rate scripts/dispatch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void empty_kernel(float *out, int spin) {
    __shared__ float lds[WAVES * 4480];  // 17,920 B per wave
    if (spin < 0) lds[threadIdx.x] = out[threadIdx.x];  // never true: keeps the allocation
    if (threadIdx.x == 0 && spin < 0) out[blockIdx.x] = lds[0];
}

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

template <int WAVES>
int run(float *out, unsigned waves) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned wgs = waves / WAVES;
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(empty_kernel<WAVES>, dim3(wgs), dim3(64 * WAVES), 0, 0, out, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(empty_kernel<WAVES>, dim3(wgs), dim3(64 * WAVES), 0, 0, out, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%u waves as %u workgroups of %d wave(s), 17.9 KB LDS per wave, no work: %.4f ms per launch = %.0f workgroups/us = %.0f waves/us\n",
           waves, wgs, WAVES, ms / 20, wgs / (ms / 20 * 1e3), waves / (ms / 20 * 1e3));
    return 0;
}

int main() {
    float *out;
    CK(hipMalloc(&out, 1 << 20));
    for (unsigned waves : {65536u, 69912u, 262144u}) {
        if (run<1>(out, waves) || run<2>(out, waves) || run<4>(out, waves)) return 1;
    }
    return 0;
}
