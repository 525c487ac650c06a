// clock_ramp.hip -- shader clock launch by launch while the library's row FFT runs C2 from a cold start (dev tool, round 3).
// After every dsp_gpu_fft_batch launch a one-wave probe kernel spins for ~4 us and compares s_memtime (shader clock
// cycles) with s_memrealtime (constant 100 MHz): the ratio is the clock the chip is running at just then.  HIP events
// give each launch's duration.  Run once per DSP_GPU_FFT_ROW_WGS_PER_CU setting (the library reads it once).
//   hipcc -O3 --offload-arch=gfx950 -o scripts/clock_ramp scripts/clock_ramp.hip -Ldsp_amd/lib -ldsp_gpu -Wl,-rpath,'$ORIGIN/../dsp_amd/lib'
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../include/dsp_gpu.h"

__global__ void clock_probe(float *mhz) {
    const unsigned long long r0 = __builtin_readcyclecounter();  // s_memtime: shader clock
    const unsigned long long t0 = wall_clock64();                // s_memrealtime: 100 MHz
    unsigned long long t1 = t0;
    while (t1 - t0 < 400) t1 = wall_clock64();                   // ~4 us
    const unsigned long long r1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) *mhz = static_cast<float>(static_cast<double>(r1 - r0) / static_cast<double>(t1 - t0) * 100.0);
}

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    const int launches = argc > 1 ? std::atoi(argv[1]) : 80;
    const size_t batch = 65536, n = 4096;
    float *in, *out, *mhz;
    CK(hipMalloc(&in, batch * n * 8));
    CK(hipMalloc(&out, batch * n * 8));
    CK(hipMalloc(&mhz, sizeof(float) * launches));
    std::vector<float> h(1 << 22);
    unsigned long long st = 88172645463325252ull;
    for (float &v : h) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        v = static_cast<float>(static_cast<double>(st >> 11) / 9007199254740992.0 - 0.5) * 3.0f;
    }
    for (size_t off = 0; off < batch * n * 2; off += h.size()) CK(hipMemcpy(in + off, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    if (dsp_gpu_fft_batch(in, out, 12, 1, DSP_GPU_FORWARD, DSP_GPU_MEM_DEVICE, nullptr) != 0) {  // first-call set-up, one transform
        printf("fft: %s\n", dsp_gpu_last_error_string());
        return 1;
    }
    CK(hipDeviceSynchronize());
    std::vector<hipEvent_t> ev(launches + 1);
    for (hipEvent_t &e : ev) CK(hipEventCreate(&e));
    for (int pass = 0; pass < 2; ++pass) {
        std::this_thread::sleep_for(std::chrono::milliseconds(1500));  // let the chip fall idle
        CK(hipEventRecord(ev[0], 0));
        for (int i = 0; i < launches; ++i) {
            if (dsp_gpu_fft_batch(in, out, 12, batch, DSP_GPU_FORWARD, DSP_GPU_MEM_DEVICE, nullptr) != 0) return 1;
            hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, 0, mhz + i);
            CK(hipEventRecord(ev[i + 1], 0));
        }
        CK(hipDeviceSynchronize());
        std::vector<float> f(launches);
        CK(hipMemcpy(f.data(), mhz, sizeof(float) * launches, hipMemcpyDeviceToHost));
        printf("pass %d (after 1.5 s idle), DSP_GPU_FFT_ROW_WGS_PER_CU=%s: launch, ms (FFT + 4 us probe), shader MHz right after it\n", pass,
               getenv("DSP_GPU_FFT_ROW_WGS_PER_CU") ? getenv("DSP_GPU_FFT_ROW_WGS_PER_CU") : "(default 3)");
        double sum25 = 0, sum_tail = 0;
        for (int i = 0; i < launches; ++i) {
            float ms = 0;
            CK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            if (i < 25) sum25 += ms;
            if (i >= launches - 20) sum_tail += ms;
            if (i < 12 || i % 5 == 0 || i >= launches - 3) printf("  %3d  %.4f  %6.0f\n", i, ms, f[i]);
        }
        printf("  mean of launches 0-24: %.4f ms; mean of the last 20: %.4f ms\n", sum25 / 25, sum_tail / 20);
    }
    return 0;
}
