// fft_wave_ab.hip -- A/B for C2 (65,536 x 4096-point complex fp32 FFT): ONE WAVE PER TRANSFORM (fft_wave_core.h: two
// radix-64 passes in a lane's registers, one 64 x 64 transpose, no workgroup barrier) against the library's row kernel
// (256 threads per transform, radix 16 x 16 x 16, two exchanges through LDS behind barriers).  VERDICT r2 item 7a.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=fast -o scripts/fft_wave_ab scripts/fft_wave_ab.hip
// The library kernel itself is timed by `python bench.py --workload fft` in the same gpurun call.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "../dsp_amd/csrc/fft_wave_core.h"

using namespace dspgpu;

// WAVES per workgroup; QUAD: 16.9 KB of LDS per wave (eight waves per CU), else the WIDE image (33 KB, four per CU)
template <int WAVES, bool QUAD, bool PERSISTENT>
__global__ __launch_bounds__(WAVES * 64) void fft_wave_rows(const c32 *__restrict__ in, c32 *__restrict__ out, size_t batch,
                                                            size_t chunk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    float *img = reinterpret_cast<float *>(lds + static_cast<size_t>(wave) * (QUAD ? kWaveLdsQuadBytes : kWaveLdsWideBytes));
    const size_t xcd = blockIdx.x & 7u;
    const size_t per_xcd = static_cast<size_t>(gridDim.x >> 3) * WAVES;
    for (size_t t = static_cast<size_t>(blockIdx.x >> 3) * WAVES + wave; t < chunk; t += per_xcd) {
        const size_t b = xcd * chunk + t;
        if (b >= batch) break;
        const c32 *src = in + b * kWaveN + lane;
        c32 *dst = out + b * kWaveN + lane;
        c32 r[64];
#pragma unroll
        for (int m = 0; m < 64; ++m) r[m] = load_stream(src + 64 * m);
        if constexpr (QUAD) fft4096_wave_quad<+1>(r, img, lane);
        else fft4096_wave<+1, true>(r, img, lane);
#pragma unroll
        for (int m = 0; m < 64; ++m) store_stream(dst + 64 * m, r[m]);
        if (!PERSISTENT) break;
    }
}

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

template <int WAVES, bool QUAD, bool PERSISTENT>
int run(const char *name, const c32 *in, c32 *out, size_t batch, int wgs_per_cu) {
    const size_t lds = static_cast<size_t>(WAVES) * (QUAD ? kWaveLdsQuadBytes : kWaveLdsWideBytes);
    auto k = fft_wave_rows<WAVES, QUAD, PERSISTENT>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    const size_t chunk = (batch + 7) / 8;
    const unsigned wgs = PERSISTENT ? 256u * wgs_per_cu : static_cast<unsigned>(8 * ((chunk + WAVES - 1) / WAVES));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(k, dim3(wgs), dim3(WAVES * 64), lds, 0, in, out, batch, chunk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    const int iters = 40;
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(k, dim3(wgs), dim3(WAVES * 64), lds, 0, in, out, batch, chunk);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    printf("%-58s %u wgs x %d waves, lds %zu: %.4f ms = %.1f GSamples/s = %.3f of 8 TB/s\n", name, wgs, WAVES, lds, ms,
           batch * 4096.0 / ms / 1e6, batch * 4096.0 * 16 / ms / 1e6 / 8000.0);
    fflush(stdout);
    return 0;
}

int main() {
    const size_t batch = 65536;
    c32 *in, *out;
    CK(hipMalloc(&in, batch * kWaveN * sizeof(c32)));
    CK(hipMalloc(&out, batch * kWaveN * sizeof(c32)));
    std::vector<float> h(1 << 22);
    unsigned long long st = 88172645463325252ull;
    for (float &v : h) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        v = static_cast<float>(static_cast<double>(st >> 11) / 9007199254740992.0 - 0.5) * 3.0f;
    }
    for (size_t off = 0; off < batch * kWaveN * 2; off += h.size())
        CK(hipMemcpy(reinterpret_cast<float *>(in) + off, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    int rc = 0;
    rc |= run<4, true, false>("(warm-up)", in, out, batch, 0);
    // correctness of what is timed: transform 5 against a direct DFT of a few bins in double
    {
        std::vector<float> x(2 * kWaveN), y(2 * kWaveN);
        CK(hipMemcpy(x.data(), in + 5 * kWaveN, x.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(y.data(), out + 5 * kWaveN, y.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int k : {0, 1, 63, 64, 2047, 4095}) {
            double re = 0, im = 0;
            for (int n = 0; n < kWaveN; ++n) {
                const double a = 2.0 * M_PI * ((static_cast<long long>(k) * n) % kWaveN) / kWaveN;  // FORWARD = e^{+i...} (fft.h:22-29)
                re += x[2 * n] * std::cos(a) - x[2 * n + 1] * std::sin(a);
                im += x[2 * n] * std::sin(a) + x[2 * n + 1] * std::cos(a);
            }
            worst = std::fmax(worst, std::hypot(re - y[2 * k], im - y[2 * k + 1]) / std::hypot(re, im));
        }
        printf("wave kernel vs direct DFT (6 bins of one transform): worst relative error %.2e\n", worst);
        if (!(worst < 1e-4)) return 2;
    }
    rc |= run<4, true, false>("QUAD, 4-wave workgroups, one transform per wave", in, out, batch, 0);
    rc |= run<8, true, false>("QUAD, 8-wave workgroups, one transform per wave", in, out, batch, 0);
    rc |= run<2, true, false>("QUAD, 2-wave workgroups, one transform per wave", in, out, batch, 0);
    rc |= run<8, true, true>("QUAD, persistent, 8 waves per CU", in, out, batch, 1);
    rc |= run<4, true, true>("QUAD, persistent, 2 x 4 waves per CU", in, out, batch, 2);
    rc |= run<4, false, false>("WIDE, 4-wave workgroups (one per CU), one transform per wave", in, out, batch, 0);
    rc |= run<4, false, true>("WIDE, persistent, 4 waves per CU", in, out, batch, 1);
    return rc;
}
