// iir_scan_ablate.hip -- where does iir_scan_kernel spend its time?  (dev tool, evidence for DESIGN.md 4.5)
// Compiles the library's own iir_kernels.hip with its ablation switches enabled and times C4b (one channel of 2^28
// samples, 4-pole Chebyshev low-pass, fc = 0.1) with the kernel's phases switched off.  With all three off the kernel is
// its memory skeleton: load tile -> LDS -> store.
//   make -C dsp_amd/csrc && hipcc -std=c++17 -O3 --offload-arch=gfx950 -ffp-contract=off -DDSPGPU_SCAN_ABLATE -c \
//       -o /tmp/iir_scan_ablate.o scripts/iir_scan_ablate.hip && hipcc --offload-arch=gfx950 -o scripts/iir_scan_ablate \
//       /tmp/iir_scan_ablate.o $(ls dsp_amd/build/*.o | grep -v iir_kernels)
#include "../dsp_amd/csrc/iir_kernels.hip"

#include <cstdio>
#include <vector>

int main() {
    const size_t n = static_cast<size_t>(1) << 28;
    // recursive::lowPassFilter<4>(0.1) (tests/golden: cheby_lp4_a / _b)
    const double a[5] = {0.0027807568676184863, 0.011123027470473947, 0.01668454120571092, 0.011123027470473947,
                         0.0027807568676184863};
    const double b[5] = {0.0, 2.764030504704419, -3.1228526783585413, 1.6645530241054276, -0.3502229603332012};
    float *x = nullptr, *y = nullptr;
    if (hipMalloc(&x, n * sizeof(float)) != hipSuccess || hipMalloc(&y, n * sizeof(float)) != hipSuccess) return 1;
    {
        std::vector<float> h(1 << 24);
        unsigned s = 12345u;
        for (float &v : h) {  // full-mantissa values: low-entropy data clocks higher
            s = s * 1664525u + 1013904223u;
            v = (static_cast<float>(s >> 8) - 8388608.0f) * (1.0f / 8388608.0f);
        }
        for (size_t off = 0; off < n; off += h.size())
            if (hipMemcpy(x + off, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return 1;
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int variants[] = {0, 1, 2, 4, 3, 7, 0};
    const char *names[] = {"everything", "no phase A (row recursion)", "no phase B (wave scan)", "no phase C (start-state correction)",
                           "no A, no B", "skeleton: load -> LDS -> store", "everything (again)"};
    for (int v = 0; v < 7; ++v) {
        if (hipMemcpyToSymbol(HIP_SYMBOL(dspgpu::g_scan_ablate), &variants[v], sizeof(int)) != hipSuccess) return 1;
        for (int i = 0; i < 3; ++i)
            if (dsp_gpu_iir(x, y, 1, n, a, b, 4, 1.0, 0.0, nullptr, DSP_GPU_IIR_SCAN, DSP_GPU_MEM_DEVICE, nullptr)) {
                std::printf("dsp_gpu_iir failed: %s\n", dsp_gpu_last_error_string());
                return 1;
            }
        hipEventRecord(e0, nullptr);
        const int iters = 20;
        for (int i = 0; i < iters; ++i) dsp_gpu_iir(x, y, 1, n, a, b, 4, 1.0, 0.0, nullptr, DSP_GPU_IIR_SCAN, DSP_GPU_MEM_DEVICE, nullptr);
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms = 0.0f;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= iters;
        std::printf("%-40s %7.4f ms  %6.1f GSamples/s  %.3f of 8 TB/s\n", names[v], ms, n / ms / 1e6, n * 8.0 / ms / 8e9);
    }
    return 0;
}
