// os_wave_ablate.hip -- where does the one-wave-per-pair overlap-save kernel spend its time?  (dev tool)
// Builds the interior path of fir_os_wave_kernel with parts switched off and times every variant on C3's shape.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=fast -o scripts/os_wave_ablate scripts/os_wave_ablate.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "../dsp_amd/csrc/fft_wave_core.h"

using namespace dspgpu;

constexpr int kThreadsWg = 512, kWaves = 8;
constexpr int kHalf = kWaveN / 2 + 1, kHalfPad = (kHalf + 7) & ~7;

__device__ __forceinline__ c32 cmul_conj(c32 a, c32 b) {
    c32 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
        : "=&v"(r)
        : "v"(a), "v"(b));
    return r;
}

// MODE bits: 1 = global loads, 2 = global stores, 4 = forward+inverse transforms, 8 = spectrum product
template <int MODE, int WAVES, bool NT = true>
__global__ __launch_bounds__(WAVES * 64) void os_kernel(const float *__restrict__ x, float *__restrict__ y, size_t n,
                                                        const c32 *__restrict__ hspec, int ntaps, unsigned bpc,
                                                        size_t tiles, size_t chunk, float *sink, long long L) {
    constexpr int T = 64, P = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    c32 *hs = reinterpret_cast<c32 *>(lds);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    float *img = reinterpret_cast<float *>(lds + kHalfPad * sizeof(c32)) + wave * kWaveLdsFloats;
    for (int i = threadIdx.x; i < kHalf; i += WAVES * 64) hs[i] = hspec[i];
    __syncthreads();
    const long long order = ntaps - 1;
    const size_t xcd = blockIdx.x & 7u, wpx = static_cast<size_t>(gridDim.x >> 3) * WAVES;
    const c32 *hlo = hs + lane, *hhi = hs + (kWaveN / 2 - lane);
    float acc = 0.0f;
    for (size_t t = static_cast<size_t>(blockIdx.x >> 3) * WAVES + wave; t < chunk; t += wpx) {
        const size_t tile = xcd * chunk + t;
        if (tile >= tiles) break;
        const size_t qa = 2 * tile, qb = qa + 1;
        const size_t ch_a = qa / bpc, ch_b = qb / bpc;
        const long long in_a = static_cast<long long>(qa % bpc) * L, in_b = static_cast<long long>(qb % bpc) * L;
        const bool interior = in_a + kWaveN <= static_cast<long long>(n) && in_b + kWaveN <= static_cast<long long>(n);
        if (!interior) continue;  // the ablation only walks interior pairs
        const float *pa = x + ch_a * n + in_a + lane, *pb = x + ch_b * n + in_b + lane;
        c32 r[P];
        if constexpr (MODE & 1) {
#pragma unroll
            for (int m = 0; m < P; ++m) r[m] = mk(__builtin_nontemporal_load(pa + T * m), __builtin_nontemporal_load(pb + T * m));
        } else {
#pragma unroll
            for (int m = 0; m < P; ++m) r[m] = mk(static_cast<float>(lane + m) * 1e-3f, static_cast<float>(tile & 255) * 1e-3f + m);
        }
        if constexpr (MODE & 4) fft4096_wave<+1>(r, img, lane);
        if constexpr (MODE & 8) {
#pragma unroll
            for (int m = 0; m < P; ++m) {
                if (m < P / 2) r[m] = cmul(r[m], hlo[T * m]);
                else r[m] = cmul_conj(r[m], hhi[-T * (m - P / 2)]);
            }
        }
        if constexpr (MODE & 4) fft4096_wave<-1>(r, img, lane);
        float *qa_ = y + ch_a * n + in_a + lane, *qb_ = y + ch_b * n + in_b + lane;
        if constexpr (MODE & 2) {
            const int keep0 = kWaveN - static_cast<int>(L);  // first kept output of the block
            int m0 = keep0 / T, l0 = keep0 % T;
            asm volatile("" : "+s"(m0), "+s"(l0));
#pragma unroll
            for (int m = 0; m < P; ++m) {
                if (m < m0) continue;
                if (m == m0 && lane < l0) continue;
                if constexpr (NT) {
                    __builtin_nontemporal_store(r[m].x, qa_ + T * m);
                    __builtin_nontemporal_store(r[m].y, qb_ + T * m);
                } else {
                    qa_[T * m] = r[m].x;
                    qb_[T * m] = r[m].y;
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < P; ++m) acc += r[m].x + r[m].y;
        }
    }
    if (!(MODE & 2) && acc == 1.2345f) sink[0] = acc;
}

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE, int WAVES, bool NT = true>
int run(const char *name, const float *x, float *y, size_t n, size_t channels, const c32 *h, float *sink, long long L = kWaveN - 510) {
    const int ntaps = 511;
    const unsigned bpc = static_cast<unsigned>((n - (ntaps - 1) + L - 1) / L);
    const size_t total = channels * bpc, tiles = (total + 1) / 2, chunk = (tiles + 7) / 8;
    const size_t lds = kHalfPad * sizeof(c32) + WAVES * kWaveLdsFloats * sizeof(float);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(os_kernel<MODE, WAVES, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int wgs = 256 * (8 / WAVES);
    for (int it = 0; it < 2; ++it)
        hipLaunchKernelGGL((os_kernel<MODE, WAVES, NT>), dim3(wgs), dim3(WAVES * 64), lds, 0, x, y, n, h, ntaps, bpc, tiles, chunk, sink, L);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    const int iters = 10;
    for (int it = 0; it < iters; ++it)
        hipLaunchKernelGGL((os_kernel<MODE, WAVES, NT>), dim3(wgs), dim3(WAVES * 64), lds, 0, x, y, n, h, ntaps, bpc, tiles, chunk, sink, L);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s waves/wg %d  %.3f ms\n", name, WAVES, ms / iters);
    return 0;
}

int main() {
    const size_t channels = 1024, n = 1 << 20;
    float *x, *y, *sink;
    c32 *h;
    CK(hipMalloc(&x, channels * n * sizeof(float)));
    CK(hipMalloc(&y, channels * n * sizeof(float)));
    CK(hipMalloc(&sink, 256));
    CK(hipMalloc(&h, kWaveN * sizeof(c32)));
    std::vector<float> hx(1 << 22);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = static_cast<float>((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f;
    for (size_t off = 0; off < channels * n; off += hx.size()) CK(hipMemcpy(x + off, hx.data(), hx.size() * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(h, hx.data(), kWaveN * sizeof(c32), hipMemcpyHostToDevice));
    int rc = 0;
    rc |= run<15, 8>("full", x, y, n, channels, h, sink);
    rc |= run<14, 8>("no global loads", x, y, n, channels, h, sink);
    rc |= run<13, 8>("no global stores", x, y, n, channels, h, sink);
    rc |= run<12, 8>("compute only (transforms + product)", x, y, n, channels, h, sink);
    rc |= run<4, 8>("transforms only", x, y, n, channels, h, sink);
    rc |= run<3, 8>("loads + stores only", x, y, n, channels, h, sink);
    rc |= run<1, 8>("loads only", x, y, n, channels, h, sink);
    rc |= run<15, 4>("full, 4 waves per workgroup (2 wg/CU)", x, y, n, channels, h, sink);
    rc |= run<3, 8>("loads + stores only, hop 3584", x, y, n, channels, h, sink, 3584);
    rc |= run<3, 8, false>("loads + plain stores only, hop 3586", x, y, n, channels, h, sink);
    rc |= run<3, 8, false>("loads + plain stores only, hop 3584", x, y, n, channels, h, sink, 3584);
    rc |= run<15, 8>("full, hop 3584", x, y, n, channels, h, sink, 3584);
    rc |= run<15, 8, false>("full, plain stores, hop 3584", x, y, n, channels, h, sink, 3584);
    rc |= run<15, 8, false>("full, plain stores, hop 3586", x, y, n, channels, h, sink);
    return rc;
}
