// os_quad_ablate.hip -- the overlap-save wave kernel with TWO waves per SIMD (dev tool, round 3).
// The 64 x 64 transpose of fft_wave_core.h in its QUAD form needs 16.9 KB per wave, so eight waves share a CU.
// Checks the QUAD transform against the WIDE one, then times the interior path of C3 with parts switched off.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=fast -o scripts/os_quad_ablate scripts/os_quad_ablate.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "../dsp_amd/csrc/fft_wave_core.h"

using namespace dspgpu;

constexpr int kHalf = kWaveN / 2 + 1, kHalfPad = (kHalf + 7) & ~7;

__device__ __forceinline__ c32 cmul_conj(c32 a, c32 b) {
    c32 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
        : "=&v"(r)
        : "v"(a), "v"(b));
    return r;
}

// one wave: the same 4096 points through the WIDE and the QUAD transform
__global__ __launch_bounds__(64) void check_kernel(const c32 *in, c32 *wide, c32 *quad) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[kWaveLdsWideBytes];
    const int lane = threadIdx.x;
    c32 r[64];
    for (int m = 0; m < 64; ++m) r[m] = in[lane + 64 * m];
    fft4096_wave<+1, true>(r, reinterpret_cast<float *>(lds), lane);
    for (int m = 0; m < 64; ++m) wide[lane + 64 * m] = r[m];
    for (int m = 0; m < 64; ++m) r[m] = in[lane + 64 * m];
    fft4096_wave_quad<+1>(r, reinterpret_cast<float *>(lds), lane);
    for (int m = 0; m < 64; ++m) quad[lane + 64 * m] = r[m];
}

// MODE bits: 1 = global loads, 2 = global stores, 4 = transforms, 8 = spectrum product
template <int MODE, int WAVES, int I0>
__global__ __launch_bounds__(WAVES * 64) void osq_kernel(const float *__restrict__ x, float *__restrict__ y, size_t n,
                                                         const c32 *__restrict__ hspec, unsigned bpc, size_t tiles,
                                                         size_t chunk, float *sink, unsigned *ctr = nullptr) {
    constexpr int T = 64, P = 64;
    constexpr long long L = kWaveN - 256 * I0;
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    c32 *hs = reinterpret_cast<c32 *>(lds);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    float *img = reinterpret_cast<float *>(lds + kHalfPad * sizeof(c32) + wave * kWaveLdsQuadBytes);
    for (int i = threadIdx.x; i < kHalf; i += WAVES * 64) hs[i] = hspec[i];
    __syncthreads();
    const size_t xcd = blockIdx.x & 7u, wpx = static_cast<size_t>(gridDim.x >> 3) * WAVES;
    const c32 *hlo = hs + lane, *hhi = hs + (kWaveN / 2 - lane);
    float acc = 0.0f;
    // ctr != nullptr: every wave DRAWS its next pair from its XCD's counter instead of striding (the order of the pairs then
    // follows the order in which waves become free, as a grid of one-pair workgroups would)
    for (size_t t = static_cast<size_t>(blockIdx.x >> 3) * WAVES + wave;; t += wpx) {
        if (ctr) {
            unsigned d = 0;
            if (lane == 0) d = atomicAdd(ctr + 32 * xcd, 1u);
            t = __builtin_amdgcn_readfirstlane(d);
        }
        if (t >= chunk) break;
        const size_t tile = xcd * chunk + t;
        if (tile >= tiles) break;
        const size_t qa = 2 * tile, qb = qa + 1;
        const size_t ch_a = qa / bpc, ch_b = qb / bpc;
        const long long in_a = static_cast<long long>(qa % bpc) * L, in_b = static_cast<long long>(qb % bpc) * L;
        const bool interior = ch_a == ch_b && in_b + kWaveN <= static_cast<long long>(n);
        if (!interior) continue;  // the ablation only walks interior pairs of one channel
        const float *pa = x + ch_a * n + in_a + lane, *pb = x + ch_b * n + in_b + lane;
        c32 r[P];
        if constexpr (MODE & 1) {
            float a[P], b[P];
#pragma unroll
            for (int m = 0; m < P; ++m) a[m] = __builtin_nontemporal_load(pa + T * m);
#pragma unroll
            for (int m = 4 * I0; m < P; ++m) b[m] = __builtin_nontemporal_load(pb + T * m);
#pragma unroll
            for (int m = 0; m < P; ++m) r[m] = mk(a[m], m < 4 * I0 ? a[m + P - 4 * I0] : b[m]);
        } else {
#pragma unroll
            for (int m = 0; m < P; ++m) r[m] = mk(static_cast<float>(lane + m) * 1e-3f, static_cast<float>(tile & 255) * 1e-3f + m);
        }
        if constexpr (MODE & 4) fft4096_wave_quad<+1>(r, img, lane);
        if constexpr (MODE & 8) {
#pragma unroll
            for (int m = 0; m < P; ++m) {
                if (m < P / 2) r[m] = cmul(r[m], hlo[T * m]);
                else r[m] = cmul_conj(r[m], hhi[-T * (m - P / 2)]);
            }
        }
        if constexpr (MODE & 4) fft4096_wave_quad<-1>(r, img, lane);
        if constexpr (MODE & 2) {
            float *ya = y + ch_a * n + in_a + 4 * lane, *yb = y + ch_b * n + in_b + 4 * lane;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int m = 4 * I0; m < P; ++m) img[(m - 4 * I0) * 64 + lane] = half ? r[m].y : r[m].x;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                f4 v[16 - I0];
#pragma unroll
                for (int i = I0; i < 16; ++i) v[i - I0] = *reinterpret_cast<const f4 *>(img + 256 * (i - I0) + 4 * lane);
#pragma unroll
                for (int i = I0; i < 16; ++i) __builtin_nontemporal_store(v[i - I0], reinterpret_cast<f4 *>((half ? yb : ya) + 256 * i));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        } else {
#pragma unroll
            for (int m = 0; m < P; ++m) acc += r[m].x + r[m].y;
        }
    }
    if (!(MODE & 2) && acc == 1.2345f) sink[0] = acc;
}

// Late prefetch: the next pair's samples are loaded into the registers that the relayout of the outputs has just
// freed (the real parts go to the image first, then the imaginary parts), so the loads run under the stores.
template <int WAVES, int I0, int STAGE>
__global__ __launch_bounds__(WAVES * 64) void osq_late_kernel(const float *__restrict__ x, float *__restrict__ y, size_t n,
                                                              const c32 *__restrict__ hspec, unsigned bpc, size_t tiles,
                                                              size_t chunk, float *sink) {
    constexpr int T = 64, P = 64;
    constexpr long long L = kWaveN - 256 * I0;
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    c32 *hs = reinterpret_cast<c32 *>(lds);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    float *img = reinterpret_cast<float *>(lds + kHalfPad * sizeof(c32) + wave * kWaveLdsQuadBytes);
    for (int i = threadIdx.x; i < kHalf; i += WAVES * 64) hs[i] = hspec[i];
    __syncthreads();
    const size_t xcd = blockIdx.x & 7u, wpx = static_cast<size_t>(gridDim.x >> 3) * WAVES;
    const c32 *hlo = hs + lane, *hhi = hs + (kWaveN / 2 - lane);
    size_t t = static_cast<size_t>(blockIdx.x >> 3) * WAVES + wave;
    // next interior pair at or after t (element offsets of its two blocks); false when the range is exhausted
    auto locate = [&](size_t &oa, size_t &ob) {
        for (; t < chunk; t += wpx) {
            const size_t tile = xcd * chunk + t;
            if (tile >= tiles) return false;
            const size_t qa = 2 * tile, qb = qa + 1;
            const size_t ch_a = qa / bpc, ch_b = qb / bpc;
            const long long in_a = static_cast<long long>(qa % bpc) * L, in_b = static_cast<long long>(qb % bpc) * L;
            if (ch_a == ch_b && in_b + kWaveN <= static_cast<long long>(n)) {
                oa = ch_a * n + in_a;
                ob = ch_b * n + in_b;
                t += wpx;
                return true;
            }
        }
        return false;
    };
    size_t oa, ob;
    bool have = locate(oa, ob);
    c32 r[P];
    if (have) {
#pragma unroll
        for (int m = 0; m < P; ++m) r[m].x = __builtin_nontemporal_load(x + oa + lane + T * m);
#pragma unroll
        for (int m = 4 * I0; m < P; ++m) r[m].y = __builtin_nontemporal_load(x + ob + lane + T * m);
    }
    while (have) {
#pragma unroll
        for (int m = 0; m < 4 * I0; ++m) r[m].y = r[m + P - 4 * I0].x;
        float *ya = y + oa + 4 * lane, *yb = y + ob + 4 * lane;
        fft4096_wave_quad<+1>(r, img, lane);
#pragma unroll
        for (int m = 0; m < P; ++m) {
            if (m < P / 2) r[m] = cmul(r[m], hlo[T * m]);
            else r[m] = cmul_conj(r[m], hhi[-T * (m - P / 2)]);
        }
        fft4096_wave_quad<-1>(r, img, lane);
        have = locate(oa, ob);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int m = 4 * I0; m < P; ++m) img[(m - 4 * I0) * 64 + lane] = half ? r[m].y : r[m].x;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (have) {
                if (half == 0) {
#pragma unroll
                    for (int m = 0; m < P; ++m) r[m].x = __builtin_nontemporal_load(x + oa + lane + T * m);
                } else {
#pragma unroll
                    for (int m = 4 * I0; m < P; ++m) r[m].y = __builtin_nontemporal_load(x + ob + lane + T * m);
                }
            }
#pragma unroll
            for (int i0 = I0; i0 < 16; i0 += STAGE) {
                f4 v[STAGE];
#pragma unroll
                for (int i = i0; i < i0 + STAGE && i < 16; ++i) v[i - i0] = *reinterpret_cast<const f4 *>(img + 256 * (i - I0) + 4 * lane);
#pragma unroll
                for (int i = i0; i < i0 + STAGE && i < 16; ++i) __builtin_nontemporal_store(v[i - i0], reinterpret_cast<f4 *>((half ? yb : ya) + 256 * i));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// Sequential walk: a wave owns a contiguous run of block pairs of ONE channel and keeps the 256 * I0 samples that the
// next pair's first block shares with this pair's second block in registers (read amplification 1.0).
// MODE bits as in osq_kernel.
template <int MODE, int WAVES, int I0>
__global__ __launch_bounds__(WAVES * 64) void osq_seq_kernel(const float *__restrict__ x, float *__restrict__ y, size_t n,
                                                             const c32 *__restrict__ hspec, unsigned bpc, unsigned runs_per_channel,
                                                             unsigned pairs_per_run, size_t total_runs, float *sink) {
    constexpr int T = 64, P = 64;
    constexpr long long L = kWaveN - 256 * I0;
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    c32 *hs = reinterpret_cast<c32 *>(lds);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    float *img = reinterpret_cast<float *>(lds + kHalfPad * sizeof(c32) + wave * kWaveLdsQuadBytes);
    for (int i = threadIdx.x; i < kHalf; i += WAVES * 64) hs[i] = hspec[i];
    __syncthreads();
    const c32 *hlo = hs + lane, *hhi = hs + (kWaveN / 2 - lane);
    const size_t run = static_cast<size_t>(blockIdx.x) * WAVES + wave;
    if (run >= total_runs) return;
    const size_t ch = run / runs_per_channel;
    const unsigned first_pair = static_cast<unsigned>(run % runs_per_channel) * pairs_per_run;
    float acc = 0.0f;
    float keep[4 * I0];  // the rows that the next pair's first block shares with this pair's second block
#pragma unroll
    for (int m = 0; m < 4 * I0; ++m) keep[m] = 0.0f;
    for (unsigned pp = 0; pp < pairs_per_run; ++pp) {
        const unsigned pair = first_pair + pp;
        const long long in_a = static_cast<long long>(2 * pair) * L, in_b = in_a + L;
        if (2 * pair + 1 >= bpc || in_b + kWaveN > static_cast<long long>(n)) break;
        const float *pa = x + ch * n + in_a + lane, *pb = x + ch * n + in_b + lane;
        c32 r[P];
        if constexpr (MODE & 1) {
            if (pp == 0) {
#pragma unroll
                for (int m = 0; m < 4 * I0; ++m) keep[m] = __builtin_nontemporal_load(pa + T * m);
            }
#pragma unroll
            for (int m = 0; m < 4 * I0; ++m) r[m].x = keep[m];
#pragma unroll
            for (int m = 4 * I0; m < P; ++m) r[m].x = __builtin_nontemporal_load(pa + T * m);
#pragma unroll
            for (int m = 4 * I0; m < P; ++m) r[m].y = __builtin_nontemporal_load(pb + T * m);
#pragma unroll
            for (int m = 0; m < 4 * I0; ++m) r[m].y = r[m + P - 4 * I0].x;
#pragma unroll
            for (int m = 0; m < 4 * I0; ++m) keep[m] = r[m + P - 4 * I0].y;
        } else {
#pragma unroll
            for (int m = 0; m < P; ++m) r[m] = mk(static_cast<float>(lane + m) * 1e-3f, static_cast<float>(pair & 255) * 1e-3f + m);
        }
        if constexpr (MODE & 4) fft4096_wave_quad<+1>(r, img, lane);
        if constexpr (MODE & 8) {
#pragma unroll
            for (int m = 0; m < P; ++m) {
                if (m < P / 2) r[m] = cmul(r[m], hlo[T * m]);
                else r[m] = cmul_conj(r[m], hhi[-T * (m - P / 2)]);
            }
        }
        if constexpr (MODE & 4) fft4096_wave_quad<-1>(r, img, lane);
        if constexpr (MODE & 2) {
            float *ya = y + ch * n + in_a + 4 * lane, *yb = y + ch * n + in_b + 4 * lane;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int m = 4 * I0; m < P; ++m) img[(m - 4 * I0) * 64 + lane] = half ? r[m].y : r[m].x;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                f4 v[16 - I0];
#pragma unroll
                for (int i = I0; i < 16; ++i) v[i - I0] = *reinterpret_cast<const f4 *>(img + 256 * (i - I0) + 4 * lane);
#pragma unroll
                for (int i = I0; i < 16; ++i) __builtin_nontemporal_store(v[i - I0], reinterpret_cast<f4 *>((half ? yb : ya) + 256 * i));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        } else {
#pragma unroll
            for (int m = 0; m < P; ++m) acc += r[m].x + r[m].y;
        }
    }
    if (!(MODE & 2) && acc == 1.2345f) sink[0] = acc;
}

template <int MODE, int WAVES>
int run_seq(const char *name, const float *x, float *y, size_t n, size_t channels, const c32 *h, float *sink, unsigned runs_per_channel) {
    constexpr int I0 = 2;
    const long long L = kWaveN - 256 * I0;
    const int ntaps = 511;
    const unsigned bpc = static_cast<unsigned>((n - (ntaps - 1) + L - 1) / L);
    const unsigned pairs = (bpc + 1) / 2, pairs_per_run = (pairs + runs_per_channel - 1) / runs_per_channel;
    const size_t total_runs = channels * runs_per_channel;
    const size_t lds = kHalfPad * sizeof(c32) + static_cast<size_t>(WAVES) * kWaveLdsQuadBytes;
    hipFuncSetAttribute(reinterpret_cast<const void *>(osq_seq_kernel<MODE, WAVES, I0>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const unsigned wgs = static_cast<unsigned>((total_runs + WAVES - 1) / WAVES);
    for (int it = 0; it < 2; ++it)
        hipLaunchKernelGGL((osq_seq_kernel<MODE, WAVES, I0>), dim3(wgs), dim3(WAVES * 64), lds, 0, x, y, n, h, bpc, runs_per_channel, pairs_per_run, total_runs, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    const int iters = 10;
    for (int it = 0; it < iters; ++it)
        hipLaunchKernelGGL((osq_seq_kernel<MODE, WAVES, I0>), dim3(wgs), dim3(WAVES * 64), lds, 0, x, y, n, h, bpc, runs_per_channel, pairs_per_run, total_runs, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s waves/wg %d  wgs %u runs/ch %u  %.3f ms (%s)\n", name, WAVES, wgs, runs_per_channel, ms / iters, hipGetErrorString(hipGetLastError()));
    fflush(stdout);
    return 0;
}

template <int WAVES, int STAGE>
int run_late(const char *name, const float *x, float *y, size_t n, size_t channels, const c32 *h, float *sink) {
    constexpr int I0 = 2;
    const long long L = kWaveN - 256 * I0;
    const int ntaps = 511;
    const unsigned bpc = static_cast<unsigned>((n - (ntaps - 1) + L - 1) / L);
    const size_t total = channels * bpc, tiles = (total + 1) / 2, chunk = (tiles + 7) / 8;
    const size_t lds = kHalfPad * sizeof(c32) + static_cast<size_t>(WAVES) * kWaveLdsQuadBytes;
    hipFuncSetAttribute(reinterpret_cast<const void *>(osq_late_kernel<WAVES, I0, STAGE>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it)
        hipLaunchKernelGGL((osq_late_kernel<WAVES, I0, STAGE>), dim3(256), dim3(WAVES * 64), lds, 0, x, y, n, h, bpc, tiles, chunk, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    const int iters = 10;
    for (int it = 0; it < iters; ++it)
        hipLaunchKernelGGL((osq_late_kernel<WAVES, I0, STAGE>), dim3(256), dim3(WAVES * 64), lds, 0, x, y, n, h, bpc, tiles, chunk, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s waves/wg %d  lds %zu  %.3f ms (%s)\n", name, WAVES, lds, ms / iters, hipGetErrorString(hipGetLastError()));
    fflush(stdout);
    return 0;
}

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE, int WAVES>
int run(const char *name, const float *x, float *y, size_t n, size_t channels, const c32 *h, float *sink, bool dyn = false) {
    static unsigned *ctr_buf = nullptr;
    if (!ctr_buf) CK(hipMalloc(&ctr_buf, 8 * 32 * sizeof(unsigned)));
    unsigned *ctr = dyn ? ctr_buf : nullptr;
    constexpr int I0 = 2;
    const long long L = kWaveN - 256 * I0;
    const int ntaps = 511;
    const unsigned bpc = static_cast<unsigned>((n - (ntaps - 1) + L - 1) / L);
    const size_t total = channels * bpc, tiles = (total + 1) / 2, chunk = (tiles + 7) / 8;
    const size_t lds = kHalfPad * sizeof(c32) + static_cast<size_t>(WAVES) * kWaveLdsQuadBytes;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(osq_kernel<MODE, WAVES, I0>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int wgs = 256;
    auto launch = [&] {
        if (ctr) (void)hipMemsetAsync(ctr, 0, 8 * 32 * sizeof(unsigned), 0);
        hipLaunchKernelGGL((osq_kernel<MODE, WAVES, I0>), dim3(wgs), dim3(WAVES * 64), lds, 0, x, y, n, h, bpc, tiles, chunk, sink, ctr);
    };
    for (int it = 0; it < 2; ++it) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    const int iters = 10;
    for (int it = 0; it < iters; ++it) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s waves/wg %d  lds %zu  %.3f ms\n", name, WAVES, lds, ms / iters);
    fflush(stdout);
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
    unsigned long long st = 88172645463325252ull;  // full-mantissa pseudo-random samples
    for (size_t i = 0; i < hx.size(); ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        hx[i] = static_cast<float>(static_cast<double>(st >> 11) / 9007199254740992.0 - 0.5) * 3.0f;
    }
    for (size_t off = 0; off < channels * n; off += hx.size()) CK(hipMemcpy(x + off, hx.data(), hx.size() * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(h, hx.data(), kWaveN * sizeof(c32), hipMemcpyHostToDevice));

    {  // QUAD against WIDE
        c32 *dw, *dq;
        CK(hipMalloc(&dw, kWaveN * sizeof(c32)));
        CK(hipMalloc(&dq, kWaveN * sizeof(c32)));
        hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, 0, h, dw, dq);
        CK(hipDeviceSynchronize());
        std::vector<float> w(2 * kWaveN), q(2 * kWaveN);
        CK(hipMemcpy(w.data(), dw, w.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(q.data(), dq, q.size() * 4, hipMemcpyDeviceToHost));
        double num = 0, den = 0;
        for (size_t i = 0; i < w.size(); ++i) num += (w[i] - q[i]) * (double)(w[i] - q[i]), den += w[i] * (double)w[i];
        printf("QUAD vs WIDE transform: rel L2 %.3g (|wide| %.3g)\n", std::sqrt(num / den), std::sqrt(den));
        if (!(num / den < 1e-12)) { printf("MISMATCH\n"); return 2; }
    }
    int rc = 0;
    rc |= run<12, 8>("(warm-up)", x, y, n, channels, h, sink);
    rc |= run<12, 8>("compute only, 8 waves/CU", x, y, n, channels, h, sink);
    rc |= run<12, 4>("compute only, 4 waves/CU", x, y, n, channels, h, sink);
    rc |= run<4, 8>("transforms only, 8 waves/CU", x, y, n, channels, h, sink);
    rc |= run<3, 8>("loads + stores only, 8 waves/CU", x, y, n, channels, h, sink);
    rc |= run<3, 4>("loads + stores only, 4 waves/CU", x, y, n, channels, h, sink);
    rc |= run<15, 8>("full, 8 waves/CU", x, y, n, channels, h, sink);
    rc |= run<15, 4>("full, 4 waves/CU", x, y, n, channels, h, sink);
    rc |= run<3, 8>("loads + stores only, 8 waves/CU, pairs drawn", x, y, n, channels, h, sink, true);
    rc |= run<15, 8>("full, 8 waves/CU, pairs drawn", x, y, n, channels, h, sink, true);
    rc |= run<15, 8>("full, 8 waves/CU (again)", x, y, n, channels, h, sink);
    rc |= run<15, 8>("full, 8 waves/CU, pairs drawn (again)", x, y, n, channels, h, sink, true);
    rc |= run<15, 6>("full, 6 waves/CU, pairs drawn", x, y, n, channels, h, sink, true);
    rc |= run<15, 4>("full, 4 waves/CU, pairs drawn", x, y, n, channels, h, sink, true);
    rc |= run<3, 6>("loads + stores only, 6 waves/CU, pairs drawn", x, y, n, channels, h, sink, true);
    rc |= run<15, 8>("full, 8 waves/CU, pairs drawn (third)", x, y, n, channels, h, sink, true);
    rc |= run_seq<3, 8>("seq walk, loads + stores only, 8 waves", x, y, n, channels, h, sink, 2);
    rc |= run_seq<3, 8>("seq walk, loads + stores only, 8 waves", x, y, n, channels, h, sink, 8);
    rc |= run_seq<3, 4>("seq walk, loads + stores only, 4 waves", x, y, n, channels, h, sink, 1);
    rc |= run_seq<15, 8>("seq walk, full, 8 waves", x, y, n, channels, h, sink, 2);
    rc |= run_seq<15, 8>("seq walk, full, 8 waves", x, y, n, channels, h, sink, 4);
    rc |= run_seq<15, 8>("seq walk, full, 8 waves", x, y, n, channels, h, sink, 8);
    for (unsigned rpc : {16u, 32u, 49u, 74u, 147u}) {  // short runs down to one pair per wave: many workgroups, dispatched as CUs free up
        rc |= run_seq<3, 8>("seq walk, loads + stores only, 8 waves", x, y, n, channels, h, sink, rpc);
        rc |= run_seq<15, 8>("seq walk, full, 8 waves", x, y, n, channels, h, sink, rpc);
    }
    rc |= run_late<8, 7>("late prefetch, 8 waves/CU, 7 staged", x, y, n, channels, h, sink);
    rc |= run_late<8, 14>("late prefetch, 8 waves/CU, 14 staged", x, y, n, channels, h, sink);
    rc |= run_late<4, 7>("late prefetch, 4 waves/CU, 7 staged", x, y, n, channels, h, sink);
    return rc;
}
