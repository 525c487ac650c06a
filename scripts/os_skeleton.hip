// os_skeleton.hip -- what limits the MEMORY side of the overlap-save wave kernel?  (dev tool, round 3)
//
// C3's shape (1,024 channels x 2^20 samples, hop 3584 = I0 2): every wave moves what fir_os_wave2_kernel moves per block
// pair -- 2 x 4096 samples in (the 512 samples two blocks share once), 2 x 3584 samples out -- and does nothing else.
// Variants: the kernel's own access shapes (dword loads at lane stride 4 B, 16-byte stores re-laid out through LDS) against
// 16-byte loads and direct 16-byte stores of the same bytes; persistent strided / sequential walks against one pair per
// wave and one workgroup per eight pairs; 8 / 4 / 2 waves per workgroup (= per CU for the persistent forms).
//
//   hipcc -O3 --offload-arch=gfx950 -o scripts/os_skeleton scripts/os_skeleton.hip && scripts/os_skeleton
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kN = 4096, kI0 = 2, kHop = kN - 256 * kI0;

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

// one block pair: WIDE_LD = 16-byte loads (lane l takes samples 4l .. 4l+3 of every 256), else dword loads (lane l takes
// sample l of every 64); VIA_LDS = outputs re-laid out through a wave-private image as the kernel does, else stored as loaded
template <bool WIDE_LD, bool VIA_LDS>
__device__ __forceinline__ void move_pair(const float *__restrict__ xa, float *__restrict__ ya, int lane, float *img) {
    const float *xb = xa + kHop;
    float *yb = ya + kHop;
    if constexpr (WIDE_LD) {
        f4 a[16], b[16 - kI0];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(xa + 256 * i) + lane);
#pragma unroll
        for (int i = kI0; i < 16; ++i) b[i - kI0] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(xb + 256 * i) + lane);
#pragma unroll
        for (int i = kI0; i < 16; ++i) __builtin_nontemporal_store(a[i], reinterpret_cast<f4 *>(ya + 256 * i) + lane);
#pragma unroll
        for (int i = kI0; i < 16; ++i) __builtin_nontemporal_store(b[i - kI0], reinterpret_cast<f4 *>(yb + 256 * i) + lane);
    } else {
        float a[64], b[64];
#pragma unroll
        for (int m = 0; m < 64; ++m) a[m] = __builtin_nontemporal_load(xa + lane + 64 * m);
#pragma unroll
        for (int m = 4 * kI0; m < 64; ++m) b[m] = __builtin_nontemporal_load(xb + lane + 64 * m);
#pragma unroll
        for (int m = 0; m < 4 * kI0; ++m) b[m] = a[m + 64 - 4 * kI0];
        if constexpr (VIA_LDS) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int m = 4 * kI0; m < 64; ++m) img[(m - 4 * kI0) * 64 + lane] = half ? b[m] : a[m];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                f4 v[16 - kI0];
#pragma unroll
                for (int i = kI0; i < 16; ++i) v[i - kI0] = *reinterpret_cast<const f4 *>(img + 256 * (i - kI0) + 4 * lane);
#pragma unroll
                for (int i = kI0; i < 16; ++i) __builtin_nontemporal_store(v[i - kI0], reinterpret_cast<f4 *>((half ? yb : ya) + 256 * i) + lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        } else {
#pragma unroll
            for (int m = 4 * kI0; m < 64; ++m) __builtin_nontemporal_store(a[m], ya + lane + 64 * m);
#pragma unroll
            for (int m = 4 * kI0; m < 64; ++m) __builtin_nontemporal_store(b[m], yb + lane + 64 * m);
        }
    }
}

// WALK 0: persistent, each XCD's waves stride through its eighth of the pairs; 1: persistent, each wave walks a run of
// consecutive pairs; 2: one pair per wave, the grid covers all pairs (workgroup w takes pairs 8w .. 8w+7, XCD-contiguous);
// 3: persistent, every wave draws its next pair (or batch of `run_len` consecutive pairs) from its XCD's counter
template <bool WIDE_LD, bool VIA_LDS, int WAVES, int WALK>
__global__ __launch_bounds__(WAVES * 64) void skel_kernel(const float *__restrict__ x, float *__restrict__ y, size_t n, unsigned ppc,
                                                          size_t pairs, size_t chunk, unsigned run_len, unsigned *ctr = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    float *img = lds + wave * (64 * 56);
    auto at = [&](size_t pair) { return (pair / ppc) * n + static_cast<size_t>(pair % ppc) * 2 * kHop; };  // ppc pairs per channel
    if constexpr (WALK == 0) {
        const size_t xcd = blockIdx.x & 7u, wpx = static_cast<size_t>(gridDim.x >> 3) * WAVES;
        for (size_t t = static_cast<size_t>(blockIdx.x >> 3) * WAVES + wave; t < chunk; t += wpx) {
            const size_t pair = xcd * chunk + t;
            if (pair >= pairs) break;
            move_pair<WIDE_LD, VIA_LDS>(x + at(pair), y + at(pair), lane, img);
        }
    } else if constexpr (WALK == 1) {
        const size_t run = (blockIdx.x & 7u) * chunk + static_cast<size_t>(blockIdx.x >> 3) * WAVES + wave;  // chunk = runs per XCD here
        for (unsigned p = 0; p < run_len; ++p) {
            const size_t pair = run * run_len + p;
            if (pair >= pairs) break;
            move_pair<WIDE_LD, VIA_LDS>(x + at(pair), y + at(pair), lane, img);
        }
    } else if constexpr (WALK == 3) {
        const size_t xcd = blockIdx.x & 7u;
        for (;;) {
            unsigned t = 0;
            if (lane == 0) t = atomicAdd(ctr + 32 * xcd, run_len);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t >= chunk) break;
            for (unsigned p = 0; p < run_len; ++p) {
                const size_t pair = xcd * chunk + t + p;
                if (t + p >= chunk || pair >= pairs) break;
                move_pair<WIDE_LD, VIA_LDS>(x + at(pair), y + at(pair), lane, img);
            }
        }
    } else {
        const size_t wg = (blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);  // chunk = workgroups per XCD here
        const size_t pair = wg * WAVES + wave;
        if (pair < pairs) move_pair<WIDE_LD, VIA_LDS>(x + at(pair), y + at(pair), lane, img);
    }
}

template <bool WIDE_LD, bool VIA_LDS, int WAVES, int WALK>
int run(const char *name, const float *x, float *y, size_t n, size_t channels, int wgs_per_cu = 1, unsigned batch = 1) {
    static unsigned *ctr = nullptr;
    if (!ctr) CK(hipMalloc(&ctr, 8 * 32 * sizeof(unsigned)));
    const unsigned ppc = static_cast<unsigned>((n - kN) / (2 * kHop));  // whole interior pairs of a channel
    const size_t pairs = channels * ppc;
    const size_t lds = VIA_LDS ? static_cast<size_t>(WAVES) * 64 * 56 * 4 : 0;
    auto k = skel_kernel<WIDE_LD, VIA_LDS, WAVES, WALK>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    size_t grid, chunk;
    unsigned run_len = 1;
    if (WALK == 0 || WALK == 3) {
        grid = 256 * wgs_per_cu;
        chunk = (pairs + 7) / 8;
        run_len = batch;
    } else if (WALK == 1) {
        grid = 256 * wgs_per_cu;
        const size_t waves = grid * WAVES;
        run_len = static_cast<unsigned>((pairs + waves - 1) / waves);
        chunk = (grid / 8) * WAVES;
    } else {
        const size_t wgs = (pairs + WAVES - 1) / WAVES;
        chunk = (wgs + 7) / 8;
        grid = 8 * chunk;
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto launch = [&] {
        if (WALK == 3) (void)hipMemsetAsync(ctr, 0, 8 * 32 * sizeof(unsigned), 0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), lds, 0, x, y, n, ppc, pairs, chunk, run_len, ctr);
    };
    for (int it = 0; it < 3; ++it) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    const int iters = 20;
    for (int it = 0; it < iters; ++it) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    const double bytes = static_cast<double>(pairs) * (2.0 * kN - 256 * kI0 + 2.0 * kHop) * 4.0;
    printf("%-78s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
    fflush(stdout);
    return 0;
}

int main() {
    const size_t channels = 1024, n = 1 << 20;
    float *x, *y;
    CK(hipMalloc(&x, channels * n * sizeof(float)));
    CK(hipMalloc(&y, channels * n * sizeof(float)));
    std::vector<float> hx(1 << 22);
    unsigned long long st = 88172645463325252ull;
    for (size_t i = 0; i < hx.size(); ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        hx[i] = static_cast<float>(static_cast<double>(st >> 11) / 9007199254740992.0 - 0.5) * 3.0f;
    }
    for (size_t off = 0; off < channels * n; off += hx.size()) CK(hipMemcpy(x + off, hx.data(), hx.size() * sizeof(float), hipMemcpyHostToDevice));
    int rc = 0;
    rc |= run<false, true, 8, 0>("(warm-up)", x, y, n, channels);
    rc |= run<false, true, 8, 0>("kernel's shapes (dword loads, stores via LDS), strided walk, 8 waves/CU", x, y, n, channels);
    rc |= run<false, true, 8, 1>("kernel's shapes, sequential runs, 8 waves/CU", x, y, n, channels);
    rc |= run<false, true, 4, 1>("kernel's shapes, sequential runs, 4 waves/CU", x, y, n, channels);
    rc |= run<false, true, 2, 1>("kernel's shapes, sequential runs, 2 waves/CU", x, y, n, channels);
    rc |= run<false, false, 8, 1>("dword loads, dword stores, sequential runs, 8 waves/CU", x, y, n, channels);
    rc |= run<true, false, 8, 1>("16-byte loads, 16-byte stores, sequential runs, 8 waves/CU", x, y, n, channels);
    rc |= run<true, false, 4, 1>("16-byte loads, 16-byte stores, sequential runs, 4 waves/CU", x, y, n, channels);
    rc |= run<true, false, 2, 1>("16-byte loads, 16-byte stores, sequential runs, 2 waves/CU", x, y, n, channels);
    rc |= run<true, false, 8, 0>("16-byte loads, 16-byte stores, strided walk, 8 waves/CU", x, y, n, channels);
    rc |= run<false, true, 8, 2>("kernel's shapes, one pair per wave, 8-wave workgroups", x, y, n, channels);
    rc |= run<false, true, 4, 2>("kernel's shapes, one pair per wave, 4-wave workgroups", x, y, n, channels);
    rc |= run<false, true, 1, 2>("kernel's shapes, one pair per wave, 1-wave workgroups", x, y, n, channels);
    rc |= run<true, false, 8, 2>("16-byte loads, 16-byte stores, one pair per wave, 8-wave workgroups", x, y, n, channels);
    rc |= run<true, false, 1, 2>("16-byte loads, 16-byte stores, one pair per wave, 1-wave workgroups", x, y, n, channels);
    rc |= run<true, false, 4, 1>("16-byte loads, 16-byte stores, sequential runs, 4 waves x 2 workgroups/CU", x, y, n, channels, 2);
    rc |= run<false, true, 8, 3>("kernel's shapes, pairs drawn from a counter per XCD, 8 waves/CU", x, y, n, channels, 1, 1);
    rc |= run<false, true, 8, 3>("kernel's shapes, batches of 2 pairs drawn from a counter, 8 waves/CU", x, y, n, channels, 1, 2);
    rc |= run<false, true, 8, 3>("kernel's shapes, batches of 4 pairs drawn from a counter, 8 waves/CU", x, y, n, channels, 1, 4);
    rc |= run<false, true, 8, 3>("kernel's shapes, batches of 8 pairs drawn from a counter, 8 waves/CU", x, y, n, channels, 1, 8);
    rc |= run<false, true, 4, 3>("kernel's shapes, pairs drawn from a counter per XCD, 4 waves/CU", x, y, n, channels, 1, 1);
    rc |= run<false, true, 8, 0>("kernel's shapes, strided walk, 8 waves/CU (again)", x, y, n, channels);
    rc |= run<false, true, 4, 2>("kernel's shapes, one pair per wave, 4-wave workgroups (again)", x, y, n, channels);
    return rc;
}
