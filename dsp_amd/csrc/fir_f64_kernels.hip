// fir_f64_kernels.hip -- direct FIR convolution on DOUBLE samples for gfx950 (MI355X).
//
// Stands in for dsp::filter::SincFilter<>::apply(SampleBuffer<double>&) (reference filter.h:123-140) -- the
// instantiation the reference's own callers use (eval/windowed_sinc_filter/main.cpp:105, eval/filter_compare/main.cpp:11;
// sample_buffer.h:20 allows any floating type) -- with the taps in double as the reference keeps them (filter.h:57,
// dsp.h:47-79 return std::vector<double>):
//     y[j] = sum_{i=0..M} h[i] x[j-i]   for j >= M,   0 before that (filter.h:133)
// The same Toeplitz formulation as the fp32 kernel (fir_kernels.hip) on the DOUBLE matrix cores,
// v_mfma_f64_16x16x4_f64: a 16x16 tile holds 256 consecutive outputs y[j0 + 16c + r]; at step q of Q the lane
// (r = lane & 15, kq = lane >> 4) feeds  A[r][kq] = h[16q + rho(r) - (4kq + kk)]  and  B[kq][r] = x[j0 + 16r + (4kq + kk) - 16q]
// for kk = 0..3, so that matrix row r accumulates output row rho(r): D[r][c] = sum_i h[i] x[j0 + 16c + rho(r) - i], 16 taps
// per step.  rho(r) = 4 (r & 3) + (r >> 2) undoes the f64 result layout (D[i][j] sits in lane j + 16 (i & 3), register
// i >> 2 -- measured, scripts/mfma_f64_layout.hip; NOT the fp32 instruction's i = 4 (lane >> 4) + register), so that a
// lane's four registers are four CONSECUTIVE outputs y[j0 + 16 (lane & 15) + 4 (lane >> 4) + reg] and leave as 16-byte stores.  Products and sums
// are exact double fmas; only the ORDER of the sum differs from the reference's loop (i ascending, one store per step),
// so results agree to a few ulp (tests/test_fir_f64_gpu.py: <= 1e-14 relative L2).  This is the widening route, not a
// tuned one: the double matrix pipe delivers 78 TFLOP/s, a quarter of what bounds it is LDS-free arithmetic.
#include <algorithm>

#include <cstdlib>

#include "common.h"

namespace dspgpu {
namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int kThreads64 = 256;
constexpr int kWaves64 = kThreads64 / 64;
constexpr int kTile64 = 256;                             // outputs per 16x16 tile
constexpr int kAcc64 = 4;                                // tiles (accumulators) per wave
constexpr int kOutPerWg64 = kWaves64 * kAcc64 * kTile64;  // 4096 outputs
constexpr size_t kMaxGridY64 = 65535;

// LDS: double xs[hist + 4096] (the workgroup's input segment, hist = 16 (Q - 1) samples of history in front) followed by
// the reversed, zero padded taps hr[w] = h[16Q - w], w in [0, 16Q + 16).
template <bool ALIGNED>
__global__ __launch_bounds__(kThreads64) void fir_direct_f64_kernel(const double *__restrict__ x, double *__restrict__ y, size_t n,
                                                                    const double *__restrict__ taps, int ntaps, int Q,
                                                                    unsigned period) {  // != 0: see launch_fir_direct_f64
    extern __shared__ __attribute__((aligned(16))) double smem64[];
    const int hist = 16 * (Q - 1);
    const int xs_len = hist + kOutPerWg64;
    const int taps_len = 16 * Q + 16;
    double *xs = smem64;
    double *hr = smem64 + xs_len;
    const int tid = threadIdx.x;
    const size_t ch = blockIdx.y;
    const long long chunk0 = static_cast<long long>(blockIdx.x) * kOutPerWg64;
    const long long nn = static_cast<long long>(n);
    const double *xc = x + ch * n;
    double *yc = y + ch * n;

    for (int w = tid; w < taps_len; w += kThreads64) {
        const int idx = 16 * Q - w;
        hr[w] = (idx >= 0 && idx < ntaps) ? taps[idx] : 0.0;
    }
    const long long seg0 = chunk0 - hist;  // even: 16-byte groups lie wholly inside or outside [0, n) when n is even
    for (int u = tid * 2; u < xs_len; u += kThreads64 * 2) {
        const long long g = seg0 + u;
        d2 v = {0.0, 0.0};
        if (ALIGNED) {  // no branch around the load: a group outside the channel is aimed at its first sample and zeroed
            const bool ok = g >= 0 && g + 1 < nn;
            v = load_stream<double>(xc + (ok ? g : 0));
            if (!ok) v = d2{0.0, 0.0};
        } else {
            if (g >= 0 && g < nn) v[0] = xc[g];
            if (g + 1 >= 0 && g + 1 < nn) v[1] = xc[g + 1];
        }
        *reinterpret_cast<d2 *>(xs + u) = v;
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int local = wave * kAcc64 * kTile64;  // first output of this wave inside the workgroup's 4096
    if (chunk0 + local >= nn) return;           // wave-uniform
    const int rho = 4 * (r & 3) + (r >> 2);                      // the output row that matrix row r stands for
    const double *a_ptr = hr + 16 * Q - rho + 4 * kq;           // minus 16 q per step; 8-byte aligned
    const double *b_ptr = xs + hist + local + 16 * r + 4 * kq;  // minus 16 q per step; 32-byte aligned
    d4 acc[kAcc64];
#pragma unroll
    for (int t = 0; t < kAcc64; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    for (int q = 0; q < Q; ++q) {
        double a4[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) a4[kk] = a_ptr[kk - 16 * q];
        d4 b4[kAcc64];
#pragma unroll
        for (int t = 0; t < kAcc64; ++t) b4[t] = *reinterpret_cast<const d4 *>(b_ptr + t * kTile64 - 16 * q);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int t = 0; t < kAcc64; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[kk], b4[t][kk], acc[t], 0, 0, 0);
        }
    }
    // D[i][j]: lane j + 16 (i & 3), register i >> 2; row i is output row rho(i)  ->  y[j0 + 16 (lane & 15) + 4 (lane >> 4) + reg]
    const long long order = ntaps - 1;
#pragma unroll
    for (int t = 0; t < kAcc64; ++t) {
        const long long j = chunk0 + local + static_cast<long long>(t) * kTile64 + 16 * r + 4 * kq;
        if (j >= nn) continue;
        double o[4];
        if (period) {  // (wave-uniform) many channels as one signal: zeros by the position inside the channel
            unsigned jm = static_cast<unsigned>(j) % period;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[i] = jm >= static_cast<unsigned>(order) ? acc[t][i] : 0.0;
                if (++jm == period) jm = 0;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (j + i >= order) ? acc[t][i] : 0.0;  // filter.h:133: first M outputs stay 0
        }
        if (ALIGNED && j + 3 < nn) {
            store_stream<double>(yc + j, d2{o[0], o[1]});
            store_stream<double>(yc + j + 2, d2{o[2], o[3]});
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (j + i < nn) yc[j + i] = o[i];
        }
    }
}

}  // namespace

int launch_fir_direct_f64(const double *x, double *y, size_t channels, size_t n, const double *d_taps, int ntaps, hipStream_t s) {
    // Many channels of less than a few workgroups' worth of samples run as ONE signal of channels * n samples, as in the fp32
    // kernel (fir_kernels.hip, PERIODIC): an output with a full window behind it never reaches across its channel's start, the
    // first ntaps - 1 outputs of every channel are zeroed by their position j mod n.  2,097,152 x 64 doubles, 7 taps: 0.04 ->
    // 0.50 of HBM; 134,217 x 1,000, 511 taps: 0.37 -> 0.76 of the fp64 MFMA rate (scripts/fir_f64_shapes.py).  DSP_GPU_FIR_NO_FLATTEN=1 keeps the channels apart.
    unsigned period = 0;
    if (channels >= 2 && n >= static_cast<size_t>(ntaps) && n < 65536 && channels * n < 0xffffffffull && !getenv("DSP_GPU_FIR_NO_FLATTEN")) {
        period = static_cast<unsigned>(n);
        n *= channels;
        channels = 1;
    }
    for (; channels > kMaxGridY64; channels -= kMaxGridY64, x += kMaxGridY64 * n, y += kMaxGridY64 * n)  // grid.y is 16 bits
        if (int rc = launch_fir_direct_f64(x, y, kMaxGridY64, n, d_taps, ntaps, s)) return rc;
    const int order = ntaps - 1;
    const int Q = (order + 15) / 16 + 1;
    const size_t lds_bytes = (static_cast<size_t>(16 * (Q - 1) + kOutPerWg64) + 16u * Q + 16u) * sizeof(double);
    if (lds_bytes > 160u * 1024) return fail(DSP_GPU_EUNSUPPORTED, "direct FIR (double): %d taps do not fit the LDS", ntaps);
    const size_t blocks = (n + kOutPerWg64 - 1) / kOutPerWg64;
    const bool aligned = (n % 2 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(y) % 16 == 0);
    auto kern = aligned ? fir_direct_f64_kernel<true> : fir_direct_f64_kernel<false>;
    if (lds_bytes > 64 * 1024)
        if (int rc = allow_full_lds(reinterpret_cast<const void *>(kern))) return rc;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks), static_cast<unsigned>(channels)), dim3(kThreads64), lds_bytes, s, x,
                       y, n, d_taps, ntaps, Q, period);
    return check_launch("fir_direct_f64_kernel");
}

}  // namespace dspgpu
