// iir_batch_ext.hip -- the one-lane-per-channel IIR recursion for the sample types and arithmetics beside the
// benchmarked one (iir_batch_kernels.hip: fp32 samples, fast arithmetic):
//   double samples, fast arithmetic       dsp_gpu_iir_f64 BATCH / AUTO -- Coefficients<P>::operator()(SampleBuffer<double>),
//                                         the type every in-repo caller of the reference uses (coefficients.h:88-104)
//   float / double samples, REF arithmetic  DSP_GPU_IIR_BATCH_REF: coefficients.h:106-130 / 172-182 operation for operation
// Alignment is a kernel argument here (one instantiation per order instead of two): these are the widening routes.
#include <algorithm>

#include "iir_batch_core.h"

namespace dspgpu {

namespace {
template <typename T, int ARITH>
int launch_any(const T *x, T *y, size_t channels, size_t n, const IirCoef &co, double *state, const IirPerChannel *pc,
               hipStream_t s) {
    using namespace iirb;
    const bool al = is_aligned<T>(x, y, n);
    if (pc && co.poles > 8) return fail(DSP_GPU_EUNSUPPORTED, "per-channel IIR coefficients: poles %d > 8", co.poles);
#define DSP_IIR_BATCH_CASE(P) return batch_dispatch<P, false, T, ARITH>(x, y, channels, n, co, state, pc, al, s)
    switch (co.poles) {  // orders without a kernel of their own run on the next one up (zero top coefficients: in the
                         // REF arithmetic the extra terms add +0.0, which changes no value)
    case 1: DSP_IIR_BATCH_CASE(1);
    case 2: DSP_IIR_BATCH_CASE(2);
    case 3: DSP_IIR_BATCH_CASE(3);
    case 4: DSP_IIR_BATCH_CASE(4);
    case 5: DSP_IIR_BATCH_CASE(5);
    case 6: DSP_IIR_BATCH_CASE(6);
    case 7: DSP_IIR_BATCH_CASE(7);
    case 8: DSP_IIR_BATCH_CASE(8);
    case 9: case 10: DSP_IIR_BATCH_CASE(10);
    case 11: case 12: DSP_IIR_BATCH_CASE(12);
    case 13: case 14: case 15: case 16: DSP_IIR_BATCH_CASE(16);
    case 17: case 18: case 19: case 20: DSP_IIR_BATCH_CASE(20);
    default: return fail(DSP_GPU_EINVAL, "poles %d outside [1,20]", co.poles);
    }
#undef DSP_IIR_BATCH_CASE
}
}  // namespace

int launch_iir_batch_f64(const double *x, double *y, size_t channels, size_t n, const IirCoef &co, double *state,
                         const IirPerChannel *pc, hipStream_t s) {
    return launch_any<double, iirb::IIR_ARITH_FMA>(x, y, channels, n, co, state, pc, s);
}

int launch_iir_batch_ref(const float *x, float *y, size_t channels, size_t n, const IirCoef &co, double *state,
                         const IirPerChannel *pc, hipStream_t s) {
    return launch_any<float, iirb::IIR_ARITH_REF>(x, y, channels, n, co, state, pc, s);
}

int launch_iir_batch_ref_f64(const double *x, double *y, size_t channels, size_t n, const IirCoef &co, double *state,
                             const IirPerChannel *pc, hipStream_t s) {
    return launch_any<double, iirb::IIR_ARITH_REF>(x, y, channels, n, co, state, pc, s);
}

// SCAN for double samples: sharded in TIME on the one-lane-per-channel kernel.  (The block-parallel scan of iir_kernels.hip
// stages its zero-state outputs as fp32 rows; a double form of it is not built.)  The reference's own callers filter ONE
// SampleBuffer<double> (eval/recursive_filters/main.cpp:35): one lane walking a million samples takes 67 ms -- slower than the
// reference's loop on a CPU core.  A stable filter forgets its state: after W = dsp_gpu_iir_settle samples the response to ANY
// state has decayed below 2^-44 of it (iir_kernels.hip, iir_settle_samples), so every channel is cut into segments of L samples,
//   1. the 2 W samples in front of every segment are copied to a scratch matrix [segments][2 W] and run from a zero state, which
//      leaves in state[g] the filter's state at segment g's first sample, exact to 2^-44 of its size (segments that begin a
//      channel take the caller's state instead),
//   2. the kernel runs over the [channels * segments][L] view of x and y, one lane per segment, from those states,
//   3. the last segment of every channel hands the caller's state back; a single channel's remainder (n not a multiple of L)
//      runs after it from that state.
// 2 W / L more arithmetic and reads (L >= 8 W), all in the reference's double arithmetic; 1 x 2^20 samples: 67 ms -> 0.16 ms, 1 x 2^27: 0.49 ms = 0.55 of HBM
// (scripts/iir_f64_long.py).
// EUNSUPPORTED (AUTO then runs one lane per channel): filters that do not settle within 8,192 samples, several channels
// whose length has no suitable divisor, signals of fewer than two segments.
namespace {
template <typename T, typename Batch>
int time_sharded(const T *x, T *y, size_t channels, size_t n, const IirCoef &co, double *state, hipStream_t s, Batch batch) {
    size_t W = 0;
    if (iir_settle_samples(co, &W) != DSP_GPU_OK || W > 8192)
        return fail(DSP_GPU_EUNSUPPORTED, "IIR in time shards: the filter does not settle within 8,192 samples; use DSP_GPU_IIR_BATCH");
    // twice the settle length: 2^-44 of the STATE can be 1e-5 of the OUTPUT for an ill-conditioned direct form (18-pole
    // Chebyshev at fc = 0.1: noise gain ~1e9, measured 5.9e-5 against one lane with W alone); 2^-88 is nothing for any of them
    W = std::max<size_t>((2 * W + 63) / 64 * 64, 64);
    // about 32,768 lanes (two one-wave workgroups per CU), segments of at least 4 W samples, whole 512-byte rows
    size_t L = std::max<size_t>(4 * W, (channels * n + 32767) / 32768);
    L = (L + 63) / 64 * 64;
    {  // the kernel walks rows of a power-of-two length fastest (profiles/r03_iir_batch_row_lengths.txt)
        size_t p2 = 64;
        while (p2 < L) p2 *= 2;
        if (p2 <= 8192) L = p2;
    }
    if (channels > 1) {  // the view [channels * S][L] needs n = S * L: whole 512-byte rows if there are such, else 16-byte ones
        size_t found = 0;
        for (size_t cand = L; cand <= 2 * L && cand <= n && !found; cand += 64)
            if (n % cand == 0) found = cand;
        constexpr size_t kGroup = 16 / sizeof(T);
        for (size_t cand = (4 * W + kGroup - 1) / kGroup * kGroup; cand <= 4 * L && cand <= n && !found; cand += kGroup)
            if (n % cand == 0) found = cand;
        if (!found) {
            // no such length (an odd or near-prime n): a few channels go one after the other, each with its remainder
            if (channels > 64) return fail(DSP_GPU_EUNSUPPORTED, "IIR in time shards: no segment length divides %zu samples; use DSP_GPU_IIR_BATCH", n);
            for (size_t c = 0; c < channels; ++c)
                if (int rc = time_sharded<T>(x + c * n, y + c * n, 1, n, co, state ? state + c * 2 * static_cast<size_t>(co.poles) : nullptr, s, batch))
                    return rc;
            return DSP_GPU_OK;
        }
        L = found;
    }
    const size_t S = n / L, tail = n - S * L;
    if (S < 2) return fail(DSP_GPU_EUNSUPPORTED, "IIR in time shards: %zu samples are fewer than two segments; use DSP_GPU_IIR_BATCH", n);
    const size_t G = channels * S;
    const size_t sp = static_cast<size_t>(co.poles), row = 2 * sp * sizeof(double);
    void *warm_v = nullptr, *st_v = nullptr;
    if (int rc = workspace(WS_SCRATCH_A, G * W * sizeof(T), &warm_v, s)) return rc;
    if (int rc = workspace(WS_SCRATCH_B, G * row, &st_v, s)) return rc;
    T *warm = static_cast<T *>(warm_v);
    double *st = static_cast<double *>(st_v);
    // 1. (segment g begins at x + g L for every g, channels being n = S L apart; a channel's first segment gets the previous
    //     channel's tail here and the caller's state below)
    DSPGPU_HIP(hipMemcpy2DAsync(warm + W, W * sizeof(T), x + L - W, L * sizeof(T), W * sizeof(T), G - 1,
                                hipMemcpyDeviceToDevice, s));
    DSPGPU_HIP(hipMemsetAsync(st, 0, G * row, s));
    if (int rc = batch(warm + W, warm + W, G - 1, W, st + 2 * sp)) return rc;
    if (state) DSPGPU_HIP(hipMemcpy2DAsync(st, S * row, state, row, row, channels, hipMemcpyDeviceToDevice, s));
    else DSPGPU_HIP(hipMemset2DAsync(st, S * row, 0, row, channels, s));
    // 2.
    if (int rc = batch(x, y, G, L, st)) return rc;
    // 3.
    if (tail)  // (channels == 1)
        if (int rc = batch(x + S * L, y + S * L, size_t{1}, tail, st + (G - 1) * 2 * sp)) return rc;
    if (state) DSPGPU_HIP(hipMemcpy2DAsync(state, row, st + (S - 1) * 2 * sp, S * row, row, channels, hipMemcpyDeviceToDevice, s));
    return DSP_GPU_OK;
}
}  // namespace

int launch_iir_scan_f64(const double *x, double *y, size_t channels, size_t n, const IirCoef &co, double *state, hipStream_t s) {
    return time_sharded<double>(x, y, channels, n, co, state, s, [&](const double *bx, double *by, size_t c, size_t len, double *st) {
        return launch_iir_batch_f64(bx, by, c, len, co, st, nullptr, s);
    });
}

// The same for fp32 samples: sections of more than eight poles, for which the block-parallel scan has no kernel (its P x P carry
// matrices live in registers) -- AUTO used to run them one lane per channel whatever the shape: 1 x 2^20 samples, 10 / 12 poles: 94 / 119 ms -> 0.8 / 6 ms.
int launch_iir_sharded(const float *x, float *y, size_t channels, size_t n, const IirCoef &co, double *state, hipStream_t s) {
    return time_sharded<float>(x, y, channels, n, co, state, s, [&](const float *bx, float *by, size_t c, size_t len, double *st) {
        return launch_iir_batch(bx, by, c, len, co, st, nullptr, s);
    });
}

}  // namespace dspgpu
