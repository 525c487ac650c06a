// iir_batch_core.h -- the one-lane-per-channel IIR recursion (device templates shared by iir_kernels.hip and
// iir_batch_ext.hip).
//
// Stands in for dsp::Coefficients<Poles>::operator()(const SampleBuffer<SampleType>&) and its BiQuad specialisation
// (reference coefficients.h:88-130, 172-195):
//     v[t] = sum_{i=0..P} a[i] x[t-i] + sum_{i=1..P} b[i] v[t-i]        (feedback ADDED)
//     y[t] = v[t]*c0 + x[t]*d0
// for SampleType float or double (sample_buffer.h:20 allows any floating type; every in-repo caller of the reference
// filters SampleBuffer<double>), in one of two arithmetics:
//   IIR_ARITH_FMA  double throughout, fused multiply-adds, the feedback term through v[t-1] applied last so that the
//                  loop-carried dependency is ONE fma.  The fast form; what AUTO / BATCH / SCAN use.
//   IIR_ARITH_REF  the reference's own arithmetic, operation for operation and in its order, without fused multiply-adds
//                  (an x86-64 build of the reference has none):
//                  generic section (coefficients.h:106-130): the accumulator has the SAMPLE type -- with float samples it is
//                  rounded to float after every `+=` -- histories are double, output
//                  SampleType(oSample * c0 + iSample * d0);
//                  Coefficients<2> = BiQuad (coefficients.h:172-182): all double, y = (a0 s + a1 x1) + a2 x2;
//                  y += (b1 y1 + b2 y2); output SampleType(y c0 + s d0).
//                  Bit-identical to the reference for both sample types (tests/test_iir_ref_arith_gpu.py).
// Both TUs are built with -ffp-contract=off (Makefile): the REF form must not be contracted, the FMA form calls fma().
#pragma once

#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace dspgpu {
namespace iirb {

#define DSP_DEV __device__ __forceinline__

constexpr int kWave = 64;

enum { IIR_ARITH_FMA = 0, IIR_ARITH_REF = 1 };

// One coefficient set per lane, in registers (dsp_gpu_iir_multi); otherwise the wave-uniform IirCoef of the kernel arguments.
template <int P>
struct LaneCoef {
    double a[P + 1], b[P + 1], c0, d0;
};

// One recursion step in double with fmas.  xh[i] = x[t-1-i], vh[i] = v[t-1-i].  The feedback term through v[t-1] is
// applied last so the loop-carried dependency is a single fma.
template <int P, typename C>
DSP_DEV double iir_step(double x0, double (&xh)[P], double (&vh)[P], const C &co) {
    double f = co.a[0] * x0;
#pragma unroll
    for (int k = 0; k < P; ++k) f = fma(co.a[k + 1], xh[k], f);
#pragma unroll
    for (int k = P - 1; k >= 1; --k) f = fma(co.b[k + 1], vh[k], f);
    const double v = fma(co.b[1], vh[0], f);
#pragma unroll
    for (int k = P - 1; k >= 1; --k) {
        xh[k] = xh[k - 1];
        vh[k] = vh[k - 1];
    }
    xh[0] = x0;
    vh[0] = v;
    return v;
}

// One step, returning the output sample.  WANT_OUT = false: only the histories advance (the scan's zero-state pass).
// UNIT (fast arithmetic only): the caller has checked c0 == 1 and d0 == 0 -- Coefficients' defaults (coefficients.h:11-12),
// what every filter designed by recursive_filters.h except the shelves carries -- and the output is v itself:
// fma(v, 1, x0 * 0) is v for every finite x0 (up to the sign of a zero); two of a four-pole step's thirteen operations less.
template <int P, int ARITH, bool WANT_OUT, bool UNIT = false, typename T, typename C>
DSP_DEV T iir_step_out(T in, double (&xh)[P], double (&vh)[P], const C &co) {
    const double x0 = static_cast<double>(in);
    if constexpr (ARITH == IIR_ARITH_FMA) {
        const double v = iir_step<P>(x0, xh, vh, co);
        if constexpr (UNIT) return WANT_OUT ? static_cast<T>(v) : T(0);
        return WANT_OUT ? static_cast<T>(fma(v, co.c0, x0 * co.d0)) : T(0);
    } else if constexpr (P == 2) {
        // coefficients.h:172-182 (Coefficients<2>): auto y = a0*sample + a1*x1 + a2*x2;  y += b1*y1 + b2*y2;
        double y = (co.a[0] * x0 + co.a[1] * xh[0]) + co.a[2] * xh[1];
        y = y + (co.b[1] * vh[0] + co.b[2] * vh[1]);
        xh[1] = xh[0];
        xh[0] = x0;
        vh[1] = vh[0];
        vh[0] = y;
        return static_cast<T>((y * co.c0) + (x0 * co.d0));
    } else {
        // coefficients.h:109-118: SampleType oSample{}; oSample += a[i] * x[i] (i = 0..P), then += b[i] * y[i] (i = 1..P):
        // every += widens the accumulator, adds the double product and rounds back to the sample type
        T o = T(0);
        o = static_cast<T>(static_cast<double>(o) + co.a[0] * x0);
#pragma unroll
        for (int k = 0; k < P; ++k) o = static_cast<T>(static_cast<double>(o) + co.a[k + 1] * xh[k]);
#pragma unroll
        for (int k = 0; k < P; ++k) o = static_cast<T>(static_cast<double>(o) + co.b[k + 1] * vh[k]);
#pragma unroll
        for (int k = P - 1; k >= 1; --k) {
            xh[k] = xh[k - 1];
            vh[k] = vh[k - 1];
        }
        xh[0] = x0;
        vh[0] = static_cast<double>(o);  // coefficients.h:127: y[1] = oSample
        return static_cast<T>(static_cast<double>(o) * co.c0 + x0 * co.d0);  // coefficients.h:129
    }
}

// Run `count` samples of one LDS row (stride 1).  STORE: overwrite the row with y.
template <int P, bool STORE, int ARITH = IIR_ARITH_FMA, bool UNIT = false, typename T, typename C>
DSP_DEV void run_row_partial(T *row, int count, double (&xh)[P], double (&vh)[P], const C &co) {
#pragma unroll 1
    for (int i = 0; i < count; ++i) {
        const T out = iir_step_out<P, ARITH, STORE, UNIT>(row[i], xh, vh, co);
        if (STORE) row[i] = out;
    }
}

// The same over LEN samples of a 16-byte aligned row, one 16-byte LDS access per group of samples.
template <int P, int LEN, bool STORE, int ARITH = IIR_ARITH_FMA, bool UNIT = false, typename T, typename C>
DSP_DEV void run_rowv(T *row, double (&xh)[P], double (&vh)[P], const C &co) {
    using V = typename Smp<T>::vec;
    constexpr int kVec = Smp<T>::kVec;
#pragma unroll
    for (int i = 0; i < LEN; i += kVec) {
        V q = *reinterpret_cast<const V *>(row + i);
#pragma unroll
        for (int k = 0; k < kVec; ++k) {
            const T out = iir_step_out<P, ARITH, STORE, UNIT>(static_cast<T>(q[k]), xh, vh, co);
            if (STORE) q[k] = out;
        }
        if (STORE) *reinterpret_cast<V *>(row + i) = q;
    }
}

// `count` samples of a 16-byte aligned row: whole 16-byte groups first, the remainder one by one (a channel's last, partial tile).
template <int P, bool STORE, int ARITH = IIR_ARITH_FMA, typename T, typename C>
DSP_DEV void run_row_count(T *row, int count, double (&xh)[P], double (&vh)[P], const C &co) {
    using V = typename Smp<T>::vec;
    constexpr int kVec = Smp<T>::kVec;
    int i = 0;
#pragma unroll 1
    for (; i + kVec <= count; i += kVec) {
        V q = *reinterpret_cast<const V *>(row + i);
#pragma unroll
        for (int k = 0; k < kVec; ++k) {
            const T out = iir_step_out<P, ARITH, STORE>(static_cast<T>(q[k]), xh, vh, co);
            if (STORE) q[k] = out;
        }
        if (STORE) *reinterpret_cast<V *>(row + i) = q;
    }
#pragma unroll 1
    for (; i < count; ++i) {
        const T out = iir_step_out<P, ARITH, STORE>(row[i], xh, vh, co);
        if (STORE) row[i] = out;
    }
}

// ---------------------------------------------------------------------------------------------
// batch: one lane per channel.  A wave owns 64 channels and walks time in tiles of 512-byte rows (128 floats / 64
// doubles) staged through a padded LDS tile, so HBM sees 512-byte runs per channel while every lane walks its own
// 16-byte aligned row with conflict-free 16-byte LDS accesses; the next tile's loads are in flight while the current
// one is computed.
// ---------------------------------------------------------------------------------------------
template <typename T>
struct BatchTile {
    static constexpr int kVec = Smp<T>::kVec;
    static constexpr int kBt = 512 / static_cast<int>(sizeof(T));  // samples per tile row: 512-byte row segments (256-byte ones cap a plain
                                                                   // copy at 4.4 TB/s, scripts/copy_patterns3.hip; 1-KB rows, two workgroups
                                                                   // per CU by their own LDS: 0.745 against 0.77 of 8 TB/s)
    static constexpr int kBtPad = kBt + kVec;                      // rows 16-byte aligned; lane stride 132 dwords: 16-byte row walks are conflict-free
    static constexpr int kIters = kBt / kVec;                      // 16-byte loads per lane per tile = 32
    static constexpr int kLanesPerRow = kBt / kVec;                // 32 lanes cover one row
    static constexpr int kRowsPerIter = kWave / kLanesPerRow;      // 2 rows per wave instruction
};

template <bool ALIGNED, typename T>
DSP_DEV void batch_load(const T *__restrict__ x, size_t channels, size_t n, size_t ch0, size_t t0, int lane,
                        typename Smp<T>::vec (&pre)[BatchTile<T>::kIters], bool rt_al) {
    using BT = BatchTile<T>;
    using V = typename Smp<T>::vec;
    const bool al = ALIGNED || rt_al;
    if (al && ch0 + kWave <= channels && t0 + BT::kBt <= n) {  // a full tile: no bounds tests between the loads
        const T *p = x + (ch0 + lane / BT::kLanesPerRow) * n + t0 + (lane % BT::kLanesPerRow) * BT::kVec;
        const size_t step = BT::kRowsPerIter * n;
#pragma unroll
        for (int it = 0; it < BT::kIters; ++it) pre[it] = load_stream<T>(p + it * step);
        return;
    }
    // a ragged tile (the last channels of the batch, the last samples of the channels): no branch around any load -- a group
    // or sample outside is aimed at the last one inside its row (the batch's last row) and zeroed afterwards, so that all the
    // tile's loads are in flight together.  (Aimed at x[0] they would all queue for one cache line: 10,000-sample rows 0.46 ->
    // 0.34 of HBM.)
    if (al) {  // n is a multiple of the group: a group lies wholly inside its row or wholly outside
#pragma unroll
        for (int it = 0; it < BT::kIters; ++it) {
            const size_t ch = ch0 + it * BT::kRowsPerIter + lane / BT::kLanesPerRow;
            const size_t g = t0 + (lane % BT::kLanesPerRow) * BT::kVec;
            const bool ok = ch < channels && g < n;
            const V v = load_stream<T>(x + (ch < channels ? ch : channels - 1) * n + (g < n ? g : n - BT::kVec));
            pre[it] = ok ? v : V(0);
        }
        return;
    }
#pragma unroll
    for (int it = 0; it < BT::kIters; ++it) {
        const size_t ch = ch0 + it * BT::kRowsPerIter + lane / BT::kLanesPerRow;
        const size_t g = t0 + (lane % BT::kLanesPerRow) * BT::kVec;
        V v;
#pragma unroll
        for (int k = 0; k < BT::kVec; ++k) {
            const bool ok = ch < channels && g + k < n;
            const T e = x[(ch < channels ? ch : channels - 1) * n + (g + k < n ? g + k : n - 1)];
            v[k] = ok ? e : T(0);
        }
        pre[it] = v;
    }
}

// P: compiled recursion order; sp <= P: the caller's order (its state is [ch][2*sp]; the top P - sp
// coefficients are zero, which leaves every sum unchanged).  PERCH: one coefficient set per channel.
// ALIGNED (compile time) or rt_al (kernel argument): rows and pointers allow 16-byte accesses.
template <int P, bool ALIGNED, bool PERCH, typename T = float, int ARITH = IIR_ARITH_FMA>
__global__ __launch_bounds__(kWave) void iir_batch_kernel(const T *__restrict__ x, T *__restrict__ y, size_t channels,
                                                           size_t n, IirCoef uco, double *__restrict__ state,
                                                           size_t groups, size_t chunk, int sp, IirPerChannel pc,
                                                           int rt_al_i) {
    using BT = BatchTile<T>;
    using V = typename Smp<T>::vec;
    constexpr int kBt = BT::kBt, kBtPad = BT::kBtPad, kBtIters = BT::kIters, kLanesPerRow = BT::kLanesPerRow,
                  kRowsPerIter = BT::kRowsPerIter, kVec = BT::kVec;
    __shared__ __attribute__((aligned(16))) T tile[kWave * kBtPad];
    const bool rt_al = rt_al_i != 0;
    const bool al = ALIGNED || rt_al;
    const int lane = threadIdx.x;
    const size_t group = xcd_tile_of(blockIdx.x, chunk);  // each XCD owns a contiguous range of channels
    if (group >= groups) return;
    const size_t ch0 = group * kWave;
    const size_t ch = ch0 + lane;
    const bool live = ch < channels;

    double xh[P], vh[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        xh[i] = (live && state && i < sp) ? state[ch * 2 * sp + i] : 0.0;
        vh[i] = (live && state && i < sp) ? state[ch * 2 * sp + sp + i] : 0.0;
    }
    typename std::conditional<PERCH, LaneCoef<P>, IirCoef>::type co;
    if constexpr (PERCH) {
#pragma unroll
        for (int i = 0; i <= P; ++i) {
            co.a[i] = (live && i <= sp) ? pc.a[ch * (sp + 1) + i] : 0.0;
            co.b[i] = (live && i <= sp && i > 0) ? pc.b[ch * (sp + 1) + i] : 0.0;
        }
        co.c0 = pc.c0 ? (live ? pc.c0[ch] : 0.0) : 1.0;  // Coefficients' defaults (coefficients.h:11-12)
        co.d0 = pc.d0 ? (live ? pc.d0[ch] : 0.0) : 0.0;
    } else {
        co = uco;
    }

    const size_t ntiles = (n + kBt - 1) / kBt;
    V pre[kBtIters];
    batch_load<ALIGNED, T>(x, channels, n, ch0, 0, lane, pre, rt_al);
    for (size_t tl = 0; tl < ntiles; ++tl) {
        const size_t t0 = tl * kBt;
#pragma unroll
        for (int it = 0; it < kBtIters; ++it) {
            *reinterpret_cast<V *>(tile + (it * kRowsPerIter + lane / kLanesPerRow) * kBtPad + (lane % kLanesPerRow) * kVec) = pre[it];
        }
        __syncthreads();
        if (tl + 1 < ntiles) batch_load<ALIGNED, T>(x, channels, n, ch0, t0 + kBt, lane, pre, rt_al);
        if (live) {
            T *row = tile + lane * kBtPad;
            if (t0 + kBt <= n) run_rowv<P, kBt, true, ARITH>(row, xh, vh, co);
            else run_row_count<P, true, ARITH>(row, static_cast<int>(n - t0), xh, vh, co);
        }
        __syncthreads();
        if (al && ch0 + kWave <= channels && t0 + kBt <= n) {  // full tile: LDS reads 16 bytes at a time, no bounds tests
            const T *src = tile + (lane / kLanesPerRow) * kBtPad + (lane % kLanesPerRow) * kVec;
            T *p = y + (ch0 + lane / kLanesPerRow) * n + t0 + (lane % kLanesPerRow) * kVec;
#pragma unroll
            for (int it0 = 0; it0 < kBtIters; it0 += 4) {
                V q[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) q[i] = *reinterpret_cast<const V *>(src + (it0 + i) * kRowsPerIter * kBtPad);
#pragma unroll
                for (int i = 0; i < 4; ++i) store_stream<T>(p + static_cast<size_t>((it0 + i) * kRowsPerIter) * n, q[i]);
            }
            __syncthreads();
            continue;
        }
#pragma unroll
        for (int it = 0; it < kBtIters; ++it) {
            const size_t c = ch0 + it * kRowsPerIter + lane / kLanesPerRow;
            const size_t g = t0 + (lane % kLanesPerRow) * kVec;
            if (c >= channels || g >= n) continue;
            const T *src = tile + (it * kRowsPerIter + lane / kLanesPerRow) * kBtPad + (lane % kLanesPerRow) * kVec;
            T *p = y + c * n + g;
            if (al && g + kVec - 1 < n) {
                store_stream<T>(p, *reinterpret_cast<const V *>(src));
            } else {
#pragma unroll
                for (int k = 0; k < kVec; ++k)
                    if (g + k < n) p[k] = src[k];
            }
        }
        __syncthreads();
    }
    if (live && state) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            if (i >= sp) break;
            state[ch * 2 * sp + i] = xh[i];
            state[ch * 2 * sp + sp + i] = vh[i];
        }
    }
}

// Dynamic LDS nobody uses: it only lowers the number of workgroups a CU holds.  Up to four poles the fast recursion is
// cheap enough for TWO one-wave workgroups per CU to keep up with memory, and with 2 x 32 KB instead of 4 x 32 KB of
// tiles in flight per CU the memory system runs faster (the in-flight lesson):
// 262,144 x 4,096, 4 poles: 1.485 -> 1.375 ms (0.72 -> 0.78 of 8 TB/s); 3 per CU 0.70, 1 per CU 0.45.  From six
// poles on the arithmetic needs all four SIMDs (6 poles: 1.49 ms at four per CU, 1.68 at two), and so does the
// reference arithmetic at every order (its dependency chain is 2P + 1 operations long).
// DSP_GPU_IIR_BATCH_LDS_PAD=<bytes> overrides (0 = four per CU).
inline unsigned batch_lds_pad(int P, int arith) {
    static const int pad_env = [] {
        const char *e = std::getenv("DSP_GPU_IIR_BATCH_LDS_PAD");
        return e ? std::atoi(e) : -1;
    }();
    if (pad_env >= 0) return static_cast<unsigned>(pad_env);
    return (P <= 4 && arith == IIR_ARITH_FMA) ? 30000u : 0u;
}

// Launch over all channels.
template <int P, bool ALIGNED, typename T, int ARITH>
int batch_dispatch(const T *x, T *y, size_t channels, size_t n, const IirCoef &co, double *state, const IirPerChannel *pc,
                   bool rt_al, hipStream_t s) {
    const unsigned pad = batch_lds_pad(P, ARITH);
    const size_t groups = (channels + kWave - 1) / kWave;
    const size_t chunk = (groups + 7) / 8;
    if (8 * chunk > 0x7fffffffull) return fail(DSP_GPU_EUNSUPPORTED, "IIR batch: too many channels");
    const dim3 grid(static_cast<unsigned>(8 * chunk));
    if constexpr (P <= 8) {
        if (pc) {
            hipLaunchKernelGGL((iir_batch_kernel<P, ALIGNED, true, T, ARITH>), grid, dim3(kWave), pad, s, x, y, channels, n, co,
                               state, groups, chunk, co.poles, *pc, rt_al ? 1 : 0);
            return check_launch("iir_batch_kernel<per-channel>");
        }
    }
    hipLaunchKernelGGL((iir_batch_kernel<P, ALIGNED, false, T, ARITH>), grid, dim3(kWave), pad, s, x, y, channels, n, co, state,
                       groups, chunk, co.poles, IirPerChannel{}, rt_al ? 1 : 0);
    return check_launch("iir_batch_kernel");
}

template <typename T>
bool is_aligned(const T *x, const T *y, size_t n) {
    return n % Smp<T>::kVec == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0;
}

}  // namespace iirb
}  // namespace dspgpu
