// iir_swept_kernels.hip -- a band-pass / band-reject BiQuad retuned at EVERY sample, for gfx950 (MI355X).
//
// Stands in for the loop of eval/signals/wind_generator.h:150-167 (Whistling::getSample): per sample
//     bp = dsp::recursive::bandPassFilter(fc[t], bw)        recursive_filters.h:165-186 (bandReject :188-209)
//     m_bp.a = bp.a; m_bp.b = bp.b;                          the history x1,x2,y1,y2 stays
//     y[t] = m_bp(x[t])                                      coefficients.h:172-182, all in double
// i.e. a two-pole recursion whose five coefficients are designed anew from the centre frequency of that sample.
// The design is elementwise (one cos and one division in double per sample), the recursion is sequential, so the
// unit of parallelism is the channel, as in iir_batch_kernel: a wave owns 64 channels and walks time in tiles staged
// through two padded LDS tiles (samples and centre frequencies), the next tile's loads in flight while this one is
// computed.  Arithmetic follows the reference operation by operation in double WITHOUT contraction (this file is
// built with -ffp-contract=off, see the Makefile): what differs from a CPU run is the last bit of cos() (swept_cos.h:
// within one ulp of the host library's, checked on the host by tests/cpp/swept_cos_check.cpp).
//
// Bound: double-precision VALU first, HBM second: ~60-70 instructions per sample (cos 18-30, the IEEE division ~12, the
// design and the recursion ~25, conversions) against 12 B/sample of traffic (x and fc in, y out); under this fp64 load
// the chip clocks at ~1.65 GHz (SQ_BUSY_CYCLES, profiles/r02_biquad_swept_pmc.txt).  262,144 voices x 4,096 samples:
// 4.57 ms with the library cos (~90 instructions, a double-double argument reduction this range never needs) and
// 64-sample rows; 3.68 ms with swept_cos.h; 3.25 ms with 32-sample rows (two waves per SIMD); 3.04 ms with the four
// cosines of a float4 evaluated together and the sine kernel skipped per wave; 2.93 ms = 367 GSamples/s, 4.4 TB/s with
// a bounds-test-free path for full tiles.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.h"
#include "swept_cos.h"

namespace dspgpu {
namespace {

#define DSP_DEV __device__ __forceinline__

constexpr int kWave = 64;

struct SweptState {
    double x1, x2, y1, y2;
};

// One sample, given c = cos(2*PI*fc[t]): design (recursive_filters.h:165-209) then the BiQuad step
// (coefficients.h:172-182), in the reference's own order of operations.  r = 1 - 3*BW and rr = r*r are the same for
// every sample and come from the host.
template <int KIND>
DSP_DEV void swept_design(double c, double r, double rr, double &a0, double &a1, double &a2, double &b1) {
    // Doubling is exact, so 2*r*c == r*(2c), 2*(k-r)*c == (k-r)*(2c) and -2*k*c == -(k*(2c)) bit for bit: 2c is formed
    // once.  Everything else is the reference's expression, operation for operation.
    const double c2 = c + c;
    b1 = r * c2;                                    // 2 * R * cos2pif
    const double k = (1 - b1 + rr) / (2 - c2);      // (1 - 2*R*cos2pif + R*R) / (2 - 2*cos2pif)
    if (KIND == 0) {
        a0 = 1 - k;
        a1 = (k - r) * c2;                          // 2 * (K - R) * cos2pif
        a2 = rr - k;
    } else {
        a0 = k;
        a1 = -(k * c2);                             // -2 * K * cos2pif
        a2 = k;
    }
}

// coefficients.h:172-182 with c0 = 1, d0 = 0 (what both designers leave, :70-71)
DSP_DEV float swept_biquad(float sample, double a0, double a1, double a2, double b1, double b2, SweptState &st) {
    const double s = static_cast<double>(sample);
    double y = a0 * s + a1 * st.x1 + a2 * st.x2;
    y += b1 * st.y1 + b2 * st.y2;
    st.x2 = st.x1;
    st.x1 = s;
    st.y2 = st.y1;
    st.y1 = y;
    return static_cast<float>(y);
}

template <int KIND>
DSP_DEV float swept_apply(float sample, double c, double r, double rr, SweptState &st) {
    double a0, a1, a2, b1;
    swept_design<KIND>(c, r, rr, a0, a1, a2, b1);
    return swept_biquad(sample, a0, a1, a2, b1, -rr, st);
}

template <int KIND>
DSP_DEV float swept_step(float sample, float cf, double r, double rr, SweptState &st) {
    return swept_apply<KIND>(sample, swept_cos(cf), r, rr, st);  // cos(2 * PI * cf), recursive_filters.h:169
}

// Four consecutive samples of a lane.  The four cosines do not depend on the recursion, so they are evaluated together
// (four independent polynomial chains to interleave); and the sine kernel is evaluated only when some lane of the WAVE
// needs it for one of the four (a wave-uniform branch): voices that stay below fc = 1/8 -- anything audible at audio
// rates -- never do.
template <int KIND>
DSP_DEV float4 swept_step4(float4 q, float4 f, double r, double rr, SweptState &st) {
    const float cf[4] = {f.x, f.y, f.z, f.w};
    double n[4], t[4], c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) swept_cos_reduce(cf[i], n[i], t[i]);
    const bool sine = n[0] == 1.0 || n[1] == 1.0 || n[2] == 1.0 || n[3] == 1.0;
    if (__builtin_amdgcn_ballot_w64(sine) == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double cs = swept_cos_kernel(t[i]);
            c[i] = n[i] == 0.0 ? cs : -cs;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = swept_cos_pick(n[i], swept_sin_kernel(t[i]), swept_cos_kernel(t[i]));
    }
    q.x = swept_apply<KIND>(q.x, c[0], r, rr, st);
    q.y = swept_apply<KIND>(q.y, c[1], r, rr, st);
    q.z = swept_apply<KIND>(q.z, c[2], r, rr, st);
    q.w = swept_apply<KIND>(q.w, c[3], r, rr, st);
    return q;
}

// T samples per tile row.  64 rows (one per lane = channel) of T + 4 floats: rows stay 16-byte aligned and lanes walking
// their own rows with 16-byte LDS accesses: lane stride T + 4 dwords (36 for T = 32: the 16 lanes of one access group
// start in 16 different 4-bank groups).
template <int T>
struct SweptShape {
    static constexpr int kPad = T + 4;
    static constexpr int kLanesPerRow = T / 4;               // lanes covering one row with float4
    static constexpr int kRowsPerIter = kWave / kLanesPerRow;  // rows per wave instruction
    static constexpr int kIters = kWave / kRowsPerIter;        // float4 loads per lane per tile and array
};

template <int T, bool ALIGNED>
DSP_DEV void swept_load(const float *__restrict__ p0, size_t stride, size_t channels, size_t n, size_t ch0, size_t t0,
                        int lane, float4 (&pre)[SweptShape<T>::kIters]) {
    using S = SweptShape<T>;
    if (ALIGNED && ch0 + kWave <= channels && t0 + T <= n) {  // a full tile: no bounds tests between the loads
        const float *p = p0 + (ch0 + lane / S::kLanesPerRow) * stride + t0 + (lane % S::kLanesPerRow) * 4;
        const size_t step = S::kRowsPerIter * stride;
#pragma unroll
        for (int it = 0; it < S::kIters; ++it) pre[it] = load_stream_f4(p + it * step);
        return;
    }
#pragma unroll
    for (int it = 0; it < S::kIters; ++it) {
        const size_t ch = ch0 + it * S::kRowsPerIter + lane / S::kLanesPerRow;
        const size_t g = t0 + (lane % S::kLanesPerRow) * 4;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (ch < channels) {
            const float *p = p0 + ch * stride + g;
            if (ALIGNED && g + 3 < n) {
                v = load_stream_f4(p);
            } else {
                if (g < n) v.x = p[0];
                if (g + 1 < n) v.y = p[1];
                if (g + 2 < n) v.z = p[2];
                if (g + 3 < n) v.w = p[3];
            }
        }
        pre[it] = v;
    }
}

template <int T>
DSP_DEV void swept_stage(float *tile, int lane, const float4 (&pre)[SweptShape<T>::kIters]) {
    using S = SweptShape<T>;
#pragma unroll
    for (int it = 0; it < S::kIters; ++it)
        *reinterpret_cast<float4 *>(tile + (it * S::kRowsPerIter + lane / S::kLanesPerRow) * S::kPad +
                                    (lane % S::kLanesPerRow) * 4) = pre[it];
}

// fc_stride: n (every channel its own sweep) or 0 (one sweep shared by all channels).
// state: [channels][4] doubles {x1, x2, y1, y2} (the BiQuad's members, coefficients.h:52-68), in/out, may be null.
template <int KIND, int T, bool ALIGNED>
__global__ __launch_bounds__(kWave) void biquad_swept_kernel(const float *__restrict__ x, const float *__restrict__ fc,
                                                              float *__restrict__ y, size_t channels, size_t n,
                                                              size_t fc_stride, double r, double rr,
                                                              double *__restrict__ state, size_t groups, size_t chunk) {
    using S = SweptShape<T>;
    __shared__ __attribute__((aligned(16))) float xt[kWave * S::kPad];
    __shared__ __attribute__((aligned(16))) float ft[kWave * S::kPad];
    const int lane = threadIdx.x;
    const size_t group = xcd_tile_of(blockIdx.x, chunk);  // each XCD owns a contiguous range of channels
    if (group >= groups) return;
    const size_t ch0 = group * kWave;
    const size_t ch = ch0 + lane;
    const bool live = ch < channels;

    SweptState st{0.0, 0.0, 0.0, 0.0};
    if (live && state) {
        st.x1 = state[ch * 4 + 0];
        st.x2 = state[ch * 4 + 1];
        st.y1 = state[ch * 4 + 2];
        st.y2 = state[ch * 4 + 3];
    }

    const size_t ntiles = (n + T - 1) / T;
    float4 px[S::kIters], pf[S::kIters];
    swept_load<T, ALIGNED>(x, n, channels, n, ch0, 0, lane, px);
    swept_load<T, ALIGNED>(fc, fc_stride, channels, n, ch0, 0, lane, pf);
    for (size_t tl = 0; tl < ntiles; ++tl) {
        const size_t t0 = tl * T;
        swept_stage<T>(xt, lane, px);
        swept_stage<T>(ft, lane, pf);
        __syncthreads();
        if (tl + 1 < ntiles) {
            swept_load<T, ALIGNED>(x, n, channels, n, ch0, t0 + T, lane, px);
            swept_load<T, ALIGNED>(fc, fc_stride, channels, n, ch0, t0 + T, lane, pf);
        }
        if (live) {
            float *xr = xt + lane * S::kPad;
            const float *fr = ft + lane * S::kPad;
            const int count = t0 + T <= n ? T : static_cast<int>(n - t0);
            if (count == T) {
#pragma unroll 2
                for (int i = 0; i < T; i += 4) {
                    float4 q = *reinterpret_cast<const float4 *>(xr + i);
                    const float4 f = *reinterpret_cast<const float4 *>(fr + i);
#ifdef SWEPT_ABLATE  // memory skeleton only (a build for measuring it, never the library's): loads, LDS, stores
                    q.x += f.x; q.y += f.y; q.z += f.z; q.w += f.w;
#else
                    q = swept_step4<KIND>(q, f, r, rr, st);
#endif
                    *reinterpret_cast<float4 *>(xr + i) = q;
                }
            } else {
#pragma unroll 1
                for (int i = 0; i < count; ++i) xr[i] = swept_step<KIND>(xr[i], fr[i], r, rr, st);
            }
        }
        __syncthreads();
        if (ALIGNED && ch0 + kWave <= channels && t0 + T <= n) {  // a full tile: LDS reads and stores without bounds tests
            const float *src = xt + (lane / S::kLanesPerRow) * S::kPad + (lane % S::kLanesPerRow) * 4;
            float *p = y + (ch0 + lane / S::kLanesPerRow) * n + t0 + (lane % S::kLanesPerRow) * 4;
            const size_t step = S::kRowsPerIter * n;
#pragma unroll
            for (int it = 0; it < S::kIters; ++it) {
                const float4 q = *reinterpret_cast<const float4 *>(src + it * S::kRowsPerIter * S::kPad);
                store_stream_f4(p + it * step, q.x, q.y, q.z, q.w);
            }
            __syncthreads();
            continue;
        }
#pragma unroll
        for (int it = 0; it < S::kIters; ++it) {
            const size_t c = ch0 + it * S::kRowsPerIter + lane / S::kLanesPerRow;
            const size_t g = t0 + (lane % S::kLanesPerRow) * 4;
            if (c >= channels || g >= n) continue;
            const float *src = xt + (it * S::kRowsPerIter + lane / S::kLanesPerRow) * S::kPad + (lane % S::kLanesPerRow) * 4;
            float *p = y + c * n + g;
            if (ALIGNED && g + 3 < n) {
                const float4 q = *reinterpret_cast<const float4 *>(src);
                store_stream_f4(p, q.x, q.y, q.z, q.w);
            } else {
                p[0] = src[0];
                if (g + 1 < n) p[1] = src[1];
                if (g + 2 < n) p[2] = src[2];
                if (g + 3 < n) p[3] = src[3];
            }
        }
        __syncthreads();
    }
    if (live && state) {
        state[ch * 4 + 0] = st.x1;
        state[ch * 4 + 1] = st.x2;
        state[ch * 4 + 2] = st.y1;
        state[ch * 4 + 3] = st.y2;
    }
}

// ONE sweep for all channels (fc_stride == 0): the design of a sample is the same in every lane, so it is done once per
// WAVE and sample instead of once per lane and sample -- lane i of the wave designs sample t0 + i of the tile and leaves
// {a0, a1, a2, b1} in LDS, from where the recursion reads them back as broadcasts (every lane the same address).  What
// is left per lane and sample is the BiQuad step itself, and the kernel is bound by its 8 B/sample of HBM traffic.
// Bit-identical to the per-lane design: it is the same function on the same value.  Rows of 128 samples as in
// iir_batch_kernel (512-byte segments; 64-sample rows stop at the 4.4 TB/s of 256-byte segments: 1.93 against 1.6 ms),
// two designs per lane and tile.
template <int KIND, bool ALIGNED>
__global__ __launch_bounds__(kWave) void biquad_swept_shared_kernel(const float *__restrict__ x, const float *__restrict__ fc,
                                                                     float *__restrict__ y, size_t channels, size_t n, double r,
                                                                     double rr, double *__restrict__ state, size_t groups,
                                                                     size_t chunk) {
    constexpr int T = 2 * kWave, kPerLane = T / kWave;
    using S = SweptShape<T>;
    __shared__ __attribute__((aligned(16))) float xt[kWave * S::kPad];
    __shared__ __attribute__((aligned(16))) double co[T * 4];
    const int lane = threadIdx.x;
    const size_t group = xcd_tile_of(blockIdx.x, chunk);
    if (group >= groups) return;
    const size_t ch0 = group * kWave;
    const size_t ch = ch0 + lane;
    const bool live = ch < channels;

    SweptState st{0.0, 0.0, 0.0, 0.0};
    if (live && state) {
        st.x1 = state[ch * 4 + 0];
        st.x2 = state[ch * 4 + 1];
        st.y1 = state[ch * 4 + 2];
        st.y2 = state[ch * 4 + 3];
    }
    const double b2 = -rr;
    const size_t ntiles = (n + T - 1) / T;
    float4 px[S::kIters];
    swept_load<T, ALIGNED>(x, n, channels, n, ch0, 0, lane, px);
    float pf[kPerLane];
#pragma unroll
    for (int j = 0; j < kPerLane; ++j)  // past the end: any harmless frequency, never used
        pf[j] = static_cast<size_t>(lane + kWave * j) < n ? fc[lane + kWave * j] : 0.25f;
    for (size_t tl = 0; tl < ntiles; ++tl) {
        const size_t t0 = tl * T;
        swept_stage<T>(xt, lane, px);
#pragma unroll
        for (int j = 0; j < kPerLane; ++j) {
            double a0, a1, a2, b1;
            swept_design<KIND>(swept_cos(pf[j]), r, rr, a0, a1, a2, b1);
            *reinterpret_cast<double2 *>(co + (lane + kWave * j) * 4) = make_double2(a0, a1);
            *reinterpret_cast<double2 *>(co + (lane + kWave * j) * 4 + 2) = make_double2(a2, b1);
        }
        __syncthreads();
        if (tl + 1 < ntiles) {
            swept_load<T, ALIGNED>(x, n, channels, n, ch0, t0 + T, lane, px);
#pragma unroll
            for (int j = 0; j < kPerLane; ++j) pf[j] = t0 + T + lane + kWave * j < n ? fc[t0 + T + lane + kWave * j] : 0.25f;
        }
        if (live) {
            float *xr = xt + lane * S::kPad;
            const int count = t0 + T <= n ? T : static_cast<int>(n - t0);
            if (count == T) {
#pragma unroll 4
                for (int i = 0; i < T; i += 4) {
                    float4 q = *reinterpret_cast<const float4 *>(xr + i);
                    float *e = reinterpret_cast<float *>(&q);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double2 c01 = *reinterpret_cast<const double2 *>(co + (i + k) * 4);
                        const double2 c23 = *reinterpret_cast<const double2 *>(co + (i + k) * 4 + 2);
                        e[k] = swept_biquad(e[k], c01.x, c01.y, c23.x, c23.y, b2, st);
                    }
                    *reinterpret_cast<float4 *>(xr + i) = q;
                }
            } else {
#pragma unroll 1
                for (int i = 0; i < count; ++i) xr[i] = swept_biquad(xr[i], co[i * 4], co[i * 4 + 1], co[i * 4 + 2], co[i * 4 + 3], b2, st);
            }
        }
        __syncthreads();
        if (ALIGNED && ch0 + kWave <= channels && t0 + T <= n) {
            const float *src = xt + (lane / S::kLanesPerRow) * S::kPad + (lane % S::kLanesPerRow) * 4;
            float *p = y + (ch0 + lane / S::kLanesPerRow) * n + t0 + (lane % S::kLanesPerRow) * 4;
            const size_t step = S::kRowsPerIter * n;
#pragma unroll
            for (int it = 0; it < S::kIters; ++it) {
                const float4 q = *reinterpret_cast<const float4 *>(src + it * S::kRowsPerIter * S::kPad);
                store_stream_f4(p + it * step, q.x, q.y, q.z, q.w);
            }
            __syncthreads();
            continue;
        }
#pragma unroll
        for (int it = 0; it < S::kIters; ++it) {
            const size_t c = ch0 + it * S::kRowsPerIter + lane / S::kLanesPerRow;
            const size_t g = t0 + (lane % S::kLanesPerRow) * 4;
            if (c >= channels || g >= n) continue;
            const float *src = xt + (it * S::kRowsPerIter + lane / S::kLanesPerRow) * S::kPad + (lane % S::kLanesPerRow) * 4;
            float *p = y + c * n + g;
            if (ALIGNED && g + 3 < n) {
                const float4 q = *reinterpret_cast<const float4 *>(src);
                store_stream_f4(p, q.x, q.y, q.z, q.w);
            } else {
                p[0] = src[0];
                if (g + 1 < n) p[1] = src[1];
                if (g + 2 < n) p[2] = src[2];
                if (g + 3 < n) p[3] = src[3];
            }
        }
        __syncthreads();
    }
    if (live && state) {
        state[ch * 4 + 0] = st.x1;
        state[ch * 4 + 1] = st.x2;
        state[ch * 4 + 2] = st.y1;
        state[ch * 4 + 3] = st.y2;
    }
}

template <int KIND, bool ALIGNED>
int swept_dispatch(const float *x, const float *fc, float *y, size_t channels, size_t n, size_t fc_stride, double bw,
                   double *state, hipStream_t s) {
    // 32-sample rows: 2 x 9 KB of LDS per one-wave workgroup, so eight of them fit a CU = two waves per SIMD, which this
    // arithmetic-bound kernel needs more than it needs long row segments (measured on 262,144 x 4,096, per-channel
    // sweeps, at the time: 32-sample rows 3.33 ms, 64: 3.76 ms, 128: 7.6 ms -- two workgroups per CU and spills).  As it
    // stands the kernel runs at 0.95 of its own memory skeleton (2.93 against 2.77 ms with the arithmetic compiled out);
    // the skeleton improves with the row length (64: 2.62 ms, 128: 2.12 ms) but the arithmetic then has one wave per SIMD
    // or less to run on (64-sample rows, everything: 3.15 ms).
#ifdef SWEPT_T  // other row lengths, for the same measurement (profiles/r02_biquad_swept_ablation.txt)
    constexpr int T = SWEPT_T;
#else
    constexpr int T = 32;
#endif
    const size_t groups = (channels + kWave - 1) / kWave;
    const size_t chunk = (groups + 7) / 8;
    if (8 * chunk > 0x7fffffffull) return fail(DSP_GPU_EUNSUPPORTED, "swept biquad: too many channels");
    const double r = 1 - 3 * bw;  // recursive_filters.h:168
    if (fc_stride == 0) {
        // (two workgroups per CU instead of four, which helps iir_batch_kernel, costs this one a third: 1.46 -> 2.14 ms)
        hipLaunchKernelGGL((biquad_swept_shared_kernel<KIND, ALIGNED>), dim3(static_cast<unsigned>(8 * chunk)), dim3(kWave), 0, s,
                           x, fc, y, channels, n, r, r * r, state, groups, chunk);
        return check_launch("biquad_swept_shared_kernel");
    }
    hipLaunchKernelGGL((biquad_swept_kernel<KIND, T, ALIGNED>), dim3(static_cast<unsigned>(8 * chunk)), dim3(kWave), 0, s,
                       x, fc, y, channels, n, fc_stride, r, r * r, state, groups, chunk);
    return check_launch("biquad_swept_kernel");
}

}  // namespace

namespace {
int swept_plain(const float *x, const float *fc, float *y, size_t channels, size_t n, size_t fc_stride, int kind,
                double bandwidth, double *state, hipStream_t s);
}

// A few long voices (the wind generator is ONE, 60 s at 44.1 kHz: eval/signals/wind_generator.h:150-167): one lane per voice walks
// 2^20 samples in 194 ms, slower than the reference's loop on a CPU core.  The section is redesigned at every sample, but its pole
// radius R = 1 - 3 BW never changes (recursive_filters.h:165-209), so the response to any state decays like R^t whatever the sweep
// does: the voice is sharded in TIME as the IIR is (iir_batch_ext.hip) -- segments of L samples, each started from the state that
// W samples of warm-up from rest leave, W = 2 * 44 ln 2 / -ln R (2^-88: the factor two covers the non-normal growth of a section
// tuned near 0 or 0.5).  Sweeps must run along with the samples: one voice, or every voice its own sweep (fc_stride == n).
int launch_biquad_swept(const float *x, const float *fc, float *y, size_t channels, size_t n, size_t fc_stride, int kind,
                        double bandwidth, double *state, hipStream_t s) {
    const double R = 1.0 - 3.0 * bandwidth;
    const bool shardable = channels <= 4096 && (channels == 1 || fc_stride == n) && R > 0.0 && R < 0.99999 && !getenv("DSP_GPU_SWEPT_NO_SHARDS");
    if (shardable) {
        size_t W = static_cast<size_t>(std::ceil(2.0 * 44.0 * 0.6931471805599453 / -std::log(R)));
        W = std::max<size_t>((W + 127) / 128 * 128, 128);
        size_t L = std::max<size_t>(4 * W, (channels * n + 32767) / 32768);
        L = (L + 127) / 128 * 128;
        {
            size_t p2 = 128;
            while (p2 < L) p2 *= 2;
            if (p2 <= 8192) L = p2;
        }
        bool ok = W <= 65536 && n >= 2 * L;
        if (ok && channels > 1) {  // the view [channels * S][L] needs n = S * L
            size_t found = 0;
            for (size_t cand = L; cand <= 2 * L && cand <= n && !found; cand += 128)
                if (n % cand == 0) found = cand;
            for (size_t cand = 4 * W; cand <= 4 * L && cand <= n && !found; cand += 4)
                if (n % cand == 0) found = cand;
            ok = found != 0;
            if (ok) L = found;
            if (!ok && channels <= 64) {  // an odd or near-prime length: the few voices one after the other, each with its remainder
                for (size_t c = 0; c < channels; ++c)
                    if (int rc = launch_biquad_swept(x + c * n, fc + c * n, y + c * n, 1, n, n, kind, bandwidth, state ? state + 4 * c : nullptr, s))
                        return rc;
                return DSP_GPU_OK;
            }
        }
        if (ok) {
            const size_t S = n / L, tail = n - S * L, G = channels * S;
            void *p = nullptr;
            if (int rc = workspace(WS_SCRATCH_B, 2 * G * W * sizeof(float) + G * 4 * sizeof(double), &p, s)) return rc;
            double *st = static_cast<double *>(p);  // (first: 8-byte aligned whatever G and W are)
            float *xw = reinterpret_cast<float *>(st + G * 4), *fw = xw + G * W;
            const size_t row = 4 * sizeof(double);
            // the W samples (and sweep values) in front of every segment but the first; a voice's first segment gets the previous
            // voice's tail here and the caller's state below
            DSPGPU_HIP(hipMemcpy2DAsync(xw + W, W * sizeof(float), x + L - W, L * sizeof(float), W * sizeof(float), G - 1, hipMemcpyDeviceToDevice, s));
            DSPGPU_HIP(hipMemcpy2DAsync(fw + W, W * sizeof(float), fc + L - W, L * sizeof(float), W * sizeof(float), G - 1, hipMemcpyDeviceToDevice, s));
            DSPGPU_HIP(hipMemsetAsync(st, 0, G * row, s));
            if (int rc = swept_plain(xw + W, fw + W, xw + W, G - 1, W, W, kind, bandwidth, st + 4, s)) return rc;
            if (state) DSPGPU_HIP(hipMemcpy2DAsync(st, S * row, state, row, row, channels, hipMemcpyDeviceToDevice, s));
            else DSPGPU_HIP(hipMemset2DAsync(st, S * row, 0, row, channels, s));
            if (int rc = swept_plain(x, fc, y, G, L, L, kind, bandwidth, st, s)) return rc;
            if (tail)  // (channels == 1)
                if (int rc = swept_plain(x + S * L, fc + S * L, y + S * L, 1, tail, tail, kind, bandwidth, st + (G - 1) * 4, s)) return rc;
            if (state) DSPGPU_HIP(hipMemcpy2DAsync(state, row, st + (S - 1) * 4, S * row, row, channels, hipMemcpyDeviceToDevice, s));
            return DSP_GPU_OK;
        }
    }
    return swept_plain(x, fc, y, channels, n, fc_stride, kind, bandwidth, state, s);
}

namespace {
int swept_plain(const float *x, const float *fc, float *y, size_t channels, size_t n, size_t fc_stride, int kind,
                double bandwidth, double *state, hipStream_t s) {
    const bool al = n % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(fc) % 16 == 0;
    if (kind == 0)
        return al ? swept_dispatch<0, true>(x, fc, y, channels, n, fc_stride, bandwidth, state, s)
                  : swept_dispatch<0, false>(x, fc, y, channels, n, fc_stride, bandwidth, state, s);
    if (kind == 1)
        return al ? swept_dispatch<1, true>(x, fc, y, channels, n, fc_stride, bandwidth, state, s)
                  : swept_dispatch<1, false>(x, fc, y, channels, n, fc_stride, bandwidth, state, s);
    return fail(DSP_GPU_EINVAL, "swept biquad: kind %d", kind);
}
}  // namespace

}  // namespace dspgpu
