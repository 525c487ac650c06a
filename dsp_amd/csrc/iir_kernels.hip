// iir_kernels.hip -- IIR recursion for gfx950 (MI355X).
//
// Stands in for dsp::Coefficients<Poles>::operator()(const SampleBuffer&) and its BiQuad
// specialisation (reference coefficients.h:88-130, 172-195):
//     v[t] = sum_{i=0..P} a[i] x[t-i] + sum_{i=1..P} b[i] v[t-i]        (feedback ADDED)
//     y[t] = v[t]*c0 + x[t]*d0
// The reference keeps coefficients and history in double (coefficients.h:9-12); so do these
// kernels (the path is HBM-bound: fp64 FMAs are affordable), samples are fp32 in HBM.
//
// Kernels
//   iir_batch_kernel   one lane per channel (C4a).  A wave owns 64 channels and walks time in
//                      tiles of 128 samples staged through a padded LDS tile, so HBM sees 512-byte
//                      runs per channel while every lane walks its own 16-byte aligned row with
//                      conflict-free 16-byte LDS accesses; the next tile's loads are in flight while
//                      the current one is computed.  Also runs one coefficient set per lane.
//   iir_scan_kernel    block-parallel linear-recurrence scan inside one channel (C4b).  A workgroup
//                      is ONE wave and owns a tile of 4096 contiguous samples = 64 rows of 64.
//                      Phase A: every lane runs its row from zero feedback state and leaves the
//                      zero-state outputs in LDS.  Phase B: Kogge-Stone scan over the 64 row-end states
//                      with the companion-matrix powers A^(64*2^j), by wave shuffles.  Phase C: every
//                      lane adds its true start state's contribution (a 64 x P table of A^(t+1) rows).
//                      Across tiles either (REDUCE -> iir_carry_kernel -> APPLY: exact carries,
//                      12 B/sample) or (FUSED: tiles overlap by W warm-up samples, chosen on the
//                      host so that ||A^W|| <= 2^-44, i.e. the truncated history is below double
//                      rounding; 8 B/sample + 4 B * W/(T-W)).  One workgroup per tile.
//   iir_carry_kernel   scan of the per-tile end states (one workgroup per channel).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <type_traits>
#include <vector>

#include "common.h"
#include "iir_batch_core.h"

namespace dspgpu {
// Ablation switches for scripts/iir_scan_ablate.hip, which compiles this file with DSPGPU_SCAN_ABLATE defined and times the
// scan kernel with phases switched off (1 = phase A, 2 = phase B, 4 = phase C).  In the library build they are
// compile-time false and leave no trace in the kernels.
#ifdef DSPGPU_SCAN_ABLATE
__device__ int g_scan_ablate = 0;
#define SCAN_ABLATED(bit) ((g_scan_ablate & (bit)) != 0)
#else
#define SCAN_ABLATED(bit) false
#endif
namespace {

using namespace iirb;  // the recursion step, the row runners and the batch kernel: iir_batch_core.h

// the float rows of the scan below, in the fast (double, fused multiply-add) arithmetic
template <int P, int LEN, bool STORE, bool UNIT = false, typename C>
DSP_DEV void run_row4(float *row, double (&xh)[P], double (&vh)[P], const C &co) {
    run_rowv<P, LEN, STORE, IIR_ARITH_FMA, UNIT>(row, xh, vh, co);
}

// ---------------------------------------------------------------------------------------------
// scan: block-parallel inside a channel
// ---------------------------------------------------------------------------------------------
constexpr int kSt = 64;             // threads = rows per tile: ONE WAVE per workgroup, so a tile needs no workgroup
                                    // barrier, no cross-wave carry, and 16+ independent waves fit a CU (8.6 KB of LDS each)
constexpr int kRow = 64;            // samples per row
constexpr int kRowPad = kRow + 4;    // 68 floats: rows stay 16-byte aligned and lanes walking their rows with 16-byte
                                    // LDS accesses (lane stride 68 = 4 mod 64 dwords) hit 16 disjoint 4-bank groups: conflict-free
constexpr int kTile = kSt * kRow;   // 4096 samples per workgroup (measured: 32-sample rows 541, 64: 560, 128: 493 GSamples/s)
constexpr int kLevels = 6;          // log2(kSt)
constexpr int kCt = 256;            // threads of the tile-carry kernel
constexpr int kCtLevels = 8;        // log2(kCt)

enum ScanMode { SCAN_REDUCE = 0, SCAN_APPLY = 1, SCAN_FUSED = 2 };

DSP_DEV double shfl_up_f64(double v, int delta) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_up(lo, delta, 64);
    hi = __shfl_up(hi, delta, 64);
    return __hiloint2double(hi, lo);
}

// ---- double-double helpers (error-free transformations) -------------------------------------------
// Contraction must stay OFF around them: with -ffp-contract=fast the backend turns `p + e` (p = x.hi*y.hi
// rounded, e = its exact error + cross terms) into fma(x.hi, y.hi, e), which adds the product's rounding
// error twice and silently degrades the result to plain double (found in the ISA).  The pragma alone does
// not stop the backend, so the Makefile builds this file with -ffp-contract=off; the recursion itself
// calls fma() explicitly.
struct DD {
    double hi, lo;
};
__host__ __device__ inline DD dd_two_sum(double a, double b) {
#pragma clang fp contract(off)

    const double s = a + b, bb = s - a;
    return DD{s, (a - (s - bb)) + (b - bb)};
}
__host__ __device__ inline DD dd_add(DD x, DD y) {
#pragma clang fp contract(off)

    const DD s = dd_two_sum(x.hi, y.hi);
    const double e = s.lo + (x.lo + y.lo);
    const double hi = s.hi + e;
    return DD{hi, e - (hi - s.hi)};
}
__host__ __device__ inline DD dd_mul(DD x, DD y) {
#pragma clang fp contract(off)

    const double p = x.hi * y.hi;
    double e = fma(x.hi, y.hi, -p);
    e += x.hi * y.lo + x.lo * y.hi;
    const double hi = p + e;
    return DD{hi, e - (hi - p)};
}

// A feedback state (v[t-1..t-P]) carried across rows / tiles: plain double, or double-double when the
// companion-matrix powers are too large for double carries (poles clustered at the unit circle).
template <int P, bool WIDE>
struct SVec {
    double hi[P];
    double lo[WIDE ? P : 1];
};
template <int P, bool WIDE>
constexpr int kWords = WIDE ? 2 * P : P;  // doubles per stored state / matrix row block

template <int P, bool WIDE>
DSP_DEV void sv_zero(SVec<P, WIDE> &s) {
#pragma unroll
    for (int i = 0; i < P; ++i) {
        s.hi[i] = 0.0;
        if (WIDE) s.lo[i] = 0.0;
    }
}
template <int P, bool WIDE>
DSP_DEV void sv_load(SVec<P, WIDE> &s, const double *p) {  // layout: P hi [, P lo]
#pragma unroll
    for (int i = 0; i < P; ++i) {
        s.hi[i] = p[i];
        if (WIDE) s.lo[i] = p[P + i];
    }
}
template <int P, bool WIDE>
DSP_DEV void sv_store(const SVec<P, WIDE> &s, double *p) {
#pragma unroll
    for (int i = 0; i < P; ++i) {
        p[i] = s.hi[i];
        if (WIDE) p[P + i] = s.lo[i];
    }
}
template <int P, bool WIDE>
DSP_DEV SVec<P, WIDE> sv_shfl_up(const SVec<P, WIDE> &s, int delta) {
    SVec<P, WIDE> o;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        o.hi[i] = shfl_up_f64(s.hi[i], delta);
        if (WIDE) o.lo[i] = shfl_up_f64(s.lo[i], delta);
    }
    return o;
}
// s += M * o.  M: P*P doubles (hi) [followed by P*P doubles (lo)], row-major.
template <int P, bool WIDE>
DSP_DEV void sv_mat_acc(SVec<P, WIDE> &s, const double *__restrict__ M, const SVec<P, WIDE> &o) {
#pragma unroll
    for (int r = 0; r < P; ++r) {
        if (WIDE) {
            // compensated dot product: exact two-product of the leading parts, cross terms and every rounding
            // error gathered in `lo`, one renormalisation per row (12 instead of 19 operations per term)
            double hi = s.hi[r], lo = s.lo[r];
#pragma unroll
            for (int c = 0; c < P; ++c) {
                const double mh = M[r * P + c], ml = M[P * P + r * P + c];
                const double p = mh * o.hi[c];
                double e = fma(mh, o.hi[c], -p);
                e = fma(mh, o.lo[c], e);
                e = fma(ml, o.hi[c], e);
                const DD t = dd_two_sum(hi, p);
                hi = t.hi;
                lo += t.lo + e;
            }
            const double h2 = hi + lo;
            s.lo[r] = lo - (h2 - hi);
            s.hi[r] = h2;
        } else {
            double acc = s.hi[r];
#pragma unroll
            for (int c = 0; c < P; ++c) acc = fma(M[r * P + c], o.hi[c], acc);
            s.hi[r] = acc;
        }
    }
}
template <int P, bool WIDE>
DSP_DEV void sv_add(SVec<P, WIDE> &s, const SVec<P, WIDE> &o) {
#pragma unroll
    for (int i = 0; i < P; ++i) {
        if (WIDE) {
            const DD r = dd_add(DD{s.hi[i], s.lo[i]}, DD{o.hi[i], o.lo[i]});
            s.hi[i] = r.hi;
            s.lo[i] = r.lo;
        } else {
            s.hi[i] += o.hi[i];
        }
    }
}

// Inclusive Kogge-Stone scan over the lanes of a wave: S_l += A^(32*2^j) S_{l-2^j}, j < levels (<= 6).
template <int P, bool WIDE>
DSP_DEV void wave_scan(SVec<P, WIDE> &S, const double *__restrict__ mats, int levels, int lane) {
    constexpr int MS = kWords<P, WIDE> * P;  // doubles per matrix slot
#pragma unroll 1
    for (int j = 0; j < levels; ++j) {
        const SVec<P, WIDE> o = sv_shfl_up<P, WIDE>(S, 1 << j);
        if (lane >= (1 << j)) sv_mat_acc<P, WIDE>(S, mats + j * MS, o);
    }
}

// mats: slots [0..8] = A^(32*2^j) (j = 8 is the whole-tile matrix), companion A in the vh ordering.
// UNIT: the host has checked c0 == 1, d0 == 0 (iir_step_out) -- a separate instantiation, not a branch inside the
// kernel: with both row runners and both correction loops in one kernel it ran 7 % slower instead of 6 % faster
// (profiles/r03_iir_scan_unit_stage_ab.txt; not an instruction-cache effect, scripts/pmc_icache.sh).
template <int P, int MODE, bool ALIGNED, bool WIDE, bool UNIT = false>
__global__ __launch_bounds__(kSt) void iir_scan_kernel(const float *__restrict__ x, float *__restrict__ y, size_t n,
                                                       IirCoef co, const double *__restrict__ mats,
                                                       const double *__restrict__ state_in,
                                                       double *__restrict__ state_out, double *__restrict__ tile_z,
                                                       const double *__restrict__ tile_c, unsigned ntiles, int warm,
                                                       int levels, const double *__restrict__ hrow, int carry_terms) {
    constexpr int W = kWords<P, WIDE>;
    constexpr int MS = W * P;
    __shared__ __attribute__((aligned(16))) float xs[(kSt + 1) * kRowPad];
    const int tid = threadIdx.x;
    const size_t ch = blockIdx.y;
    const float *xc = x + ch * n;
    float *yc = y + ch * n;
    const double *sin_c = state_in ? state_in + ch * 2 * P : nullptr;
    // Persistent workgroups: XCD x = blockIdx.x & 7 owns the contiguous tile range [x*chunk, (x+1)*chunk) and its
    // workgroups take those tiles round-robin, loading the NEXT tile's samples into registers while the
    // current one is scanned (a tile's life was mostly waiting: SQ_WAIT_ANY 0.61 of wave cycles).
    // (channels with only a few tiles skip the XCD map: its grid of at least 8 workgroups per channel would be
    // mostly workgroups that exit at once, and one-wave workgroups are dispatch-rate bound)
    const bool xcd_map = (gridDim.x & 7u) == 0 && ntiles >= 64u;
    const unsigned chunk = xcd_map ? (ntiles + 7u) / 8u : ntiles, wgs_per_xcd = xcd_map ? gridDim.x >> 3 : gridDim.x;
    const unsigned t_lo = xcd_map ? (blockIdx.x & 7u) * chunk : 0u, t_hi = min(ntiles, t_lo + chunk);
    unsigned tile = t_lo + (xcd_map ? blockIdx.x >> 3 : blockIdx.x);
    if (tile >= t_hi) return;

    // tile geometry: samples [start, start + kTile), outputs kept from out_begin on
    auto tile_start = [&](unsigned tl) -> size_t {
        if (MODE == SCAN_FUSED && tl > 0) return static_cast<size_t>(kTile) + static_cast<size_t>(tl - 1) * (kTile - warm) - warm;
        return static_cast<size_t>(tl) * kTile;
    };
    constexpr int kPre = kTile / (kSt * 4);  // float4 loads per thread per tile = 16
    auto tile_load = [&](float4 (&pre)[kPre], size_t start) {
        if (ALIGNED && start + kTile <= n) {  // whole tile inside the channel: no bounds tests between the loads
#pragma unroll
            for (int q = 0; q < kPre; ++q) pre[q] = load_stream_f4(xc + start + tid * 4 + q * kSt * 4);
            return;
        }
#pragma unroll
        for (int q = 0; q < kPre; ++q) {
            const size_t g = start + static_cast<size_t>(tid * 4 + q * kSt * 4);
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (ALIGNED && g + 3 < n) {
                v = load_stream_f4(xc + g);
            } else {
                if (g < n) v.x = xc[g];
                if (g + 1 < n) v.y = xc[g + 1];
                if (g + 2 < n) v.z = xc[g + 2];
                if (g + 3 < n) v.w = xc[g + 3];
            }
            pre[q] = v;
        }
    };
    float4 pre[kPre];
    tile_load(pre, tile_start(tile));

    for (; tile < t_hi; tile += wgs_per_xcd) {
    const size_t start = tile_start(tile);
    const size_t out_begin = (MODE == SCAN_FUSED && tile > 0) ? start + warm : start;

    // stage the tile (zeros past the end of the channel) and the P samples of history before it
#pragma unroll
    for (int q = 0; q < kPre; ++q) {
        const int u = tid * 4 + q * kSt * 4;
        *reinterpret_cast<float4 *>(xs + (u / kRow + 1) * kRowPad + (u % kRow)) = pre[q];
    }
    if (tid < P) {  // row "-1", column 31 - i holds x[start - 1 - i]
        float h = 0.0f;
        if (start > static_cast<size_t>(tid)) h = xc[start - 1 - tid];
        else if (sin_c) h = static_cast<float>(sin_c[tid - start]);  // start == 0: the caller's x history
        xs[kRow - 1 - tid] = h;
    }
    __syncthreads();
    if (tile + wgs_per_xcd < t_hi) tile_load(pre, tile_start(tile + wgs_per_xcd));  // in flight during the scan

    float *row = xs + (tid + 1) * kRowPad;
    double xh0[P], xh[P], vh[P];
#pragma unroll
    for (int i = 0; i < P; ++i) xh0[i] = static_cast<double>(xs[tid * kRowPad + kRow - 1 - i]);
    __syncthreads();  // phase A overwrites the rows: every lane has its x history (the row before's tail) first

    // phase A: zero feedback state.  Rows other than the one holding the channel's last sample overwrite
    // their samples with the zero-state OUTPUT (fp32): phase C then only adds the start state's
    // contribution instead of running the recursion a second time.
    const size_t g0 = start + static_cast<size_t>(tid) * kRow;
    const bool last_row = g0 < n && n <= g0 + kRow;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        xh[i] = xh0[i];
        vh[i] = 0.0;
    }
    if (SCAN_ABLATED(1)) {
    } else if (MODE == SCAN_REDUCE || last_row) run_row4<P, kRow, false>(row, xh, vh, co);
    else run_row4<P, kRow, true, UNIT>(row, xh, vh, co);
    SVec<P, WIDE> S;
    sv_zero<P, WIDE>(S);
#pragma unroll
    for (int i = 0; i < P; ++i) S.hi[i] = vh[i];

    // carry into the tile
    SVec<P, WIDE> carry;
    sv_zero<P, WIDE>(carry);
    if (MODE == SCAN_APPLY) {
        if (carry_terms > 0) {
            // C_k = Z_{k-1} + MT (Z_{k-2} + MT (Z_{k-3} + ...)): the tile matrix MT = A^4096 of a stable filter decays so
            // fast that a handful of terms is exact to 2^-60 (host), so every tile builds its own carry from the
            // end states the REDUCE pass left and the serial carry kernel is not needed ("Z_{-1}" = caller's state)
            for (int j = carry_terms; j >= 1; --j) {
                const long long kk = static_cast<long long>(tile) - j;
                SVec<P, WIDE> nxt;
                sv_zero<P, WIDE>(nxt);
                if (kk >= 0) sv_load<P, WIDE>(nxt, tile_z + (ch * ntiles + kk) * W);
                else if (kk == -1 && sin_c)
#pragma unroll
                    for (int i = 0; i < P; ++i) nxt.hi[i] = sin_c[P + i];
                sv_mat_acc<P, WIDE>(nxt, mats + 9 * MS, carry);
                carry = nxt;
            }
        } else {
            sv_load<P, WIDE>(carry, tile_c + (ch * ntiles + tile) * W);
        }
    } else if (MODE == SCAN_FUSED && tile == 0 && sin_c)
#pragma unroll
        for (int i = 0; i < P; ++i) carry.hi[i] = sin_c[P + i];
    if (tid == 0) sv_mat_acc<P, WIDE>(S, mats, carry);

    // phase B: inclusive scan of the row-end states, S_l = Z_l + A^32 S_{l-1}: a Kogge-Stone scan over the
    // lanes of the wave with shuffles.  `levels` is the number of doubling levels whose matrix is not
    // negligible (host: ||A^(32*2^j)|| > 2^-60), so a fast-decaying filter stops early.
    const int lane = tid;
    if (!SCAN_ABLATED(2)) wave_scan<P, WIDE>(S, mats, levels < kLevels ? levels : kLevels, lane);
    const SVec<P, WIDE> cw = carry;  // true state at the start of the tile
    if (MODE == SCAN_REDUCE) {
        if (tid == kSt - 1) sv_store<P, WIDE>(S, tile_z + (ch * ntiles + tile) * W);
        continue;  // the barrier above ordered this tile's last LDS reads before the next tile's staging
    }

    // phase C: true start state = inclusive state of the row before (the carry for the first row of a
    // wave), rounded to double exactly as the sequential recursion would hold it
    {
        const SVec<P, WIDE> prev = sv_shfl_up<P, WIDE>(S, 1);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            xh[i] = xh0[i];
            const double ph = lane == 0 ? cw.hi[i] : prev.hi[i];
            const double pl = WIDE ? (lane == 0 ? cw.lo[i] : prev.lo[i]) : 0.0;
            vh[i] = ph + pl;
        }
    }
    if (last_row) {  // the exact recursion from the true start state; also yields the state handed back
        run_row_partial<P, true, IIR_ARITH_FMA, UNIT>(row, static_cast<int>(n - g0), xh, vh, co);
        if (state_out) {
            double *so = state_out + ch * 2 * P;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                so[i] = xh[i];
                so[P + i] = vh[i];
            }
        }
    } else if (g0 < n && !SCAN_ABLATED(4)) {
        // v[t] = v_zero_state[t] + sum_i hrow[t][i] * v[-1-i]: hrow[t] is the first row of A^(t+1) (host, from
        // the recursion run on unit states), uniform across lanes (scalar loads), no loop-carried dependency
#pragma unroll
        for (int t4 = 0; t4 < kRow; t4 += 4) {
            float4 q = *reinterpret_cast<const float4 *>(row + t4);
            float *e = reinterpret_cast<float *>(&q);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if constexpr (UNIT) {  // c0 == 1: the correction accumulates onto the sample itself
                    double acc = static_cast<double>(e[k]);
#pragma unroll
                    for (int i = 0; i < P; ++i) acc = fma(hrow[(t4 + k) * P + i], vh[i], acc);
                    e[k] = static_cast<float>(acc);
                } else {
                    double corr = 0.0;
#pragma unroll
                    for (int i = 0; i < P; ++i) corr = fma(hrow[(t4 + k) * P + i], vh[i], corr);
                    e[k] = static_cast<float>(fma(corr, co.c0, static_cast<double>(e[k])));
                }
            }
            *reinterpret_cast<float4 *>(row + t4) = q;
        }
    }
    __syncthreads();

    if (ALIGNED && start + kTile <= n) {  // whole tile inside the channel: LDS reads four at a time, no bounds tests
#pragma unroll
        for (int q0 = 0; q0 < kPre; q0 += 4) {
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int u = tid * 4 + (q0 + i) * kSt * 4;
                v[i] = *reinterpret_cast<const float4 *>(xs + (u / kRow + 1) * kRowPad + (u % kRow));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const size_t g = start + static_cast<size_t>(tid * 4 + (q0 + i) * kSt * 4);
                if (g >= out_begin) store_stream_f4(yc + g, v[i].x, v[i].y, v[i].z, v[i].w);
            }
        }
    } else {
        for (int u = tid * 4; u < kTile; u += kSt * 4) {
            const size_t g = start + u;
            if (g < out_begin || g >= n) continue;
            const float *src = xs + (u / kRow + 1) * kRowPad + (u % kRow);
            if (ALIGNED && g + 3 < n) {
                const float4 q = *reinterpret_cast<const float4 *>(src);
                store_stream_f4(yc + g, q.x, q.y, q.z, q.w);
            } else {
                yc[g] = src[0];
                if (g + 1 < n) yc[g + 1] = src[1];
                if (g + 2 < n) yc[g + 2] = src[2];
                if (g + 3 < n) yc[g + 3] = src[3];
            }
        }
    }
    __syncthreads();  // the tile has been read out before the next one is staged
    }  // tile loop
}

// Exclusive scan of the tile end states: C_0 = initial state, C_k = Z_{k-1} + MT C_{k-1}.
// mats2: slot [0] = MT = A^kTile, [1 + j] = MT^(G * 2^j).  Workgroup (ch, w) owns tiles [w*tiles_per_wg, ...), its
// thread t the G tiles from w*tiles_per_wg + t*G.  The carry into a workgroup's first tile comes from `warm`
// tiles of warm-up (host: ||MT^warm|| <= 2^-60, so the dropped older terms are below rounding) -- with one
// workgroup per channel (warm = 0, the fallback for filters that decay slower than that) it is the caller's state.
template <int P, bool WIDE>
__global__ __launch_bounds__(kCt) void iir_carry_kernel(const double *__restrict__ tile_z, double *__restrict__ tile_c,
                                                         const double *__restrict__ mats2,
                                                         const double *__restrict__ state_in, unsigned ntiles,
                                                         unsigned G, unsigned tiles_per_wg, unsigned warm) {
    constexpr int W = kWords<P, WIDE>;
    constexpr int MS = W * P;
    __shared__ double sbuf[W * kCt];
    const int tid = threadIdx.x;
    const size_t ch = blockIdx.x;
    const double *z = tile_z + ch * ntiles * W;
    double *c = tile_c + ch * ntiles * W;
    const unsigned base = blockIdx.y * tiles_per_wg, end = min(ntiles, base + tiles_per_wg);
    const unsigned k0 = min(end, base + tid * G), k1 = min(end, k0 + G);

    SVec<P, WIDE> s;
    sv_zero<P, WIDE>(s);
    if (tid == 0) {
        if (base == 0) {
            if (state_in)
#pragma unroll
                for (int i = 0; i < P; ++i) s.hi[i] = state_in[ch * 2 * P + P + i];
        } else {
            for (unsigned k = base - warm; k < base; ++k) {  // base >= tiles_per_wg >= warm
                SVec<P, WIDE> nxt;
                sv_load<P, WIDE>(nxt, z + static_cast<size_t>(k) * W);
                sv_mat_acc<P, WIDE>(nxt, mats2, s);
                s = nxt;
            }
        }
    }
    const SVec<P, WIDE> first = s;
    for (unsigned k = k0; k < k1; ++k) {
        SVec<P, WIDE> nxt;
        sv_load<P, WIDE>(nxt, z + static_cast<size_t>(k) * W);
        sv_mat_acc<P, WIDE>(nxt, mats2, s);
        s = nxt;
    }
#pragma unroll 1
    for (int j = 0; j < kCtLevels; ++j) {
#pragma unroll
        for (int i = 0; i < W; ++i) sbuf[i * kCt + tid] = i < P ? s.hi[i] : s.lo[WIDE ? i - P : 0];
        __syncthreads();
        const int off = 1 << j;
        if (tid >= off) {
            SVec<P, WIDE> o;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                o.hi[i] = sbuf[i * kCt + tid - off];
                if (WIDE) o.lo[i] = sbuf[(P + i) * kCt + tid - off];
            }
            sv_mat_acc<P, WIDE>(s, mats2 + (1 + j) * MS, o);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < W; ++i) sbuf[i * kCt + tid] = i < P ? s.hi[i] : s.lo[WIDE ? i - P : 0];
    __syncthreads();
    s = first;
    if (tid > 0) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            s.hi[i] = sbuf[i * kCt + tid - 1];
            if (WIDE) s.lo[i] = sbuf[(P + i) * kCt + tid - 1];
        }
    }
    for (unsigned k = k0; k < k1; ++k) {
        sv_store<P, WIDE>(s, c + static_cast<size_t>(k) * W);
        SVec<P, WIDE> nxt;
        sv_load<P, WIDE>(nxt, z + static_cast<size_t>(k) * W);
        sv_mat_acc<P, WIDE>(nxt, mats2, s);
        s = nxt;
    }
}

// ---- host side: companion-matrix powers in double-double ------------------------------------------
using Mat = std::vector<DD>;

Mat mat_mul(const Mat &a, const Mat &b, int P) {
    Mat c(static_cast<size_t>(P) * P, DD{0.0, 0.0});
    for (int i = 0; i < P; ++i)
        for (int k = 0; k < P; ++k)
            for (int j = 0; j < P; ++j) c[i * P + j] = dd_add(c[i * P + j], dd_mul(a[i * P + k], b[k * P + j]));
    return c;
}

Mat mat_identity(int P) {
    Mat m(static_cast<size_t>(P) * P, DD{0.0, 0.0});
    for (int i = 0; i < P; ++i) m[i * P + i].hi = 1.0;
    return m;
}

Mat mat_pow(Mat base, unsigned long long e, int P) {
    Mat r = mat_identity(P);
    while (e) {
        if (e & 1) r = mat_mul(r, base, P);
        e >>= 1;
        if (e) base = mat_mul(base, base, P);
    }
    return r;
}

double mat_norm_inf(const Mat &m, int P) {
    double worst = 0.0;
    for (int i = 0; i < P; ++i) {
        double row = 0.0;
        for (int j = 0; j < P; ++j) row += std::fabs(m[i * P + j].hi);
        if (!(row <= worst)) worst = row;  // also catches NaN
    }
    return worst;
}

// slot layout: P*P hi values [then P*P lo values when wide]
void store_mat(std::vector<double> &table, size_t slot, const Mat &m, int P, bool wide) {
    const size_t ms = static_cast<size_t>(wide ? 2 : 1) * P * P;
    for (int i = 0; i < P * P; ++i) {
        table[slot * ms + i] = m[i].hi;
        if (wide) table[slot * ms + P * P + i] = m[i].lo;
    }
}

// sum_k |h_k| of the all-pole part 1 / (1 - sum b_i z^-i): how much one rounding error injected into
// the recursion can grow in the output
double feedback_noise_gain(const IirCoef &co) {
    const int P = co.poles;
    double h[21] = {0};  // (poles <= 20: dsp_gpu_iir_settle takes every order the library filters, not only the scan's eight)
    double total = 1.0;
    h[0] = 1.0;  // h[i] = response i steps ago
    for (int k = 1; k < (1 << 22); ++k) {
        double v = 0.0;
        for (int i = 1; i <= P; ++i) v += co.b[i] * h[i - 1];
        for (int i = P - 1; i >= 1; --i) h[i] = h[i - 1];
        h[0] = v;
        total += std::fabs(v);
        if (!(total < 1e30)) return 1e30;  // unstable filter
        if ((k & 255) == 0) {
            double tail = 0.0;
            for (int i = 0; i < P; ++i) tail += std::fabs(h[i]);
            if (tail < 1e-18 * total) break;
        }
    }
    return total;
}

// What the host works out once per coefficient set (cached): the companion powers and the decisions
// that follow from them.
struct ScanPlan {
    IirCoef co;
    std::vector<Mat> pow32;  // A^(32*2^j), j = 0..8
    std::vector<double> hrow;  // [kRow][P]: v[t] of the zero-input run started from the unit state e_i
    int levels = 0;          // doubling levels whose matrix still matters (norm > 2^-60)
    int warm = -1;           // smallest multiple of 32 with ||A^W||_inf <= 2^-44 (<= kTile/4), else -1
    int decay = 0;           // smallest j <= 64 with ||(A^kTile)^j|| <= 2^-60 (0: slower than that)
    bool wide = false;       // carry states in double-double
    double predicted = 0.0;  // predicted relative error of the chosen carry arithmetic
    bool usable = false;
    unsigned long long serial = 0;  // identifies the plan in the device-table cache
};

// The matrix table a scan launch reads lives on the device with its plan: built and uploaded ONCE per (plan, carry
// arithmetic, carry-scan grouping) and device, then only looked up -- a streamed filter no longer pays a host-side
// rebuild (up to 64 double-double matrix products) and a pageable upload per block.  Entries are never evicted while
// the library is loaded (a few KB each, at most kMaxTables per device; beyond that a call uploads its own table into
// the shared workspace slot as before), so a pointer handed to a launch stays valid.
struct DeviceTable {
    unsigned long long serial;
    int wide, g;  // g = 0: fused / local-carry forms (no G-dependent slots)
    double *ptr;
    hipEvent_t ready;  // recorded on the uploading stream; other streams wait for it
    hipStream_t stream;
};
constexpr size_t kMaxTables = 256;
constexpr int kMaxDev = 64;
std::mutex g_table_mutex;
std::vector<DeviceTable> g_tables[kMaxDev];

bool tables_in_use(int dev) {
    std::lock_guard<std::mutex> lock(g_table_mutex);
    return dev >= 0 && dev < kMaxDev && !g_tables[dev].empty();
}

void release_tables(int dev) {  // with device `dev` current and idle
    if (dev < 0 || dev >= kMaxDev) return;
    std::lock_guard<std::mutex> lock(g_table_mutex);
    for (DeviceTable &t : g_tables[dev]) {
        (void)hipFree(t.ptr);
        (void)hipEventDestroy(t.ready);
    }
    g_tables[dev].clear();
}

template <int P, bool ALIGNED, bool WIDE>
int scan_launch(const float *x, float *y, size_t channels, size_t n, const IirCoef &co, const double *state_in,
                double *state, const ScanPlan &plan, hipStream_t s) {
    constexpr int W = kWords<P, WIDE>;
    const size_t ms = static_cast<size_t>(W) * P;
    const std::vector<Mat> &pow32 = plan.pow32;
    const int levels = plan.levels;
    int warm = plan.warm;
    const bool fused = warm > 0 || n <= static_cast<size_t>(kTile);
    if (warm < 0) warm = 0;
    unsigned ntiles;
    if (fused) ntiles = n <= static_cast<size_t>(kTile) ? 1u : 1u + static_cast<unsigned>((n - kTile + (kTile - warm) - 1) / (kTile - warm));
    else ntiles = static_cast<unsigned>((n + kTile - 1) / kTile);
    unsigned G = (ntiles + kCt - 1) / kCt, tiles_per_wg = ntiles, carry_warm = 0;
    int carry_terms = 0;  // > 0: MT^carry_terms is negligible and small: tiles sum their own carries, no carry kernel
    if (!fused) {
        const int decay = plan.decay;
        if (decay > 0 && decay <= 2) carry_terms = decay;
        constexpr unsigned kCarryG = 4;  // tiles per thread when the carry scan is cut into workgroups
        if (decay > 0 && ntiles > 4 * kCt * kCarryG) {
            G = kCarryG;
            tiles_per_wg = kCt * kCarryG;  // 1024 >= 64 >= decay
            carry_warm = static_cast<unsigned>(decay);
        }
    }
    const int gkey = (fused || carry_terms > 0) ? 0 : static_cast<int>(G);  // only the carry kernel reads the G-dependent slots

    // 18 matrix slots (0..8: A^(32*2^j); 9: MT = A^kTile; 10..: MT^(G*2^j)), then hrow[kRow][P]
    auto build_table = [&](std::vector<double> &table) {
        table.assign(static_cast<size_t>(9 + 9) * ms + plan.hrow.size(), 0.0);
        std::copy(plan.hrow.begin(), plan.hrow.end(), table.begin() + 18 * ms);
        for (int j = 0; j <= kLevels; ++j) store_mat(table, j, pow32[j], P, WIDE);
        const Mat &MT = pow32[kLevels];
        store_mat(table, 9, MT, P, WIDE);
        if (gkey) {
            Mat mg = mat_pow(MT, G, P);
            for (int j = 0; j < kCtLevels; ++j) {
                store_mat(table, 10 + j, mg, P, WIDE);
                mg = mat_mul(mg, mg, P);
            }
        }
    };
    double *d_table = nullptr;
    {
        int dev = 0;
        DSPGPU_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= kMaxDev) return fail(DSP_GPU_EUNSUPPORTED, "device index %d out of range", dev);
        std::lock_guard<std::mutex> lock(g_table_mutex);
        std::vector<DeviceTable> &tabs = g_tables[dev];
        for (const DeviceTable &t : tabs)
            if (t.serial == plan.serial && t.wide == (WIDE ? 1 : 0) && t.g == gkey) {
                d_table = t.ptr;
                if (t.stream != s) DSPGPU_HIP(hipStreamWaitEvent(s, t.ready, 0));
                break;
            }
        if (!d_table) {
            std::vector<double> table;
            build_table(table);
            if (tabs.size() < kMaxTables) {
                DeviceTable t{plan.serial, WIDE ? 1 : 0, gkey, nullptr, nullptr, s};
                if (hipMalloc(&t.ptr, table.size() * sizeof(double)) != hipSuccess)
                    return fail(DSP_GPU_ENOMEM, "IIR scan: table allocation failed");
                // pageable source: the runtime stages the bytes before returning, `table` may go out of scope
                hipError_t e = hipMemcpyAsync(t.ptr, table.data(), table.size() * sizeof(double), hipMemcpyHostToDevice, s);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&t.ready, hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventRecord(t.ready, s);
                if (e != hipSuccess) {
                    (void)hipFree(t.ptr);
                    return fail(DSP_GPU_EHIP, "IIR scan: table upload failed: %s", hipGetErrorString(e));
                }
                tabs.push_back(t);
                d_table = t.ptr;
            } else {  // cache full: this call's own copy in the shared workspace slot
                void *d_table_v = nullptr;
                if (int rc = workspace(WS_IIR_TABLE, table.size() * sizeof(double), &d_table_v, s)) return rc;
                d_table = static_cast<double *>(d_table_v);
                DSPGPU_HIP(hipMemcpyAsync(d_table, table.data(), table.size() * sizeof(double), hipMemcpyHostToDevice, s));
            }
        }
    }

    // One workgroup (= one wave) per tile, XCD x taking the contiguous eighth [x*chunk, (x+1)*chunk) of a channel's tiles.
    // Until round 2 the grid was persistent (8 workgroups per CU, each prefetching its next tile while it
    // scanned the current one); plain copies showed that persistent grid-stride streaming is the slowest way to move
    // data on this chip (profiles/r02_copy_ceiling.txt: 5.3-5.7 TB/s against 5.9-6.5 for one tile per workgroup: what
    // counts is how little a CU keeps in flight), and the scan followed: 0.439 -> 0.409 ms on C4b, same box.  The
    // kernel's loop still takes any grid (DSP_GPU_IIR_SCAN_WGS_PER_CU=8 restores the persistent launch for A/B).
    static const unsigned wgs_per_cu = [] {
        const char *e = getenv("DSP_GPU_IIR_SCAN_WGS_PER_CU");
        return e && atoi(e) > 0 ? static_cast<unsigned>(atoi(e)) : 0u;
    }();
    const unsigned per_xcd = wgs_per_cu ? std::min((ntiles + 7u) / 8u, 32u * wgs_per_cu) : (ntiles + 7u) / 8u;
    const dim3 grid(ntiles >= 64u ? 8u * per_xcd : ntiles, static_cast<unsigned>(channels));
    // (capping the one-wave workgroups per CU with unused LDS, which helps iir_batch_kernel, costs the scan: 14 per CU
    // 0.446 ms, 8 per CU 0.502, 4 per CU 0.641 on a box where the plain launch took 0.436-0.446)
    if (fused) {
        if constexpr (P <= 4 && ALIGNED && !WIDE) {  // the default output stage (iir_step_out), the common orders only
            if (co.c0 == 1.0 && co.d0 == 0.0) {
                hipLaunchKernelGGL((iir_scan_kernel<P, SCAN_FUSED, ALIGNED, WIDE, true>), grid, dim3(kSt), 0, s, x, y, n, co, d_table,
                                   state_in, state, nullptr, nullptr, ntiles, warm, levels, d_table + 18 * ms, 0);
                return check_launch("iir_scan_kernel<fused, unit output stage>");
            }
        }
        hipLaunchKernelGGL((iir_scan_kernel<P, SCAN_FUSED, ALIGNED, WIDE>), grid, dim3(kSt), 0, s, x, y, n, co, d_table,
                           state_in, state, nullptr, nullptr, ntiles, warm, levels, d_table + 18 * ms, 0);
        return check_launch("iir_scan_kernel<fused>");
    }
    void *z_v = nullptr, *c_v = nullptr;
    const size_t carry_bytes = channels * ntiles * W * sizeof(double);
    if (int rc = workspace(WS_SCRATCH_A, carry_bytes, &z_v, s)) return rc;
    if (int rc = workspace(WS_SCRATCH_B, carry_bytes, &c_v, s)) return rc;
    double *tz = static_cast<double *>(z_v), *tc = static_cast<double *>(c_v);
    hipLaunchKernelGGL((iir_scan_kernel<P, SCAN_REDUCE, ALIGNED, WIDE>), grid, dim3(kSt), 0, s, x, y, n, co, d_table,
                       state_in, nullptr, tz, nullptr, ntiles, 0, levels, d_table + 18 * ms, 0);
    if (int rc = check_launch("iir_scan_kernel<reduce>")) return rc;
    if (carry_terms > 0) {  // every tile sums its own few carry terms
        hipLaunchKernelGGL((iir_scan_kernel<P, SCAN_APPLY, ALIGNED, WIDE>), grid, dim3(kSt), 0, s, x, y, n, co, d_table,
                           state_in, state, tz, nullptr, ntiles, 0, levels, d_table + 18 * ms, carry_terms);
        return check_launch("iir_scan_kernel<apply, local carries>");
    }
    const unsigned carry_wgs = (ntiles + tiles_per_wg - 1) / tiles_per_wg;
    hipLaunchKernelGGL((iir_carry_kernel<P, WIDE>), dim3(static_cast<unsigned>(channels), carry_wgs), dim3(kCt), 0, s, tz, tc,
                       d_table + 9 * ms, state_in, ntiles, G, tiles_per_wg, carry_warm);
    if (int rc = check_launch("iir_carry_kernel")) return rc;
    hipLaunchKernelGGL((iir_scan_kernel<P, SCAN_APPLY, ALIGNED, WIDE>), grid, dim3(kSt), 0, s, x, y, n, co, d_table,
                       state_in, state, nullptr, tc, ntiles, 0, levels, d_table + 18 * ms, 0);
    return check_launch("iir_scan_kernel<apply>");
}

// Powers of the companion matrix by RUNNING the recursion: column c of A^k is the feedback state after k
// zero-input steps from the unit state e_c.  Done in double-double this is as well conditioned as the
// filter itself, whereas repeated squaring of this highly non-normal matrix loses three digits per
// squaring (measured at fc = 0.001: A^1024 only 3e-13 accurate even in double-double).
void build_plan(const IirCoef &co, ScanPlan &plan) {
    const int P = co.poles;
    plan.co = co;
    plan.pow32.assign(kLevels + 1, Mat(static_cast<size_t>(P) * P, DD{0.0, 0.0}));
    std::vector<DD> st(static_cast<size_t>(P) * P);  // st[c*P + i] = state component i of the run started from e_c
    for (int c = 0; c < P; ++c)
        for (int i = 0; i < P; ++i) st[c * P + i] = DD{i == c ? 1.0 : 0.0, 0.0};
    plan.hrow.assign(static_cast<size_t>(kRow) * P, 0.0);
    double amp = 0.0;
    plan.levels = kLevels + 1;
    plan.warm = -1;
    int next_level = 0;
    for (int k = 1; k <= kTile; ++k) {
        for (int c = 0; c < P; ++c) {
            DD v{0.0, 0.0};
            for (int i = 0; i < P; ++i) v = dd_add(v, dd_mul(DD{co.b[i + 1], 0.0}, st[c * P + i]));
            for (int i = P - 1; i >= 1; --i) st[c * P + i] = st[c * P + i - 1];
            st[c * P] = v;
            if (k <= kRow) plan.hrow[static_cast<size_t>(k - 1) * P + c] = v.hi + v.lo;
        }
        if (k % kRow) continue;
        double nrm = 0.0;  // ||A^k||_inf: rows are state components, columns the start vectors
        for (int i = 0; i < P; ++i) {
            double row = 0.0;
            for (int c = 0; c < P; ++c) row += std::fabs(st[c * P + i].hi);
            if (!(row <= nrm)) nrm = row;
        }
        if (plan.warm < 0 && k <= kTile / 4 && nrm <= 0x1p-44) plan.warm = k;
        if (k == (kRow << next_level)) {
            Mat &m = plan.pow32[next_level];
            for (int i = 0; i < P; ++i)
                for (int c = 0; c < P; ++c) m[i * P + c] = st[c * P + i];
            amp = std::max(amp, nrm);
            if (!(nrm == nrm)) amp = 1e300;
            if (nrm <= 0x1p-60 && plan.levels > next_level) plan.levels = next_level;
            ++next_level;
            if (nrm <= 0x1p-200) break;  // everything later is zero for all purposes (left as zeros)
        }
    }
    {  // how fast the tile matrix MT = A^kTile decays: decides how the tile carries are scanned (scan_launch)
        const Mat &MT = plan.pow32[kLevels];
        Mat pw = MT;
        plan.decay = 0;
        for (int j = 1; j <= 64; ++j) {
            if (mat_norm_inf(pw, P) <= 0x1p-60) {
                plan.decay = j;
                break;
            }
            pw = mat_mul(pw, MT, P);
        }
    }
    // Conditioning.  A carry S += M S' in double rounds at ||M|| * 2^-53 relative to the state and the
    // recursion amplifies an injected error by the feedback noise gain (calibrated on the 4-pole
    // Chebyshev at fc = 0.01: predicted 5e-5, measured 5.4e-5 with double carries).  Direct-form
    // sections with poles clustered at the unit circle therefore carry their states in double-double
    // (matrices included); what even that cannot hold inside the 1e-4 budget -- unstable filters -- is
    // refused (DSP_GPU_IIR_AUTO then runs the channel sequentially in BATCH mode).
    const double gain = feedback_noise_gain(co);
    const double predicted_double = 64.0 * amp * gain * 0x1p-52;
    plan.wide = !(predicted_double <= 2e-7);
    plan.predicted = plan.wide ? 64.0 * amp * gain * 0x1p-100 : predicted_double;
    plan.usable = plan.predicted <= 1e-4 && amp < 1e200;
}

std::mutex g_plan_mutex;
std::vector<std::shared_ptr<const ScanPlan>> g_plans;  // small cache: a filter is usually applied many times
unsigned long long g_plan_serial = 0;

std::shared_ptr<const ScanPlan> cached_plan(const IirCoef &co) {
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    for (const auto &p : g_plans)
        if (p->co.poles == co.poles && std::memcmp(p->co.a, co.a, sizeof co.a) == 0 && std::memcmp(p->co.b, co.b, sizeof co.b) == 0)
            return p;  // shared: the cache may be rotated by another thread while this call runs
    auto plan = std::make_shared<ScanPlan>();
    build_plan(co, *plan);
    plan->serial = ++g_plan_serial;
    if (g_plans.size() >= 64) g_plans.erase(g_plans.begin());
    g_plans.push_back(plan);
    return plan;
}

template <int P, bool ALIGNED>
int scan_dispatch(const float *x, float *y, size_t channels, size_t n, const IirCoef &co, double *state,
                  hipStream_t s) {
    constexpr size_t kMaxGridY = 65535;  // channels ride grid.y; longer batches are cut into several launches
    for (; channels > kMaxGridY; channels -= kMaxGridY, x += kMaxGridY * n, y += kMaxGridY * n) {
        if (int rc = scan_dispatch<P, ALIGNED>(x, y, kMaxGridY, n, co, state, s)) return rc;
        if (state) state += kMaxGridY * 2 * P;
    }
    const std::shared_ptr<const ScanPlan> plan = cached_plan(co);
    if (!plan->usable)
        return fail(DSP_GPU_EUNSUPPORTED,
                    "IIR scan: filter too ill-conditioned for block-parallel carries "
                    "(predicted relative error %.2e > 1e-4); use DSP_GPU_IIR_BATCH", plan->predicted);
    // tile 0 reads the caller's history while the last tile writes the new one: read from a snapshot
    const double *state_in = state;
    if (state) {
        void *snap = nullptr;
        const size_t sb = channels * 2 * P * sizeof(double);
        if (int rc = workspace(WS_STATE_SNAP, sb, &snap, s)) return rc;
        DSPGPU_HIP(hipMemcpyAsync(snap, state, sb, hipMemcpyDeviceToDevice, s));
        state_in = static_cast<const double *>(snap);
    }
    return plan->wide ? scan_launch<P, ALIGNED, true>(x, y, channels, n, co, state_in, state, *plan, s)
                      : scan_launch<P, ALIGNED, false>(x, y, channels, n, co, state_in, state, *plan, s);
}

}  // namespace

// Samples after which the response to ANY initial state has decayed below 2^-44 of it (||A^k||_inf <= 2^-44 for the
// companion matrix A of the feedback part): what a time-sharded run needs to know -- a segment started `samples` early,
// or re-run over its first `samples` outputs from the true state, is exact to double rounding from there on.
int iir_settle_samples(const IirCoef &co, size_t *samples) {
    const std::shared_ptr<const ScanPlan> plan = cached_plan(co);
    if (plan->warm >= 0) {
        *samples = static_cast<size_t>(plan->warm);
        return DSP_GPU_OK;
    }
    if (plan->decay > 0) {  // (A^4096)^j <= 2^-60: a coarser but safe bound
        *samples = static_cast<size_t>(plan->decay) * kTile;
        return DSP_GPU_OK;
    }
    return fail(DSP_GPU_EUNSUPPORTED, "dsp_gpu_iir_settle: the filter's state does not decay within %d samples", 64 * kTile);
}

bool iir_tables_in_use(int dev) { return tables_in_use(dev); }
void release_iir_tables(int dev) { release_tables(dev); }

int launch_iir_scan(const float *x, float *y, size_t channels, size_t n, const IirCoef &co, double *state,
                    hipStream_t s) {
    const bool al = iirb::is_aligned<float>(x, y, n);
#define DSP_IIR_SCAN_CASE(P) \
    return al ? scan_dispatch<P, true>(x, y, channels, n, co, state, s) : scan_dispatch<P, false>(x, y, channels, n, co, state, s)
    switch (co.poles) {
    case 1: DSP_IIR_SCAN_CASE(1);
    case 2: DSP_IIR_SCAN_CASE(2);
    case 3: DSP_IIR_SCAN_CASE(3);
    case 4: DSP_IIR_SCAN_CASE(4);
    case 5: DSP_IIR_SCAN_CASE(5);
    case 6: DSP_IIR_SCAN_CASE(6);
    case 7: DSP_IIR_SCAN_CASE(7);
    case 8: DSP_IIR_SCAN_CASE(8);
    default:  // P x P carry matrices in registers: beyond 8 poles the channel runs sequentially (BATCH)
        return fail(DSP_GPU_EUNSUPPORTED, "IIR scan: poles %d > 8; use DSP_GPU_IIR_BATCH", co.poles);
    }
#undef DSP_IIR_SCAN_CASE
}

}  // namespace dspgpu
