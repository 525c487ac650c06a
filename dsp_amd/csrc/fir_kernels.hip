// fir_kernels.hip -- FIR convolution and centred moving average for gfx950 (MI355X).
//
// Stands in for dsp::filter::SincFilter<>::apply (reference filter.h:123-140), dsp::convolve
// (convolution.h:8-29, incl. the Frequency branch the reference only declares) and
// dsp::filter::MovingAverageFilter::apply (filter.h:81-101).
//
// Kernels
//   fir_direct_kernel        y[j] = sum_i h[i] x[j-i] as a Toeplitz product on the fp32 matrix
//                            cores (v_mfma_f32_16x16x4_f32: exact fp32 fma chains at the fp32
//                            vector rate).  A 16x16 output tile holds 256 consecutive outputs
//                            y[j0 + 16c + r]; with m = r - i = 16p + t the sum becomes
//                              D[r][c] = sum_p sum_t h[r - 16p - t] * x[j0 + 16(c+p) + t]
//                            i.e. one 16x16 Toeplitz block of taps times the data block shifted
//                            by p columns.  Both operands come from LDS as one ds_read_b128 per
//                            four MFMAs; a wave keeps 4 tiles (independent accumulators) going.
//   fir_overlap_save_kernel  4096-point overlap-save: two real blocks ride one complex FFT
//                            (a + i b; valid because the taps are real), multiplied by the cached
//                            tap spectrum, inverse FFT, real/imag parts stored.  One workgroup per
//                            block pair, the transform never leaves registers + LDS.
//                            Streaming (carried input tail) and partitioned (filters longer than one
//                            block, OsExtra) forms share the kernel.
//   fir_direct_valu_kernel   the direct form on the vector ALU (packed fp32, sliding register window):
//                            the matrix-core kernel's measured A/B partner, not picked by AUTO.
//   fir_history_kernel       new input tail for the streaming form.
//   moving_average_kernel    windows summed from 16-sample partial sums held in LDS (no running
//                            sum along the signal, so no accumulated drift).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "fft_lds_core.h"
#include "fft_wave_core.h"

namespace dspgpu {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// ---------------------------------------------------------------------------------------------
// direct FIR on fp32 MFMA
// ---------------------------------------------------------------------------------------------
constexpr size_t kMaxGridY = 65535;  // channels ride grid.y; longer batches are cut into several launches
constexpr int kFirThreads = 256;
constexpr int kFirWaves = kFirThreads / 64;
constexpr int kFirTile = 256;      // outputs per 16x16 tile
constexpr int kFirAcc = 4;         // tiles (accumulators) per wave per pass
constexpr int kFirPasses = 2;
constexpr int kFirSeg = 16;        // Toeplitz blocks (of 16 taps) per accumulation segment
constexpr int kFirOutPerWg = kFirWaves * kFirAcc * kFirTile * kFirPasses;  // 8192 outputs

// LDS: `slots` input segments float xs[hist + S] (slots * S = kFirOutPerWg) followed by 4 shifted, reversed, zero
// padded copies of the taps, each taps_len = 16*Q + 16 floats:  copy s holds As[w] = h[16Q - s - w].
// Long channels: one slot, S = 8192 outputs of channel blockIdx.y starting at blockIdx.x * 8192.  Channels of at most
// 4096 samples: S = n rounded up to 256 (one tile; until late round 3: 1024, what one wave covers per pass) and the workgroup takes 8192 / S channels at
// once, each with its own history in front -- otherwise three of the four waves would idle on a 1024-sample channel.
// PERIODIC (one-shot calls on many channels shorter than a few workgroups' worth): the channels lie one behind the other in
// memory, and an output with a full window behind it never reaches across its channel's start, so the batch is filtered as ONE
// signal of channels * period samples (n is that product, one "channel") and the outputs whose position inside their channel
// is below the filter's order -- the ones that would have seen the previous channel's tail -- are zeroed as filter.h:126,133
// asks anyway.  No slot rounded up to 1,024 outputs, no last workgroup of a 10,000-sample channel that is three quarters empty.
template <bool ALIGNED, bool MULTI, bool PERIODIC = false>  // MULTI: more than one slot (kept apart: the slot arithmetic cost the long-channel path 3 %)
__global__ __launch_bounds__(kFirThreads) void fir_direct_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                                  size_t n, size_t channels, const float *__restrict__ taps,
                                                                  int ntaps, int Q, int slots, int S,
                                                                  const float *__restrict__ prev, int zero_head, unsigned period) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int hist = 16 * (Q - 1);
    const int slot_len = hist + S;
    const int xs_len = slots * slot_len;
    const int taps_len = 16 * Q + 16;
    float *xs = smem;
    float *as = smem + xs_len;

    const int tid = threadIdx.x;
    const size_t ch0 = MULTI ? static_cast<size_t>(blockIdx.y) * slots : blockIdx.y;  // first channel of this workgroup
    const long long chunk0 = static_cast<long long>(blockIdx.x) * kFirOutPerWg;  // (blockIdx.x == 0 when slots > 1)
    const long long nn = static_cast<long long>(n);
    // streaming form: prev[ch][ntaps-1] holds the samples before x[0] (oldest first) and every output is kept
    const long long hlen = ntaps - 1;

    // taps: 4 shifted reversed copies
    for (int i = tid; i < 4 * taps_len; i += kFirThreads) {
        const int s = i / taps_len, w = i - s * taps_len;
        const int idx = 16 * Q - s - w;
        as[i] = (idx >= 0 && idx < ntaps) ? taps[idx] : 0.0f;
    }
    // input segment [chunk0 - hist, chunk0 + S) of every slot's channel, zeros outside the channel
    const long long seg0 = chunk0 - hist;
    int u0 = tid * 4;
    if (!MULTI && ALIGNED && (!prev || seg0 >= 0)) {
        // long channel: four loads in flight per thread, no branch between them; a float4 outside the channel (n and
        // seg0 are multiples of 4: wholly outside) is aimed at the channel's first sample and zeroed afterwards
        constexpr int kStep = kFirThreads * 4;
        const float *xc = x + ch0 * n;
        if (seg0 >= 0 && seg0 + xs_len <= nn) {  // interior chunk: nothing to clamp
            for (; u0 + 3 * kStep < xs_len; u0 += 4 * kStep) {
                float4 v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4 *>(xc + seg0 + u0 + i * kStep);
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<float4 *>(xs + u0 + i * kStep) = v[i];
            }
        }
        for (; u0 + 3 * kStep < xs_len; u0 += 4 * kStep) {
            float4 v[4];
            bool ok[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long long g = seg0 + u0 + i * kStep;
                ok[i] = g >= 0 && g < nn;
                v[i] = *reinterpret_cast<const float4 *>(xc + (ok[i] ? g : 0));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<float4 *>(xs + u0 + i * kStep) = ok[i] ? v[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
    if (MULTI && ALIGNED && !prev) {
        // short channels, one-shot: every float4 is wholly inside its channel or wholly outside (n, hist multiples of
        // 4); outside ones are aimed at x[0] and zeroed afterwards, so four loads go out together without a branch
        constexpr int kStep = kFirThreads * 4;
        for (; u0 < xs_len; u0 += 4 * kStep) {
            float4 v[4];
            bool ok[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int u = u0 + i * kStep;
                const int k = u / slot_len, off = u - k * slot_len;
                const long long g = seg0 + off;
                ok[i] = u < xs_len && ch0 + k < channels && g >= 0 && g < nn;
                v[i] = *reinterpret_cast<const float4 *>(x + (ok[i] ? (ch0 + k) * n + g : 0));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (u0 + i * kStep < xs_len)
                    *reinterpret_cast<float4 *>(xs + u0 + i * kStep) = ok[i] ? v[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
    for (int u = u0; u < xs_len; u += kFirThreads * 4) {
        const int k = MULTI ? u / slot_len : 0, off = u - k * slot_len;  // slot_len is a multiple of 4: a float4 stays in its slot
        const size_t ch = ch0 + k;
        const long long g = seg0 + off;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (!MULTI || ch < channels) {
            const float *xc = x + ch * n;
            const float *hc = prev ? prev + ch * hlen : nullptr;
            if (ALIGNED && g >= 0 && g + 3 < nn) {
                v = *reinterpret_cast<const float4 *>(xc + g);
            } else {
                float e[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const long long gi = g + q;
                    e[q] = (gi >= 0 && gi < nn) ? xc[gi] : ((hc && gi < 0 && gi >= -hlen) ? hc[hlen + gi] : 0.0f);
                }
                v = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
        *reinterpret_cast<float4 *>(xs + u) = v;
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kq = lane >> 4;  // r: A row / B column, kq: k slot
    const int s = (4 - (r & 3)) & 3;
    const float *a_ptr = as + s * taps_len + (16 * Q - s - r + 4 * kq);  // minus 16 q per step
    const int order = zero_head;  // outputs before the stream's first full window stay zero (filter.h:133)

#pragma unroll 1
    for (int pass = 0; pass < kFirPasses; ++pass) {
        const int tile0 = (pass * kFirWaves + wave) * kFirAcc;  // first of this wave's tiles in the workgroup's 8192 outputs
        // MULTI: S is a multiple of one tile's 256 outputs, so each of the wave's four tiles has a slot and a place in it of its own
        // (a streamed 256-sample chunk is one tile: rounded up to the 1,024 outputs of a wave-pass it cost four times the work)
        const int k = 0;
        const int local = tile0 * kFirTile;  // (!MULTI) first output inside the workgroup's chunk
        int kt[kFirAcc], lt[kFirAcc];
        bool live[kFirAcc];
        const float *bp[kFirAcc];
        if (MULTI) {
            bool any = false;
#pragma unroll
            for (int t = 0; t < kFirAcc; ++t) {
                const int o = (tile0 + t) * kFirTile;
                kt[t] = o / S;
                lt[t] = o - kt[t] * S;
                live[t] = kt[t] < slots && ch0 + kt[t] < channels && lt[t] < nn;  // wave-uniform
                any = any || live[t];
                bp[t] = xs + (live[t] ? kt[t] * slot_len + lt[t] : 0) + hist + 16 * r + 4 * kq;  // (a dead tile reads slot 0 and stores nothing)
            }
            if (!any) continue;
        } else if (chunk0 + local >= nn) {
            break;  // wave-uniform
        }
        float *yc = y + ch0 * n;
        const float *b_ptr = xs + k * slot_len + hist + local + 16 * r + 4 * kq;
        // The taps are summed in segments of kFirSeg blocks (256 taps) with a fresh accumulator each,
        // so the fp32 rounding of a long filter grows like a blocked sum, not like one long chain.
        f32x4 acc[kFirAcc];
#pragma unroll
        for (int t = 0; t < kFirAcc; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
        for (int q0 = 0; q0 < Q; q0 += kFirSeg) {
            const int q1 = q0 + kFirSeg < Q ? q0 + kFirSeg : Q;
            f32x4 seg[kFirAcc];
#pragma unroll
            for (int t = 0; t < kFirAcc; ++t) seg[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            for (int q = q0; q < q1; ++q) {
                const f32x4 a4 = *reinterpret_cast<const f32x4 *>(a_ptr - 16 * q);
                f32x4 b4[kFirAcc];
#pragma unroll
                for (int t = 0; t < kFirAcc; ++t)
                    b4[t] = *reinterpret_cast<const f32x4 *>((MULTI ? bp[t] : b_ptr + t * kFirTile) - 16 * q);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int t = 0; t < kFirAcc; ++t)
                        seg[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[kk], b4[t][kk], seg[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < kFirAcc; ++t) acc[t] += seg[t];
        }
        // D: column = lane & 15, rows 4*(lane >> 4) + i  ->  y[j0 + 16*col + 4*(lane>>4) + i]
#pragma unroll
        for (int t = 0; t < kFirAcc; ++t) {
            if (MULTI && !live[t]) continue;
            if (MULTI) yc = y + (ch0 + kt[t]) * n;
            const long long j = MULTI ? static_cast<long long>(lt[t]) + 16 * r + 4 * kq
                                      : chunk0 + local + static_cast<long long>(t) * kFirTile + 16 * r + 4 * kq;
            if (j >= nn) continue;
            float o[4];
            if constexpr (PERIODIC) {  // (the host has checked that channels * period fits 32 bits)
                unsigned jm = static_cast<unsigned>(j) % period;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o[i] = jm >= static_cast<unsigned>(order) ? acc[t][i] : 0.0f;
                    if (++jm == period) jm = 0;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (j + i >= order) ? acc[t][i] : 0.0f;  // filter.h:133: first M outputs stay 0
            }
            if (ALIGNED && j + 3 < nn) {
                *reinterpret_cast<float4 *>(yc + j) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (j + i < nn) yc[j + i] = o[i];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// direct FIR on the vector ALU (the A/B partner of the MFMA kernel: BASELINE's north_star wants the matrix
// cores used only if they beat the VALU formulation in rocprof; DSP_GPU_CONV_DIRECT_VALU selects this one)
// ---------------------------------------------------------------------------------------------
// A thread owns 16 consecutive outputs as 8 packed pairs and slides a 20-sample register window along the
// taps, four taps per step: 32 v_pk_fma_f32 per 3 LDS reads.  v_pk_fma_f32 wants even-aligned register pairs,
// but (x[q], x[q+1]) starts on an odd sample for every other tap, so the workgroup keeps TWO images of its
// samples in LDS, xs0[u] = x[u] and xs1[u] = x[u+1], and odd taps read their pairs from the second one.
// The window is a ring of five 4-sample groups; unrolling five steps makes every register index static.
// Plain running fp32 sums (no segmentation): meant for filters up to ~1k taps.
constexpr int kVfOut = 16;                       // outputs per thread
constexpr int kVfTile = kThreads * kVfOut;       // 4096 outputs per workgroup
typedef float v2f __attribute__((ext_vector_type(2)));

DSP_DEV int vf_pad(int u) { return u + ((u >> 4) << 2); }  // 4 pad floats per 16: 16-byte reads at lane stride 20 dwords

__global__ __launch_bounds__(kThreads) void fir_direct_valu_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                                   size_t n, const float *__restrict__ taps, int ntaps,
                                                                   int taps_p, int H) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int S = H + kVfTile;          // staged samples: logical u <-> sample tile0 - H + u
    float *xs0 = smem;
    float *xs1 = xs0 + vf_pad(S);
    float *hp = xs1 + vf_pad(S);        // taps_p floats, zero padded
    const int tid = threadIdx.x;
    const size_t ch = blockIdx.y;
    const long long nn = static_cast<long long>(n);
    const long long tile0 = static_cast<long long>(blockIdx.x) * kVfTile;
    const float *xc = x + ch * n;
    float *yc = y + ch * n;

    for (int u = tid * 4; u < S; u += kThreads * 4) {
        const long long g = tile0 - H + u;
        float4 v;
        if (g >= 0 && g + 3 < nn && (n & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
            v = load_stream_f4(xc + g);
        } else {
            v.x = (g >= 0 && g < nn) ? xc[g] : 0.0f;
            v.y = (g + 1 >= 0 && g + 1 < nn) ? xc[g + 1] : 0.0f;
            v.z = (g + 2 >= 0 && g + 2 < nn) ? xc[g + 2] : 0.0f;
            v.w = (g + 3 >= 0 && g + 3 < nn) ? xc[g + 3] : 0.0f;
        }
        *reinterpret_cast<float4 *>(xs0 + vf_pad(u)) = v;
    }
    for (int i = tid; i < taps_p; i += kThreads) hp[i] = i < ntaps ? taps[i] : 0.0f;
    __syncthreads();
    for (int u = tid * 4; u < S; u += kThreads * 4) {  // the image shifted by one sample
        float4 v = *reinterpret_cast<const float4 *>(xs0 + vf_pad(u));
        const float nxt = u + 4 < S ? xs0[vf_pad(u + 4)] : 0.0f;
        *reinterpret_cast<float4 *>(xs1 + vf_pad(u)) = make_float4(v.y, v.z, v.w, nxt);
    }
    __syncthreads();

    // ring of five groups (4 samples = 2 pairs each) per image; at step s the group at window position q
    // (samples base - 4 + 4q .. +3) lives in slot (q - s + 5) % 5
    v2f w0[10], w1[10], acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = v2f{0.0f, 0.0f};
    int base = H + kVfOut * tid;  // logical index of this thread's first output; decreases by 4 per step
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const float4 a = *reinterpret_cast<const float4 *>(xs0 + vf_pad(base - 4 + 4 * q));
        const float4 b = *reinterpret_cast<const float4 *>(xs1 + vf_pad(base - 4 + 4 * q));
        w0[2 * q] = v2f{a.x, a.y};
        w0[2 * q + 1] = v2f{a.z, a.w};
        w1[2 * q] = v2f{b.x, b.y};
        w1[2 * q + 1] = v2f{b.z, b.w};
    }
    for (int c0 = 0; c0 < taps_p; c0 += 20) {
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const float4 h = *reinterpret_cast<const float4 *>(hp + c0 + 4 * s);  // uniform address: broadcast
            // next step's lowest group, into the slot of this step's highest one (position 4)
            const int nb = base - 8;
            const float4 na = *reinterpret_cast<const float4 *>(xs0 + vf_pad(nb));
            const float4 nbv = *reinterpret_cast<const float4 *>(xs1 + vf_pad(nb));
            const float hk[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const v2f hh = v2f{hk[k], hk[k]};
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    // even tap: pair (x[base+2m-k], +1) from image 0; odd tap: the same pair starts on an odd
                    // sample = aligned pair (base+2m-k-1) of image 1
                    const int d = (k & 1) ? 2 * m - k - 1 : 2 * m - k;  // even, in [-4, 14]
                    const int P = (d + 4) / 2, q = P / 2, half = P % 2;
                    const int slot = (q - s + 5) % 5;
                    const v2f pr = (k & 1) ? w1[2 * slot + half] : w0[2 * slot + half];
                    acc[m] = __builtin_elementwise_fma(hh, pr, acc[m]);
                }
            }
            const int hs = (4 - s + 5) % 5;  // slot of the highest group of this step
            w0[2 * hs] = v2f{na.x, na.y};
            w0[2 * hs + 1] = v2f{na.z, na.w};
            w1[2 * hs] = v2f{nbv.x, nbv.y};
            w1[2 * hs + 1] = v2f{nbv.z, nbv.w};
            base -= 4;
        }
    }
    const long long j0 = tile0 + kVfOut * tid;
    const long long order = ntaps - 1;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const long long j = j0 + 2 * m;
        if (j < nn) yc[j] = j >= order ? acc[m].x : 0.0f;  // filter.h:133: first M outputs stay 0
        if (j + 1 < nn) yc[j + 1] = j + 1 >= order ? acc[m].y : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// overlap-save FIR
// ---------------------------------------------------------------------------------------------

// hspec: FORWARD spectrum of the zero padded taps (unscaled).  Block b of a channel produces
// outputs [order + b*L, order + (b+1)*L) from inputs [b*L, b*L + 4096), L = 4096 - order.
// `ex` serves the partitioned form for filters longer than one block allows (launch_fir_partitioned): the rows
// of x and y are `stride` apart while only n samples of each take part, outputs land `yoff` samples further on and
// are ADDED to what is there, and the history in front of x is all zeros without a buffer to prove it.
struct OsExtra {
    size_t stride;   // distance between channels in x and y (n in the plain forms)
    size_t yoff;     // output j is stored at y[ch*stride + yoff + j]
    int accumulate;  // y += result instead of y = result
    int zero_hist;   // streaming form over an all-zero history (no `hist` buffer)
    int edge_pairs;  // the grid walks only the pairs holding a channel's first or last block: workgroup 2c -> the pair of
                     // channel c's block 0, 2c + 1 -> the pair of its last block (what fir_os_wave_kernel leaves out)
};

// Workgroup: 256 threads (1,024 for 16384-point blocks); a transform takes T = N/16 of them, so blocks of 256 ... 2,048
// points (channels short enough to fit one 4096-point block, capi.hip) go sixteen ... two pairs to a workgroup.
template <int LOG2N>
__global__ __launch_bounds__(((1 << LOG2N) / kPoints) < 256 ? 256 : ((1 << LOG2N) / kPoints)) void fir_overlap_save_kernel(
    const float *__restrict__ x, float *__restrict__ y, size_t n, const c32 *__restrict__ hspec, int ntaps,
    unsigned blocks_per_channel, size_t total_blocks, size_t tiles, size_t wg_tiles, size_t chunk,
    const float *__restrict__ hist, int zero_head, OsExtra ex, int hop) {
    constexpr int kOsN = 1 << LOG2N;    // block (transform) length
    constexpr int T = kOsN / kPoints;   // threads per transform
    constexpr int G = T < 256 ? 256 / T : 1;  // transforms (block pairs) per workgroup
    extern __shared__ __attribute__((aligned(16))) c32 lds[];
    const int lt = static_cast<int>(threadIdx.x) % T, grp = static_cast<int>(threadIdx.x) / T;
    const int base = grp * kOsN;
    size_t wtile = xcd_tile_of(blockIdx.x, chunk);  // each XCD streams a contiguous range of block pairs
    if (ex.edge_pairs)  // G == 1 here (4096-point blocks)
        wtile = (static_cast<size_t>(blockIdx.x >> 1) * blocks_per_channel + ((blockIdx.x & 1u) ? blocks_per_channel - 1 : 0)) >> 1;
    if (wtile >= wg_tiles) return;
    const size_t tile = wtile * G + grp;
    const bool has_a = tile < tiles;  // a workgroup's last transforms may have no pair left: they idle through the barriers
    // the two real blocks riding one complex transform are consecutive entries of the flattened (channel, block)
    // list: with an odd block count per channel (short channels: often just one) a pair straddles two channels
    const size_t qa = has_a ? 2 * tile : 0, qb = qa + 1;
    const bool has_b = has_a && qb < total_blocks;
    const size_t ch_a = qa / blocks_per_channel, ch_b = has_b ? qb / blocks_per_channel : ch_a;
    const unsigned blk_a = static_cast<unsigned>(qa % blocks_per_channel), blk_b = static_cast<unsigned>(qb % blocks_per_channel);
    const long long order = ntaps - 1;
    // Block b reads inputs [b*L, b*L + N) and keeps its outputs k >= N - L (block 0: k >= order).  L <= N - order; the
    // launcher rounds it down to a multiple of 64 samples where that costs next to nothing, so that on channels whose
    // length is a multiple of 64 every block starts on a 256-byte boundary and a wave's stores are whole 128-byte
    // lines (with L = N - order = 3586 the same kernel's stores ran at 2.7 instead of 4.3 TB/s).
    const long long L = hop;
    const long long keep0 = kOsN - L;
    // streaming form: the blocks run over the virtual signal history ++ x (hoff = ntaps-1 samples of
    // history in front), so every output of x is kept; one-shot form: hoff = 0 and y[0, order) = 0
    const long long hoff = (hist || ex.zero_hist) ? order : 0;
    const long long nn = static_cast<long long>(n) + hoff;
    const float *xa = x + ch_a * ex.stride - hoff;  // xa[v] valid for v >= hoff
    const float *xb = x + ch_b * ex.stride - hoff;
    float *ya = y + ch_a * ex.stride + ex.yoff - hoff;
    float *yb = y + ch_b * ex.stride + ex.yoff - hoff;
    const float *ha = hist ? hist + ch_a * order : nullptr;
    const float *hb = hist ? hist + ch_b * order : nullptr;
    // a missing block starts at nn: nothing is loaded for it and nothing stored
    const long long in_a = has_a ? static_cast<long long>(blk_a) * L : nn;
    const long long in_b = has_b ? static_cast<long long>(blk_b) * L : nn;

    // interior pairs (all but the first of a streamed chunk and the last one or two of a channel) take
    // the unchecked path: the bounds handling costs more VALU work than a radix-16 pass
    const bool interior = in_a >= hoff && in_a + kOsN <= nn && in_b >= hoff && in_b + kOsN <= nn;
    c32 r[kPoints];
    if (interior) {
        const float *pa = xa + in_a + lt, *pb = xb + in_b + lt;
#pragma unroll
        for (int m = 0; m < kPoints; ++m)
            r[m] = mk(__builtin_nontemporal_load(pa + T * m), __builtin_nontemporal_load(pb + T * m));
    } else if (!(hist && (in_a < hoff || in_b < hoff))) {
        // ragged pair (a channel's last blocks, short channels): every load goes out, aimed at the channel's first
        // sample when its own index is outside the signal, and is zeroed afterwards -- no branch between the loads
#pragma unroll
        for (int m = 0; m < kPoints; ++m) {
            const long long ia = in_a + lt + T * m, ib = in_b + lt + T * m;
            const bool oka = ia >= hoff && ia < nn, okb = ib >= hoff && ib < nn;
            const float va = __builtin_nontemporal_load(xa + (oka ? ia : hoff));
            const float vb = __builtin_nontemporal_load(xb + (okb ? ib : hoff));
            r[m] = mk(oka ? va : 0.0f, okb ? vb : 0.0f);
        }
    } else {  // the first pair of a streamed chunk reaches into the carried history
#pragma unroll
        for (int m = 0; m < kPoints; ++m) {
            const long long ia = in_a + lt + T * m, ib = in_b + lt + T * m;
            const float va = ia >= nn ? 0.0f : (ia >= hoff ? __builtin_nontemporal_load(xa + ia) : (ha ? ha[ia] : 0.0f));
            const float vb = ib >= nn ? 0.0f : (ib >= hoff ? __builtin_nontemporal_load(xb + ib) : (hb ? hb[ib] : 0.0f));
            r[m] = mk(va, vb);
        }
    }
    fft_passes<LOG2N, +1>(r, lds, lt, base);
    constexpr float scale = 1.0f / kOsN;
#pragma unroll
    for (int m = 0; m < kPoints; ++m) {
        const c32 h = hspec[lt + T * m];
        const c32 v = cmul(r[m], h);
        r[m] = mk(v.x * scale, v.y * scale);
    }
    __syncthreads();  // the forward passes' last LDS image has been read by everyone
    fft_passes<LOG2N, -1>(r, lds, lt, base);

    if (hoff == 0) {  // filter.h:126,133: the first `order` outputs of a channel stay zero
        if (has_a && blk_a == 0)
            for (long long j = lt; j < order && j < nn; j += T) ya[j] = 0.0f;
        if (has_b && blk_b == 0)
            for (long long j = lt; j < order && j < nn; j += T) yb[j] = 0.0f;
    }
    // first kept output of each block: what lies before it is the wrapped (circular) part or the previous block's
    const long long first_a = blk_a == 0 ? order : keep0, first_b = blk_b == 0 ? order : keep0;
    if (ex.accumulate) {  // partitioned form: this segment's contribution on top of the earlier ones
#pragma unroll
        for (int m = 0; m < kPoints; ++m) {
            const long long k = lt + T * m;
            const long long ja = in_a + k, jb = in_b + k;
            if (k >= first_a && ja < nn) ya[ja] += r[m].x;
            if (k >= first_b && jb < nn) yb[jb] += r[m].y;
        }
        return;
    }
    if (interior && blk_a != 0 && blk_b != 0) {  // every kept output is in range and past the zero head
        float *pa = ya + in_a + lt, *pb = yb + in_b + lt;
        const int k0 = static_cast<int>(keep0);
#pragma unroll
        for (int m = 0; m < kPoints; ++m) {
            if (lt + T * m < k0) continue;
            __builtin_nontemporal_store(r[m].x, pa + T * m);
            __builtin_nontemporal_store(r[m].y, pb + T * m);
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < kPoints; ++m) {
        const long long k = lt + T * m;
        const long long ja = in_a + k, jb = in_b + k;
        // streaming: outputs before the stream's first full window stay zero like the one-shot head
        if (k >= first_a && ja < nn) __builtin_nontemporal_store(ja - hoff < zero_head ? 0.0f : r[m].x, ya + ja);
        if (k >= first_b && jb < nn) __builtin_nontemporal_store(jb - hoff < zero_head ? 0.0f : r[m].y, yb + jb);
    }
}

// The same overlap-save step with ONE WAVE per block pair (fft_wave_core.h): 64 lanes x 64 points, two radix-64
// passes per transform, one wave-private transpose each way, no workgroup barrier after the set-up.  A workgroup is
// four independent persistent waves, ONE PER SIMD (measured: two per SIMD transform slower, 1.68 against 1.50 ms of
// pure arithmetic on C3, and leave no registers for what follows), sharing only the tap spectrum: taps are real, so
// H[4096 - k] = conj(H[k]) and half of it (2049 bins, pre-scaled by 1/4096) sits in LDS.
//
// Software pipeline.  With one wave per SIMD nothing else hides HBM latency, so the NEXT pair's 2 x 4096 samples are
// fetched while the current pair is transformed -- into the 128 accumulator registers (AGPRs) the wave does not use
// otherwise, by asm loads the compiler knows nothing about: it never sees a pending value, because every use sits
// behind the asm wait statement that names all 128 registers.  vmcnt counts loads and stores together, in order, and
// the stores of the current pair are issued AFTER the prefetch; they are therefore 16-byte stores (re-laid out through
// the wave's LDS image), few enough (<= 30) for `s_waitcnt vmcnt(#stores)` to wait for the prefetch alone.
// One-shot calls on long signals only (4096-point blocks, >= kOsWaveMinTiles block pairs, hop a multiple of 256, n a
// multiple of 4): one wave walks a pair in ~10 us, so few pairs are better off spread over 256 threads each.
constexpr int kOsWaveThreads = 256;
constexpr int kOsWaveWaves = kOsWaveThreads / 64;
constexpr int kOsWaveHalf = kWaveN / 2 + 1;                      // spectrum bins kept in LDS
constexpr int kOsWaveHalfPad = (kOsWaveHalf + 7) & ~7;
constexpr size_t kOsWaveLdsBytes = kOsWaveHalfPad * sizeof(c32) + kOsWaveWaves * kWaveLdsWideBytes;  // 149,568 B: one workgroup per CU
constexpr size_t kOsWaveMinTiles = 4096;

// a * conj(b) in two packed instructions (see cmul)
DSP_DEV c32 cmul_conj(c32 a, c32 b) {
    c32 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
        : "=&v"(r)
        : "v"(a), "v"(b));
    return r;
}

// dword loads into AGPRs, 16 rows of 64 samples per base pointer (the immediate offset reaches 4 KB)
#define OSW_LOAD(dst, base, off) asm volatile("global_load_dword %0, %1, off offset:" #off " nt" : "=a"(dst) : "v"(base))
#define OSW_ROW8(arr, k, base)                                                                                           \
    OSW_LOAD(arr[8 * (k) + 0], base, 0); OSW_LOAD(arr[8 * (k) + 1], base, 256); OSW_LOAD(arr[8 * (k) + 2], base, 512);   \
    OSW_LOAD(arr[8 * (k) + 3], base, 768); OSW_LOAD(arr[8 * (k) + 4], base, 1024); OSW_LOAD(arr[8 * (k) + 5], base, 1280); \
    OSW_LOAD(arr[8 * (k) + 6], base, 1536); OSW_LOAD(arr[8 * (k) + 7], base, 1792)
#define OSW_ROW16(arr, k, base)                                                                                          \
    OSW_LOAD(arr[16 * k + 0], base, 0); OSW_LOAD(arr[16 * k + 1], base, 256); OSW_LOAD(arr[16 * k + 2], base, 512);       \
    OSW_LOAD(arr[16 * k + 3], base, 768); OSW_LOAD(arr[16 * k + 4], base, 1024); OSW_LOAD(arr[16 * k + 5], base, 1280);   \
    OSW_LOAD(arr[16 * k + 6], base, 1536); OSW_LOAD(arr[16 * k + 7], base, 1792); OSW_LOAD(arr[16 * k + 8], base, 2048);  \
    OSW_LOAD(arr[16 * k + 9], base, 2304); OSW_LOAD(arr[16 * k + 10], base, 2560); OSW_LOAD(arr[16 * k + 11], base, 2816); \
    OSW_LOAD(arr[16 * k + 12], base, 3072); OSW_LOAD(arr[16 * k + 13], base, 3328); OSW_LOAD(arr[16 * k + 14], base, 3584); \
    OSW_LOAD(arr[16 * k + 15], base, 3840)
// ties 16 registers to the (volatile, hence ordered) wait in front: no use of them can move above it
#define OSW_TIE16(arr, k)                                                                                              \
    asm volatile("" : "+a"(arr[16 * k + 0]), "+a"(arr[16 * k + 1]), "+a"(arr[16 * k + 2]), "+a"(arr[16 * k + 3]),        \
                 "+a"(arr[16 * k + 4]), "+a"(arr[16 * k + 5]), "+a"(arr[16 * k + 6]), "+a"(arr[16 * k + 7]),            \
                 "+a"(arr[16 * k + 8]), "+a"(arr[16 * k + 9]), "+a"(arr[16 * k + 10]), "+a"(arr[16 * k + 11]),          \
                 "+a"(arr[16 * k + 12]), "+a"(arr[16 * k + 13]), "+a"(arr[16 * k + 14]), "+a"(arr[16 * k + 15]))

// I0 = (4096 - hop) / 256: the first kept 256-sample row group of a block.  A template parameter so that the 16 - I0
// LDS reads and 16-byte stores per block are straight-line code (a test per store would serialise read -> store).
template <int I0>
__global__ __launch_bounds__(kOsWaveThreads) void fir_os_wave_kernel(
    const float *__restrict__ x, float *__restrict__ y, size_t n, const c32 *__restrict__ hspec, int ntaps,
    unsigned blocks_per_channel, size_t total_blocks, size_t tiles, size_t chunk) {
    constexpr int T = 64, P = kWavePoints;
    constexpr int hop = kWaveN - 256 * I0;
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char os_wave_lds[];
    c32 *hs = reinterpret_cast<c32 *>(os_wave_lds);
    const int lane = static_cast<int>(threadIdx.x) & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    float *img = reinterpret_cast<float *>(os_wave_lds + kOsWaveHalfPad * sizeof(c32) + wave * kWaveLdsWideBytes);
    constexpr float scale = 1.0f / kWaveN;  // the 1/N of the inverse transform rides the spectrum
    for (int i = threadIdx.x; i < kOsWaveHalf; i += kOsWaveThreads) {
        const c32 h = hspec[i];
        hs[i] = mk(h.x * scale, h.y * scale);
    }
    __syncthreads();  // the only workgroup barrier: from here on every wave runs on its own

    const long long L = hop;   // see fir_overlap_save_kernel; here a multiple of 256
    constexpr int i0 = I0;     // 16 - i0 16-byte stores per block
    const long long nn = static_cast<long long>(n);
    // each XCD streams one contiguous range of block pairs; its waves stride through it
    const size_t xcd = blockIdx.x & 7u;
    const size_t waves_per_xcd = static_cast<size_t>(gridDim.x >> 3) * kOsWaveWaves;
    const c32 *hlo = hs + lane;                  // H[lane + 64 m],         m < 32
    const c32 *hhi = hs + (kWaveN / 2 - lane);   // H[4096 - (lane + 64 m)], m >= 32:  hhi[-64 (m - 32)]

    // Where a block pair lives (wave-uniform): element offsets of the two block starts in x / y.  `plain`: both blocks
    // exist, lie wholly inside their channels and are not block 0 of a channel -- the only pairs this kernel transforms.
    // The others (the pairs holding a channel's first or last block: zero head, outputs from `order` on, ragged end) are
    // left to fir_overlap_save_kernel, launched over exactly those pairs (OsExtra::edge_pairs), which keeps every rare
    // path out of this loop.
    struct Pair {
        size_t oa, ob;
        bool valid, plain;
    };
    // (channel, block) of block q = 2 * tile, stepped along with t instead of divided out for every pair
    const size_t t0 = static_cast<size_t>(blockIdx.x >> 3) * kOsWaveWaves + wave;
    const size_t q0 = 2 * (xcd * chunk + t0), dq = 2 * waves_per_xcd;
    size_t ch_q = q0 / blocks_per_channel;
    unsigned blk_q = static_cast<unsigned>(q0 % blocks_per_channel);
    const size_t dch = dq / blocks_per_channel;
    const unsigned dblk = static_cast<unsigned>(dq % blocks_per_channel);
    auto locate = [&](size_t t) {  // must be called with t = t0, t0 + waves_per_xcd, ... in turn
        Pair p{};
        const size_t tile = xcd * chunk + t;
        p.valid = t < chunk && tile < tiles;
        if (!p.valid) return p;
        // the two real blocks riding one complex transform are consecutive entries of the flattened (channel, block)
        // list: with an odd block count per channel a pair straddles two channels
        const size_t ch_a = ch_q;
        const unsigned blk_a = blk_q;
        const bool has_b = 2 * tile + 1 < total_blocks;
        const bool wrap = blk_a + 1 == blocks_per_channel;
        const size_t ch_b = wrap ? ch_a + 1 : ch_a;
        const unsigned blk_b = wrap ? 0u : blk_a + 1;
        ch_q += dch;
        blk_q += dblk;
        if (blk_q >= blocks_per_channel) blk_q -= blocks_per_channel, ++ch_q;
        const long long in_a = static_cast<long long>(blk_a) * L, in_b = static_cast<long long>(blk_b) * L;
        p.oa = ch_a * n + static_cast<size_t>(in_a);
        p.ob = ch_b * n + static_cast<size_t>(in_b);
        p.plain = has_b && blk_a != 0 && blk_b != 0 && in_a + kWaveN <= nn && in_b + kWaveN <= nn;
        return p;
    };

    float na[P], nb[P];  // the prefetched samples (AGPRs)
#pragma unroll
    for (int m = 0; m < P; ++m) na[m] = nb[m] = 0.0f;
    auto prefetch = [&](const Pair &p) {
        const float *a0 = x + p.oa + lane, *a1 = a0 + 1024, *a2 = a0 + 2048, *a3 = a0 + 3072;
        const float *b0 = x + p.ob + lane, *b1 = b0 + 1024, *b2 = b0 + 2048, *b3 = b0 + 3072;
        OSW_ROW16(na, 0, a0); OSW_ROW16(na, 1, a1); OSW_ROW16(na, 2, a2); OSW_ROW16(na, 3, a3);
        OSW_ROW16(nb, 0, b0); OSW_ROW16(nb, 1, b1); OSW_ROW16(nb, 2, b2); OSW_ROW16(nb, 3, b3);
    };
    // the same loads in 32 instalments of four (instalment k: rows 4k' .. 4k'+3 of block a for k < 16, of block b after)
    // (the first 4 * I0 rows of block b are the last 4 * I0 rows of block a -- a plain pair is two consecutive blocks of one
    // channel, hop = 4096 - 256 * I0 samples = whole rows -- and sit in the same lanes: they are not loaded a second time)
    auto prefetch_part = [&](const Pair &p, int k) {
        if (k >= 16 && k < 16 + I0) return;
#define OSW_PART(K)                                                                                       \
    case K: {                                                                                            \
        const float *base = x + ((K) < 16 ? p.oa : p.ob) + lane + 256 * ((K) & 15);                       \
        if ((K) < 16) {                                                                                  \
            OSW_LOAD(na[4 * ((K) & 15) + 0], base, 0); OSW_LOAD(na[4 * ((K) & 15) + 1], base, 256);        \
            OSW_LOAD(na[4 * ((K) & 15) + 2], base, 512); OSW_LOAD(na[4 * ((K) & 15) + 3], base, 768);      \
        } else {                                                                                         \
            OSW_LOAD(nb[4 * ((K) & 15) + 0], base, 0); OSW_LOAD(nb[4 * ((K) & 15) + 1], base, 256);        \
            OSW_LOAD(nb[4 * ((K) & 15) + 2], base, 512); OSW_LOAD(nb[4 * ((K) & 15) + 3], base, 768);      \
        }                                                                                                \
    } break
        switch (k) {
            OSW_PART(0); OSW_PART(1); OSW_PART(2); OSW_PART(3); OSW_PART(4); OSW_PART(5); OSW_PART(6); OSW_PART(7);
            OSW_PART(8); OSW_PART(9); OSW_PART(10); OSW_PART(11); OSW_PART(12); OSW_PART(13); OSW_PART(14); OSW_PART(15);
            OSW_PART(16); OSW_PART(17); OSW_PART(18); OSW_PART(19); OSW_PART(20); OSW_PART(21); OSW_PART(22); OSW_PART(23);
            OSW_PART(24); OSW_PART(25); OSW_PART(26); OSW_PART(27); OSW_PART(28); OSW_PART(29); OSW_PART(30); OSW_PART(31);
        }
#undef OSW_PART
    };

    size_t t = t0;
    Pair nxt = locate(t);
    while (nxt.valid && !nxt.plain) nxt = locate(t += waves_per_xcd);
    if (nxt.valid) prefetch(nxt);
    bool stores_behind = false;  // exactly 2 * (16 - I0) stores were issued after the pending prefetch (else: wait for everything)
    while (nxt.valid) {
        float *const ya = y + nxt.oa + 4 * lane, *const yb = y + nxt.ob + 4 * lane;
        // wait for the prefetch only: the youngest 2 * (16 - I0) operations are the previous pair's stores
        if (stores_behind) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (16 - I0)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        OSW_TIE16(na, 0); OSW_TIE16(na, 1); OSW_TIE16(na, 2); OSW_TIE16(na, 3);
        OSW_TIE16(nb, 0); OSW_TIE16(nb, 1); OSW_TIE16(nb, 2); OSW_TIE16(nb, 3);
        c32 r[P];
#pragma unroll
        for (int m = 0; m < P; ++m) r[m] = mk(na[m], m < 4 * I0 ? na[m + P - 4 * I0] : nb[m]);
        asm volatile("" ::: "memory");  // the prefetch below overwrites na/nb: keep it behind the copies above
        do nxt = locate(t += waves_per_xcd);
        while (nxt.valid && !nxt.plain);
        // the next pair's loads go out eight at a time at sixteen points spread over the two transforms (all at once they
        // hold the wave up while the memory queue is full: 1.89 ms on C3 against 1.82; 32 instalments of four: 1.85)
        const bool more = nxt.valid;

        // (the hooks behind the 16-point sub-transforms: i = 0..3 and 8..11 of each transform)
        auto instalment = [&](int i, int half) {
            if (!more || (i & 4)) return;
            const int j = (i & 3) + 4 * (i >> 3) + 8 * half;
            prefetch_part(nxt, 2 * j);
            prefetch_part(nxt, 2 * j + 1);
        };
        fft4096_wave<+1, true>(r, img, lane, [&](int i) { instalment(i, 0); });
#pragma unroll
        for (int m = 0; m < P; ++m) {  // bin k = lane + 64 m
            if (m < P / 2) r[m] = cmul(r[m], hlo[T * m]);
            else r[m] = cmul_conj(r[m], hhi[-T * (m - P / 2)]);
        }
        fft4096_wave<-1, true>(r, img, lane, [&](int i) { instalment(i, 1); });

        // Outputs k in [256 * I0, 4096) of both blocks are kept.  They leave as 16-byte stores: each plane is re-laid out
        // through the LDS image (rows of 64 floats, linear), a lane then owns 4 consecutive outputs and a wave
        // instruction writes 1 KB.
        // (both planes at once: block a's in the first 4096 floats of the image, block b's in the next 4096)
#pragma unroll
        for (int m = 4 * i0; m < P; ++m) {
            img[m * 64 + lane] = r[m].x;
            img[kWaveN + m * 64 + lane] = r[m].y;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            f4 va[16 - i0], vb[16 - i0];
#pragma unroll
            for (int i = i0; i < 16; ++i) {
                va[i - i0] = *reinterpret_cast<const f4 *>(img + 256 * i + 4 * lane);
                vb[i - i0] = *reinterpret_cast<const f4 *>(img + kWaveN + 256 * i + 4 * lane);
            }
#pragma unroll
            for (int i = i0; i < 16; ++i) {
                __builtin_nontemporal_store(va[i - i0], reinterpret_cast<f4 *>(ya + 256 * i));
                __builtin_nontemporal_store(vb[i - i0], reinterpret_cast<f4 *>(yb + 256 * i));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        stores_behind = true;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no prefetch is pending here; belt and braces
}

// Round 3: the same work with TWO waves per SIMD and no hand-scheduled prefetch.  The transform's 64 x 64 transpose in its
// QUAD form (fft_wave_core.h: v_permlane32_swap trades the off-diagonal blocks, two 32 x 33 images) needs 16.9 KB per
// wave instead of 33 KB, so a CU holds the images of eight waves; with <= 256 registers per wave two of them share a
// SIMD and cover each other's loads, LDS phases and issue bubbles -- plain HIP, scheduled by the compiler, none of the
// AGPR / vmcnt bookkeeping above.  A wave walks a RUN of consecutive block pairs of the flattened (channel, block) list
// and keeps the 256 * I0 samples that the next pair's first block shares with this pair's second block in registers, so
// inside a run every input sample is read once (read amplification 4096 / hop only at the start of a run).
// Same pairs as fir_os_wave_kernel (plain ones; the edge pairs go to fir_overlap_save_kernel), same arithmetic per pair:
// the two kernels give identical bits.
constexpr int kOsWave2Waves = 8;
constexpr int kOsWave2Threads = 64 * kOsWave2Waves;
constexpr size_t kOsWave2LdsBytes = kOsWaveHalfPad * sizeof(c32) + kOsWave2Waves * kWaveLdsQuadBytes;  // 151,616 B

template <int I0>
__global__ __launch_bounds__(kOsWave2Threads) void fir_os_wave2_kernel(
    const float *__restrict__ x, float *__restrict__ y, size_t n, const c32 *__restrict__ hspec, unsigned blocks_per_channel,
    size_t total_blocks, size_t tiles, unsigned run_len, size_t runs_per_xcd, unsigned *__restrict__ draw, unsigned chunk) {
    constexpr int T = 64, P = kWavePoints;
    constexpr long long L = kWaveN - 256 * I0;
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char os_wave_lds[];
    c32 *hs = reinterpret_cast<c32 *>(os_wave_lds);
    const int lane = static_cast<int>(threadIdx.x) & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    float *img = reinterpret_cast<float *>(os_wave_lds + kOsWaveHalfPad * sizeof(c32) + wave * kWaveLdsQuadBytes);
    constexpr float scale = 1.0f / kWaveN;  // the 1/N of the inverse transform rides the spectrum
    for (int i = threadIdx.x; i < kOsWaveHalf; i += kOsWave2Threads) {
        const c32 h = hspec[i];
        hs[i] = mk(h.x * scale, h.y * scale);
    }
    __syncthreads();  // the only workgroup barrier
    const c32 *hlo = hs + lane;                  // H[lane + 64 m],         m < 32
    const c32 *hhi = hs + (kWaveN / 2 - lane);   // H[4096 - (lane + 64 m)], m >= 32:  hhi[-64 (m - 32)]
    const long long nn = static_cast<long long>(n);
    // Which pairs a wave takes.  draw != nullptr (the default, late round 3): the eight XCDs own the eighths [x chunk,
    // (x + 1) chunk) of the pair list and every wave DRAWS its next `run_len` consecutive pairs from its XCD's counter when
    // it is free -- the pairs are then worked in the order in which waves become free, as a grid of one-pair workgroups would
    // be dispatched, and the waves of a SIMD stay out of step.  With the pairs dealt out in advance (draw == nullptr: each
    // XCD a contiguous range of runs, workgroups b and b + 8 sharing an XCD) the same kernel is 9-12 % slower and its
    // memory side alone 6-14 % (scripts/os_skeleton.hip, scripts/os_quad_ablate.hip, profiles/r03_fir_os_drawn_pairs.txt).
    // Every wave leaves the loop: the counter only grows.
    const unsigned xcd = blockIdx.x & 7u;  // (one counter for the whole chip instead of one per XCD: 1.91 ms against 1.62)
    // rows 64 - 4 I0 .. 63 of the last pair's second block = rows 0 .. 4 I0 - 1 of the next pair's first: kept in registers
    // while they fit (hops from 2560 samples up; below that 28 or 32 more registers would spill, and the rows are read again)
    constexpr bool KEEP = I0 <= 6;
    constexpr int kKeep = KEEP ? 4 * I0 : 1;
    float keep[kKeep];
#pragma unroll
    for (int m = 0; m < kKeep; ++m) keep[m] = 0.0f;
    for (bool first = true;; first = false) {
        size_t tile0, tile_end;
        if (draw) {
            unsigned t = 0;
            if (lane == 0) t = atomicAdd(draw + 32 * xcd, run_len);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t >= chunk) break;
            tile0 = static_cast<size_t>(xcd) * chunk + t;
            tile_end = static_cast<size_t>(xcd) * chunk + (t + run_len < chunk ? t + run_len : chunk);
        } else {
            if (!first) break;
            const size_t run = xcd * runs_per_xcd + static_cast<size_t>(blockIdx.x >> 3) * kOsWave2Waves + wave;
            tile0 = run * run_len;
            tile_end = tile0 + run_len;
            if ((static_cast<size_t>(blockIdx.x >> 3) * kOsWave2Waves + wave) >= runs_per_xcd) break;
        }
        if (tile_end > tiles) tile_end = tiles;
        if (tile0 >= tile_end) continue;
        // (channel, block) of the first block of the pair, stepped by two blocks per pair (drawn pairs: the host has checked
        // that block numbers fit 32 bits)
        size_t ch;
        unsigned blk;
        if (draw) {
            const unsigned q = 2u * static_cast<unsigned>(tile0);
            ch = q / blocks_per_channel;
            blk = q % blocks_per_channel;
        } else {
            ch = (2 * tile0) / blocks_per_channel;
            blk = static_cast<unsigned>((2 * tile0) % blocks_per_channel);
        }
        bool have_keep = false;
        for (size_t tile = tile0; tile < tile_end; ++tile) {
            const size_t ch_a = ch;
            const unsigned blk_a = blk;
            const bool wrap = blk_a + 1 == blocks_per_channel;
            const unsigned blk_b = wrap ? 0u : blk_a + 1;
            blk += 2;
            if (blk >= blocks_per_channel) blk -= blocks_per_channel, ++ch;
            const long long in_a = static_cast<long long>(blk_a) * L, in_b = in_a + L;
            // plain: both blocks exist, lie wholly inside ONE channel and neither is its block 0 (see fir_os_wave_kernel)
            const bool plain = 2 * tile + 1 < total_blocks && blk_a != 0 && blk_b != 0 && in_b + kWaveN <= nn;
            if (!plain) {
                have_keep = false;
                continue;
            }
            const float *pa = x + ch_a * n + static_cast<size_t>(in_a) + lane, *pb = pa + L;
            c32 r[P];
            if (KEEP && have_keep) {
#pragma unroll
                for (int m = 0; m < kKeep; ++m) r[m].x = keep[m];
            } else {
                // (these rows are the last rows of the pair before, which a neighbouring wave of the same XCD reads at about the
                // same time, and the last 4 I0 rows of block b are the next pair's first: both through the cache, the rest streams)
#pragma unroll
                for (int m = 0; m < 4 * I0; ++m) r[m].x = pa[T * m];
            }
#pragma unroll
            for (int m = 4 * I0; m < P; ++m) r[m].x = __builtin_nontemporal_load(pa + T * m);
#pragma unroll
            for (int m = 4 * I0; m < P - 4 * I0; ++m) r[m].y = __builtin_nontemporal_load(pb + T * m);
#pragma unroll
            for (int m = P - 4 * I0; m < P; ++m) r[m].y = pb[T * m];
#pragma unroll
            for (int m = 0; m < 4 * I0; ++m) r[m].y = r[m + P - 4 * I0].x;  // hop = whole rows: block b starts inside block a
            if constexpr (KEEP) {
#pragma unroll
                for (int m = 0; m < 4 * I0; ++m) keep[m] = r[m + P - 4 * I0].y;
                have_keep = true;
            }

            fft4096_wave_quad<+1>(r, img, lane);
#pragma unroll
            for (int m = 0; m < P; ++m) {  // bin k = lane + 64 m
                if (m < P / 2) r[m] = cmul(r[m], hlo[T * m]);
                else r[m] = cmul_conj(r[m], hhi[-T * (m - P / 2)]);
            }
            fft4096_wave_quad<-1>(r, img, lane);

            // outputs k in [256 * I0, 4096) of both blocks are kept; each plane is re-laid out through the image so that a lane
            // owns 4 consecutive outputs and a wave instruction writes 1 KB
            float *const ya = y + ch_a * n + static_cast<size_t>(in_a) + 4 * lane, *const yb = ya + L;
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
        }
    }  // draws / the one dealt run
}

// ---------------------------------------------------------------------------------------------
// centred moving average
// ---------------------------------------------------------------------------------------------
constexpr int kMaPerMax = 16;  // outputs per thread: 16 (4096 outputs per workgroup), fewer for channels shorter than that

DSP_DEV double shfl_up_f64(double v, int delta) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_up(lo, delta, 64);
    hi = __shfl_up(hi, delta, 64);
    return __hiloint2double(hi, lo);
}
// Inclusive prefix sum of a double over the 64 lanes by data-parallel-primitive moves, no LDS round trip: Kogge-Stone inside
// every row of 16 lanes (row_shr 1, 2, 4, 8; lanes shifted in from outside the row read 0), then lane 15 of rows 0 and 2
// into rows 1 and 3 (row_bcast15, row mask 0xa) and lane 31 into rows 2 and 3 (row_bcast31, row mask 0xc).  Three vector
// instructions a step against the eight (two of them LDS permutes with their latency) of a shuffle-based step.
template <int CTRL, int ROW_MASK>
DSP_DEV double dpp_move_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
DSP_DEV double wave_inclusive_sum_f64(double v) {
    v += dpp_move_f64<0x111, 0xf>(v);
    v += dpp_move_f64<0x112, 0xf>(v);
    v += dpp_move_f64<0x114, 0xf>(v);
    v += dpp_move_f64<0x118, 0xf>(v);
    v += dpp_move_f64<0x142, 0xa>(v);
    v += dpp_move_f64<0x143, 0xc>(v);
    return v;
}
// 16 pad BYTES per 16 samples (4 floats / 2 doubles): conflict-free ds_read_b128 for consecutive lanes on consecutive chunks
template <typename T>
DSP_DEV int ma_pad(int i) { return i + ((i >> 4) * Smp<T>::kVec); }

// The reference's running sum (filter.h:88-98) starts from x[0..P) and then adds x[i+mid], drops
// x[i-mid-1] with mid = P/2.  Algebraically that is
//     y[i] = (sum_{k=i-mid}^{i+mid} x[k]  -  (P even ? x[P] : 0)) / P      for mid <= i < n-mid
// (for odd P the plain centred mean; for even P -- reachable through numPoints(), which does not
// re-force odd, filter.h:72-74 -- a 2*mid+1 window minus the constant x[P]).  Reproduced as is.
//
// A workgroup stages tile + window samples (aligned down to a 16-byte group so the loads are 16 bytes per lane)
// and turns them into a two-level exclusive prefix sum in LDS: a prefix in the sample type inside every chunk of
// 16 samples (overwriting the samples) plus a DOUBLE prefix over the chunk sums (block scan with wave
// shuffles).  y[j] = (G[j + window] - G[j]) / P with G[u] = chunk_prefix[u >> 4] + in_chunk[u];
// consecutive lanes take consecutive outputs, so the LDS reads are conflict-free and the global
// stores coalesced.  Nothing is carried along the signal, hence none of the reference's accumulated
// drift.  LDS: T xs[ma_pad(span)], double pc[chunks], double wave_tot[8 per pair of rounds], double d1[tile / 16 + 1].
// T = float (BASELINE's type) or double (SampleBuffer<double>, what the reference's own callers filter:
// eval/moving_average_filter_demo/main.cpp:14).
//
// PERIODIC (many short channels): the channels lie one behind the other in memory, so the batch is filtered as ONE signal of
// channels * period samples in full 4096-output tiles -- a kept output's window never leaves its channel (the first and last
// `mid` outputs of a channel are zero, filter.h:88-98) -- and only the edge test and the even-P constant go by the position
// inside the channel (j mod period).  One workgroup per channel would stage and scan 256-sample tiles, mostly launch overhead.
template <bool ALIGNED, int kMaPer, typename T = float, bool PERIODIC = false>
__global__ __launch_bounds__(kThreads) void moving_average_kernel(const T *__restrict__ x, T *__restrict__ y,
                                                                  size_t n, int npoints, int span, int cpt, int skip_lo,
                                                                  int skip_n, double inv, int ld_mode, int halo,  // inv = 1.0 / npoints (host)
                                                                  unsigned period) {
    using V = typename Smp<T>::vec;
    constexpr int kVec = Smp<T>::kVec;
    constexpr int kMaTile = kThreads * kMaPer;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int chunks = span >> 4;  // span is a multiple of 16
    T *xs = reinterpret_cast<T *>(smem_raw);
    double *pc = reinterpret_cast<double *>(xs + ma_pad<T>(span));  // ma_pad(span) * sizeof(T) is a multiple of 16
    double *wave_tot = pc + chunks;
    const int tid = threadIdx.x;
    const size_t ch = blockIdx.y;
    const long long nn = static_cast<long long>(n);
    const long long mid = npoints / 2;
    const int window = 2 * static_cast<int>(mid) + 1;
    // XCD-contiguous tiles along the channel (workgroups x and x + 8 share an XCD).  Worth 1 %; plain (cached) loads
    // would let the neighbouring tile's window halo come out of L2 (traffic 8.55 -> 8.01 GiB) but run 3 % slower
    // than streaming loads
    // (skip_n > 0: this launch covers only the tiles outside [skip_lo, skip_lo + skip_n), which moving_average_seq_kernel takes)
    const unsigned bx = skip_n > 0 ? blockIdx.x + (static_cast<int>(blockIdx.x) >= skip_lo ? skip_n : 0)
                                   : ((gridDim.x & 7u) == 0 ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x);
    const long long out0 = static_cast<long long>(bx) * kMaTile;  // first output of the tile
    const long long first = out0 - mid;                                   // first sample the tile needs
    const long long seg0 = first & ~static_cast<long long>(kVec - 1);     // staged from here (16-byte aligned)
    const int shift = static_cast<int>(first - seg0);                     // 0..kVec-1
    const T *xc = x + ch * n;
    T *yc = y + ch * n;
    const T corr = (!PERIODIC && (npoints & 1) == 0 && npoints < nn) ? xc[npoints] : T(0);

    int u0 = tid * kVec;
    if (ALIGNED) {
        // four loads in flight per thread, no branch between them (a bounds branch around every load makes the compiler
        // wait for each one before issuing the next).  n and seg0 are multiples of the group, so a group lies wholly inside
        // the channel or wholly outside; outside ones are aimed at the channel's first sample and zeroed afterwards.
        constexpr int kStep = kThreads * kVec;
        if (seg0 >= 0 && seg0 + span <= nn) {  // interior tile: nothing to clamp
            for (; u0 + 3 * kStep < span; u0 += 4 * kStep) {
                V v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int u = u0 + i * kStep;
                    const bool cached = ld_mode == 2 || (ld_mode == 1 && (u < halo || u >= span - halo));
                    v[i] = cached ? *reinterpret_cast<const V *>(xc + seg0 + u) : load_stream<T>(xc + seg0 + u);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<V *>(xs + ma_pad<T>(u0 + i * kStep)) = v[i];
            }
        }
        for (; u0 + 3 * kStep < span; u0 += 4 * kStep) {
            V v[4];
            bool ok[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long long g = seg0 + u0 + i * kStep;
                ok[i] = g >= 0 && g < nn;
                v[i] = load_stream<T>(xc + (ok[i] ? g : 0));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<V *>(xs + ma_pad<T>(u0 + i * kStep)) = ok[i] ? v[i] : V(0);
        }
    }
    for (int u = u0; u < span; u += kThreads * kVec) {
        const long long g = seg0 + u;
        V v;
        if (ALIGNED && g >= 0 && g + kVec - 1 < nn) {
            v = load_stream<T>(xc + g);
        } else {
#pragma unroll
            for (int k = 0; k < kVec; ++k) v[k] = (g + k >= 0 && g + k < nn) ? xc[g + k] : T(0);
        }
        *reinterpret_cast<V *>(xs + ma_pad<T>(u)) = v;
    }
    __syncthreads();

    // round k: thread t owns chunk t + 256*k (consecutive lanes on consecutive chunks: the 20-dword lane
    // stride keeps the 16-byte LDS accesses conflict-free).  In-chunk exclusive prefixes (sample type, in place),
    // then a block-wide exclusive scan of the chunk totals in double, rounds chained through `base`.
    // Rounds go in PAIRS behind one barrier (C3: 289 chunks = one full round and one of 33 chunks, whose scan only
    // wave 0 runs): the two wave scans are independent chains, and a wave without a chunk in a round skips its scan.
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto chunk_prefix = [&](int c) -> double {  // exclusive prefixes inside chunk c, in place; returns the chunk's sum
        T run = T(0);
        if (c < chunks) {
#pragma unroll
            for (int q = 0; q < 16 / kVec; ++q) {
                V *p = reinterpret_cast<V *>(xs + ma_pad<T>(16 * c + kVec * q));
                const V v = *p;
                V e;
#pragma unroll
                for (int i = 0; i < kVec; ++i) {
                    e[i] = run;
                    run = run + v[i];
                }
                *p = e;
            }
        }
        return static_cast<double>(run);
    };
    double base = 0.0;
    for (int k = 0; k < cpt; k += 2) {
        const int c0 = tid + kThreads * k, c1 = c0 + kThreads;
        const double m0 = chunk_prefix(c0), m1 = chunk_prefix(c1);
        double i0 = m0, i1 = m1;
        if (kThreads * k + 64 * wave < chunks) i0 = wave_inclusive_sum_f64(m0);        // (wave-uniform)
        if (kThreads * (k + 1) + 64 * wave < chunks) i1 = wave_inclusive_sum_f64(m1);  // (wave-uniform)
        if (lane == 63) {
            wave_tot[4 * k + wave] = i0;
            wave_tot[4 * k + 4 + wave] = i1;
        }
        __syncthreads();
        double t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = wave_tot[4 * k + q];
        double r0 = base + (i0 - m0);  // exclusive prefix of the chunk sums: rounds before, waves before, lanes before
        if (wave > 0) r0 += t[0];
        if (wave > 1) r0 += t[1];
        if (wave > 2) r0 += t[2];
        if (c0 < chunks) pc[c0] = r0;
        base += t[0] + t[1] + t[2] + t[3];
        double r1 = base + (i1 - m1);
        if (wave > 0) r1 += t[4];
        if (wave > 1) r1 += t[5];
        if (wave > 2) r1 += t[6];
        if (c1 < chunks) pc[c1] = r1;
        base += t[4] + t[5] + t[6] + t[7];
    }
    __syncthreads();

    const double dcorr = static_cast<double>(corr);
    const long long hi = (nn - mid > mid + 1) ? nn - mid : mid + 1;  // out[mid] is always written (filter.h:93)
    // double samples: divide as the reference does (sum / M, filter.h:93,97) rather than multiply by the reciprocal
    const double ncorr_inv = -dcorr * inv;
    auto mean = [&](double sum) -> T {
        if constexpr (sizeof(T) == 8) return (sum - dcorr) / static_cast<double>(npoints);
        else return static_cast<T>(fma(sum, inv, ncorr_inv));  // (sum - corr) / P in one operation; corr = 0 for odd P
    };
    if constexpr (PERIODIC) {  // (the window fits the channel: the host has checked 2 mid + 1 <= period)
        const unsigned long long j0 = static_cast<unsigned long long>(out0) + tid;
        const size_t c0 = j0 / period;
        const unsigned r0 = static_cast<unsigned>(j0 - c0 * period);
        const unsigned umid = static_cast<unsigned>(mid), uhi = period - umid;
        const bool even = (npoints & 1) == 0;
#pragma unroll
        for (int i = 0; i < kMaPer; ++i) {
            const int jrel = kThreads * i + tid;
            const long long j = out0 + jrel;
            if (j >= nn) continue;
            unsigned r = r0 + kThreads * i;
            const unsigned q = r / period;
            r -= q * period;
            T v = T(0);
            if (r >= umid && r < uhi) {
                const int lo = jrel + shift, up = lo + window;
                const double sum = (pc[up >> 4] - pc[lo >> 4]) +
                                   (static_cast<double>(xs[ma_pad<T>(up)]) - static_cast<double>(xs[ma_pad<T>(lo)]));
                const double dc = even ? static_cast<double>(xc[(c0 + q) * period + npoints]) : 0.0;
                if constexpr (sizeof(T) == 8) v = (sum - dc) / static_cast<double>(npoints);
                else v = static_cast<T>(fma(sum, inv, -dc * inv));
            }
            __builtin_nontemporal_store(v, yc + j);
        }
        return;
    }
    if (out0 >= mid && out0 + kMaTile <= hi) {  // interior tile: every output has its full window
        // Output i of a thread sits kThreads = 16 chunks further on than output i - 1, so all four LDS addresses move by
        // constants: they are computed once and the loop body is reads at immediate offsets plus the arithmetic (left to
        // the compiler the padding's shift-and-add was redone for every output: 26 vector instructions per output against
        // 8, in a kernel whose vector ALU was 87 % busy -- profiles/r03_mavg_pmc.txt).
        // The chunk-prefix difference pc[up >> 4] - pc[lo >> 4] depends on lo's chunk and on one bit (does lo's offset in
        // its chunk carry into the next chunk when the window is added): it is formed once per chunk and bit -- D0 over pc
        // in place, D1 behind it -- so that an output costs three LDS reads instead of four.  LDS is what this kernel
        // waits for once the vector ALU is out of the way (its index unit was busy 0.78 of the time: 24 bytes read per
        // 4-byte output); the subtraction is the same one, so are the bits.
        {
            const int w16 = window >> 4, n_d = (kMaTile >> 4) + 1;  // lo >> 4 takes the values 0 .. kMaTile / 16
            double *e1 = wave_tot + 4 * ((cpt + 1) & ~1);
            double a[2] = {0.0, 0.0}, b0[2] = {0.0, 0.0}, b1[2] = {0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int c = tid + kThreads * r;
                if (c < n_d) {
                    a[r] = pc[c];
                    b0[r] = pc[min(c + w16, chunks - 1)];
                    b1[r] = pc[min(c + w16 + 1, chunks - 1)];  // (clamped entries belong to combinations no output uses)
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int c = tid + kThreads * r;
                if (c < n_d) {
                    pc[c] = b0[r] - a[r];
                    e1[c] = b1[r] - a[r];
                }
            }
            __syncthreads();
        }
        T *yt = yc + out0 + tid;
        const int lo0 = tid + shift, up0 = lo0 + window;  // window of output tid = staged samples [lo0, up0)
        const T *plo = xs + ma_pad<T>(lo0), *pup = xs + ma_pad<T>(up0);
        const double *qd = ((((lo0 & 15) + (window & 15)) >> 4) ? wave_tot + 4 * ((cpt + 1) & ~1) : pc) + (lo0 >> 4);
        constexpr int kRow = kThreads + (kThreads >> 4) * kVec;  // ma_pad(u + kThreads) - ma_pad(u)
#pragma unroll
        for (int i = 0; i < kMaPer; ++i) {
            const double sum = qd[(kThreads >> 4) * i] + (static_cast<double>(pup[kRow * i]) - static_cast<double>(plo[kRow * i]));
            __builtin_nontemporal_store(mean(sum), yt + kThreads * i);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < kMaPer; ++i) {
        const int jrel = kThreads * i + tid;
        const long long j = out0 + jrel;
        if (j >= nn) continue;
        T v = T(0);
        if (j >= mid && j < hi) {
            const int lo = jrel + shift, up = lo + window;
            const double sum = (pc[up >> 4] - pc[lo >> 4]) +
                               (static_cast<double>(xs[ma_pad<T>(up)]) - static_cast<double>(xs[ma_pad<T>(lo)]));
            v = mean(sum);
        }
        __builtin_nontemporal_store(v, yc + j);
    }
}

// Round 3: the same filter with every sample read ONCE.  A workgroup walks a run of consecutive interior tiles of one
// channel (4096 outputs each) and carries the window's worth of samples that two neighbouring tiles share from tile to
// tile inside LDS -- in the prefix form they are already in (an in-chunk exclusive prefix depends on its chunk alone; the
// chunk prefixes are rebased) -- instead of staging them again from HBM / L2 (traffic 1.04x -> 1.01x of 8 B/sample, reads at
// the L2 1.125x -> 1.008x).  Only the 4096 new samples of a tile are loaded, scanned (one chunk per thread) and the next
// tile's are already on their way (registers) while this tile's outputs are formed.  Tiles whose window leaves the channel
// (the first one or two and the last ones) stay with moving_average_kernel.  Same arithmetic per output, except that the
// double chunk prefixes of a run's later tiles carry the rebased values of the earlier one (differences of doubles: exact
// to ~1e-16 relative, far inside the 1e-6 budget).
template <typename T>
__global__ __launch_bounds__(kThreads) void moving_average_seq_kernel(const T *__restrict__ x, T *__restrict__ y, size_t n,
                                                                      int npoints, int span, int tile_lo, int tile_n, int run) {
    using V = typename Smp<T>::vec;
    constexpr int kVec = Smp<T>::kVec;
    constexpr int kTile = kThreads * kMaPerMax;  // 4096 outputs
    constexpr int kNew = kTile / 16;             // 256 new chunks per tile: one per thread
    constexpr int kLoads = 16 / kVec;            // 16-byte loads per thread and tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int chunks = span >> 4, carried = chunks - kNew;
    T *xs = reinterpret_cast<T *>(smem_raw);
    double *pc = reinterpret_cast<double *>(xs + ma_pad<T>(span));  // chunks + 1 entries: pc[chunks] = total of the span
    double *wave_tot = pc + chunks + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t ch = blockIdx.y;
    const int t0 = tile_lo + static_cast<int>(blockIdx.x) * run;
    const int count = min(run, tile_lo + tile_n - t0);
    const long long mid = npoints / 2;
    const int window = 2 * static_cast<int>(mid) + 1;
    const T *xc = x + ch * n;
    T *yc = y + ch * n;
    const long long nn = static_cast<long long>(n);
    const T corr = ((npoints & 1) == 0 && npoints < nn) ? xc[npoints] : T(0);
    const double inv = 1.0 / static_cast<double>(npoints), dcorr = static_cast<double>(corr);
    const long long first0 = static_cast<long long>(t0) * kTile - mid;
    const long long seg0 = first0 & ~static_cast<long long>(kVec - 1);
    const int shift = static_cast<int>(first0 - seg0);  // the same for every tile: kTile is a multiple of the group

    // in-chunk exclusive prefixes of chunks [c_first + tid + 256 k], k < rounds, then the block scan of their totals in
    // double; pc[c] = base + (sum of the chunks before c); returns base + (sum of all of them)
    auto scan_chunks = [&](int c_first, int rounds, double base) {
        for (int k = 0; k < rounds; ++k) {
            const int c = c_first + tid + kThreads * k;
            T run_sum = T(0);
            if (c < chunks) {
#pragma unroll
                for (int q = 0; q < 16 / kVec; ++q) {
                    V *p = reinterpret_cast<V *>(xs + ma_pad<T>(16 * c + kVec * q));
                    const V v = *p;
                    V e;
#pragma unroll
                    for (int i = 0; i < kVec; ++i) {
                        e[i] = run_sum;
                        run_sum = run_sum + v[i];
                    }
                    *p = e;
                }
            }
            const double mine = static_cast<double>(run_sum);
            double incl = mine;
            if (c_first + kThreads * k + 64 * wave < chunks) incl = wave_inclusive_sum_f64(mine);  // (wave-uniform)
            if (lane == 63) wave_tot[4 * k + wave] = incl;
            __syncthreads();
            const double t0 = wave_tot[4 * k], t1 = wave_tot[4 * k + 1], t2 = wave_tot[4 * k + 2], t3 = wave_tot[4 * k + 3];
            double running = base + (incl - mine);
            if (wave > 0) running += t0;
            if (wave > 1) running += t1;
            if (wave > 2) running += t2;
            if (c < chunks) pc[c] = running;
            base += t0 + t1 + t2 + t3;
        }
        if (tid == 0) pc[chunks] = base;
        return base;
    };
    auto emit = [&](long long out0) {  // as moving_average_kernel's interior tiles: four LDS addresses, the rest immediates
        T *yt = yc + out0 + tid;
        const int lo0 = tid + shift, up0 = lo0 + window;
        const T *plo = xs + ma_pad<T>(lo0), *pup = xs + ma_pad<T>(up0);
        const double *qlo = pc + (lo0 >> 4), *qup = pc + (up0 >> 4);
        constexpr int kRow = kThreads + (kThreads >> 4) * kVec;
#pragma unroll
        for (int i = 0; i < kMaPerMax; ++i) {
            const double sum = (qup[(kThreads >> 4) * i] - qlo[(kThreads >> 4) * i]) +
                               (static_cast<double>(pup[kRow * i]) - static_cast<double>(plo[kRow * i]));
            if constexpr (sizeof(T) == 8) __builtin_nontemporal_store((sum - dcorr) / static_cast<double>(npoints), yt + kThreads * i);
            else __builtin_nontemporal_store(static_cast<T>(fma(sum, inv, -dcorr * inv)), yt + kThreads * i);
        }
    };
    const T *newp = xc + seg0 + 16 * carried + tid * kVec;  // this thread's first new sample of tile 0 (as if it were a later tile)
    auto load_new = [&](V (&v)[kLoads], int k) {
#pragma unroll
        for (int i = 0; i < kLoads; ++i) v[i] = load_stream<T>(newp + static_cast<long long>(k) * kTile + i * kThreads * kVec);
    };

    // tile 0 of the run: the whole span from memory
    {
        constexpr int kStep = kThreads * kVec;
        int u0 = tid * kVec;
        for (; u0 + 3 * kStep < span; u0 += 4 * kStep) {
            V v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = load_stream<T>(xc + seg0 + u0 + i * kStep);
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<V *>(xs + ma_pad<T>(u0 + i * kStep)) = v[i];
        }
        for (int u = u0; u < span; u += kStep) *reinterpret_cast<V *>(xs + ma_pad<T>(u)) = load_stream<T>(xc + seg0 + u);
    }
    V nxt[kLoads];
    if (count > 1) load_new(nxt, 1);
    __syncthreads();
    scan_chunks(0, (chunks + kThreads - 1) / kThreads, 0.0);
    __syncthreads();
    emit(static_cast<long long>(t0) * kTile);

    for (int k = 1; k < count; ++k) {
        __syncthreads();  // every output of the previous tile has been formed
        // carry: chunks [256, chunks) become chunks [0, carried): their in-chunk prefixes as they are, chunk prefixes rebased
        // (carried < 256, so source and destination never overlap)
        {
            const int rec = 16 + kVec;  // samples + pad of one chunk in the padded image
            const int nvec = carried * rec / kVec;
            for (int v = tid; v < nvec; v += kThreads)
                *reinterpret_cast<V *>(xs + v * kVec) = *reinterpret_cast<const V *>(xs + kNew * rec + v * kVec);
            if (tid <= carried) pc[tid] = pc[kNew + tid] - pc[kNew];
        }
        // the new samples (in flight since the previous tile) behind them
#pragma unroll
        for (int i = 0; i < kLoads; ++i) *reinterpret_cast<V *>(xs + ma_pad<T>(16 * carried + tid * kVec + i * kThreads * kVec)) = nxt[i];
        if (k + 1 < count) load_new(nxt, k + 1);
        __syncthreads();
        scan_chunks(carried, 1, pc[carried]);
        __syncthreads();
        emit(static_cast<long long>(t0 + k) * kTile);
    }
}

// new_hist[ch][k] = virtual[n + k], k < hlen, where virtual = old_hist ++ x (old_hist == nullptr: zeros)
__global__ __launch_bounds__(kThreads) void fir_history_kernel(const float *__restrict__ x, size_t n,
                                                               const float *__restrict__ old_hist,
                                                               float *__restrict__ new_hist, size_t channels, size_t hlen) {
    const size_t total = channels * hlen;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
         i += static_cast<size_t>(gridDim.x) * kThreads) {
        const size_t ch = i / hlen, k = i % hlen;
        const size_t v = n + k;  // index into old_hist ++ x
        new_hist[i] = v >= hlen ? x[ch * n + (v - hlen)] : (old_hist ? old_hist[ch * hlen + v] : 0.0f);
    }
}

}  // namespace

int launch_fir_history(const float *x, size_t n, const float *old_hist, float *new_hist, size_t channels, int ntaps,
                       hipStream_t s) {
    const size_t hlen = static_cast<size_t>(ntaps - 1);
    if (hlen == 0 || channels == 0) return DSP_GPU_OK;
    const unsigned grid = static_cast<unsigned>(std::min<size_t>((channels * hlen + kThreads - 1) / kThreads, 65536));
    hipLaunchKernelGGL(fir_history_kernel, dim3(grid), dim3(kThreads), 0, s, x, n, old_hist, new_hist, channels, hlen);
    return check_launch("fir_history_kernel");
}

int launch_fir_direct_valu(const float *x, float *y, size_t channels, size_t n, const float *d_taps, int ntaps,
                           hipStream_t s) {
    if (ntaps > 1024) return fail(DSP_GPU_EUNSUPPORTED, "direct FIR (VALU): at most 1024 taps, got %d", ntaps);
    for (; channels > kMaxGridY; channels -= kMaxGridY, x += kMaxGridY * n, y += kMaxGridY * n)  // grid.y is 16 bits
        if (int rc = launch_fir_direct_valu(x, y, kMaxGridY, n, d_taps, ntaps, s)) return rc;
    const int taps_p = ((ntaps + 19) / 20) * 20;
    const int H = ((taps_p + 8 + 15) / 16) * 16;  // staged history: the window reaches base - 8 on its last prefetch
    const int S = H + kVfTile;
    const size_t lds_bytes = (2 * static_cast<size_t>(S + (S / 16) * 4 + 8) + taps_p) * sizeof(float);
    const size_t blocks = (n + kVfTile - 1) / kVfTile;
    hipLaunchKernelGGL(fir_direct_valu_kernel, dim3(static_cast<unsigned>(blocks), static_cast<unsigned>(channels)),
                       dim3(kThreads), lds_bytes, s, x, y, n, d_taps, ntaps, taps_p, H);
    return check_launch("fir_direct_valu_kernel");
}

int launch_fir_direct(const float *x, float *y, size_t channels, size_t n, const float *d_taps, int ntaps,
                      const float *hist, int zero_head, hipStream_t s) {
    const int order = ntaps - 1;
    const int Q = (order + 15) / 16 + 1;
    // One-shot calls on many channels of less than a few workgroups' worth of samples: as one signal (PERIODIC above).
    // 7 taps, 2^28 samples: 4,194,304 x 64 samples 0.07 -> 0.60-0.64 of HBM, 1,048,576 x 256 0.25 -> 0.68, 26,843 x 10,000 0.54 -> 0.67
    // (scripts/fir_short_filters.py, profiles/r03_fir_short_filters.txt).
    // DSP_GPU_FIR_NO_FLATTEN=1 (A/B runs, tests) keeps the channels apart.
    if (!hist && zero_head == order && channels >= 2 && n >= static_cast<size_t>(ntaps) && n < 65536 &&
        channels * n < 0xffffffffull && !getenv("DSP_GPU_FIR_NO_FLATTEN")) {
        const size_t total = channels * n;
        const bool al = (total % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(y) % 16 == 0);
        auto kern = al ? fir_direct_kernel<true, false, true> : fir_direct_kernel<false, false, true>;
        const size_t lds = (static_cast<size_t>(16 * (Q - 1) + kFirOutPerWg) + 4u * (16 * Q + 16)) * sizeof(float);
        if (lds > 64 * 1024)
            if (int rc = allow_full_lds(reinterpret_cast<const void *>(kern))) return rc;
        hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>((total + kFirOutPerWg - 1) / kFirOutPerWg), 1), dim3(kFirThreads), lds, s, x, y,
                           total, size_t{1}, d_taps, ntaps, Q, 1, kFirOutPerWg, nullptr, zero_head, static_cast<unsigned>(n));
        return check_launch("fir_direct_kernel<periodic>");
    }
    // channels of at most 4096 samples: several to a workgroup (slots), each rounded up to one tile's 256 outputs
    const int quantum = kFirTile;  // (one tile; until late round 3 the 1,024 outputs of a wave-pass)
    const int S = n * 2 <= static_cast<size_t>(kFirOutPerWg) ? static_cast<int>((n + quantum - 1) / quantum) * quantum : kFirOutPerWg;
    int slots = kFirOutPerWg / S;
    // every slot carries its own history: keep the workgroup's LDS near 80 KB (two workgroups per CU) for long filters
    while (slots > 1 && (static_cast<size_t>(slots) * (16 * (Q - 1) + S) + 4u * (16 * Q + 16)) * sizeof(float) > 80u * 1024) slots /= 2;
    const size_t per_launch = kMaxGridY * static_cast<size_t>(slots);  // grid.y is 16 bits
    for (; channels > per_launch; channels -= per_launch, x += per_launch * n, y += per_launch * n) {
        if (int rc = launch_fir_direct(x, y, per_launch, n, d_taps, ntaps, hist, zero_head, s)) return rc;
        if (hist) hist += per_launch * static_cast<size_t>(order);
    }
    const size_t lds_bytes = (static_cast<size_t>(slots) * (16 * (Q - 1) + S) + 4u * (16 * Q + 16)) * sizeof(float);
    const size_t blocks = slots > 1 ? 1 : (n + kFirOutPerWg - 1) / kFirOutPerWg;
    const size_t groups = (channels + slots - 1) / slots;
    const bool aligned = (n % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(y) % 16 == 0);
    auto kern = slots > 1 ? (aligned ? fir_direct_kernel<true, true> : fir_direct_kernel<false, true>)
                          : (aligned ? fir_direct_kernel<true, false> : fir_direct_kernel<false, false>);
    if (lds_bytes > 64 * 1024)
        if (int rc = allow_full_lds(reinterpret_cast<const void *>(kern))) return rc;
    hipLaunchKernelGGL(kern, dim3(blocks, groups), dim3(kFirThreads), lds_bytes, s, x, y, n, channels, d_taps, ntaps, Q, slots, S,
                       hist, zero_head, 0u);
    return check_launch("fir_direct_kernel");
}

int launch_fir_spectrum(const float *d_taps, int ntaps, float2 *d_hspec_f2, int log2n, hipStream_t s) {
    // FORWARD transform of the taps zero padded to the block length (the real-input wrapper does the padding)
    return launch_fft_real_batch(d_taps, static_cast<size_t>(ntaps), static_cast<size_t>(ntaps), d_hspec_f2, log2n, 1,
                                 DSP_GPU_FORWARD, 0, s);
}

namespace {
// Block pairs from which the one-wave-per-pair kernel takes over.  DSP_GPU_OS_WAVE_MIN_TILES overrides it (read per
// call): 1 sends every 4096-point-block call to the wave kernel, a huge value none -- for A/B runs and for parity
// tests that exercise both kernels on small inputs.
size_t os_wave_min_tiles() {
    const char *e = getenv("DSP_GPU_OS_WAVE_MIN_TILES");
    if (e && *e) return static_cast<size_t>(strtoull(e, nullptr, 10));
    return kOsWaveMinTiles;
}

// Blocks per channel from which it does.  The wave kernel takes the pairs that lie wholly inside one channel and hold neither
// its first nor its last block; the others go to the workgroup kernel, two workgroups per channel.  On channels of a few blocks
// most pairs are such edge pairs and that split costs more than it gains -- 65,536 x 4,096 (one block per channel): 1.82 ms split
// against 0.47 ms on the workgroup kernel alone; 8,192 x 32,768 (10 blocks): 0.55 / 0.49; 4,096 x 65,536 (19 blocks): 0.47 /
// 0.53 (scripts/fir_mid_channels.py, profiles/r03_fir_mid_channels.txt).  A DSP_GPU_OS_WAVE_MIN_TILES override lifts this too.
// zeros over the first `count` samples of `rows` rows that lie `stride` samples apart: one wave per row
__global__ __launch_bounds__(256) void zero_heads_kernel(float *__restrict__ y, size_t stride, unsigned count, size_t rows) {
    const size_t row = static_cast<size_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float *p = y + row * stride;
    for (unsigned i = threadIdx.x & 63; i < count; i += 64) p[i] = 0.0f;
}

size_t os_wave_min_blocks_per_channel() {
    const char *e = getenv("DSP_GPU_OS_WAVE_MIN_TILES");
    return (e && *e) ? 1 : 14;
}

int launch_os(const float *x, float *y, size_t channels, size_t n, const c32 *d_hspec, int ntaps, const float *hist,
              int zero_head, const OsExtra &ex, int log2n, hipStream_t s) {
    const long long order = ntaps - 1;
    const long long kOsN = 1ll << log2n;
    const long long nv = static_cast<long long>(n) + ((hist || ex.zero_hist) ? order : 0);  // length of the (virtual) signal
    const long long valid = nv > order ? nv - order : 0;
    // block 0 delivers N - order outputs, every later block L (block 0 also zero-fills y[0, order))
    auto blocks_at = [&](long long hop_v) { return valid > kOsN - order ? 1 + (valid - (kOsN - order) + hop_v - 1) / hop_v : 1ll; };
    // One-shot calls on several channels of up to some tens of blocks each (where many pairs hold a channel's first or last
    // block, which the wave kernel leaves to the workgroup kernel): as ONE signal of channels * n samples -- an output with a full window behind it never reaches across
    // its channel's start -- and then zeros over the first `order` outputs of every channel but the first (filter.h:126,133; those
    // would have seen the previous channel's tail).  Costs order / n more written bytes and buys the long-signal kernels:
    // 511 taps, 2^28 samples: 87,381 x 3,072 0.46 -> 0.60 of HBM, 32,768 x 8,192 0.41 -> 0.62, 4,096 x 65,536 0.56 -> 0.64
    // (profiles/r03_fir_mid_channels.txt).  DSP_GPU_FIR_NO_FLATTEN=1 keeps the channels apart.
    if (log2n == 12 && channels >= 2 && !hist && !ex.accumulate && !ex.zero_hist && !ex.edge_pairs && ex.stride == n && ex.yoff == 0 &&
        zero_head == order && n >= 1536 && n < 262144 && channels * n < 0xffffffffull && !getenv("DSP_GPU_FIR_NO_FLATTEN")) {
        const size_t total = channels * n;
        if (int rc = launch_os(x, y, 1, total, d_hspec, ntaps, nullptr, zero_head, OsExtra{total, 0, 0, 0, 0}, log2n, s)) return rc;
        if (order > 0) {  // (hipMemset2DAsync takes ~1.8 ns per row whatever its length: 120 us for 65,535 rows of 2 KB)
            const size_t heads = channels - 1;
            hipLaunchKernelGGL(zero_heads_kernel, dim3(static_cast<unsigned>((heads + 3) / 4)), dim3(256), 0, s, y + n, n,
                               static_cast<unsigned>(order), heads);
            return check_launch("zero_heads_kernel");
        }
        return DSP_GPU_OK;
    }
    // hop between blocks: N - order at most
    long long L = kOsN - order;
    // The one-wave-per-pair kernel wants a hop of whole 1-KB rows (256 samples; its 16-byte stores, the rows it shares between
    // pairs) and is worth up to 1/16 more blocks for it: 127 taps, 1,024 x 2^20: 1.80 ms on the workgroup kernel at hop 3968,
    // 1.6 ms on the wave kernel at hop 3840 (scripts/fir_crossover.py).
    bool wave = false;
    if (log2n == 12 && !hist && !ex.accumulate && !ex.zero_hist && ex.stride == n && ex.yoff == 0 && (n & 3) == 0 &&
        (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(x) & 3) == 0) {
        const long long L256 = L & ~255ll;
        if (L256 >= 2048 && L256 <= 3840 && (L - L256) * 16 <= L) {
            const long long nb = blocks_at(L256);
            const size_t tl = (channels * static_cast<size_t>(nb) + 1) / 2;
            if (tl >= os_wave_min_tiles() && static_cast<size_t>(nb) >= os_wave_min_blocks_per_channel()) {
                wave = true;
                L = L256;
            }
        }
    }
    // otherwise one-shot calls round it down to a multiple of 64 samples (256 if that is as cheap) when that costs at most 1/32
    // of it (block starts, and with them every wave's loads and stores, then sit on 256-byte boundaries)
    if (!wave && !hist && !ex.zero_hist) {
        if ((L & 255) * 32 <= L && (L & ~255ll) >= 256) L &= ~255ll;
        else if ((L & 63) * 32 <= L && (L & ~63ll) >= 64) L &= ~63ll;
    }
    const long long nblocks = blocks_at(L);
    const int hop = static_cast<int>(L);
    const unsigned bpc = static_cast<unsigned>(nblocks);
    const size_t total_blocks = channels * static_cast<size_t>(nblocks);
    const size_t tiles = (total_blocks + 1) / 2;  // two real blocks per complex transform, across channels if need be
    if (wave) {
        static const int cus = [] {
            int dev = 0, v = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
            return v;
        }();
        // DSP_GPU_OS_WAVE_FORM=agpr: round 2's kernel (one wave per SIMD, WIDE transpose, prefetch into AGPRs by hand);
        // default: two waves per SIMD, QUAD transpose, sequential runs (fir_os_wave2_kernel).  Identical bits.
        const char *form = getenv("DSP_GPU_OS_WAVE_FORM");
        if (!(form && form[0] == 'a')) {
            // runs: whole rounds of one 8-wave workgroup per CU (a last round that fills a quarter of the chip costs as much
            // as a full one), about 32 pairs to a run
            const size_t per_round = static_cast<size_t>(cus) * kOsWave2Waves;
            size_t rounds = (tiles + per_round * 16) / (per_round * 32);
            if (rounds < 1) rounds = 1;
            unsigned run_len = static_cast<unsigned>((tiles + per_round * rounds - 1) / (per_round * rounds));
            if (const char *e = getenv("DSP_GPU_OS_WAVE_RUN"))  // tests: runs of several pairs on inputs too small to get them
                if (atoi(e) > 0) run_len = static_cast<unsigned>(atoi(e));
            const size_t runs = (tiles + run_len - 1) / run_len;
            size_t runs_per_xcd = ((runs + 7) / 8 + kOsWave2Waves - 1) / kOsWave2Waves * kOsWave2Waves;
            dim3 grid2(static_cast<unsigned>(runs_per_xcd / kOsWave2Waves * 8));
            // Default: the pairs are DRAWN, one workgroup per CU, from eight counters (one per XCD, 128 bytes apart) that a
            // memset on the caller's stream clears in front of the launch.  DSP_GPU_OS_WAVE_DRAW=0: dealt out in runs as until
            // late round 3; =<k>: k consecutive pairs per draw (inside a draw the shared rows stay in registers).
            unsigned *d_draw = nullptr;
            unsigned chunk2 = 0;
            int draw_len = getenv("DSP_GPU_OS_WAVE_RUN") ? 0 : 1;  // (a test that sets the run length means the dealt form)
            if (const char *e = getenv("DSP_GPU_OS_WAVE_DRAW")) draw_len = atoi(e);
            if (draw_len > 0 && total_blocks < 0xffffffffull) {
                void *p = nullptr;
                if (int rc = workspace(WS_COUNTERS, 8 * 32 * sizeof(unsigned), &p, s)) return rc;
                d_draw = static_cast<unsigned *>(p);
                DSPGPU_HIP(hipMemsetAsync(d_draw, 0, 8 * 32 * sizeof(unsigned), s));
                chunk2 = static_cast<unsigned>((tiles + 7) / 8);
                run_len = static_cast<unsigned>(draw_len);
                runs_per_xcd = 0;
                grid2 = dim3(static_cast<unsigned>((cus + 7) / 8 * 8));
            }
#define DSPGPU_OSW2_CASE(I0)                                                                                            \
    case I0: {                                                                                                          \
        if (int rc = allow_full_lds(reinterpret_cast<const void *>(fir_os_wave2_kernel<I0>))) return rc;                \
        hipLaunchKernelGGL(fir_os_wave2_kernel<I0>, grid2, dim3(kOsWave2Threads), kOsWave2LdsBytes, s, x, y, n, d_hspec, bpc, \
                           total_blocks, tiles, run_len, runs_per_xcd, d_draw, chunk2);                                 \
    } break
            switch (static_cast<int>((kOsN - L) / 256)) {
                DSPGPU_OSW2_CASE(1); DSPGPU_OSW2_CASE(2); DSPGPU_OSW2_CASE(3); DSPGPU_OSW2_CASE(4);
                DSPGPU_OSW2_CASE(5); DSPGPU_OSW2_CASE(6); DSPGPU_OSW2_CASE(7); DSPGPU_OSW2_CASE(8);
            default: return fail(DSP_GPU_EINVAL, "fir_os_wave2_kernel: hop %lld", L);
            }
#undef DSPGPU_OSW2_CASE
            if (int rc = check_launch("fir_os_wave2_kernel")) return rc;
        } else {
        // long signals: one wave per block pair, persistent workgroups (one per CU: four waves, one per SIMD)
        const size_t chunkw = (tiles + 7) / 8;
        size_t wgs = (static_cast<size_t>(cus) + 7) / 8 * 8;
        wgs = std::min(wgs, (chunkw + kOsWaveWaves - 1) / kOsWaveWaves * 8);
        const dim3 wgrid(static_cast<unsigned>(wgs));
#define DSPGPU_OSW_CASE(I0)                                                                                              \
    case I0: {                                                                                                          \
        if (int rc = allow_full_lds(reinterpret_cast<const void *>(fir_os_wave_kernel<I0>))) return rc;                 \
        hipLaunchKernelGGL(fir_os_wave_kernel<I0>, wgrid, dim3(kOsWaveThreads), kOsWaveLdsBytes, s, x, y, n, d_hspec, ntaps, \
                           bpc, total_blocks, tiles, chunkw);                                                           \
    } break
        switch (static_cast<int>((kOsN - L) / 256)) {
            DSPGPU_OSW_CASE(1); DSPGPU_OSW_CASE(2); DSPGPU_OSW_CASE(3); DSPGPU_OSW_CASE(4);
            DSPGPU_OSW_CASE(5); DSPGPU_OSW_CASE(6); DSPGPU_OSW_CASE(7); DSPGPU_OSW_CASE(8);
        default: return fail(DSP_GPU_EINVAL, "fir_os_wave_kernel: hop %lld", L);
        }
#undef DSPGPU_OSW_CASE
        if (int rc = check_launch("fir_os_wave_kernel")) return rc;
        }
        // the pairs it left out -- those holding a channel's first or last block -- on the workgroup kernel: two per channel
        OsExtra edge = ex;
        edge.edge_pairs = 1;
        if (2 * channels > 0x7fffffffull) return fail(DSP_GPU_EUNSUPPORTED, "overlap-save FIR: grid too large");
        hipLaunchKernelGGL(fir_overlap_save_kernel<12>, dim3(static_cast<unsigned>(2 * channels)), dim3(256),
                           (4096 + 4096 / 16) * sizeof(c32), s, x, y, n, d_hspec, ntaps, bpc, total_blocks, tiles, tiles,
                           (tiles + 7) / 8, hist, zero_head, edge, hop);
        return check_launch("fir_overlap_save_kernel (edge pairs)");
    }
    const size_t per_wg = kOsN < 4096 ? static_cast<size_t>(4096 / kOsN) : 1;  // transforms per workgroup
    const size_t wg_tiles = (tiles + per_wg - 1) / per_wg;
    const size_t chunk = (wg_tiles + 7) / 8;
    if (8 * chunk > 0x7fffffffull) return fail(DSP_GPU_EUNSUPPORTED, "overlap-save FIR: grid too large");
    const size_t points = static_cast<size_t>(kOsN < 4096 ? 4096 : kOsN);
    const size_t lds_bytes = (points + points / 16) * sizeof(c32);
    const dim3 grid(static_cast<unsigned>(8 * chunk));
#define DSPGPU_OS_LAUNCH(LOG2N, THREADS)                                                                               \
    hipLaunchKernelGGL(fir_overlap_save_kernel<LOG2N>, grid, dim3(THREADS), lds_bytes, s, x, y, n, d_hspec, ntaps, bpc, \
                       total_blocks, tiles, wg_tiles, chunk, hist, zero_head, ex, hop)
    switch (log2n) {
    case 8: DSPGPU_OS_LAUNCH(8, 256); break;
    case 9: DSPGPU_OS_LAUNCH(9, 256); break;
    case 10: DSPGPU_OS_LAUNCH(10, 256); break;
    case 11: DSPGPU_OS_LAUNCH(11, 256); break;
    case 12: DSPGPU_OS_LAUNCH(12, 256); break;
    case 14:
        if (int rc = allow_full_lds(reinterpret_cast<const void *>(fir_overlap_save_kernel<14>))) return rc;
        DSPGPU_OS_LAUNCH(14, 1024);
        break;
    default: return fail(DSP_GPU_EINVAL, "overlap-save FIR: no kernel for 2^%d-point blocks", log2n);
    }
#undef DSPGPU_OS_LAUNCH
    return check_launch("fir_overlap_save_kernel");
}
}  // namespace

int launch_fir_overlap_save(const float *x, float *y, size_t channels, size_t n, const float2 *d_hspec_f2, int ntaps,
                            const float *hist, int zero_head, int log2n, hipStream_t s) {
    return launch_os(x, y, channels, n, reinterpret_cast<const c32 *>(d_hspec_f2), ntaps, hist, zero_head,
                     OsExtra{n, 0, 0, 0, 0}, log2n, s);
}

// One segment of a partitioned long filter (taps [seg0, seg0 + seg_len) of it): y[.][seg0 + j] (+)= sum_i h_seg[i] x[.][j - i]
// for j < n - seg0, every output kept (the samples before x[0] are zeros).
int launch_fir_os_segment(const float *x, float *y, size_t channels, size_t n, const float2 *d_hspec_f2, int seg_len,
                          size_t seg0, int accumulate, int log2n, hipStream_t s) {
    if (seg0 >= n) return DSP_GPU_OK;
    return launch_os(x, y, channels, n - seg0, reinterpret_cast<const c32 *>(d_hspec_f2), seg_len, nullptr, 0,
                     OsExtra{n, seg0, accumulate, 1, 0}, log2n, s);
}

namespace {
template <typename T>
int launch_moving_average_t(const T *x, T *y, size_t channels, size_t n, int npoints, hipStream_t s, unsigned period = 0) {
    if (npoints > 8192) return fail(DSP_GPU_EUNSUPPORTED, "moving average: npoints %d > 8192", npoints);
    constexpr int kVec = Smp<T>::kVec;
    const int window = 2 * (npoints / 2) + 1;
    // Many short channels: as one signal of channels * n samples in full tiles (PERIODIC above).  2^29 samples, 5 / 511 points
    // (scripts/mavg_mid_channels.py, profiles/r03_mavg_shapes.txt): 512-sample channels 0.37 / 0.41 -> 0.52 / 0.72 of HBM, 1,024:
    // 0.52 / 0.60 -> 0.60 / 0.67; equal at 1,536, and from there the per-channel tiles (one, two, four or eight outputs per thread
    // fewer) win: 2,048: 0.57 / 0.69 against 0.52 / 0.60.  Also channels that leave their second 4096-output tile mostly empty:
    // 4,500 samples 0.48 / 0.44 -> 0.61, 5,000: 0.53 / 0.48 -> 0.61, 6,000: 0.59 / 0.56 -> 0.61 / 0.59 (three tiles and more: not).
    // DSP_GPU_MAVG_NO_FLATTEN=1 (A/B runs, tests) keeps a workgroup per channel.
    if (period == 0 && channels >= 2 && (n < 1536 || (n > 4096 && n <= 6144)) && static_cast<size_t>(window) <= n &&
        channels * n / 4096 < 0x7fffffffull && !getenv("DSP_GPU_MAVG_NO_FLATTEN"))
        return launch_moving_average_t<T>(x, y, 1, channels * n, npoints, s, static_cast<unsigned>(n));
    for (; channels > kMaxGridY; channels -= kMaxGridY, x += kMaxGridY * n, y += kMaxGridY * n)  // grid.y is 16 bits
        if (int rc = launch_moving_average_t<T>(x, y, kMaxGridY, n, npoints, s)) return rc;
    // outputs per thread: the full 16 unless the whole channel fits a smaller tile (many short channels would
    // otherwise stage and scan mostly padding)
    int per = kMaPerMax;
    while (per > 1 && static_cast<size_t>(kThreads) * (per / 2) >= n) per /= 2;
    if (const char *e = getenv("DSP_GPU_MAVG_PER"))  // A/B runs: outputs per thread (1, 2, 4, 8, 16)
        if (atoi(e) >= 1 && atoi(e) <= per) per = atoi(e);
    const int tile = kThreads * per;
    // staged samples: tile + window, + kVec - 1 for the aligned start, + 1 so index `up` of the last output exists
    const int span = ((tile + window + kVec + 15) / 16) * 16;
    const int chunks = span / 16;
    const int cpt = (chunks + kThreads - 1) / kThreads;
    const size_t lds_bytes = static_cast<size_t>(span + (span / 16) * kVec) * sizeof(T) + static_cast<size_t>(chunks + 4 * ((cpt + 1) & ~1) + tile / 16 + 1) * sizeof(double) +
                             [] { const char *e = getenv("DSP_GPU_MAVG_LDS_PAD"); return e ? static_cast<size_t>(atoi(e)) : size_t(0); }();  // A/B runs: fewer workgroups per CU
    const size_t blocks = (n + tile - 1) / tile;
    const double inv = 1.0 / static_cast<double>(npoints);
    // loads of an interior tile: the window halo (shared with the neighbouring tiles, which run on the same XCD at about the
    // same time) through the cache, the rest streaming.  Same speed as all-streaming, 4 % less HBM traffic
    // (profiles/r03_mavg_sweep.txt).  DSP_GPU_MAVG_LOADS=0 all streaming, 2 all cached (A/B runs).
    const int ld_mode = [] { const char *e = getenv("DSP_GPU_MAVG_LOADS"); return e ? atoi(e) : 1; }();
    const int halo = ((window + 15) / 16) * 16 + 16;
    const bool aligned = (n % kVec == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0);
    void (*kern)(const T *, T *, size_t, int, int, int, int, int, double, int, int, unsigned) = nullptr;
#define DSPGPU_MA_PICK(PER)                                                                                              \
    kern = period ? (aligned ? moving_average_kernel<true, PER, T, true> : moving_average_kernel<false, PER, T, true>)  \
                  : (aligned ? moving_average_kernel<true, PER, T> : moving_average_kernel<false, PER, T>)
    switch (per) {
    case 1: DSPGPU_MA_PICK(1); break;
    case 2: DSPGPU_MA_PICK(2); break;
    case 4: DSPGPU_MA_PICK(4); break;
    case 8: DSPGPU_MA_PICK(8); break;
    default: DSPGPU_MA_PICK(16); break;
    }
#undef DSPGPU_MA_PICK
    if (lds_bytes > 64 * 1024)
        if (int rc = allow_full_lds(reinterpret_cast<const void *>(kern))) return rc;
    // DSP_GPU_MAVG_SEQ=1 (opt-in, read per call): on long channels the interior tiles -- window wholly inside the channel --
    // go to the sequential kernel in runs of 16 (every sample read once: HBM traffic 1.04x -> 1.01x of the algorithmic 8 B per
    // sample), the first and last few stay here.  Not the default: it moves fewer bytes but runs 9-13 % slower on C3's shape
    // (1.61-1.66 ms against 1.47: profiles/r03_mavg_sequential_ab.txt) -- the tile loop serialises what independent
    // one-tile workgroups overlap.
    const char *seq_env = getenv("DSP_GPU_MAVG_SEQ");
    const bool seq_on = seq_env && seq_env[0] == '1';
    if (seq_on && !period && aligned && per == kMaPerMax && window + kVec <= tile / 2) {
        const long long mid = npoints / 2;
        const long long lo = (mid + kVec + tile - 1) / tile;  // first tile whose staged span starts inside the channel
        // last tile t with t*tile - mid - (kVec-1) + span <= n (the span ends inside) and (t+1)*tile <= n - mid (full windows)
        const long long hi_a = (static_cast<long long>(n) - span + mid) / tile;
        const long long hi_b = (static_cast<long long>(n) - mid) / tile - 1;
        const long long hi = std::min(hi_a, hi_b);
        if (hi - lo + 1 >= 8) {
            const int tile_n = static_cast<int>(hi - lo + 1), run = 16;
            const unsigned runs = static_cast<unsigned>((tile_n + run - 1) / run);
            const size_t lds_seq = static_cast<size_t>(span + (span / 16) * kVec) * sizeof(T) + static_cast<size_t>(chunks + 1 + 4 * cpt) * sizeof(double);
            auto kseq = moving_average_seq_kernel<T>;
            if (lds_seq > 64 * 1024)
                if (int rc = allow_full_lds(reinterpret_cast<const void *>(kseq))) return rc;
            hipLaunchKernelGGL(kseq, dim3(runs, channels), dim3(kThreads), lds_seq, s, x, y, n, npoints, span, static_cast<int>(lo), tile_n, run);
            if (int rc = check_launch("moving_average_seq_kernel")) return rc;
            const size_t edge = blocks - static_cast<size_t>(tile_n);
            if (edge) hipLaunchKernelGGL(kern, dim3(edge, channels), dim3(kThreads), lds_bytes, s, x, y, n, npoints, span, cpt, static_cast<int>(lo), tile_n, inv, ld_mode, halo, 0u);
            return check_launch("moving_average_kernel (edge tiles)");
        }
    }
    hipLaunchKernelGGL(kern, dim3(blocks, channels), dim3(kThreads), lds_bytes, s, x, y, n, npoints, span, cpt, 0, 0, inv, ld_mode, halo, period);
    return check_launch("moving_average_kernel");
}
}  // namespace

int launch_moving_average(const float *x, float *y, size_t channels, size_t n, int npoints, hipStream_t s) {
    return launch_moving_average_t<float>(x, y, channels, n, npoints, s);
}
int launch_moving_average_f64(const double *x, double *y, size_t channels, size_t n, int npoints, hipStream_t s) {
    return launch_moving_average_t<double>(x, y, channels, n, npoints, s);
}

}  // namespace dspgpu
