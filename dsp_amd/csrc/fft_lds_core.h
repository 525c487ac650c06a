// fft_lds_core.h -- the workgroup-level 4096-point-tile FFT used by the batched FFT kernel
// and by the fused overlap-save FIR kernel: 256 threads x 16 register-resident points, radix-16
// Stockham passes, points exchanged between passes through one padded LDS image.
#pragma once

#include "fft_device.h"

namespace dspgpu {

constexpr int kThreads = 256;
constexpr int kPoints = 16;                    // points per thread
constexpr int kTile = kThreads * kPoints;      // 4096 points per workgroup
constexpr int kLdsElems = kTile + kTile / 16;  // one pad element per 16: conflict-free radix-16 scatter

DSP_DEV int padded(int i) { return i + (i >> 4); }

// Scatter the outputs of the pass (radix R, sub-length NS) to their Stockham positions, then
// gather r[m] = element lt + T*m for the next pass.  `base` = first point of this thread's
// transform inside the workgroup tile.
// Where the padded index is linear in the register slot -- (x + 16k) >> 4 == (x >> 4) + k -- it is written as ONE
// base plus compile-time offsets, so the slots become immediate offsets of ds_read/ds_write instead of sixteen
// separately computed address registers (the compiler does not find this: ~100 integer VALU instructions per
// 4096-point transform; the overlap-save FIR, which is VALU-bound, gained 7 %).
template <int R, int NS, int LOG2N>
DSP_DEV void lds_exchange(c32 (&r)[kPoints], c32 *lds, int lt, int base) {
    constexpr int T = (1 << LOG2N) / kPoints;
    constexpr int B = kPoints / R;
#pragma unroll
    for (int s = 0; s < B; ++s) {
        const int j = lt + T * s;
        if constexpr (NS % 16 == 0) {
            c32 *w = lds + padded(base + stockham_dest<R, NS>(j, 0));
#pragma unroll
            for (int u = 0; u < R; ++u) w[u * (NS + NS / 16)] = r[s + u * B];
        } else if constexpr (NS == 1 && R == 16) {  // first exchange: 16 j + u, and base is a multiple of 16: one padding group
            c32 *w = lds + padded(base + 16 * j);
#pragma unroll
            for (int u = 0; u < R; ++u) w[u] = r[s + u * B];
        } else {
#pragma unroll
            for (int u = 0; u < R; ++u) lds[padded(base + stockham_dest<R, NS>(j, u))] = r[s + u * B];
        }
    }
    __syncthreads();
    if constexpr (T % 16 == 0) {
        const c32 *g = lds + padded(base + lt);
#pragma unroll
        for (int m = 0; m < kPoints; ++m) r[m] = g[m * (T + T / 16)];
    } else {
#pragma unroll
        for (int m = 0; m < kPoints; ++m) r[m] = lds[padded(base + lt + T * m)];
    }
}

// All passes of one transform.  In: r[m] = x[lt + T*m].  Out: r[m] = X[lt + T*m].
// The caller must make sure no thread still reads `lds` from an earlier use (barrier).
template <int LOG2N, int SIGN>
DSP_DEV void fft_passes(c32 (&r)[kPoints], c32 *lds, int lt, int base) {
    constexpr int R0 = radix_at(LOG2N, 0), R1 = radix_at(LOG2N, 1), R2 = radix_at(LOG2N, 2), R3 = radix_at(LOG2N, 3);
    stockham_pass<R0, 1, LOG2N, SIGN, kPoints>(r, lt);
    if constexpr (R1 > 1) {
        lds_exchange<R0, 1, LOG2N>(r, lds, lt, base);
        stockham_pass<R1, R0, LOG2N, SIGN, kPoints>(r, lt);
        if constexpr (R2 > 1) {
            __syncthreads();  // every thread has finished reading the image of exchange 1
            lds_exchange<R1, R0, LOG2N>(r, lds, lt, base);
            stockham_pass<R2, R0 * R1, LOG2N, SIGN, kPoints>(r, lt);
            if constexpr (R3 > 1) {  // 2^13 and 2^14 points: one workgroup of N/16 threads (fft_big_kernel)
                __syncthreads();
                lds_exchange<R2, R0 * R1, LOG2N>(r, lds, lt, base);
                stockham_pass<R3, R0 * R1 * R2, LOG2N, SIGN, kPoints>(r, lt);
            }
        }
    }
}

// ---- the same with 32 points per thread (fft_big_kernel: 2^13 / 2^14 points in one workgroup) -------------------------
// Radix 32 first: 2^14 = 32 * 32 * 16 and 2^13 = 32 * 16 * 16 take two exchanges where radix 16 takes three, and an
// exchange of the whole transform through LDS is what these kernels pay most for.  The image is padded by one element
// per 32 so that the radix-32 scatter strides by 33 elements.
DSP_DEV int padded32(int i) { return i + (i >> 5); }

template <int R, int NS, int LOG2N>
DSP_DEV void lds_exchange32(c32 (&r)[32], c32 *lds, int lt) {
    constexpr int P = 32, T = (1 << LOG2N) / P, B = P / R;
    static_assert(T % 32 == 0 && ((NS == 1 && R == 32) || NS % 32 == 0), "exchange shape");
#pragma unroll
    for (int s = 0; s < B; ++s) {
        const int j = lt + T * s;
        // padded32(dest + u NS) is linear in u: 32 j + u -> 33 j + u (NS = 1), x + 32 k -> padded32(x) + 33 k otherwise
        c32 *w = lds + padded32(stockham_dest<R, NS>(j, 0));
#pragma unroll
        for (int u = 0; u < R; ++u) w[NS == 1 ? u : u * (NS + NS / 32)] = r[s + u * B];
    }
    __syncthreads();
    const c32 *g = lds + padded32(lt);
#pragma unroll
    for (int m = 0; m < P; ++m) r[m] = g[m * (T + T / 32)];
}

// In: r[m] = x[lt + T*m], T = N/32.  Out: r[m] = X[lt + T*m].  LOG2N = 13 (32 * 16 * 16) or 14 (32 * 32 * 16).
template <int LOG2N, int SIGN>
DSP_DEV void fft_passes32(c32 (&r)[32], c32 *lds, int lt) {
    static_assert(LOG2N == 13 || LOG2N == 14, "size");
    constexpr int R1 = LOG2N == 14 ? 32 : 16;
    stockham_pass<32, 1, LOG2N, SIGN, 32>(r, lt);
    lds_exchange32<32, 1, LOG2N>(r, lds, lt);
    stockham_pass<R1, 32, LOG2N, SIGN, 32>(r, lt);
    __syncthreads();  // every thread has finished reading the image of exchange 1
    lds_exchange32<R1, 32, LOG2N>(r, lds, lt);
    stockham_pass<16, 32 * R1, LOG2N, SIGN, 32>(r, lt);
}

}  // namespace dspgpu
