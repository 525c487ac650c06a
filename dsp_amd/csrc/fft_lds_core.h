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
template <int R, int NS, int LOG2N>
DSP_DEV void lds_exchange(c32 (&r)[kPoints], c32 *lds, int lt, int base) {
    constexpr int T = (1 << LOG2N) / kPoints;
    constexpr int B = kPoints / R;
#pragma unroll
    for (int s = 0; s < B; ++s) {
        const int j = lt + T * s;
#pragma unroll
        for (int u = 0; u < R; ++u) lds[padded(base + stockham_dest<R, NS>(j, u))] = r[s + u * B];
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < kPoints; ++m) r[m] = lds[padded(base + lt + T * m)];
}

// All passes of one transform.  In: r[m] = x[lt + T*m].  Out: r[m] = X[lt + T*m].
// The caller must make sure no thread still reads `lds` from an earlier use (barrier).
template <int LOG2N, int SIGN>
DSP_DEV void fft_passes(c32 (&r)[kPoints], c32 *lds, int lt, int base) {
    constexpr int R0 = radix_at(LOG2N, 0), R1 = radix_at(LOG2N, 1), R2 = radix_at(LOG2N, 2);
    stockham_pass<R0, 1, LOG2N, SIGN, kPoints>(r, lt);
    if constexpr (R1 > 1) {
        lds_exchange<R0, 1, LOG2N>(r, lds, lt, base);
        stockham_pass<R1, R0, LOG2N, SIGN, kPoints>(r, lt);
        if constexpr (R2 > 1) {
            __syncthreads();  // every thread has finished reading the image of exchange 1
            lds_exchange<R1, R0, LOG2N>(r, lds, lt, base);
            stockham_pass<R2, R0 * R1, LOG2N, SIGN, kPoints>(r, lt);
        }
    }
}

}  // namespace dspgpu
