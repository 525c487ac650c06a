// fft_wave_core.h -- a 4096-point transform owned by ONE wavefront (gfx950).
//
// 64 lanes x 64 register-resident points: 4096 = 64 x 64, two radix-64 passes done entirely in a lane's
// registers, ONE twiddled step between them and ONE 64x64 transpose through a wave-private LDS image.  Against
// the workgroup-level radix-16 transform (fft_lds_core.h: three passes, two twiddled steps, two exchanges behind
// workgroup barriers) this is ~20 % fewer vector instructions per point (the twiddle work halves), half the LDS
// traffic and no barrier at all -- a wave never waits for another wave.  The price is registers: 128 VGPRs of
// data, so two waves per SIMD.  Used by the overlap-save FIR (fir_kernels.hip), which is bound by exactly the
// instruction count and the exchanges.  Stands in for the butterfly loop of dsp::fft0 (reference fft.h:40-54);
// SIGN = +1 is the reference's FORWARD (e^{+i 2 pi k n / N}, fft.h:22-29), nothing normalises.
#pragma once

#include "fft_device.h"

namespace dspgpu {

constexpr int kWavePoints = 64;                 // points per lane
constexpr int kWaveN = 64 * kWavePoints;        // 4096 points per wave
constexpr int kWaveRow = 65;                    // dwords per row of the transpose image: odd, so both the row-wise
                                                // stores and the column-wise loads hit 32 distinct banks per half wave
constexpr int kWaveLdsFloats = 64 * kWaveRow;   // one fp32 plane (re and im cross one after the other): 16,640 B

// multiply by exp(SIGN * 2 pi i * q / 64); q is a constant once the caller's loops are unrolled
template <int SIGN>
DSP_DEV c32 mul_w64(c32 a, int q) {
    q &= 63;
    if (q == 0) return a;
    if (q == 16) return mul_i<SIGN>(a);
    if (q == 32) return -a;
    if (q == 48) return mul_i<-SIGN>(a);
    // The constant is materialised in a scalar register pair right here (the empty volatile asm cannot be hoisted):
    // left to itself the compiler keeps all ~60 distinct constants of the two directions live across the whole
    // persistent loop and spills scalar registers.
    c32 k = c32{cos64(q), SIGN * sin64(q)};
    asm volatile("" : "+s"(k));
    return cmul_k(a, k);
}

// 64 = 16 x 4: n = 4*n1 + n2, k = k1 + 16*k2.  Natural order in and out.
template <int SIGN>
struct Dft<64, SIGN> {
    static DSP_DEV void run(c32 (&v)[64]) {
        run(v, [](int) {});
    }
    // `hook(0..7)` runs after each of the four 16-point sub-transforms and after every fourth 4-point one: callers hang
    // other work there (the overlap-save
    // kernel issues its prefetch loads a few at a time, so that a full memory queue never holds up the arithmetic long)
    template <typename Hook>
    static DSP_DEV void run(c32 (&v)[64], Hook hook) {
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) {  // A[k1][n2] = sum_n1 v[4 n1 + n2] W16^{n1 k1}, left at v[4 k1 + n2]
            c32 t[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) t[i] = v[4 * i + n2];
            Dft<16, SIGN>::run(t);
#pragma unroll
            for (int i = 0; i < 16; ++i) v[4 * i + n2] = t[i];
            hook(n2);
        }
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) {
#pragma unroll
            for (int n2 = 1; n2 < 4; ++n2) v[4 * k1 + n2] = mul_w64<SIGN>(v[4 * k1 + n2], n2 * k1);
        }
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {
            dft4<SIGN>(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
            if ((k1 & 3) == 3) hook(4 + (k1 >> 2));
        }
        c32 o[64];  // X[k1 + 16 k2] sits at v[4 k1 + k2]: a renaming, no instruction
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) o[k1 + 16 * k2] = v[4 * k1 + k2];
        }
#pragma unroll
        for (int i = 0; i < 64; ++i) v[i] = o[i];
    }
};

// Lane l holds r[j] = M[j][l]; afterwards r[j] = M[l][j].  A wave's LDS instructions execute in order, so the loads
// see the stores of every lane without any hardware barrier; the wavefront-scope fences only keep the compiler from
// reordering memory accesses across them.
// Two forms.  WIDE: `img` is a wave-private image of 64 x 65 complex values (33,280 B) and the points cross whole, one
// store and one load phase of 8-byte accesses (row stride 65 complex = 130 dwords: lane l reads dword bank
// (2l + 2j) mod 64, distinct over each half wave).  Otherwise `img` is one fp32 plane (16,640 B) that the real and the
// imaginary parts cross one after the other -- half the LDS, twice the dependent phases.
constexpr int kWaveLdsWideBytes = 64 * kWaveRow * 8;

// WRITTEN (WIDE only): the caller has already stored r[j] at row j, column `lane` of the image
template <bool WIDE, bool WRITTEN = false>
DSP_DEV void wave_transpose(c32 (&r)[kWavePoints], float *img, int lane) {
    if constexpr (WIDE) {
        c32 *im2 = reinterpret_cast<c32 *>(img);
        const c32 *rd = im2 + lane * kWaveRow;
        if constexpr (!WRITTEN) {
            c32 *wr = im2 + lane;
#pragma unroll
            for (int j = 0; j < kWavePoints; ++j) wr[j * kWaveRow] = r[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // in the order the next pass consumes them (its first 16-point sub-transform takes elements 0, 4, 8, ...), so that
        // it can start while the rest are still on their way
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) r[4 * i + n2] = rd[4 * i + n2];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        return;
    }
    float *wr = img + lane;             // row j, column lane:  wr[j * kWaveRow]
    float *rd = img + lane * kWaveRow;  // row lane, column j:  rd[j]
#pragma unroll
    for (int j = 0; j < kWavePoints; ++j) wr[j * kWaveRow] = r[j].x;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < kWavePoints; ++j) r[j].x = rd[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < kWavePoints; ++j) wr[j * kWaveRow] = r[j].y;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < kWavePoints; ++j) r[j].y = rd[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// QUAD: the same 64 x 64 transpose in HALF the LDS (16,896 B per wave), so that a CU's 160 KB hold the images of eight
// waves -- two per SIMD.  The matrix is cut into four 32 x 32 blocks.  v_permlane32_swap_b32 (gfx950) first trades the
// off-diagonal blocks between the two halves of the wave (lanes 32..63 of r[jj] <-> lanes 0..31 of r[32 + jj]); what is
// left is a 32 x 32 transpose inside each (lane half, register half), done for the register halves one after the other
// through one 32 x 33 image per lane half.  Same 64 stores + 64 loads of 8 bytes as the WIDE form, 64 swaps on top.
constexpr int kWaveQuadRow = 33;
constexpr int kWaveLdsQuadBytes = 2 * 32 * kWaveQuadRow * 8;

// lanes 32..63 of a <-> lanes 0..31 of b
DSP_DEV void swap_lane_halves(c32 &a, c32 &b) {
    // (asm, not __builtin_amdgcn_permlane32_swap: with the builtin called once per component hipcc 7.2 emitted the swap
    // of the real parts only and copied its result into the imaginary parts)
    float ax = a.x, ay = a.y, bx = b.x, by = b.y;
    asm("v_permlane32_swap_b32 %0, %2\n\t"
        "v_permlane32_swap_b32 %1, %3"
        : "+v"(ax), "+v"(ay), "+v"(bx), "+v"(by));
    a = mk(ax, ay);
    b = mk(bx, by);
}

DSP_DEV void wave_transpose_quad(c32 (&r)[kWavePoints], float *img, int lane) {
    c32 *q = reinterpret_cast<c32 *>(img) + (lane >> 5) * (32 * kWaveQuadRow);
    const int c = lane & 31;
    c32 *wr = q + c;
    const c32 *rd = q + c * kWaveQuadRow;
#pragma unroll
    for (int jj = 0; jj < 32; ++jj) swap_lane_halves(r[jj], r[32 + jj]);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) wr[jj * kWaveQuadRow] = r[32 * g + jj];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) r[32 * g + jj] = rd[jj];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// In: r[j] = x[lane + 64 j].  Out: r[j] = X[lane + 64 j], X[k] = sum_n x[n] exp(SIGN 2 pi i n k / 4096).
//   n = 64 n1 + n2, k = k1 + 64 k2:
//   X[k1 + 64 k2] = sum_n2 W64^{n2 k2} ( W4096^{n2 k1} sum_n1 x[64 n1 + n2] W64^{n1 k1} )
// pass 1 over n1 in the registers of lane n2; twiddle W4096^{lane * k1}; transpose; pass 2 over n2 in lane k1.
template <int SIGN, bool WIDE, typename Hook>
DSP_DEV void fft4096_wave(c32 (&r)[kWavePoints], float *img, int lane, Hook hook);

template <int SIGN, bool WIDE = false>
DSP_DEV void fft4096_wave(c32 (&r)[kWavePoints], float *img, int lane) {
    fft4096_wave<SIGN, WIDE>(r, img, lane, [](int) {});
}

// `hook(0..15)` runs at sixteen points spread over the transform (see Dft<64>::run)
template <int SIGN, bool WIDE, typename Hook>
DSP_DEV void fft4096_wave(c32 (&r)[kWavePoints], float *img, int lane, Hook hook) {
    Dft<64, SIGN>::run(r, [&](int i) { hook(i); });
    {
        // W^{lane * k1}, k1 = 8a + b:  (W^{8 lane})^a * (W^{lane})^b -- one sincospif, two short product trees
        const c32 w = unit_root<SIGN>(static_cast<unsigned>(lane), 12);
        c32 pb[8], pa[8];
        pb[1] = w;
#pragma unroll
        for (int t = 2; t < 8; ++t) pb[t] = cmul(pb[t >> 1], pb[t - (t >> 1)]);
        pa[1] = cmul(pb[4], pb[4]);
#pragma unroll
        for (int t = 2; t < 8; ++t) pa[t] = cmul(pa[t >> 1], pa[t - (t >> 1)]);
        c32 *wr = reinterpret_cast<c32 *>(img) + lane;  // WIDE: every point leaves for the transpose image as soon as it is ready
        if constexpr (WIDE) wr[0] = r[0];
#pragma unroll
        for (int a = 0; a < 8; ++a) {
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (a == 0 && b == 0) continue;
                const c32 tw = a == 0 ? pb[b] : (b == 0 ? pa[a] : cmul(pa[a], pb[b]));
                r[8 * a + b] = cmul(r[8 * a + b], tw);
                if constexpr (WIDE) wr[(8 * a + b) * kWaveRow] = r[8 * a + b];
            }
        }
    }
    wave_transpose<WIDE, WIDE>(r, img, lane);
    Dft<64, SIGN>::run(r, [&](int i) { hook(8 + i); });
}

// the same transform with the QUAD transpose (half the LDS: two waves per SIMD fit a CU)
template <int SIGN, typename Hook>
DSP_DEV void fft4096_wave_quad(c32 (&r)[kWavePoints], float *img, int lane, Hook hook) {
    Dft<64, SIGN>::run(r, [&](int i) { hook(i); });
    {
        const c32 w = unit_root<SIGN>(static_cast<unsigned>(lane), 12);
        c32 pb[8], pa[8];
        pb[1] = w;
#pragma unroll
        for (int t = 2; t < 8; ++t) pb[t] = cmul(pb[t >> 1], pb[t - (t >> 1)]);
        pa[1] = cmul(pb[4], pb[4]);
#pragma unroll
        for (int t = 2; t < 8; ++t) pa[t] = cmul(pa[t >> 1], pa[t - (t >> 1)]);
#pragma unroll
        for (int a = 0; a < 8; ++a) {
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (a == 0 && b == 0) continue;
                const c32 tw = a == 0 ? pb[b] : (b == 0 ? pa[a] : cmul(pa[a], pb[b]));
                r[8 * a + b] = cmul(r[8 * a + b], tw);
            }
        }
    }
    wave_transpose_quad(r, img, lane);
    Dft<64, SIGN>::run(r, [&](int i) { hook(8 + i); });
}

template <int SIGN>
DSP_DEV void fft4096_wave_quad(c32 (&r)[kWavePoints], float *img, int lane) {
    fft4096_wave_quad<SIGN>(r, img, lane, [](int) {});
}

}  // namespace dspgpu
