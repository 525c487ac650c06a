// fft_kernels.hip -- batched complex fp32 FFT/IFFT for gfx950 (MI355X).
//
// Stands in for dsp::fft0 / dsp::fft / dsp::shift (reference fft.h:31-94).
//
// Kernel family
//   fft_lds_kernel<LOG2N>      4 <= log2n <= 12.  256 threads x 16 register-resident points =
//                              4096 points per workgroup (1 transform of 4096 ... 256 of 16).
//                              Radix-16 Stockham passes; between passes the points cross the
//                              workgroup through one 34 KiB padded LDS image (4 workgroups/CU).
//                              HBM traffic is exactly one coalesced 8 B read and one 8 B write
//                              per point: 16 B/sample, the figure the roofline is quoted on.
//   fft_tiny_kernel<LOG2N>     log2n <= 3: one thread per transform, registers only.
//   fft_big_kernel<LOG2N>      log2n = 13, 14: the whole transform in one workgroup's LDS (N/16 threads, 70 / 139 KB),
//                              four radix passes, still a single 16 B/sample trip through HBM; persistent
//                              workgroups that prefetch the next transform.
//   fft_column_kernel<LOG2L>   building block of large transforms (13 <= log2n <= 30): FFTs of
//                              length L <= 256 down the middle axis of an [A][L][B] view, 32-64
//                              adjacent columns per workgroup (256-512 byte runs), XCD-contiguous
//                              tile order, streaming accesses; fuses the inter-step twiddle and the
//                              transposed store (staged through LDS so a wave writes 512 contiguous
//                              bytes), optionally the real -> complex packing of dsp::fft on the first
//                              pass and the per-rank block layouts of the distributed transform
//                              (ColumnPlan).  A transform of 2^13..2^16 points takes two such passes,
//                              2^17..2^24 three, 2^25..2^30 four, all on one GPU.
#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "common.h"
#include "fft_lds_core.h"

namespace dspgpu {
namespace {

constexpr bool kRealHalf14Default = true;

// REAL_IN: rows of real samples, zero padded (fft.h:65-74).  SHIFTED: dsp::shift fused (fft.h:85-94).
template <int LOG2N, int SIGN, bool REAL_IN, bool SHIFTED>
__global__ __launch_bounds__(kThreads) void fft_lds_kernel(const void *__restrict__ in_, c32 *__restrict__ out,
                                                           size_t batch, size_t n_in, size_t in_stride,
                                                           size_t tiles, size_t chunk) {
    constexpr int N = 1 << LOG2N;
    constexpr int T = N / kPoints;     // threads per transform
    constexpr int G = kThreads / T;    // transforms per workgroup
    __shared__ c32 lds[kLdsElems];

    const size_t tile = xcd_tile(blockIdx.x, chunk);
    if (tile >= tiles) return;  // whole workgroup
    const int tid = threadIdx.x;
    const int g = tid / T;
    const int lt = tid % T;
    const size_t xf = tile * G + g;
    const bool live = xf < batch;

    // The tile is contiguous in memory and moves with 16-byte accesses (two points per lane, 1 KB per wave
    // instruction), redistributed through the LDS image.  Short transforms (T < 16 threads each) need that anyway --
    // a thread's own points are only T*8 bytes apart from its neighbour's, direct accesses would touch up to 64
    // lines per wave instruction -- and for the long ones the wider accesses are worth more than the two extra LDS
    // round trips cost (A/B on one box: N = 4096 369 -> 373, real input 485 -> 499 GSamples/s; N = 16 ... 128
    // 367 -> 380 against 8-byte accesses).
    constexpr bool kStaged = true;
    const size_t tile_points = (batch - tile * G < static_cast<size_t>(G) ? batch - tile * G : static_cast<size_t>(G)) * N;

    c32 r[kPoints];
    if constexpr (kStaged && !REAL_IN) {
        const c32 *src = static_cast<const c32 *>(in_) + tile * kTile;
        if (tile_points == static_cast<size_t>(kTile)) {  // full tile: sixteen unconditional loads in flight at once
            // 16 bytes (two points) per lane and load: a wave instruction covers 1 KB
            float4 v[kPoints / 2];
#pragma unroll
            for (int m = 0; m < kPoints / 2; ++m)
                v[m] = load_stream_f4(reinterpret_cast<const float *>(src + 2 * (tid + kThreads * m)));
            // padded(2 tid + 512 m) = padded(2 tid) + 544 m: one address, the rest immediate offsets (e and e + 1 share a
            // padding group); likewise the gather below and the way out -- left to the compiler every slot got its own
            // shift-and-add address, ~200 of this kernel's ~640 vector instructions per thread, which is what it runs on while
            // the shader clock is down
            c32 *w = lds + padded(2 * tid);
#pragma unroll
            for (int m = 0; m < kPoints / 2; ++m) {
                w[m * (2 * kThreads + 2 * kThreads / 16)] = mk(v[m].x, v[m].y);
                w[m * (2 * kThreads + 2 * kThreads / 16) + 1] = mk(v[m].z, v[m].w);
            }
        } else {
#pragma unroll
            for (int m = 0; m < kPoints; ++m) {
                const int e = tid + kThreads * m;
                lds[padded(e)] = static_cast<size_t>(e) < tile_points ? load_stream(src + e) : mk(0.0f, 0.0f);
            }
        }
        __syncthreads();
        if constexpr (T % 16 == 0) {
            const c32 *gsrc = lds + padded(g * N + lt);
#pragma unroll
            for (int m = 0; m < kPoints; ++m) r[m] = gsrc[m * (T + T / 16)];
        } else {
#pragma unroll
            for (int m = 0; m < kPoints; ++m) r[m] = lds[padded(g * N + lt + T * m)];
        }
        __syncthreads();  // the passes reuse the image
    } else if (live) {
        if constexpr (REAL_IN) {
            const float *src = static_cast<const float *>(in_) + xf * in_stride;
            if (n_in >= static_cast<size_t>(N)) {  // no zero padding: no bounds tests between the loads
#pragma unroll
                for (int m = 0; m < kPoints; ++m)  // short rows: a line is shared by several loads, keep it cached
                    r[m] = mk(T >= 16 ? __builtin_nontemporal_load(src + lt + T * m) : src[lt + T * m], 0.0f);
            } else {
#pragma unroll
                for (int m = 0; m < kPoints; ++m) {
                    const size_t i = lt + T * m;
                    r[m] = mk(i < n_in ? src[i] : 0.0f, 0.0f);
                }
            }
        } else {
            const c32 *src = static_cast<const c32 *>(in_) + xf * N;
#pragma unroll
            for (int m = 0; m < kPoints; ++m) r[m] = load_stream(src + lt + T * m);
        }
    } else {
#pragma unroll
        for (int m = 0; m < kPoints; ++m) r[m] = mk(0.0f, 0.0f);
    }

    fft_passes<LOG2N, SIGN>(r, lds, lt, g * N);

    if constexpr (kStaged) {
        __syncthreads();  // the last pass's image has been read by everyone
        if constexpr (T % 16 == 0) {
            c32 *w = lds + padded(g * N + lt);
#pragma unroll
            for (int m = 0; m < kPoints; ++m) {
                // dsp::shift: bin k goes to (k + N/2) mod N, and N/2 = 8 T: slot m lands where slot (m + 8) mod 16 would
                const int ms = SHIFTED ? ((m + kPoints / 2) & (kPoints - 1)) : m;
                w[ms * (T + T / 16)] = r[m];
            }
        } else {
#pragma unroll
            for (int m = 0; m < kPoints; ++m) {
                int k = lt + T * m;
                if constexpr (SHIFTED) k = (k + N / 2) & (N - 1);
                lds[padded(g * N + k)] = r[m];
            }
        }
        __syncthreads();
        c32 *dst = out + tile * kTile;
        if (tile_points == static_cast<size_t>(kTile)) {
            const c32 *rd = lds + padded(2 * tid);
#pragma unroll
            for (int m = 0; m < kPoints / 2; ++m) {
                const c32 p0 = rd[m * (2 * kThreads + 2 * kThreads / 16)], p1 = rd[m * (2 * kThreads + 2 * kThreads / 16) + 1];
                store_stream_f4(reinterpret_cast<float *>(dst + 2 * (tid + kThreads * m)), p0.x, p0.y, p1.x, p1.y);
            }
        } else {
#pragma unroll
            for (int m = 0; m < kPoints; ++m) {
                const int e = tid + kThreads * m;
                if (static_cast<size_t>(e) < tile_points) store_stream(dst + e, lds[padded(e)]);
            }
        }
    } else if (live) {
        c32 *dst = out + xf * N;
#pragma unroll
        for (int m = 0; m < kPoints; ++m) {
            int k = lt + T * m;
            if constexpr (SHIFTED) k = (k + N / 2) & (N - 1);
            store_stream(dst + k, r[m]);
        }
    }
}

// 2^13 and 2^14 points in ONE pass through HBM: a workgroup of N/16 threads holds the whole transform in LDS
// (70 / 139 KB: two / one workgroup per CU), four radix passes, three exchanges.  With so few workgroups per CU
// nothing else hides the loads, so the workgroups are persistent and fetch the next transform into registers while
// the current one is transformed.  XCD x owns the contiguous range [x*chunk, (x+1)*chunk) of the batch.
// REAL_HALF: a real row of 2N samples rides an N-point complex transform (z[m] = x[2m] + i x[2m+1], which is just the
// row read as complex values) and the 2N-bin spectrum the reference returns is unfolded from it in the same workgroup:
// with E / O the transforms of the even / odd samples, Z[k] = E[k] + i O[k] and, x being real, conj Z[N-k] = E[k] - i O[k];
// X[k] = E[k] + W_2N^k O[k], X[k+N] = E[k] - W_2N^k O[k].  Half the butterflies and half the LDS of the full-length
// transform, and rows of 2^15 samples still take one trip through HBM (4 B read + 8 B written per sample).
template <int LOG2N, int SIGN, bool REAL_IN, bool SHIFTED, bool REAL_HALF = false, int P = kPoints>
__global__ __launch_bounds__((1 << LOG2N) / P) void fft_big_kernel(const void *__restrict__ in_, c32 *__restrict__ out,
                                                                         size_t batch, size_t chunk, size_t n_in,
                                                                         size_t in_stride) {
    constexpr int N = 1 << LOG2N;
    constexpr int T = N / P;  // threads
    auto pad = [](int i) { return P == 32 ? padded32(i) : padded(i); };
    extern __shared__ __attribute__((aligned(16))) c32 big_lds[];
    const int lt = threadIdx.x;
    const size_t per_xcd = gridDim.x >> 3;
    const size_t lo = static_cast<size_t>(blockIdx.x & 7u) * chunk, hi = lo + chunk < batch ? lo + chunk : batch;
    size_t t = lo + (blockIdx.x >> 3);
    if (t >= hi) return;
    auto fetch = [&](c32 (&v)[P], size_t row) {
        if constexpr (REAL_HALF) {
            const float *src = static_cast<const float *>(in_) + row * in_stride;
            const bool pairs = (in_stride & 1) == 0 && (reinterpret_cast<uintptr_t>(in_) & 7) == 0;
            if (pairs && n_in >= static_cast<size_t>(2 * N)) {  // no padding, 8-byte aligned rows: the row IS the complex input
#pragma unroll
                for (int m = 0; m < P; ++m) v[m] = load_stream(reinterpret_cast<const c32 *>(src) + lt + T * m);
            } else {
                // zero padding (fft.h:65-74) by the hardware: buffer loads of a descriptor that ends with the row return 0
                // beyond it -- no branch around any load, no lane masks
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0,
                                                                    static_cast<int>(n_in < 2 * static_cast<size_t>(N) ? n_in : 2 * N) * 4, 0x00020000);
                int lo8 = 8 * lt;  // laundered: the loop-invariant offsets are not worth registers held across the persistent loop
                asm volatile("" : "+v"(lo8));
#pragma unroll
                for (int m = 0; m < P; ++m) {
                    const int off = lo8 + 8 * T * m;
                    v[m] = mk(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0)),
                              __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off + 4, 0, 0)));
                }
            }
        } else if constexpr (REAL_IN) {  // rows of real samples, zero padded (fft.h:65-74) by the descriptor's bounds check
            const float *src = static_cast<const float *>(in_) + row * in_stride;
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0,
                                                                static_cast<int>(n_in < static_cast<size_t>(N) ? n_in : N) * 4, 0x00020000);
            int lo4 = 4 * lt;
            asm volatile("" : "+v"(lo4));
#pragma unroll
            for (int m = 0; m < P; ++m)
                v[m] = mk(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lo4 + 4 * T * m, 0, 0)), 0.0f);
        } else {
            const c32 *src = static_cast<const c32 *>(in_) + row * N;
#pragma unroll
            for (int m = 0; m < P; ++m) v[m] = load_stream(src + lt + T * m);
        }
    };
    c32 nxt[P];
    fetch(nxt, t);
    for (; t < hi; t += per_xcd) {
        c32 r[P];
#pragma unroll
        for (int m = 0; m < P; ++m) r[m] = nxt[m];
        if (t + per_xcd < hi) fetch(nxt, t + per_xcd);
        if constexpr (P == 32) fft_passes32<LOG2N, SIGN>(r, big_lds, lt);
        else fft_passes<LOG2N, SIGN>(r, big_lds, lt, 0);
        __syncthreads();  // the last image has been read before the next transform's first exchange
        if constexpr (REAL_HALF) {
            // Z in natural order through the image once more: bin k needs its mirror N - k
#pragma unroll
            for (int m = 0; m < P; ++m) big_lds[pad(lt + T * m)] = r[m];
            __syncthreads();
            // W_2N^k, k = lt + T m:  W^lt (W^T)^m -- two sincos and a product tree
            const c32 w0 = unit_root<SIGN>(static_cast<unsigned>(lt), LOG2N + 1);
            c32 pw[P];
            pw[1] = unit_root<SIGN>(static_cast<unsigned>(T), LOG2N + 1);
#pragma unroll
            for (int m = 2; m < P; ++m) pw[m] = cmul(pw[m >> 1], pw[m - (m >> 1)]);
            c32 *row = out + t * (2 * N);
#pragma unroll
            for (int m = 0; m < P; ++m) {
                const int k = lt + T * m;
                const c32 zp = big_lds[pad((N - k) & (N - 1))];
                const c32 zc = mk(zp.x, -zp.y);                       // conj Z[N - k] = E - i O
                const c32 e = (r[m] + zc) * 0.5f;
                const c32 d = (r[m] - zc) * 0.5f;                      // i O
                const c32 o = mk(d.y, -d.x);                           // O = -i (i O)
                const c32 wo = cmul(o, m == 0 ? w0 : cmul(w0, pw[m]));
                // dsp::shift fused (fft.h:85-94): the halves of the 2N-bin row change places
                store_stream(row + (SHIFTED ? k + N : k), e + wo);
                store_stream(row + (SHIFTED ? k : k + N), e - wo);
            }
            __syncthreads();  // the image is the next transform's exchange buffer
            continue;
        }
#pragma unroll
        for (int m = 0; m < P; ++m) {
            int k = lt + T * m;
            if constexpr (SHIFTED) k = (k + N / 2) & (N - 1);  // dsp::shift fused (fft.h:85-94)
            store_stream(out + t * N + k, r[m]);
        }
    }
}

// ---- 2^15 points in ONE pass through HBM -----------------------------------------------------------------------
// 512 threads x 64 register-resident points; the complex image of the transform (256 KB) does not fit a CU's LDS, so the
// points cross between passes one fp32 plane at a time (real parts, then imaginary parts) through a 132 KB image:
// radix 32, 32, 32 -- three passes, two exchanges of two planes each (the exchanges are what costs: 512 KB of LDS traffic
// each; radix 16, 16, 16, 8 with three of them ran 7 % slower).  One workgroup per CU, persistent, XCD x owns
// the contiguous range [x*chunk, (x+1)*chunk) of the batch.  Against the two-pass route (16 B + 16 B per point) this
// is one read and one write of every point.
// REAL_HALF: rows of 2^16 real samples ride the 2^15-point transform and the 2^16-bin spectrum is unfolded in the same
// workgroup (see fft_big_kernel): 4 B read + 8 B written per sample instead of 8 + 12.
constexpr int kPlanePoints = 64;
constexpr int kPlaneLog2N = 15;
constexpr int kPlaneThreads = (1 << kPlaneLog2N) / kPlanePoints;                        // 512
constexpr size_t kPlaneLdsBytes = ((1 << kPlaneLog2N) + (1 << kPlaneLog2N) / 32) * sizeof(float);  // 135,168
DSP_DEV int plane_padded(int i) { return i + (i >> 5); }  // one pad word per 32: the radix-32 scatter strides by 33 words

// scatter the outputs of the pass (radix 32 after sub-length NS), gather r[m] = element lt + T m; one plane at a time
template <int NS>
DSP_DEV void plane_exchange(c32 (&r)[kPlanePoints], float *lds, int lt) {
    constexpr int R = 32, T = kPlaneThreads, B = kPlanePoints / R;
    static_assert(NS == 1 || NS % 32 == 0, "sub-length");
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        __syncthreads();  // the image of the previous plane / exchange / transform has been read by everyone
#pragma unroll
        for (int s = 0; s < B; ++s) {
            const int j = lt + T * s;
            // plane_padded(dest + u NS) is linear in u: 32 j + u -> 33 j + u (NS = 1), x + 32 k -> plane_padded(x) + 33 k otherwise
            float *w = lds + plane_padded(stockham_dest<R, NS>(j, 0));
#pragma unroll
            for (int u = 0; u < R; ++u) w[NS == 1 ? u : u * (NS + NS / 32)] = part ? r[s + u * B].y : r[s + u * B].x;
        }
        __syncthreads();
        const float *g = lds + plane_padded(lt);
#pragma unroll
        for (int m = 0; m < kPlanePoints; ++m) {
            const float v = g[m * (T + T / 32)];
            if (part) r[m].y = v;
            else r[m].x = v;
        }
    }
}

template <int SIGN, bool REAL_HALF, bool SHIFTED>
__global__ __launch_bounds__(kPlaneThreads) void fft_plane_kernel(const void *__restrict__ in_, c32 *__restrict__ out,
                                                                  size_t batch, size_t chunk, size_t n_in, size_t in_stride) {
    constexpr int LOG2N = kPlaneLog2N, N = 1 << LOG2N, P = kPlanePoints, T = kPlaneThreads;
    extern __shared__ __attribute__((aligned(16))) float plane_lds[];
    const int lt = threadIdx.x;
    const size_t per_xcd = gridDim.x >> 3;
    const size_t lo = static_cast<size_t>(blockIdx.x & 7u) * chunk, hi = lo + chunk < batch ? lo + chunk : batch;
    for (size_t t = lo + (blockIdx.x >> 3); t < hi; t += per_xcd) {
        c32 r[P];
        if constexpr (REAL_HALF) {
            const float *src = static_cast<const float *>(in_) + t * in_stride;
            const bool pairs = (in_stride & 1) == 0 && (reinterpret_cast<uintptr_t>(in_) & 7) == 0;
            if (pairs && n_in >= static_cast<size_t>(2 * N)) {  // no padding, 8-byte aligned rows: the row IS the complex input
#pragma unroll
                for (int i = 0; i < P; ++i) {  // even points first: the first radix-32 butterfly takes r[0], r[2], ... and
                    const int m = 2 * (i % (P / 2)) + i / (P / 2);  // starts while the odd ones are still on their way (+3 %)
                    r[m] = load_stream(reinterpret_cast<const c32 *>(src) + lt + T * m);
                }
            } else {
                // zero padding (fft.h:65-74) by the hardware: buffer loads of a descriptor that ends with the row return 0
                // beyond it -- no branch around any load, no lane masks, all loads in flight together
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0,
                                                                    static_cast<int>(n_in < 2 * static_cast<size_t>(N) ? n_in : 2 * N) * 4, 0x00020000);
                int lo8 = 8 * lt;  // laundered: 128 loop-invariant offsets hoisted out of the persistent loop would be spilled
                asm volatile("" : "+v"(lo8));
#pragma unroll
                for (int m = 0; m < P; ++m) {
                    const int off = lo8 + 8 * T * m;
                    r[m] = mk(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0)),
                              __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off + 4, 0, 0)));
                }
            }
        } else {
            const c32 *src = static_cast<const c32 *>(in_) + t * N;
#pragma unroll
            for (int i = 0; i < P; ++i) {  // even points first (see above)
                const int m = 2 * (i % (P / 2)) + i / (P / 2);
                r[m] = load_stream(src + lt + T * m);
            }
        }
        stockham_pass<32, 1, LOG2N, SIGN, P>(r, lt);
        plane_exchange<1>(r, plane_lds, lt);
        stockham_pass<32, 32, LOG2N, SIGN, P>(r, lt);
        plane_exchange<32>(r, plane_lds, lt);
        if constexpr (REAL_HALF) {
            stockham_pass<32, 1024, LOG2N, SIGN, P>(r, lt);
        } else {
            // the outputs of each of the last pass's two butterflies leave as soon as they exist
            stockham_pass<32, 1024, LOG2N, SIGN, P>(r, lt, [&](int sb) {
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    int k = lt + T * (sb + 2 * u);
                    if constexpr (SHIFTED) k = (k + N / 2) & (N - 1);
                    store_stream(out + t * N + k, r[sb + 2 * u]);
                }
            });
            continue;
        }
        if constexpr (REAL_HALF) {
            // Z[k], k = lt + T m, in natural order through the image once more, a plane at a time: bin k needs Z[N - k].
            // With zc = conj Z[N - k]:  E = (Z + zc)/2, i O = (Z - zc)/2, X[k] = E + W_2N^k O, X[k + N] = E - W_2N^k O.
            // The image addresses do not depend on the transform; recomputed from a laundered copy of the thread index so
            // that they are not hoisted out of the persistent loop (64 + 64 address registers held across all passes).
            int lv = lt;
            asm volatile("" : "+v"(lv));
            float *own = plane_lds + plane_padded(lv);
            // N - k = (T - lt) + T (63 - m) for lt > 0; thread 0: T (64 - m), and bin 0 mirrors itself.
            // plane_padded(x + T j) = plane_padded(x) + (T + T/32) j.
            const float *mir = plane_lds + plane_padded(T - lv);
            constexpr int kRow = T + T / 32;
            float dx[P];
            __syncthreads();
#pragma unroll
            for (int m = 0; m < P; ++m) own[m * kRow] = r[m].x;
            __syncthreads();
#pragma unroll
            for (int m = 0; m < P; ++m) {
                if (m % 8 == 0) __builtin_amdgcn_sched_barrier(0);  // as below: eight image reads in flight, not 64
                const float zx = m == 0 ? (lv ? mir[63 * kRow] : plane_lds[0]) : mir[(63 - m) * kRow];
                dx[m] = 0.5f * (r[m].x - zx);
                r[m].x = 0.5f * (r[m].x + zx);
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < P; ++m) own[m * kRow] = r[m].y;
            __syncthreads();
            // W_2N^k = W_2N^lt * W_128^m (T / 2N = 1/128): one sincos per thread, the rest compile-time constants
            const c32 w0 = unit_root<SIGN>(static_cast<unsigned>(lv), LOG2N + 1);
            c32 *row = out + t * (2 * static_cast<size_t>(N)) + lv;
#pragma unroll
            for (int m = 0; m < P; ++m) {
                // keep the image reads next to their use: all 64 hoisted to the top would need 64 more registers
                if (m % 8 == 0) __builtin_amdgcn_sched_barrier(0);
                const float zy = m == 0 ? (lv ? mir[63 * kRow] : plane_lds[0]) : mir[(63 - m) * kRow];
                const c32 e = mk(r[m].x, 0.5f * (r[m].y - zy));
                const c32 o = mk(0.5f * (r[m].y + zy), -dx[m]);   // O = -i (i O)
                c32 w = w0;
                if (m) {
                    c32 k = c32{cos128(m), SIGN * sin128(m)};
                    asm volatile("" : "+s"(k));  // materialised here, not kept in scalar registers across the loop
                    w = cmul_k(w0, k);
                }
                const c32 wo = cmul(o, w);
                // dsp::shift fused (fft.h:85-94): the halves of the 2N-bin row change places
                store_stream(row + T * m + (SHIFTED ? N : 0), e + wo);
                store_stream(row + T * m + (SHIFTED ? 0 : N), e - wo);
            }
        } else {
#pragma unroll
            for (int m = 0; m < P; ++m) {
                int k = lt + T * m;
                if constexpr (SHIFTED) k = (k + N / 2) & (N - 1);
                store_stream(out + t * N + k, r[m]);
            }
        }
    }
}

template <int LOG2N, int SIGN, bool REAL_IN, bool SHIFTED>
__global__ __launch_bounds__(kThreads) void fft_tiny_kernel(const void *__restrict__ in_, c32 *__restrict__ out,
                                                            size_t batch, size_t n_in, size_t in_stride) {
    constexpr int N = 1 << LOG2N;
    const size_t xf = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x;
    if (xf >= batch) return;
    c32 v[N];
    if constexpr (REAL_IN) {
        const float *src = static_cast<const float *>(in_) + xf * in_stride;
#pragma unroll
        for (int m = 0; m < N; ++m) v[m] = mk(static_cast<size_t>(m) < n_in ? src[m] : 0.0f, 0.0f);
    } else {
        const c32 *src = static_cast<const c32 *>(in_) + xf * N;
#pragma unroll
        for (int m = 0; m < N; ++m) v[m] = src[m];
    }
    Dft<N, SIGN>::run(v);
    c32 *dst = out + xf * N;
#pragma unroll
    for (int m = 0; m < N; ++m) dst[SHIFTED ? ((m + N / 2) & (N - 1)) : m] = v[m];
}

// ---- column pass of a large transform -----------------------------------------------------
// View: in[a][l][b], l < L = 2^LOG2L, b < B (contiguous).  For every (a, b) column: FFT over l,
// output element k multiplied by exp(SIGN*2 pi i * k*(b/Bi) / 2^tw_log2) when tw_log2 > 0,
// stored at out[a*L*B + (b/Bi)*s_hi + k*s_k + (b%Bi)].  A workgroup owns C = 4096/L adjacent
// columns (C >= 16 because L <= 256); the LDS image is [position][column] so neighbouring
// lanes (columns) touch neighbouring banks in every access and no padding is needed.
struct ColumnPlan {
    size_t B;      // inner contiguous extent
    size_t Bi;     // b is split as (b / Bi, b % Bi)
    size_t s_hi;   // output stride of b / Bi
    size_t s_k;    // output stride of k
    int tw_log2;   // log2 of the twiddle modulus, 0 = no twiddle
    // Optional generalisations (0 = the plain layout above), used by the fused "unpack . twiddle -> row FFT -> pack"
    // phase of the distributed transform, where the row axis is cut into per-rank blocks on both sides:
    size_t in_a = 0;      // input stride of a                                   (default L*B)
    size_t in_lo = 0;     // input stride of l & (2^in_split - 1)                  (default B)
    size_t in_hi = 0;     // input stride of l >> (in_split + in_mid_bits)       (0: l is not split)
    int in_split = 0;
    size_t in_mid = 0;    // input stride of the in_mid_bits bits of l above in_split (pipelined exchange: the column chunk)
    int in_mid_bits = 0;
    int in_tw_log2 = 0;   // input element (a, l, b) is first multiplied by W^{(in_tw_row0 + a) * (l*B + b)}
    unsigned long long in_tw_row0 = 0;
    size_t out_a = 0;     // output stride of a                                  (default L*B)
    size_t out_hi = 0;    // natural-order store: k -> (k >> out_split)*out_hi + (k & (2^out_split - 1))*s_k
    int out_split = 0;
    // first pass of a large real-input transform (dsp::fft, fft.h:57-83): `in` is really float rows of real_n_in
    // samples (row stride real_stride); element (a, l, b) = sample l*B + b of row a, zero beyond real_n_in
    size_t real_n_in = 0, real_stride = 0;
    // Half-length form of a large real-input transform (see fft_big_kernel, REAL_HALF): the 2M real samples of a row ride
    // an M-point complex transform.  real_pairs (first pass): element (a, l, b) = (x[2i], x[2i+1]), i = l*B + b.
    // unfold (last pass, natural-order output Z[b + B k]): the workgroup takes C/2 columns and their mirrors B - b, so that
    // every bin finds Z[M - k] in its own LDS image, and stores the 2M-bin spectrum X[k] = E + W_2M^k O, X[k + M] = E - W O
    // (rows out_a apart); unfold == 2 swaps the two halves of the row (dsp::shift, fft.h:85-94).
    int real_pairs = 0, unfold = 0;
};

template <int R, int NS, int LOG2L, int C>
DSP_DEV void column_exchange(c32 (&r)[kPoints], c32 *lds, int lt, int c) {
    constexpr int T = (1 << LOG2L) / kPoints;
    constexpr int Bf = kPoints / R;
#pragma unroll
    for (int s = 0; s < Bf; ++s) {
        const int j = lt + T * s;
#pragma unroll
        for (int u = 0; u < R; ++u) lds[stockham_dest<R, NS>(j, u) * C + c] = r[s + u * Bf];
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < kPoints; ++m) r[m] = lds[(lt + T * m) * C + c];
}

// TH threads per workgroup: 256, or 512 for the long columns so that a workgroup still covers 32-64 adjacent
// columns (256-512 contiguous bytes per access; with 128-256 the pass ran at 4.4 TB/s against 5.8 for rows).
template <int LOG2L, int SIGN, int TH>
__global__ __launch_bounds__(TH) void fft_column_kernel(const c32 *__restrict__ in, c32 *__restrict__ out,
                                                        ColumnPlan plan, size_t columns_per_a) {
    constexpr int L = 1 << LOG2L;
    constexpr int T = L / kPoints;   // threads per column
    constexpr int C = TH / T;        // columns per workgroup
    __shared__ c32 lds[TH * kPoints + TH];  // + one pad element per column for the transposed store

    const int tid = threadIdx.x;
    const int c = tid % C;
    const int lt = tid / C;
    const size_t ntile = gridDim.x;
    const size_t chunkx = (ntile + 7) / 8;
    const size_t tile = (ntile & 7) == 0 ? xcd_tile(blockIdx.x, chunkx) : blockIdx.x;  // tiles never straddle an `a` slab: B % C == 0
    const size_t tiles_per_a = columns_per_a / C;
    const size_t a = tile / tiles_per_a;
    size_t b = (tile % tiles_per_a) * C + c;
    if (plan.unfold) {  // C/2 columns j = 1, 2, ... and their mirrors B - j; the self-mirrored columns 0 and B/2 share "pair" 0
        const size_t j = (tile % tiles_per_a) * (C / 2) + (c < C / 2 ? c : c - C / 2);
        b = c < C / 2 ? j : (j == 0 ? plan.B / 2 : plan.B - j);
    }
    const size_t in_lo = plan.in_lo ? plan.in_lo : plan.B;
    const c32 *src = in + a * (plan.in_a ? plan.in_a : L * plan.B) + b;

    c32 r[kPoints];
    if (plan.real_stride && plan.real_pairs) {
        const float *row = reinterpret_cast<const float *>(in) + a * plan.real_stride;
#pragma unroll
        for (int m = 0; m < kPoints; ++m) {
            const size_t i = 2 * (static_cast<size_t>(lt + T * m) * plan.B + b);
            r[m] = mk(i < plan.real_n_in ? row[i] : 0.0f, i + 1 < plan.real_n_in ? row[i + 1] : 0.0f);
        }
    } else if (plan.real_stride) {
        const float *row = reinterpret_cast<const float *>(in) + a * plan.real_stride;
#pragma unroll
        for (int m = 0; m < kPoints; ++m) {
            const size_t i = static_cast<size_t>(lt + T * m) * plan.B + b;
            r[m] = mk(i < plan.real_n_in ? row[i] : 0.0f, 0.0f);
        }
    } else if (plan.in_hi && plan.in_mid_bits) {
        const unsigned mask = (1u << plan.in_split) - 1u, mmask = (1u << plan.in_mid_bits) - 1u;
#pragma unroll
        for (int m = 0; m < kPoints; ++m) {
            const unsigned l = lt + T * m;
            r[m] = load_stream(src + static_cast<size_t>(l >> (plan.in_split + plan.in_mid_bits)) * plan.in_hi +
                               static_cast<size_t>((l >> plan.in_split) & mmask) * plan.in_mid + static_cast<size_t>(l & mask) * in_lo);
        }
    } else if (plan.in_hi) {
        const unsigned mask = (1u << plan.in_split) - 1u;
#pragma unroll
        for (int m = 0; m < kPoints; ++m) {
            const unsigned l = lt + T * m;
            r[m] = load_stream(src + static_cast<size_t>(l >> plan.in_split) * plan.in_hi + static_cast<size_t>(l & mask) * in_lo);
        }
    } else {
#pragma unroll
        for (int m = 0; m < kPoints; ++m) r[m] = load_stream(src + static_cast<size_t>(lt + T * m) * in_lo);
    }
    if (plan.in_tw_log2 > 0) {
        // W^{k1 * n2}, n2 = (lt + T*m)*B + b:  W^{k1*(lt*B + b)} * (W^{k1*T*B})^m, the two roots from exact
        // double angles (exponents reach 2^30), the powers by a product tree
        const unsigned long long mask = (1ull << plan.in_tw_log2) - 1ull;
        const unsigned long long k1 = plan.in_tw_row0 + a;
        const c32 w0 = unit_root_wide<SIGN>((k1 * (static_cast<unsigned long long>(lt) * plan.B + b)) & mask, plan.in_tw_log2);
        c32 pw[kPoints];
        pw[1] = unit_root_wide<SIGN>((k1 * (static_cast<unsigned long long>(T) * plan.B)) & mask, plan.in_tw_log2);
#pragma unroll
        for (int m = 2; m < kPoints; ++m) pw[m] = cmul(pw[m >> 1], pw[m - (m >> 1)]);
        r[0] = cmul(r[0], w0);
#pragma unroll
        for (int m = 1; m < kPoints; ++m) r[m] = cmul(r[m], cmul(w0, pw[m]));
    }

    constexpr int R0 = radix_at(LOG2L, 0), R1 = radix_at(LOG2L, 1);
    stockham_pass<R0, 1, LOG2L, SIGN, kPoints>(r, lt);
    if constexpr (R1 > 1) {
        column_exchange<R0, 1, LOG2L, C>(r, lds, lt, c);
        stockham_pass<R1, R0, LOG2L, SIGN, kPoints>(r, lt);
    }

    const size_t bhi = b / plan.Bi, blo = b % plan.Bi;
    if (plan.tw_log2 > 0) {
        // W^(k*bhi), k = lt + T*m:  W^(lt*bhi) * (W^(T*bhi))^m -- two sincos and a product tree (depth 4) instead
        // of sixteen sincos, which cost more VALU work than the transform itself
        const unsigned mask = (1u << plan.tw_log2) - 1u;
        const unsigned ub = static_cast<unsigned>(bhi);
        // moduli beyond 2^24 (transforms of more than 2^24 points) need the angle in double
        const unsigned e0 = (static_cast<unsigned>(lt) * ub) & mask, e1 = (static_cast<unsigned>(T) * ub) & mask;
        const c32 w0 = plan.tw_log2 > 24 ? unit_root_wide<SIGN>(e0, plan.tw_log2) : unit_root<SIGN>(e0, plan.tw_log2);
        c32 pw[kPoints];
        pw[1] = plan.tw_log2 > 24 ? unit_root_wide<SIGN>(e1, plan.tw_log2) : unit_root<SIGN>(e1, plan.tw_log2);
#pragma unroll
        for (int m = 2; m < kPoints; ++m) pw[m] = cmul(pw[m >> 1], pw[m - (m >> 1)]);
        r[0] = cmul(r[0], w0);
#pragma unroll
        for (int m = 1; m < kPoints; ++m) r[m] = cmul(r[m], cmul(w0, pw[m]));
    }
    if (plan.unfold) {
        // Z[b + B k] of the tile's columns, as [column slot][k], so that bin k finds its mirror M - k = (B - b) + B (L - 1 - k)
        // (column 0: B (L - k)) in the partner slot
        __syncthreads();  // the last exchange's image has been read by everyone
#pragma unroll
        for (int m = 0; m < kPoints; ++m) lds[c * (L + 1) + lt + T * m] = r[m];
        __syncthreads();
        const size_t j = (tile % tiles_per_a) * (C / 2) + (c < C / 2 ? c : c - C / 2);
        const int cp = j == 0 ? c : (c < C / 2 ? c + C / 2 : c - C / 2);
        const size_t M = static_cast<size_t>(L) * plan.B;
        int log2b = 0;
        while ((static_cast<size_t>(1) << log2b) < plan.B) ++log2b;
        // W_2M^{b + B k} = W_2M^b * W_2L^k, k = lt + T m: three roots and a product tree
        const c32 wcol = LOG2L + log2b + 1 <= 24 ? unit_root<SIGN>(static_cast<unsigned>(b), LOG2L + log2b + 1)
                                                 : unit_root_wide<SIGN>(b, LOG2L + log2b + 1);
        const c32 w0 = cmul(wcol, unit_root<SIGN>(static_cast<unsigned>(lt), LOG2L + 1));
        c32 pw[kPoints];
        pw[1] = unit_root<SIGN>(static_cast<unsigned>(T), LOG2L + 1);
#pragma unroll
        for (int m = 2; m < kPoints; ++m) pw[m] = cmul(pw[m >> 1], pw[m - (m >> 1)]);
        c32 *rowp = out + a * plan.out_a;
        const size_t lo_off = plan.unfold == 2 ? M : 0, hi_off = plan.unfold == 2 ? 0 : M;
#pragma unroll
        for (int m = 0; m < kPoints; ++m) {
            const int k = lt + T * m;
            const int kp = b == 0 ? ((L - k) & (L - 1)) : L - 1 - k;
            const c32 zp = lds[cp * (L + 1) + kp];
            const c32 zc = mk(zp.x, -zp.y);             // conj Z[M - k] = E - i O
            const c32 e = (r[m] + zc) * 0.5f;
            const c32 d = (r[m] - zc) * 0.5f;            // i O
            const c32 wo = cmul(mk(d.y, -d.x), m == 0 ? w0 : cmul(w0, pw[m]));
            const size_t kk = b + plan.B * static_cast<size_t>(k);
            store_stream(rowp + lo_off + kk, e + wo);
            store_stream(rowp + hi_off + kk, e - wo);
        }
        return;
    }
    if (plan.s_k == 1 && plan.Bi == 1) {
        // transposed store (k contiguous per column): stage the tile as [column][k] so that a wave writes 512
        // contiguous bytes of one column per instruction instead of 4 elements of each of 16 columns
        __syncthreads();  // the last exchange's image has been read by everyone
#pragma unroll
        for (int m = 0; m < kPoints; ++m) lds[c * (L + 1) + lt + T * m] = r[m];
        __syncthreads();
        const size_t b0 = (tile % tiles_per_a) * C;
        c32 *base = out + a * (plan.out_a ? plan.out_a : L * plan.B) + b0 * plan.s_hi;
        if ((plan.s_hi & 1) == 0 && (reinterpret_cast<uintptr_t>(base) & 15) == 0) {
            // 16 bytes (two consecutive k of one column) per lane and store: a wave instruction writes 1 KB
#pragma unroll
            for (int m = 0; m < kPoints / 2; ++m) {
                const int i = 2 * (tid + TH * m);
                const int cc = i / L, k = i % L;  // L is even: k and k + 1 lie in the same column
                const c32 p0 = lds[cc * (L + 1) + k], p1 = lds[cc * (L + 1) + k + 1];
                store_stream_f4(reinterpret_cast<float *>(base + static_cast<size_t>(cc) * plan.s_hi + k), p0.x, p0.y, p1.x, p1.y);
            }
            return;
        }
#pragma unroll
        for (int m = 0; m < kPoints; ++m) {
            const int i = tid + TH * m;
            const int cc = i / L, k = i % L;
            store_stream(base + static_cast<size_t>(cc) * plan.s_hi + k, lds[cc * (L + 1) + k]);
        }
        return;
    }
    c32 *dst = out + a * (plan.out_a ? plan.out_a : L * plan.B) + bhi * plan.s_hi + blo;
    if (plan.out_hi) {
        const unsigned mask = (1u << plan.out_split) - 1u;
#pragma unroll
        for (int m = 0; m < kPoints; ++m) {
            const unsigned k = lt + T * m;
            store_stream(dst + static_cast<size_t>(k >> plan.out_split) * plan.out_hi + static_cast<size_t>(k & mask) * plan.s_k, r[m]);
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < kPoints; ++m) {
        const unsigned k = lt + T * m;
        store_stream(dst + static_cast<size_t>(k) * plan.s_k, r[m]);
    }
}

// dsp::shift (fft.h:85-94) for the large path: rotating a power-of-two row by n/2 is a swap
// of its halves, done in place (one thread per pair).
__global__ __launch_bounds__(kThreads) void swap_halves_kernel(c32 *__restrict__ data, size_t n, size_t pairs) {
    const size_t half = n / 2;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < pairs;
         i += static_cast<size_t>(gridDim.x) * kThreads) {
        const size_t row = i / half, col = i % half;
        c32 *p = data + row * n + col;
        const c32 lo = p[0], hi = p[half];
        p[0] = hi;
        p[half] = lo;
    }
}

// ---- host-side dispatch ---------------------------------------------------------------------
// Workgroups of fft_lds_kernel per CU: the 34 KB image allows four; unused dynamic LDS lowers that to THREE (default) or
// two (DSP_GPU_FFT_ROW_WGS_PER_CU=2; =4 restores the old launch).  With fewer 32 KB tiles in flight per CU the memory
// system runs faster once the chip has been busy for ~40 launches -- 65,536 x 4096, sustained: four per CU 0.719 ms
// = 0.745 of 8 TB/s, three 0.699 = 0.768, two 0.675 = 0.795, and every size from 32 points up gains alike -- but in the
// first ~25 launches after idle the order is the reverse (0.725 / 0.730 / 0.776 ms).  Three per CU is the setting that
// loses nothing cold (-0.5 %) and gains warm (+3 %); two is for pipelines that stay busy (DESIGN.md section 4.1,
// profiles/r02_fft_wgs_per_cu.txt).
// Complex transforms of up to 256 points keep four (three reads 1.5 % slower there; real-input rows gain at every size).
static unsigned row_lds_pad(int log2n, bool real_in) {
    static const int forced = [] {
        const char *e = getenv("DSP_GPU_FFT_ROW_WGS_PER_CU");
        return e ? atoi(e) : 0;
    }();
    const int w = forced ? forced : ((log2n >= 9 || real_in) ? 3 : 4);
    return w == 3 ? 8192u : (w == 2 ? 20480u : 0u);
}

template <int LOG2N, bool REAL_IN, bool SHIFTED>
int launch_small(const void *in, c32 *out, size_t batch, size_t n_in, size_t in_stride, int direction,
                 hipStream_t s) {
    if constexpr (LOG2N >= 4) {
        constexpr int G = kThreads / ((1 << LOG2N) / kPoints);
        const size_t tiles = (batch + G - 1) / G;
        const size_t chunk = (tiles + 7) / 8;  // tiles per XCD
        if (8 * chunk > 0x7fffffffull) return fail(DSP_GPU_EUNSUPPORTED, "batch too large for one launch");
        const dim3 grid(static_cast<unsigned>(8 * chunk));
        if (direction == DSP_GPU_FORWARD)
            hipLaunchKernelGGL((fft_lds_kernel<LOG2N, +1, REAL_IN, SHIFTED>), grid, dim3(kThreads), row_lds_pad(LOG2N, REAL_IN), s, in, out, batch,
                               n_in, in_stride, tiles, chunk);
        else
            hipLaunchKernelGGL((fft_lds_kernel<LOG2N, -1, REAL_IN, SHIFTED>), grid, dim3(kThreads), row_lds_pad(LOG2N, REAL_IN), s, in, out, batch,
                               n_in, in_stride, tiles, chunk);
    } else {
        const size_t blocks = (batch + kThreads - 1) / kThreads;
        if (direction == DSP_GPU_FORWARD)
            hipLaunchKernelGGL((fft_tiny_kernel<LOG2N, +1, REAL_IN, SHIFTED>), dim3(blocks), dim3(kThreads), 0, s, in,
                               out, batch, n_in, in_stride);
        else
            hipLaunchKernelGGL((fft_tiny_kernel<LOG2N, -1, REAL_IN, SHIFTED>), dim3(blocks), dim3(kThreads), 0, s, in,
                               out, batch, n_in, in_stride);
    }
    return check_launch("fft_lds_kernel");
}

template <bool REAL_IN, bool SHIFTED>
int dispatch_small(int log2n, const void *in, c32 *out, size_t batch, size_t n_in, size_t in_stride, int direction,
                   hipStream_t s) {
    switch (log2n) {
#define DSP_CASE(K) \
    case K: return launch_small<K, REAL_IN, SHIFTED>(in, out, batch, n_in, in_stride, direction, s);
        DSP_CASE(1) DSP_CASE(2) DSP_CASE(3) DSP_CASE(4) DSP_CASE(5) DSP_CASE(6) DSP_CASE(7) DSP_CASE(8)
        DSP_CASE(9) DSP_CASE(10) DSP_CASE(11) DSP_CASE(12)
#undef DSP_CASE
    default: return fail(DSP_GPU_EINVAL, "log2n %d outside the single-pass range", log2n);
    }
}

// DSP_GPU_REAL_SHORT_LAST=0 (A/B): the balanced factorisation for the half-length real-input route too
bool real_half_short_last() {
    const char *e = getenv("DSP_GPU_REAL_SHORT_LAST");
    return !e || e[0] != '0';
}

int launch_column(int log2l, const c32 *in, c32 *out, const ColumnPlan &plan, size_t slabs, int direction,
                  hipStream_t s) {
    const int L = 1 << log2l;
    // 512 threads: twice the columns.  The unfolding pass (mirrored half tiles) also for 64-point columns.
    const bool wide = (log2l >= 7 || (log2l == 6 && plan.unfold)) && plan.B % (2 * kTile / L) == 0;
    const int C = (wide ? 2 : 1) * kTile / L;
    if (plan.B % C != 0) return fail(DSP_GPU_EINVAL, "column pass: inner extent %zu not a multiple of %d", plan.B, C);
    const size_t blocks = slabs * (plan.B / C);
    switch (log2l) {
#define DSP_LAUNCH(K, TH)                                                                                          \
        if (direction == DSP_GPU_FORWARD)                                                                        \
            hipLaunchKernelGGL((fft_column_kernel<K, +1, TH>), dim3(blocks), dim3(TH), 0, s, in, out, plan, plan.B); \
        else                                                                                                     \
            hipLaunchKernelGGL((fft_column_kernel<K, -1, TH>), dim3(blocks), dim3(TH), 0, s, in, out, plan, plan.B);
#define DSP_CASE(K) \
    case K: DSP_LAUNCH(K, kThreads) break;
#define DSP_CASE_WIDE(K) \
    case K: if (wide) { DSP_LAUNCH(K, 2 * kThreads) } else { DSP_LAUNCH(K, kThreads) } break;
        DSP_CASE(4) DSP_CASE(5) DSP_CASE_WIDE(6) DSP_CASE_WIDE(7) DSP_CASE_WIDE(8)
#undef DSP_CASE
#undef DSP_CASE_WIDE
#undef DSP_LAUNCH
    default: return fail(DSP_GPU_EINVAL, "column pass length 2^%d unsupported", log2l);
    }
    return check_launch("fft_column_kernel");
}

// Large transform: N = N1*N2*...*NK, K = 2 (N <= 2^16), 3 (<= 2^24) or 4 (<= 2^30) column passes, every factor
// <= 256.  Before step i the data is laid out [n_i][n_{i+1} .. n_K][k_{i-1} .. k_1]; the step transforms over
// n_i, multiplies by W_{n_i * rest}^{k_i * r} (r = the index into n_{i+1} .. n_K, rest = their product; nothing on
// the last step) and stores [n_{i+1} .. n_K][k_i][k_{i-1} .. k_1] -- after step K that is natural order.  The
// passes ping-pong between `out` and one scratch buffer so that the input is never modified.
// `fixed_scratch` != nullptr: the caller's buffer of at least batch * n points is the scratch (the allocation-free
// entry point); otherwise the library's growing workspace.
// real_half (real_in only; 0 off, 1 on, 2 on + dsp::shift): log2n is then the length of the COMPLEX transform the rows ride
// (half their padded length) and the last pass unfolds the full spectrum, rows of 2^(log2n+1) bins.
int launch_large(const c32 *in, c32 *out, int log2n, size_t batch, int direction, hipStream_t s,
                 const float *real_in = nullptr, size_t real_n_in = 0, size_t real_stride = 0,
                 c32 *fixed_scratch = nullptr, int real_half = 0) {
    const size_t n = static_cast<size_t>(1) << log2n;
    // (rows of 2^17 real samples: two passes would leave the unfolding one 256-point columns, 16 + 16 to a tile and
    // 128-byte runs -- three passes with 512-byte runs are faster, 165 -> 183 GSamples/s)
    const int steps = log2n <= 16 && !(real_half && log2n == 16 && real_half_short_last()) ? 2 : (log2n <= 24 ? 3 : 4);
    int lg[4] = {0, 0, 0, 0};
    for (int i = 0, left = log2n; i < steps; ++i) {
        lg[i] = (left + (steps - i) - 1) / (steps - i);
        left -= lg[i];
    }
    if (real_half && real_half_short_last() && lg[steps - 1] >= 7 && log2n - 6 <= 8 * (steps - 1)) {
        // The unfolding last pass moves its tile as two mirrored halves, so its memory runs are half as long as a plain
        // pass's: 128-point columns leave it 256-byte runs.  Give it 64-point columns (512-byte runs with the wide tile)
        // where the earlier passes can take up the difference (2^22 real samples: 140 -> 164 GSamples/s).
        lg[steps - 1] = 6;
        for (int i = 0, left = log2n - 6; i + 1 < steps; ++i) {
            lg[i] = (left + (steps - 1 - i) - 1) / (steps - 1 - i);
            left -= lg[i];
        }
    }
    c32 *scratch = fixed_scratch;
    if (!scratch) {
        void *scratch_v = nullptr;
        if (int rc = workspace(WS_SCRATCH_A, batch * n * sizeof(c32), &scratch_v, s)) return rc;
        scratch = static_cast<c32 *>(scratch_v);
    }
    // destination of pass i: the last one writes `out` and they alternate backwards; in place, the first pass
    // may not write what it reads, so the chain starts in scratch and a copy finishes it if it ends there
    const bool in_place = in == out && !real_in;
    const c32 *src = real_in ? reinterpret_cast<const c32 *>(real_in) : in;
    int done_log2 = 0, rest_log2 = log2n;
    for (int i = 0; i < steps; ++i) {
        rest_log2 -= lg[i];
        const size_t ni = static_cast<size_t>(1) << lg[i], done = static_cast<size_t>(1) << done_log2,
                     rest = static_cast<size_t>(1) << rest_log2;
        const bool to_out = in_place ? (i % 2 == 1) : ((steps - 1 - i) % 2 == 0);
        c32 *dst = to_out ? out : scratch;
        ColumnPlan p{rest * done, done, ni * done, done, i + 1 < steps ? lg[i] + rest_log2 : 0};
        if (i == 0 && real_in) {  // the real -> complex packing rides the first pass's loads
            p.real_n_in = real_n_in;
            p.real_stride = real_stride;
            p.real_pairs = real_half ? 1 : 0;
        }
        if (real_half && i + 1 == steps) {
            p.unfold = real_half;
            p.out_a = 2 * n;
        }
        if (int rc = launch_column(lg[i], src, dst, p, batch, direction, s)) return rc;
        src = dst;
        done_log2 += lg[i];
    }
    if (src != out) DSPGPU_HIP(hipMemcpyAsync(out, src, batch * n * sizeof(c32), hipMemcpyDeviceToDevice, s));
    return DSP_GPU_OK;
}

}  // namespace

namespace {
// DSP_GPU_BIG_P32=0 (A/B): 16 points per thread and radix-16 passes (three exchanges) instead of 32 and two
bool big_p32() {
    const char *e = getenv("DSP_GPU_BIG_P32");
    return !e || e[0] != '0';
}

template <int LOG2N, bool REAL_IN, bool SHIFTED, bool REAL_HALF, int P>
int launch_big_p(const void *in, c32 *out, size_t batch, size_t n_in, size_t in_stride, int direction, hipStream_t s) {
    constexpr int N = 1 << LOG2N;
    const size_t lds_bytes = static_cast<size_t>(N + N / P) * sizeof(c32);
    const size_t chunk = (batch + 7) / 8;
    // Resident workgroups per XCD, each walking its share of the batch with the next transform prefetched -- except complex
    // 2^13, which runs one workgroup per transform: 307 -> 336 GSamples/s (A/B on one box; plain copies stream slowest from
    // persistent grids, profiles/r02_copy_ceiling.txt).  The others lost 3-10 % that way: with one workgroup per CU, or
    // the unfold's extra LDS phase, the prefetch is what hides their loads.
    const size_t per_xcd = (LOG2N == 13 && !REAL_IN) ? chunk : std::min<size_t>(chunk, 32 * (LOG2N == 12 ? 4 : (LOG2N == 13 ? 2 : 1)));
    auto kf = fft_big_kernel<LOG2N, +1, REAL_IN, SHIFTED, REAL_HALF, P>;
    auto ki = fft_big_kernel<LOG2N, -1, REAL_IN, SHIFTED, REAL_HALF, P>;
    auto kern = direction == DSP_GPU_FORWARD ? kf : ki;
    if (int rc = allow_full_lds(reinterpret_cast<const void *>(kern))) return rc;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(8 * per_xcd)), dim3(N / P), lds_bytes, s, in, out, batch, chunk,
                       n_in, in_stride);
    return check_launch("fft_big_kernel");
}

template <int LOG2N, bool REAL_IN, bool SHIFTED, bool REAL_HALF = false>
int launch_big(const void *in, c32 *out, size_t batch, size_t n_in, size_t in_stride, int direction, hipStream_t s) {
    // 32 points per thread where it was measured faster: 2^14 points (complex 309 -> 341 GSamples/s, rows of 2^15 real
    // samples 430 -> 441).  2^13 lost 2 % with it (two workgroups per CU already hide its exchanges).
    if constexpr (LOG2N == 14)
        if (big_p32()) return launch_big_p<LOG2N, REAL_IN, SHIFTED, REAL_HALF, 32>(in, out, batch, n_in, in_stride, direction, s);
    return launch_big_p<LOG2N, REAL_IN, SHIFTED, REAL_HALF, 16>(in, out, batch, n_in, in_stride, direction, s);
}
}  // namespace

namespace {
template <bool REAL_HALF, bool SHIFTED>
int launch_plane(const void *in, c32 *out, size_t batch, size_t n_in, size_t in_stride, int direction, hipStream_t s) {
    const size_t chunk = (batch + 7) / 8;
    const size_t per_xcd = std::min<size_t>(chunk, 32);  // one resident workgroup per CU (one workgroup per transform: no gain)
    auto kf = fft_plane_kernel<+1, REAL_HALF, SHIFTED>;
    auto ki = fft_plane_kernel<-1, REAL_HALF, SHIFTED>;
    auto kern = direction == DSP_GPU_FORWARD ? kf : ki;
    if (int rc = allow_full_lds(reinterpret_cast<const void *>(kern))) return rc;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(8 * per_xcd)), dim3(kPlaneThreads), kPlaneLdsBytes, s, in, out, batch, chunk,
                       n_in, in_stride);
    return check_launch("fft_plane_kernel");
}

// DSP_GPU_FFT_PLANE=0 (A/B, tests of both): 2^15-point transforms (and rows of 2^16 real samples) take the two-pass route
bool plane_route() {
    const char *e = getenv("DSP_GPU_FFT_PLANE");
    return !e || e[0] != '0';
}
}  // namespace

int launch_fft_batch(const float2 *in_f2, float2 *out_f2, int log2n, size_t batch, int direction, hipStream_t s) {
    const c32 *in = reinterpret_cast<const c32 *>(in_f2);
    c32 *out = reinterpret_cast<c32 *>(out_f2);
    if (batch == 0) return DSP_GPU_OK;
    if (log2n <= 12) return dispatch_small<false, false>(log2n, in, out, batch, 0, 0, direction, s);
    if (log2n == 13) return launch_big<13, false, false>(in, out, batch, 0, 0, direction, s);  // one HBM pass: 338 against 187 GSamples/s
    if (log2n == 14) return launch_big<14, false, false>(in, out, batch, 0, 0, direction, s);  //               341 against 190
    if (log2n == 15 && plane_route()) return launch_plane<false, false>(in, out, batch, 0, 0, direction, s);
    return launch_large(in, out, log2n, batch, direction, s);
}

// FFT along axis 0 of a row-major [n][cols] matrix (n = 2^log2n <= 65536), natural order out.
//   n <= 256: one column pass.  Otherwise n = na*nb:
//   pass 1  [na][nb*cols]   FFT over na, twiddle W_n^{ka*nb_index}, store -> [nb][ka][cols]
//   pass 2  [nb][ka*cols]   FFT over nb                              -> [kb][ka][cols] = [k][cols]
int launch_fft_columns(const float2 *in_f2, float2 *out_f2, int log2n, size_t cols, int direction, hipStream_t s) {
    const c32 *in = reinterpret_cast<const c32 *>(in_f2);
    c32 *out = reinterpret_cast<c32 *>(out_f2);
    if (log2n < 4 || log2n > 16) return fail(DSP_GPU_EINVAL, "fft_columns: log2n %d outside [4,16]", log2n);
    if (in == out) return fail(DSP_GPU_EINVAL, "fft_columns: in-place is not supported");
    {
        // A column pass of length L moves 4096 / L adjacent columns per workgroup.  Narrower matrices (small transforms
        // spread over many ranks: 2^10 points on one rank give 32 columns against a tile of 128) take the general route:
        // transpose, transform the rows, transpose back.
        if (!fft_columns_tiled(log2n, cols)) {
            const size_t n = static_cast<size_t>(1) << log2n;
            void *tv = nullptr;
            if (int rc = workspace(WS_SCRATCH_B, n * cols * sizeof(c32), &tv, s)) return rc;
            float2 *t = static_cast<float2 *>(tv);
            if (int rc = launch_transpose_blocks(in_f2, t, 1, n, cols, s)) return rc;   // t[col][row]
            if (int rc = launch_fft_batch(t, t, log2n, cols, direction, s)) return rc;  // in place (scratch slot A above 2^14)
            return launch_transpose_blocks(t, out_f2, 1, cols, n, s);                   // out[row][col]
        }
    }
    if (log2n <= 8) {
        const size_t n = static_cast<size_t>(1) << log2n;
        const ColumnPlan p{cols, cols, 0, cols, 0};
        (void)n;
        return launch_column(log2n, in, out, p, 1, direction, s);
    }
    const int la = (log2n + 1) / 2, lb = log2n - la;
    const size_t na = static_cast<size_t>(1) << la, nb = static_cast<size_t>(1) << lb;
    void *scratch_v = nullptr;
    if (int rc = workspace(WS_SCRATCH_B, (na * nb * cols) * sizeof(c32), &scratch_v, s)) return rc;
    c32 *scratch = static_cast<c32 *>(scratch_v);
    const ColumnPlan p1{nb * cols, cols, na * cols, cols, log2n};
    if (int rc = launch_column(la, in, scratch, p1, 1, direction, s)) return rc;
    const ColumnPlan p2{na * cols, na * cols, 0, na * cols, 0};
    return launch_column(lb, scratch, out, p2, 1, direction, s);
}

// The middle phase of the distributed transform in two passes instead of four (dsp_amd/dist_fft.py):
//   in  [G][R][C]  block g = the columns n2 in [g*C, (g+1)*C) of my R rows (row r is k1 = row0 + r)
//   out [G][R][C]  block g = the columns k2 in [g*C, (g+1)*C) of  Z[r][k2] = sum_n2 W_N2^{n2 k2} W_N^{n2 k1} in[r][n2]
// N2 = G*C = N2a*N2b.  Pass 1 reads straight out of the received blocks (row axis split per rank), applies the
// inter-step twiddle on load, transforms over n2a and stores [r][n2b][k2a]; pass 2 transforms over n2b and
// stores straight into the blocks to send.
namespace {
struct RowsExchangeShape {
    int la, lb;
    size_t na, nb, C, cols1, cols2;
    bool ok;
};
RowsExchangeShape rows_exchange_shape(int log2n2, size_t world, size_t chunks) {
    RowsExchangeShape q{};
    q.la = (log2n2 + 1) / 2, q.lb = log2n2 - q.la;
    const size_t n2 = static_cast<size_t>(1) << log2n2;
    q.na = static_cast<size_t>(1) << q.la, q.nb = static_cast<size_t>(1) << q.lb;
    q.ok = false;
    if (q.la > 8 || q.lb < 4 || world == 0 || (world & (world - 1)) || n2 % world || chunks == 0 || (chunks & (chunks - 1))) return q;
    q.C = n2 / world;
    q.cols1 = (q.la >= 7 && q.nb % (2 * kTile / q.na) == 0 ? 2 : 1) * kTile / q.na;  // columns per workgroup, pass 1
    q.cols2 = (q.lb >= 7 && q.na % (2 * kTile / q.nb) == 0 ? 2 : 1) * kTile / q.nb;
    // a column chunk must hold whole runs of n2b (the chunk index is then a bit field of n2a)
    q.ok = q.C >= q.na && q.nb % q.cols1 == 0 && q.na % q.cols2 == 0 && q.C % chunks == 0 && (q.C / chunks) % q.nb == 0;
    return q;
}
}  // namespace

bool fft_rows_exchange_covers(int log2n2, size_t world, size_t chunks) { return rows_exchange_shape(log2n2, world, chunks).ok; }

bool fft_columns_tiled(int log2n, size_t cols) {
    const int la = log2n <= 8 ? log2n : (log2n + 1) / 2, lb = log2n <= 8 ? 0 : log2n - la;
    const size_t na = static_cast<size_t>(1) << la, nb = static_cast<size_t>(1) << lb;
    return (nb * cols) % (kTile >> la) == 0 && (lb == 0 || (na * cols) % (kTile >> lb) == 0);
}

int launch_fft_rows_exchange_chunk(const float2 *in_f2, float2 *out_f2, int log2n2, size_t rows, size_t world, int tw_log2n,
                                   unsigned long long row0, int direction, hipStream_t s, size_t chunks, size_t rows_total,
                                   size_t row_off) {
    const c32 *in = reinterpret_cast<const c32 *>(in_f2);
    c32 *out = reinterpret_cast<c32 *>(out_f2);
    const RowsExchangeShape q = rows_exchange_shape(log2n2, world, chunks);
    if (q.la > 8 || q.lb < 4 || world == 0 || (world & (world - 1)) || (static_cast<size_t>(1) << log2n2) % world)
        return fail(DSP_GPU_EUNSUPPORTED, "fft_rows_exchange: 2^%d columns over %zu ranks", log2n2, world);
    if (!q.ok || in == out || row_off + rows > rows_total)
        return fail(DSP_GPU_EUNSUPPORTED, "fft_rows_exchange: shape not covered by the fused passes");
    const size_t n2 = static_cast<size_t>(1) << log2n2, na = q.na, nb = q.nb, C = q.C, Cc = C / chunks;
    void *scratch_v = nullptr;
    if (int rc = workspace(WS_SCRATCH_A, rows * n2 * sizeof(c32), &scratch_v, s)) return rc;
    c32 *scratch = static_cast<c32 *>(scratch_v);
    int rlog = 0, klog = 0, olog = 0;
    while ((nb << rlog) < Cc) ++rlog;                        // n2a values per (sender, column chunk)
    while ((static_cast<size_t>(1) << klog) < chunks) ++klog;  // column chunks per sender
    while ((na << olog) < C) ++olog;                         // k2b values per block to send
    // n2 = n2a*nb + n2b = g*C + i*Cc + rest*nb + n2b  ->  in[i][g][row][rest*nb + n2b]
    ColumnPlan p1{nb, 1, na, 1, log2n2};
    p1.in_a = Cc;
    p1.in_lo = nb;
    p1.in_split = rlog;
    p1.in_mid = world * rows_total * Cc;
    p1.in_mid_bits = klog;
    p1.in_hi = rows_total * Cc;
    p1.in_tw_log2 = tw_log2n;
    p1.in_tw_row0 = row0 + row_off;
    if (int rc = launch_column(q.la, in + row_off * Cc, scratch, p1, rows, direction, s)) return rc;
    ColumnPlan p2{na, na, 0, na, 0};
    p2.out_a = C;
    p2.out_hi = rows * C;
    p2.out_split = olog;
    return launch_column(q.lb, scratch, out, p2, rows, direction, s);
}

int launch_fft_rows_exchange(const float2 *in_f2, float2 *out_f2, int log2n2, size_t rows, size_t world, int tw_log2n,
                             unsigned long long row0, int direction, hipStream_t s) {
    return launch_fft_rows_exchange_chunk(in_f2, out_f2, log2n2, rows, world, tw_log2n, row0, direction, s, 1, rows, 0);
}

// DSP_GPU_REAL_HALF_14=0/1 (A/B): whether rows of 2^13 and 2^14 real samples take the half-length form too
static bool real_half_14() {
    const char *e = getenv("DSP_GPU_REAL_HALF_14");
    return e ? e[0] == '1' : kRealHalf14Default;
}

static bool real_half_large() {  // DSP_GPU_REAL_HALF_LARGE=0: the full-length route (A/B, tests of both)
    const char *e = getenv("DSP_GPU_REAL_HALF_LARGE");
    return !e || e[0] != '0';
}

int launch_fft_real_batch(const float *in, size_t n_in, size_t in_stride, float2 *out_f2, int log2n, size_t batch,
                          int direction, int shifted, hipStream_t s) {
    c32 *out = reinterpret_cast<c32 *>(out_f2);
    if (batch == 0) return DSP_GPU_OK;
    if (log2n <= 12) {
        if (shifted) return dispatch_small<true, true>(log2n, in, out, batch, n_in, in_stride, direction, s);
        return dispatch_small<true, false>(log2n, in, out, batch, n_in, in_stride, direction, s);
    }
    if (log2n == 15)  // 2^15 real samples ride one 2^14-point complex transform: still a single HBM pass
        return shifted ? launch_big<14, true, true, true>(in, out, batch, n_in, in_stride, direction, s)
                       : launch_big<14, true, false, true>(in, out, batch, n_in, in_stride, direction, s);
    if (log2n == 13 && real_half_14())  // 2^13 real samples ride a 4096-point transform: 382 -> 458 GSamples/s
        return shifted ? launch_big<12, true, true, true>(in, out, batch, n_in, in_stride, direction, s)
                       : launch_big<12, true, false, true>(in, out, batch, n_in, in_stride, direction, s);
    if (log2n == 14 && real_half_14())
        return shifted ? launch_big<13, true, true, true>(in, out, batch, n_in, in_stride, direction, s)
                       : launch_big<13, true, false, true>(in, out, batch, n_in, in_stride, direction, s);
    if (log2n == 13 || log2n == 14) {  // one HBM pass
        if (log2n == 13)
            return shifted ? launch_big<13, true, true>(in, out, batch, n_in, in_stride, direction, s)
                           : launch_big<13, true, false>(in, out, batch, n_in, in_stride, direction, s);
        return shifted ? launch_big<14, true, true>(in, out, batch, n_in, in_stride, direction, s)
                       : launch_big<14, true, false>(in, out, batch, n_in, in_stride, direction, s);
    }
    if (log2n == 16 && real_half_large() && plane_route())  // 2^16 real samples ride one 2^15-point transform: a single HBM pass
        return shifted ? launch_plane<true, true>(in, out, batch, n_in, in_stride, direction, s)
                       : launch_plane<true, false>(in, out, batch, n_in, in_stride, direction, s);
    if (real_half_large())  // the rows ride a half-length complex transform; the last pass unfolds (and shifts) the spectrum
        return launch_large(nullptr, out, log2n - 1, batch, direction, s, in, n_in, in_stride, nullptr, shifted ? 2 : 1);
    const size_t n = static_cast<size_t>(1) << log2n;
    const size_t total = batch * n;
    const unsigned grid = static_cast<unsigned>(std::min<size_t>((total + kThreads - 1) / kThreads, 2048 * 8));
    if (int rc = launch_large(nullptr, out, log2n, batch, direction, s, in, n_in, in_stride)) return rc;
    if (!shifted) return DSP_GPU_OK;
    hipLaunchKernelGGL(swap_halves_kernel, dim3(grid), dim3(kThreads), 0, s, out, n, total / 2);
    return check_launch("swap_halves_kernel");
}

// ---- the allocation-free form (dsp_gpu_fft_batch_noalloc) -------------------------------------------------------
// Transforms of more than 2^14 points ping-pong between `out` and a scratch buffer.  Callers that cannot hipMalloc
// (HIP graph capture, serving loops) get that buffer from this file-scope device array instead of the growing
// workspace: 2^21 complex points = 16 MiB, held by every GPU the library runs on until the process ends.  The name
// `scratch` is only the host-side handle of the symbol; its per-device address comes from hipGetSymbolAddress, once
// per device, and only that pointer is handed to kernels.  Nothing in it has to start at zero (every pass overwrites
// what the next one reads); all calls on one GPU share it, so calls that overlap on one GPU race.
constexpr size_t kNoAllocPoints = static_cast<size_t>(1) << 21;
static __device__ __align__(256) __attribute__((used)) unsigned char scratch[kNoAllocPoints * sizeof(c32)];

namespace {
constexpr int kNoAllocDevices = 64;
std::mutex g_noalloc_mutex;
c32 *g_noalloc_ptr[kNoAllocDevices] = {};

int noalloc_scratch(c32 **ptr) {
    int dev = 0;
    DSPGPU_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kNoAllocDevices) return fail(DSP_GPU_EUNSUPPORTED, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (!g_noalloc_ptr[dev]) {  // once per device, with that device current: every GPU holds its own copy of the array
        void *p = nullptr;
        DSPGPU_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(DSP_GPU_EHIP, "hipGetSymbolAddress returned a null scratch address");
        g_noalloc_ptr[dev] = static_cast<c32 *>(p);
    }
    *ptr = g_noalloc_ptr[dev];
    return DSP_GPU_OK;
}
}  // namespace

size_t fft_noalloc_max_points() { return kNoAllocPoints; }

int launch_fft_batch_noalloc(const float2 *in_f2, float2 *out_f2, int log2n, size_t batch, int direction, hipStream_t s) {
    if (batch == 0) return DSP_GPU_OK;
    if (log2n <= 14 || (log2n == 15 && plane_route()))  // single pass: no scratch at all
        return launch_fft_batch(in_f2, out_f2, log2n, batch, direction, s);
    if (batch > (kNoAllocPoints >> log2n))
        return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_batch_noalloc: %zu transforms of 2^%d points exceed the fixed scratch of 2^21 points",
                    batch, log2n);
    c32 *ptr = nullptr;
    if (int rc = noalloc_scratch(&ptr)) return rc;
    return launch_large(reinterpret_cast<const c32 *>(in_f2), reinterpret_cast<c32 *>(out_f2), log2n, batch, direction, s,
                        nullptr, 0, 0, ptr);
}

}  // namespace dspgpu
