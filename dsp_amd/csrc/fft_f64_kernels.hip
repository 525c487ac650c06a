// fft_f64_kernels.hip -- dsp::fft0 in the reference's own sample type, std::complex<double> (reference fft.h:31-32
// declares fft0 for double only; everything else in this library computes in fp32, which is what BASELINE's metric is
// quoted on).  Same conventions as the fp32 path: FORWARD = e^{+i 2 pi k n / N}, nothing normalises (fft.h:22-29,40-54),
// natural order in and out (Stockham autosort instead of the bit-reversal copy of fft.h:37-39).
//
//   N <= 4096   fft64_tile_kernel: 256 threads x 16 register-resident points = one 4096-point tile (4096/N transforms),
//               radix-16 Stockham passes, points exchanged through a padded LDS image of 68 KB (two workgroups per CU).
//               32 B of HBM traffic per point: bound by HBM like its fp32 sibling (fp64 FMA rate is not the limit).
//               Like the fp32 kernel: each XCD walks a contiguous eighth of the tiles, every global access is a use-once
//               (non-temporal) 16-byte one, and rows shorter than 128 points move their tile coalesced through the image.
//   N <= 2^18   N = N1*N2, both <= 512: two passes of the same kernel, the transposes folded into its loads and stores
//               (a tile of 4096/N1 adjacent columns, then a tile of 4096/N2 adjacent rows written transposed): 64 B/point.
//   N <= 2^24   N = N1*N2*N3, each 2^6 .. 2^8: three passes of that kernel (columns over n1, columns over n2 inside every
//               row k1, rows over n3 written as [k3][k2][k1]): 96 B/point.
//   DSP_GPU_F64_TRANSPOSES=1 (A/B partner, the first form of this route): N = N1*N2, both <= 4096, explicit transposes:
//               [N1][N2] -T-> [N2][N1] -rows N1, x W_N^{n2 k1}-> -T-> [N1][N2] -rows N2-> -T-> X[k1 + N1 k2]
//               five passes of 32 B per point (the twiddle rides the first row pass's stores).  Plain and correct rather
//               than fast: this is the widening route, the fused column passes of the fp32 path are not repeated here.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace dspgpu {
namespace {

#define D64_DEV __device__ __forceinline__

struct c64 {
    double x, y;
};
D64_DEV c64 operator+(c64 a, c64 b) { return c64{a.x + b.x, a.y + b.y}; }
D64_DEV c64 operator-(c64 a, c64 b) { return c64{a.x - b.x, a.y - b.y}; }
D64_DEV c64 cmul64(c64 a, c64 b) { return c64{fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x)}; }
template <int SIGN>
D64_DEV c64 mul_i64(c64 a) {  // a * (SIGN i)
    return SIGN > 0 ? c64{-a.y, a.x} : c64{a.y, -a.x};
}

constexpr int kT64 = 256;                      // threads per workgroup
constexpr int kP64 = 16;                       // points per thread
constexpr int kTile64 = kT64 * kP64;           // 4096 points per workgroup
constexpr size_t kLds64Bytes = (kTile64 + kTile64 / 16) * sizeof(c64);  // 69,632
D64_DEV int padded64(int i) { return i + (i >> 4); }
D64_DEV c64 load_stream64(const c64 *p) {
    const auto v = load_stream<double>(reinterpret_cast<const double *>(p));
    return c64{v.x, v.y};
}
D64_DEV void store_stream64(c64 *p, c64 v) {
    typename Smp<double>::vec w = {v.x, v.y};
    store_stream<double>(reinterpret_cast<double *>(p), w);
}

constexpr double cos16d(int q) {
    constexpr double t[16] = {1.0, 0.92387953251128673848, 0.70710678118654752440, 0.38268343236508977173,
                              0.0, -0.38268343236508977173, -0.70710678118654752440, -0.92387953251128673848,
                              -1.0, -0.92387953251128673848, -0.70710678118654752440, -0.38268343236508977173,
                              0.0, 0.38268343236508977173, 0.70710678118654752440, 0.92387953251128673848};
    return t[q & 15];
}
constexpr double sin16d(int q) { return cos16d(q + 12); }

template <int SIGN>
D64_DEV c64 mul_w16d(c64 a, int q) {  // a * exp(SIGN 2 pi i q / 16), q constant after unrolling
    q &= 15;
    if (q == 0) return a;
    if (q == 4) return mul_i64<SIGN>(a);
    if (q == 8) return c64{-a.x, -a.y};
    if (q == 12) return mul_i64<-SIGN>(a);
    return cmul64(a, c64{cos16d(q), SIGN * sin16d(q)});
}

template <int SIGN>
D64_DEV void dft2d(c64 &a, c64 &b) {
    const c64 t = a;
    a = t + b;
    b = t - b;
}
template <int SIGN>
D64_DEV void dft4d(c64 &a, c64 &b, c64 &c, c64 &d) {
    const c64 t0 = a + c, t1 = a - c, t2 = b + d, t3 = mul_i64<SIGN>(b - d);
    a = t0 + t2;
    b = t1 + t3;
    c = t0 - t2;
    d = t1 - t3;
}

// in-register DFT of v[0..R), natural order in and out
template <int R, int SIGN>
D64_DEV void dft_regs(c64 (&v)[R]) {
    if constexpr (R == 2) {
        dft2d<SIGN>(v[0], v[1]);
    } else if constexpr (R == 4) {
        dft4d<SIGN>(v[0], v[1], v[2], v[3]);
    } else if constexpr (R == 8) {  // 4 x 2: n = 2 n1 + n2, k = k1 + 4 k2
        dft4d<SIGN>(v[0], v[2], v[4], v[6]);
        dft4d<SIGN>(v[1], v[3], v[5], v[7]);
        v[3] = mul_w16d<SIGN>(v[3], 2);
        v[5] = mul_w16d<SIGN>(v[5], 4);
        v[7] = mul_w16d<SIGN>(v[7], 6);
        dft2d<SIGN>(v[0], v[1]);
        dft2d<SIGN>(v[2], v[3]);
        dft2d<SIGN>(v[4], v[5]);
        dft2d<SIGN>(v[6], v[7]);
        c64 o[8];  // X[k1 + 4 k2] sits at v[2 k1 + k2]
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            o[k1] = v[2 * k1];
            o[k1 + 4] = v[2 * k1 + 1];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = o[i];
    } else {  // 16 = 4 x 4: n = 4 n1 + n2, k = k1 + 4 k2
        static_assert(R == 16, "radix");
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) dft4d<SIGN>(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);  // A[k1][n2] at v[4 k1 + n2]
#pragma unroll
        for (int k1 = 1; k1 < 4; ++k1) {
#pragma unroll
            for (int n2 = 1; n2 < 4; ++n2) v[4 * k1 + n2] = mul_w16d<SIGN>(v[4 * k1 + n2], n2 * k1);
        }
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) dft4d<SIGN>(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
        c64 o[16];  // X[k1 + 4 k2] sits at v[4 k1 + k2]
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) o[k1 + 4 * k2] = v[4 * k1 + k2];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = o[i];
    }
}

// exp(SIGN 2 pi i num / 2^log2den); the angle is an exact dyadic fraction in double
template <int SIGN>
D64_DEV c64 root64(unsigned long long num, int log2den) {
    double sn, cs;
    sincospi(static_cast<double>(num) * __builtin_ldexp(2.0, -log2den), &sn, &cs);
    return c64{cs, SIGN > 0 ? sn : -sn};
}

constexpr int radix64_at(int log2n, int pass) {  // greedy 16, 16, ..., remainder
    const int left = log2n - 4 * pass;
    return left >= 4 ? 16 : (left > 0 ? (1 << left) : 1);
}

// One Stockham pass on the 16 points of a thread (see fft_device.h, stockham_pass): T = N/16 threads per transform,
// r[m] = element lt + T m; the thread owns B = 16/R butterflies j = lt + T s.
template <int R, int NS, int LOG2N, int SIGN>
D64_DEV void pass64(c64 (&r)[kP64], int lt) {
    constexpr int T = (1 << LOG2N) / kP64, B = kP64 / R;
#pragma unroll
    for (int s = 0; s < B; ++s) {
        c64 v[R];
#pragma unroll
        for (int t = 0; t < R; ++t) v[t] = r[s + t * B];
        if constexpr (NS > 1) {
            constexpr int LOG2DEN = __builtin_ctz(NS * R);
            const unsigned k = static_cast<unsigned>(lt + T * s) & (NS - 1);
            c64 pw[R];
            pw[1] = root64<SIGN>(k, LOG2DEN);
#pragma unroll
            for (int t = 2; t < R; ++t) pw[t] = cmul64(pw[t >> 1], pw[t - (t >> 1)]);
#pragma unroll
            for (int t = 1; t < R; ++t) v[t] = cmul64(v[t], pw[t]);
        }
        dft_regs<R, SIGN>(v);
#pragma unroll
        for (int u = 0; u < R; ++u) r[s + u * B] = v[u];
    }
}

template <int R, int NS, int LOG2N>
D64_DEV void exchange64(c64 (&r)[kP64], c64 *lds, int lt, int base) {
    constexpr int T = (1 << LOG2N) / kP64, B = kP64 / R;
#pragma unroll
    for (int s = 0; s < B; ++s) {
        const int j = lt + T * s;
#pragma unroll
        for (int u = 0; u < R; ++u) lds[padded64(base + ((j & ~(NS - 1)) * R) + (j & (NS - 1)) + u * NS)] = r[s + u * B];
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < kP64; ++m) r[m] = lds[padded64(base + lt + T * m)];
}

// 16 <= N <= 4096.  tw_log2 > 0: element k of transform q is also multiplied by exp(SIGN 2 pi i k (q % tw_rows) / 2^tw_log2),
// the inter-step twiddle of the large routes.
// MODE F64_ROWS    transform q is the contiguous row q of `in` / `out` (`batch` rows in all)
//      F64_COLS    `batch` matrices [N][other]: the tile's transforms are 4096/N adjacent COLUMNS, read and written in
//                  place of one another through the LDS image so that HBM sees runs of 4096/N elements; the twiddle row
//                  is the column index (pass 1 of the two-pass route: over n1, times W^{n2 k1})
//      F64_ROWS_T  `batch` arrays [other][groups][N]: the tile's transforms are the rows (r, g) of 4096/N adjacent r at one g,
//                  element k of row (r, g) goes to out[k][g][r] (the last pass: groups == 1 is the plain transposed
//                  store of the two-pass route, X[k1 + N1 k2] at [k2][k1]; with groups == N2 the three-pass route's
//                  X[k1 + N1 (k2 + N2 k3)] at [k3][k2][k1]); in != out
enum F64Mode { F64_ROWS = 0, F64_COLS = 1, F64_ROWS_T = 2 };

template <int LOG2N, int SIGN, int MODE>
__global__ __launch_bounds__(kT64) void fft64_tile_kernel(const c64 *in, c64 *out, size_t batch, int tw_log2, unsigned tw_rows,
                                                          unsigned other, unsigned groups, size_t blocks, size_t chunk) {
    constexpr int N = 1 << LOG2N, T = N / kP64, PER_TILE = kTile64 / N;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    c64 *lds = reinterpret_cast<c64 *>(lds_raw);
    const int tid = threadIdx.x, xf = tid / T, lt = tid % T, base = xf * N;
    const size_t block = xcd_tile_of(blockIdx.x, chunk);  // every XCD walks one contiguous eighth of the tiles
    if (block >= blocks) return;
    // Rows shorter than 128 points give a thread's neighbours runs of less than 128 bytes: their tile (one contiguous run of
    // 4096 points) is moved with coalesced accesses and redistributed through the image instead.
    constexpr bool STAGED = MODE == F64_ROWS && T < 8;
    const size_t tile_e0 = block * kTile64, tile_end = batch * N;
    size_t q;          // F64_ROWS: transform index; otherwise the twiddle row (column index) / unused
    bool live = true;
    const c64 *mat_in = in;
    c64 *mat_out = out;
    unsigned first = 0;  // first column (F64_COLS) / first row (F64_ROWS_T) of the tile inside its matrix
    unsigned grp = 0;    // F64_ROWS_T: the tile's g
    if constexpr (MODE == F64_ROWS) {
        q = block * PER_TILE + xf;
        live = q < batch;
    } else {
        const unsigned tiles = other / PER_TILE * groups;  // the launcher guarantees other % PER_TILE == 0
        const size_t m = block / tiles;
        const unsigned t = static_cast<unsigned>(block % tiles);
        first = (t / groups) * PER_TILE;
        grp = t % groups;
        mat_in = in + m * static_cast<size_t>(N) * other * groups;
        mat_out = out + m * static_cast<size_t>(N) * other * groups;
        q = first + xf;
    }
    c64 r[kP64];
    if constexpr (MODE == F64_COLS) {
#pragma unroll
        for (int i = 0; i < kP64; ++i) {  // element e of column first + c sits at [e][first + c]: threads run along c
            const int idx = tid + kT64 * i, c = idx % PER_TILE, e = idx / PER_TILE;
            lds[padded64(c * N + e)] = load_stream64(mat_in + static_cast<size_t>(e) * other + first + c);
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < kP64; ++m) r[m] = lds[padded64(base + lt + T * m)];
        __syncthreads();
    } else if constexpr (STAGED) {
        c64 v[kP64];
#pragma unroll
        for (int i = 0; i < kP64; ++i) {  // no branch around the loads: a point behind the batch is aimed at in[0] and zeroed
            const size_t e = tile_e0 + tid + kT64 * i;
            v[i] = load_stream64(in + (e < tile_end ? e : 0));
            if (e >= tile_end) v[i] = c64{0.0, 0.0};
        }
#pragma unroll
        for (int i = 0; i < kP64; ++i) lds[padded64(tid + kT64 * i)] = v[i];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < kP64; ++m) r[m] = lds[padded64(base + lt + T * m)];
        __syncthreads();
    } else if (live) {
        const c64 *src = (MODE == F64_ROWS ? in + q * N : mat_in + (static_cast<size_t>(first + xf) * groups + grp) * N) + lt;
#pragma unroll
        for (int m = 0; m < kP64; ++m) r[m] = load_stream64(src + T * m);
    } else {
#pragma unroll
        for (int m = 0; m < kP64; ++m) r[m] = c64{0.0, 0.0};
    }
    constexpr int R0 = radix64_at(LOG2N, 0), R1 = radix64_at(LOG2N, 1), R2 = radix64_at(LOG2N, 2);
    pass64<R0, 1, LOG2N, SIGN>(r, lt);
    if constexpr (R1 > 1) {
        exchange64<R0, 1, LOG2N>(r, lds, lt, base);
        pass64<R1, R0, LOG2N, SIGN>(r, lt);
        if constexpr (R2 > 1) {
            __syncthreads();
            exchange64<R1, R0, LOG2N>(r, lds, lt, base);
            pass64<R2, R0 * R1, LOG2N, SIGN>(r, lt);
        }
    }
    if (MODE == F64_ROWS && !STAGED && !live) return;
    if (tw_log2 > 0) {
        // W^{row k}, k = lt + T m:  W^{row lt} (W^{row T})^m -- two sincos and a product tree
        const unsigned long long mask = (1ull << tw_log2) - 1ull, row = q % tw_rows;
        const c64 w0 = root64<SIGN>((row * static_cast<unsigned long long>(lt)) & mask, tw_log2);
        c64 pw[kP64];
        pw[1] = root64<SIGN>((row * static_cast<unsigned long long>(T)) & mask, tw_log2);
#pragma unroll
        for (int m = 2; m < kP64; ++m) pw[m] = cmul64(pw[m >> 1], pw[m - (m >> 1)]);
        r[0] = cmul64(r[0], w0);
#pragma unroll
        for (int m = 1; m < kP64; ++m) r[m] = cmul64(r[m], cmul64(w0, pw[m]));
    }
    if constexpr (STAGED) {
        __syncthreads();  // the last exchange's image has been read by everyone
#pragma unroll
        for (int m = 0; m < kP64; ++m) lds[padded64(base + lt + T * m)] = r[m];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kP64; ++i) {
            const size_t e = tile_e0 + tid + kT64 * i;
            if (e < tile_end) store_stream64(out + e, lds[padded64(tid + kT64 * i)]);
        }
    } else if constexpr (MODE == F64_ROWS) {
        c64 *dst = out + q * N + lt;
#pragma unroll
        for (int m = 0; m < kP64; ++m) store_stream64(dst + T * m, r[m]);
    } else {
        // through the image once more so that threads run along the tile's columns (F64_COLS) / rows (F64_ROWS_T), which
        // are adjacent in the output: element e of transform c goes to out[e][first + c], rows `other` long
        __syncthreads();
#pragma unroll
        for (int m = 0; m < kP64; ++m) lds[padded64(base + lt + T * m)] = r[m];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kP64; ++i) {
            const int idx = tid + kT64 * i, c = idx % PER_TILE, e = idx / PER_TILE;
            const size_t row = MODE == F64_COLS ? static_cast<size_t>(e) : static_cast<size_t>(e) * groups + grp;
            store_stream64(mat_out + row * other + first + c, lds[padded64(c * N + e)]);
        }
    }
}

// N = 2, 4, 8: one thread per transform
template <int LOG2N, int SIGN>
__global__ __launch_bounds__(kT64) void fft64_tiny_kernel(const c64 *in, c64 *out, size_t batch) {
    constexpr int N = 1 << LOG2N;
    const size_t q = static_cast<size_t>(blockIdx.x) * kT64 + threadIdx.x;
    if (q >= batch) return;
    c64 v[N];
#pragma unroll
    for (int m = 0; m < N; ++m) v[m] = in[q * N + m];
    dft_regs<N, SIGN>(v);
#pragma unroll
    for (int m = 0; m < N; ++m) out[q * N + m] = v[m];
}

// out[b][col][row] = in[b][row][col]; 32 x 32 tiles through LDS
__global__ __launch_bounds__(kT64) void transpose64_kernel(const c64 *__restrict__ in, c64 *__restrict__ out, size_t rows,
                                                           size_t cols) {
    __shared__ c64 tile[32][33];
    const size_t tiles_c = (cols + 31) / 32, tiles_r = (rows + 31) / 32;
    const size_t t = blockIdx.x;
    const size_t b = t / (tiles_r * tiles_c), rem = t % (tiles_r * tiles_c);
    const size_t r0 = (rem / tiles_c) * 32, c0 = (rem % tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const c64 *src = in + b * rows * cols;
    c64 *dst = out + b * rows * cols;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t r = r0 + ty + 8 * k, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + 8 * k][tx] = src[r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t c = c0 + ty + 8 * k, r = r0 + tx;
        if (r < rows && c < cols) dst[c * rows + r] = tile[tx][ty + 8 * k];
    }
}

template <int LOG2N, int MODE>
int launch_tile(const c64 *in, c64 *out, size_t batch, int direction, int tw_log2, unsigned tw_rows, unsigned other,
                unsigned groups, hipStream_t s) {
    constexpr size_t per_tile = kTile64 >> LOG2N;
    size_t blocks;
    if (MODE == F64_ROWS) {
        blocks = (batch + per_tile - 1) / per_tile;
    } else {
        if (other % per_tile != 0) return fail(DSP_GPU_EINVAL, "fft64_tile_kernel: %u is not a multiple of the tile's %zu", other, per_tile);
        blocks = batch * (other / per_tile) * groups;
    }
    const size_t chunk = (blocks + 7) / 8;  // tiles per XCD
    if (8 * chunk > 0x7fffffffull) return fail(DSP_GPU_EUNSUPPORTED, "fft_batch_f64: batch too large for one launch");
    auto kf = fft64_tile_kernel<LOG2N, +1, MODE>;
    auto ki = fft64_tile_kernel<LOG2N, -1, MODE>;
    auto kern = direction == DSP_GPU_FORWARD ? kf : ki;
    if (int rc = allow_full_lds(reinterpret_cast<const void *>(kern))) return rc;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(8 * chunk)), dim3(kT64), kLds64Bytes, s, in, out, batch, tw_log2, tw_rows, other,
                       groups, blocks, chunk);
    return check_launch("fft64_tile_kernel");
}

template <int LOG2N>
int launch_tiny(const c64 *in, c64 *out, size_t batch, int direction, hipStream_t s) {
    const size_t blocks = (batch + kT64 - 1) / kT64;
    if (blocks > 0x7fffffffull) return fail(DSP_GPU_EUNSUPPORTED, "fft_batch_f64: batch too large for one launch");
    if (direction == DSP_GPU_FORWARD)
        hipLaunchKernelGGL((fft64_tiny_kernel<LOG2N, +1>), dim3(static_cast<unsigned>(blocks)), dim3(kT64), 0, s, in, out, batch);
    else
        hipLaunchKernelGGL((fft64_tiny_kernel<LOG2N, -1>), dim3(static_cast<unsigned>(blocks)), dim3(kT64), 0, s, in, out, batch);
    return check_launch("fft64_tiny_kernel");
}

int rows64(const c64 *in, c64 *out, int log2n, size_t batch, int direction, int tw_log2, unsigned tw_rows, hipStream_t s) {
    switch (log2n) {
    case 1: return launch_tiny<1>(in, out, batch, direction, s);
    case 2: return launch_tiny<2>(in, out, batch, direction, s);
    case 3: return launch_tiny<3>(in, out, batch, direction, s);
#define D64_CASE(K) \
    case K: return launch_tile<K, F64_ROWS>(in, out, batch, direction, tw_log2, tw_rows, 0, 1, s);
        D64_CASE(4) D64_CASE(5) D64_CASE(6) D64_CASE(7) D64_CASE(8) D64_CASE(9) D64_CASE(10) D64_CASE(11) D64_CASE(12)
#undef D64_CASE
    }
    return fail(DSP_GPU_EINVAL, "fft_batch_f64: row length 2^%d", log2n);
}

// The two passes of the two-pass route (N1, N2 in 2^4 .. 2^9 so that a tile spans at least 8 adjacent columns / rows):
// `mats` matrices [2^log2n][other], transforms along the columns, in place of one another, times W_N^{col k} ...
int cols64(const c64 *in, c64 *out, int log2n, size_t mats, unsigned other, int direction, int tw_log2, hipStream_t s) {
    switch (log2n) {
#define D64_CASE(K) \
    case K: return launch_tile<K, F64_COLS>(in, out, mats, direction, tw_log2, other, other, 1, s);
        D64_CASE(4) D64_CASE(5) D64_CASE(6) D64_CASE(7) D64_CASE(8) D64_CASE(9)
#undef D64_CASE
    }
    return fail(DSP_GPU_EINVAL, "fft_batch_f64: column length 2^%d", log2n);
}
// ... and `mats` arrays [other][groups][2^log2n], transforms along the rows, written as [k][groups][other]
int rows_t64(const c64 *in, c64 *out, int log2n, size_t mats, unsigned other, unsigned groups, int direction, hipStream_t s) {
    switch (log2n) {
#define D64_CASE(K) \
    case K: return launch_tile<K, F64_ROWS_T>(in, out, mats, direction, 0, 1, other, groups, s);
        D64_CASE(4) D64_CASE(5) D64_CASE(6) D64_CASE(7) D64_CASE(8) D64_CASE(9)
#undef D64_CASE
    }
    return fail(DSP_GPU_EINVAL, "fft_batch_f64: row length 2^%d", log2n);
}

int transpose64(const c64 *in, c64 *out, size_t batch, size_t rows, size_t cols, hipStream_t s) {
    const size_t tiles = batch * ((rows + 31) / 32) * ((cols + 31) / 32);
    if (tiles > 0x7fffffffull) return fail(DSP_GPU_EUNSUPPORTED, "fft_batch_f64: batch too large for one launch");
    hipLaunchKernelGGL(transpose64_kernel, dim3(static_cast<unsigned>(tiles)), dim3(kT64), 0, s, in, out, rows, cols);
    return check_launch("transpose64_kernel");
}

}  // namespace

// complex double interleaved, transform b at in + b * 2^log2n; in == out allowed
int launch_fft_batch_f64(const double *in_d, double *out_d, int log2n, size_t batch, int direction, hipStream_t s) {
    const c64 *in = reinterpret_cast<const c64 *>(in_d);
    c64 *out = reinterpret_cast<c64 *>(out_d);
    if (batch == 0) return DSP_GPU_OK;
    if (log2n <= 12) return rows64(in, out, log2n, batch, direction, 0, 1, s);  // a tile holds whole transforms: in place is safe
    if (log2n > 24) return fail(DSP_GPU_EUNSUPPORTED, "fft_batch_f64: 2^%d points (the double route stops at 2^24)", log2n);
    const int l1 = (log2n + 1) / 2, l2 = log2n - l1;
    const size_t n1 = static_cast<size_t>(1) << l1, n2 = static_cast<size_t>(1) << l2, n = n1 * n2;
    void *scratch_v = nullptr;
    if (int rc = workspace(WS_SCRATCH_A, batch * n * sizeof(c64), &scratch_v, s)) return rc;
    c64 *scratch = static_cast<c64 *>(scratch_v);
    static const bool with_transposes = [] {  // DSP_GPU_F64_TRANSPOSES=1: the five-pass form below (A/B partner)
        const char *e = std::getenv("DSP_GPU_F64_TRANSPOSES");
        return e && e[0] == '1';
    }();
    if (!with_transposes && log2n <= 18) {
        // two passes, the transposes folded into the tiles' loads and stores: x[n1][n2] -> over n1 down the columns, times
        // W_N^{n2 k1} -> scratch[k1][n2] -> over n2 along the rows, written transposed -> out[k2][k1] = X[k1 + N1 k2].
        // Pass 1 has consumed `in` before pass 2 writes `out`, so in == out needs nothing extra.
        if (int rc = cols64(in, scratch, l1, batch, static_cast<unsigned>(n2), direction, log2n, s)) return rc;
        return rows_t64(scratch, out, l2, batch, static_cast<unsigned>(n1), 1, direction, s);
    }
    if (!with_transposes) {
        // three passes, N = N1 * N2 * N3 (each 2^6 .. 2^8 here): x[n1][m], m = n2 N3 + n3.  Down the columns over n1,
        // times W_N^{m k1} -> scratch[k1][n2][n3]; inside every row k1, down the columns over n2, times W_M^{n3 k2}, in
        // place -> scratch[k1][k2][n3]; along the rows over n3, written as out[k3][k2][k1] = X[k1 + N1 (k2 + N2 k3)].
        const int t3 = log2n / 3, t2 = (log2n - t3) / 2, t1 = log2n - t3 - t2;
        const size_t m = static_cast<size_t>(1) << (t2 + t3);
        if (int rc = cols64(in, scratch, t1, batch, static_cast<unsigned>(m), direction, log2n, s)) return rc;
        if (int rc = cols64(scratch, scratch, t2, batch << t1, 1u << t3, direction, t2 + t3, s)) return rc;
        return rows_t64(scratch, out, t3, batch, 1u << t1, 1u << t2, direction, s);
    }
    // x[n1][n2]: transpose, transform over n1 (rows of the transposed matrix) with the twiddle W_N^{n2 k1} on the way out,
    // transpose, transform over n2, transpose: X[k2][k1] = X[k1 + N1 k2].  a and b alternate so that the last hop lands
    // in `out`; in place, the first hop goes to scratch and a copy finishes the chain.
    c64 *a = in == out ? scratch : out, *b = in == out ? out : scratch;
    if (int rc = transpose64(in, a, batch, n1, n2, s)) return rc;                                                  // a[n2][n1]
    if (int rc = rows64(a, a, l1, batch * n2, direction, log2n, static_cast<unsigned>(n2), s)) return rc;         // a[n2][k1]
    if (int rc = transpose64(a, b, batch, n2, n1, s)) return rc;                                                   // b[k1][n2]
    if (int rc = rows64(b, b, l2, batch * n1, direction, 0, 1, s)) return rc;                                      // b[k1][k2]
    if (int rc = transpose64(b, a, batch, n1, n2, s)) return rc;                                                   // a[k2][k1]
    if (a != out) DSPGPU_HIP(hipMemcpyAsync(out, a, batch * n * sizeof(c64), hipMemcpyDeviceToDevice, s));
    return DSP_GPU_OK;
}

}  // namespace dspgpu
