// dist_kernels.hip -- local steps of the multi-GPU (six-step) FFT for gfx950.
//
// One transform of N = N1*N2 points, larger than one GPU wants to hold (BASELINE config 5: 2^30),
// is dsp::fft0 (reference fft.h:31-55) decomposed as   n = n1*N2 + n2,  k = k1 + N1*k2:
//     X[k1 + N1 k2] = sum_n2 W_N2^{n2 k2} * W_N^{n2 k1} * ( sum_n1 x[n1 N2 + n2] W_N1^{n1 k1} )
// Every rank owns a contiguous N/G slice.  The three all-to-all exchanges are done by the host layer
// (dsp_amd/dist_fft.py, RCCL through torch.distributed); the kernels here are what runs between them:
//   fft columns      FFT along axis 0 of a row-major [n][cols] matrix (the sum over n1), built from
//                    the column passes of fft_kernels.hip
//   swap_axes        out[b][a][c] = in[a][b][c], optionally times W_N^{(row0+b)*(a*C+c)}: the pack /
//                    unpack around an exchange with the inter-step twiddle fused
//   transpose_blocks out[col][blk][row] = in[blk][row][col]: the final reordering to natural order
#include <algorithm>

#include "common.h"
#include "fft_device.h"

namespace dspgpu {
namespace {

constexpr int kDistThreads = 256;

// two complex values per thread (16-byte accesses); C is even.  The innermost rows of `in` / `out` may be windows of
// longer rows (in_cs / out_cs float4 apart; the pointers already aim at the window): the pipelined distributed
// transform packs and unpacks a chunk of columns at a time.  The twiddle column of element (a, ., c) is a*tw_c + tw_c0 + c.
template <int SIGN, bool TW>
__global__ __launch_bounds__(kDistThreads) void swap_axes_kernel(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                                 size_t A, size_t B, size_t C2, size_t in_cs, size_t out_cs,
                                                                 int tw_log2, unsigned long long row0,
                                                                 unsigned long long tw_c, unsigned long long tw_c0) {
    const size_t total = A * B * C2;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kDistThreads + threadIdx.x; i < total;
         i += static_cast<size_t>(gridDim.x) * kDistThreads) {
        const size_t c2 = i % C2, ab = i / C2;
        const size_t a = ab / B, b = ab % B;
        float4 v = in[ab * in_cs + c2];
        if (TW) {
            const unsigned long long mask = (1ull << tw_log2) - 1ull;
            const unsigned long long col = a * tw_c + tw_c0 + 2 * c2;
            const unsigned long long row = row0 + b;
            const c32 w0 = unit_root_wide<SIGN>((row * col) & mask, tw_log2);
            const c32 w1 = unit_root_wide<SIGN>((row * (col + 1)) & mask, tw_log2);
            const c32 p0 = cmul(mk(v.x, v.y), w0), p1 = cmul(mk(v.z, v.w), w1);
            v = make_float4(p0.x, p0.y, p1.x, p1.y);
        }
        out[(b * A + a) * out_cs + c2] = v;
    }
}

// 32x32 tiles through LDS; in[blk][row][col] -> out[col][blk][row_off + row], the output rows out_rows long
__global__ __launch_bounds__(kDistThreads) void transpose_blocks_kernel(const c32 *__restrict__ in, c32 *__restrict__ out,
                                                                        size_t blocks, size_t rows, size_t cols,
                                                                        size_t out_rows, size_t row_off) {
    __shared__ c32 tile[32][33];
    const size_t tiles_c = (cols + 31) / 32, tiles_r = (rows + 31) / 32;
    const size_t t = blockIdx.x;
    const size_t blk = t / (tiles_r * tiles_c), rem = t % (tiles_r * tiles_c);
    const size_t r0 = (rem / tiles_c) * 32, c0 = (rem % tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const c32 *src = in + blk * rows * cols;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t r = r0 + ty + 8 * k, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + 8 * k][tx] = src[r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t c = c0 + ty + 8 * k, r = r0 + tx;
        if (r < rows && c < cols) out[(c * blocks + blk) * out_rows + row_off + r] = tile[tx][ty + 8 * k];
    }
}

}  // namespace

int launch_swap_axes_window(const float2 *in_f2, float2 *out_f2, size_t a, size_t b, size_t c, size_t in_row, size_t out_row,
                            int tw_log2, unsigned long long row0, size_t tw_c, size_t tw_c0, int direction, hipStream_t s) {
    const c32 *in = reinterpret_cast<const c32 *>(in_f2);
    c32 *out = reinterpret_cast<c32 *>(out_f2);
    if (c % 2 || in_row % 2 || out_row % 2 || tw_c0 % 2 || in_row < c || out_row < c)
        return fail(DSP_GPU_EINVAL, "swap_axes: inner extent %zu (rows of %zu / %zu) must be even and fit its rows", c, in_row, out_row);
    const size_t total = a * b * (c / 2);
    if (total == 0) return DSP_GPU_OK;
    const unsigned grid = static_cast<unsigned>(std::min<size_t>((total + kDistThreads - 1) / kDistThreads, 1u << 20));
    const float4 *i4 = reinterpret_cast<const float4 *>(in);
    float4 *o4 = reinterpret_cast<float4 *>(out);
    if (tw_log2 <= 0)
        hipLaunchKernelGGL((swap_axes_kernel<+1, false>), dim3(grid), dim3(kDistThreads), 0, s, i4, o4, a, b, c / 2, in_row / 2,
                           out_row / 2, 0, 0ull, 0ull, 0ull);
    else if (direction == DSP_GPU_FORWARD)
        hipLaunchKernelGGL((swap_axes_kernel<+1, true>), dim3(grid), dim3(kDistThreads), 0, s, i4, o4, a, b, c / 2, in_row / 2,
                           out_row / 2, tw_log2, row0, static_cast<unsigned long long>(tw_c), static_cast<unsigned long long>(tw_c0));
    else
        hipLaunchKernelGGL((swap_axes_kernel<-1, true>), dim3(grid), dim3(kDistThreads), 0, s, i4, o4, a, b, c / 2, in_row / 2,
                           out_row / 2, tw_log2, row0, static_cast<unsigned long long>(tw_c), static_cast<unsigned long long>(tw_c0));
    return check_launch("swap_axes_kernel");
}

int launch_swap_axes(const float2 *in_f2, float2 *out_f2, size_t a, size_t b, size_t c, int tw_log2, unsigned long long row0,
                     int direction, hipStream_t s) {
    return launch_swap_axes_window(in_f2, out_f2, a, b, c, c, c, tw_log2, row0, c, 0, direction, s);
}

int launch_transpose_blocks_into(const float2 *in_f2, float2 *out_f2, size_t blocks, size_t rows, size_t cols, size_t out_rows,
                                 size_t row_off, hipStream_t s) {
    const c32 *in = reinterpret_cast<const c32 *>(in_f2);
    c32 *out = reinterpret_cast<c32 *>(out_f2);
    if (row_off + rows > out_rows) return fail(DSP_GPU_EINVAL, "transpose_blocks: rows %zu + %zu beyond %zu", row_off, rows, out_rows);
    const size_t tiles = blocks * ((rows + 31) / 32) * ((cols + 31) / 32);
    if (tiles == 0) return DSP_GPU_OK;
    if (tiles > 0x7fffffffull) return fail(DSP_GPU_EUNSUPPORTED, "transpose_blocks: grid too large");
    hipLaunchKernelGGL(transpose_blocks_kernel, dim3(static_cast<unsigned>(tiles)), dim3(kDistThreads), 0, s, in, out, blocks,
                       rows, cols, out_rows, row_off);
    return check_launch("transpose_blocks_kernel");
}

int launch_transpose_blocks(const float2 *in_f2, float2 *out_f2, size_t blocks, size_t rows, size_t cols, hipStream_t s) {
    return launch_transpose_blocks_into(in_f2, out_f2, blocks, rows, cols, rows, 0, s);
}

}  // namespace dspgpu
