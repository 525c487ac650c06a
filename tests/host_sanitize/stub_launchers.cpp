// stub_launchers.cpp -- host stand-ins for the kernel launchers of dsp_amd/csrc/common.h (sanitizer builds only, see
// hip/hip_runtime.h beside this file).  They compute nothing meaningful; each one READS every byte its kernel reads and
// WRITES every byte its kernel writes, so that a staging buffer, workspace or history array of the wrong size under the
// real host code in capi.hip / host_path.hip shows up in AddressSanitizer.
#include <cstdint>
#include <cstring>

#include "../../dsp_amd/csrc/common.h"

namespace dspgpu {
namespace {

// folds `bytes` at p into a checksum (a read ASan sees) and returns a byte derived from it
unsigned char touch(const void *p, size_t bytes) {
    const unsigned char *q = static_cast<const unsigned char *>(p);
    unsigned acc = 0;
    for (size_t i = 0; i < bytes; ++i) acc = acc * 31u + q[i];
    return static_cast<unsigned char>(acc);
}
void fill(void *p, size_t bytes, unsigned char v) { std::memset(p, v, bytes); }
// y <- f(x): reads all of x, then writes all of y (memmove semantics: in-place calls are legal)
void pass(const void *x, size_t xb, void *y, size_t yb) {
    if (x == y && xb == yb) { (void)touch(x, xb); return; }
    const unsigned char v = touch(x, xb);
    if (xb == yb) std::memmove(y, x, yb);
    else fill(y, yb, v);
}

}  // namespace

int launch_fft_batch(const float2 *in, float2 *out, int log2n, size_t batch, int, hipStream_t) {
    pass(in, (batch << log2n) * sizeof(float2), out, (batch << log2n) * sizeof(float2));
    return DSP_GPU_OK;
}
int launch_fft_batch_f64(const double *in, double *out, int log2n, size_t batch, int, hipStream_t) {
    pass(in, (batch << log2n) * 16, out, (batch << log2n) * 16);
    return DSP_GPU_OK;
}
int launch_fft_batch_noalloc(const float2 *in, float2 *out, int log2n, size_t batch, int d, hipStream_t s) {
    return launch_fft_batch(in, out, log2n, batch, d, s);
}
size_t fft_noalloc_max_points() { return size_t(1) << 21; }
int launch_fft_real_batch(const float *in, size_t n_in, size_t in_stride, float2 *out, int log2n, size_t batch, int, int, hipStream_t) {
    unsigned char v = 0;
    const size_t n = size_t(1) << log2n;
    for (size_t b = 0; b < batch; ++b) v ^= touch(in + b * in_stride, (n_in < n ? n_in : n) * sizeof(float));
    fill(out, batch * n * sizeof(float2), v);
    return DSP_GPU_OK;
}
int launch_fft_columns(const float2 *in, float2 *out, int log2n, size_t cols, int, hipStream_t) {
    pass(in, (cols << log2n) * sizeof(float2), out, (cols << log2n) * sizeof(float2));
    return DSP_GPU_OK;
}
int launch_fft_rows_exchange(const float2 *in, float2 *out, int log2n2, size_t rows, size_t, int, unsigned long long, int, hipStream_t) {
    pass(in, (rows << log2n2) * sizeof(float2), out, (rows << log2n2) * sizeof(float2));
    return DSP_GPU_OK;
}
int launch_swap_axes(const float2 *in, float2 *out, size_t a, size_t b, size_t c, int, unsigned long long, int, hipStream_t) {
    pass(in, a * b * c * sizeof(float2), out, a * b * c * sizeof(float2));
    return DSP_GPU_OK;
}
int launch_transpose_blocks(const float2 *in, float2 *out, size_t blocks, size_t rows, size_t cols, hipStream_t) {
    pass(in, blocks * rows * cols * sizeof(float2), out, blocks * rows * cols * sizeof(float2));
    return DSP_GPU_OK;
}
bool fft_rows_exchange_covers(int, size_t, size_t) { return true; }
bool fft_columns_tiled(int, size_t) { return true; }

int launch_fir_direct(const float *x, float *y, size_t channels, size_t n, const float *d_taps, int ntaps, const float *hist,
                      int, hipStream_t) {
    (void)touch(d_taps, sizeof(float) * ntaps);
    if (hist) (void)touch(hist, sizeof(float) * channels * (ntaps - 1));
    pass(x, channels * n * sizeof(float), y, channels * n * sizeof(float));
    return DSP_GPU_OK;
}
int launch_fir_direct_valu(const float *x, float *y, size_t channels, size_t n, const float *d_taps, int ntaps, hipStream_t s) {
    return launch_fir_direct(x, y, channels, n, d_taps, ntaps, nullptr, 0, s);
}
int launch_fir_history(const float *x, size_t n, const float *old_hist, float *new_hist, size_t channels, int ntaps, hipStream_t) {
    const size_t hb = sizeof(float) * channels * (ntaps - 1);
    unsigned char v = touch(x, channels * n * sizeof(float));
    if (old_hist) v ^= touch(old_hist, hb);
    fill(new_hist, hb, v);
    return DSP_GPU_OK;
}
int launch_fir_overlap_save(const float *x, float *y, size_t channels, size_t n, const float2 *d_hspec, int ntaps, const float *hist,
                            int, int log2n, hipStream_t) {
    (void)touch(d_hspec, sizeof(float2) << log2n);
    if (hist) (void)touch(hist, sizeof(float) * channels * (ntaps - 1));
    pass(x, channels * n * sizeof(float), y, channels * n * sizeof(float));
    return DSP_GPU_OK;
}
int launch_fir_os_segment(const float *x, float *y, size_t channels, size_t n, const float2 *d_hspec, int, size_t, int, int log2n,
                          hipStream_t) {
    (void)touch(d_hspec, sizeof(float2) << log2n);
    pass(x, channels * n * sizeof(float), y, channels * n * sizeof(float));
    return DSP_GPU_OK;
}
int launch_fir_spectrum(const float *d_taps, int ntaps, float2 *d_hspec, int log2n, hipStream_t) {
    fill(d_hspec, sizeof(float2) << log2n, touch(d_taps, sizeof(float) * ntaps));
    return DSP_GPU_OK;
}
int launch_moving_average(const float *x, float *y, size_t channels, size_t n, int, hipStream_t) {
    pass(x, channels * n * sizeof(float), y, channels * n * sizeof(float));
    return DSP_GPU_OK;
}
int launch_moving_average_f64(const double *x, double *y, size_t channels, size_t n, int, hipStream_t) {
    pass(x, channels * n * sizeof(double), y, channels * n * sizeof(double));
    return DSP_GPU_OK;
}
int launch_fir_direct_f64(const double *x, double *y, size_t channels, size_t n, const double *d_taps, int ntaps, hipStream_t) {
    (void)touch(d_taps, sizeof(double) * ntaps);
    pass(x, channels * n * sizeof(double), y, channels * n * sizeof(double));
    return DSP_GPU_OK;
}

namespace {
template <typename T>
int iir_any(const T *x, T *y, size_t channels, size_t n, const IirCoef &co, double *state, const IirPerChannel *pc) {
    if (pc) {
        (void)touch(pc->a, sizeof(double) * channels * (co.poles + 1));
        (void)touch(pc->b, sizeof(double) * channels * (co.poles + 1));
        (void)touch(pc->c0, sizeof(double) * channels);
        (void)touch(pc->d0, sizeof(double) * channels);
    }
    if (state) fill(state, sizeof(double) * channels * 2 * co.poles, touch(state, sizeof(double) * channels * 2 * co.poles));
    pass(x, channels * n * sizeof(T), y, channels * n * sizeof(T));
    return DSP_GPU_OK;
}
}  // namespace
int launch_iir_batch(const float *x, float *y, size_t c, size_t n, const IirCoef &co, double *st, const IirPerChannel *pc, hipStream_t) { return iir_any(x, y, c, n, co, st, pc); }
int launch_iir_batch_f64(const double *x, double *y, size_t c, size_t n, const IirCoef &co, double *st, const IirPerChannel *pc, hipStream_t) { return iir_any(x, y, c, n, co, st, pc); }
int launch_iir_batch_ref(const float *x, float *y, size_t c, size_t n, const IirCoef &co, double *st, const IirPerChannel *pc, hipStream_t) { return iir_any(x, y, c, n, co, st, pc); }
int launch_iir_batch_ref_f64(const double *x, double *y, size_t c, size_t n, const IirCoef &co, double *st, const IirPerChannel *pc, hipStream_t) { return iir_any(x, y, c, n, co, st, pc); }
int launch_iir_scan(const float *x, float *y, size_t c, size_t n, const IirCoef &co, double *st, hipStream_t) { return iir_any(x, y, c, n, co, st, nullptr); }
int launch_iir_scan_f64(const double *x, double *y, size_t c, size_t n, const IirCoef &co, double *st, hipStream_t) { return iir_any(x, y, c, n, co, st, nullptr); }
int launch_iir_sharded(const float *x, float *y, size_t c, size_t n, const IirCoef &co, double *st, hipStream_t) { return iir_any(x, y, c, n, co, st, nullptr); }
int iir_settle_samples(const IirCoef &, size_t *samples) { *samples = 256; return DSP_GPU_OK; }
int launch_biquad_swept(const float *x, const float *fc, float *y, size_t channels, size_t n, size_t fc_stride, int, double, double *state,
                        hipStream_t) {
    (void)touch(fc, sizeof(float) * (fc_stride ? (channels - 1) * fc_stride + n : n));
    if (state) fill(state, sizeof(double) * channels * 4, touch(state, sizeof(double) * channels * 4));
    pass(x, channels * n * sizeof(float), y, channels * n * sizeof(float));
    return DSP_GPU_OK;
}
bool iir_tables_in_use(int) { return false; }
void release_iir_tables(int) {}

}  // namespace dspgpu
