// common.h -- shared host-side helpers of libdsp_gpu (error plumbing, launch helpers).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <functional>

#include "../../include/dsp_gpu.h"

namespace dspgpu {

// Per-thread description of the last failure (returned by dsp_gpu_last_error_string).
char *last_error_buffer();
constexpr size_t kLastErrorBytes = 512;

inline int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buffer(), kLastErrorBytes, fmt, ap);
    va_end(ap);
    return code;
}

#define DSPGPU_HIP(expr)                                                                        \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return ::dspgpu::fail(DSP_GPU_EHIP, "%s failed: %s (%s:%d)", #expr,                \
                                  hipGetErrorString(e_), __FILE__, __LINE__);                   \
    } while (0)

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DSP_GPU_EHIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return DSP_GPU_OK;
}

// ---- streaming helpers shared by the kernels (device side) ------------------------------------
#if defined(__HIPCC__)
typedef float dsp_f32x4 __attribute__((ext_vector_type(4)));
// use-once global accesses, 16 bytes per lane (p must be 16-byte aligned)
__device__ __forceinline__ float4 load_stream_f4(const float *p) {
    const dsp_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const dsp_f32x4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store_stream_f4(float *p, float a, float b, float c, float d) {
    dsp_f32x4 v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<dsp_f32x4 *>(p));
}
// 16-byte groups of samples of either type (SampleType float or double, sample_buffer.h:20)
template <typename T>
struct Smp;
template <>
struct Smp<float> {
    typedef float vec __attribute__((ext_vector_type(4)));
    static constexpr int kVec = 4;
};
template <>
struct Smp<double> {
    typedef double vec __attribute__((ext_vector_type(2)));
    static constexpr int kVec = 2;
};
template <typename T>
__device__ __forceinline__ typename Smp<T>::vec load_stream(const T *p) {  // use-once access, 16 bytes per lane (p 16-byte aligned)
    return __builtin_nontemporal_load(reinterpret_cast<const typename Smp<T>::vec *>(p));
}
template <typename T>
__device__ __forceinline__ void store_stream(T *p, typename Smp<T>::vec v) {
    __builtin_nontemporal_store(v, reinterpret_cast<typename Smp<T>::vec *>(p));
}
// workgroup -> tile map giving each XCD one contiguous eighth of the tiles (see fft_device.h)
__device__ __forceinline__ size_t xcd_tile_of(unsigned block, size_t chunk) {
    return static_cast<size_t>(block & 7u) * chunk + (block >> 3);
}
#endif

// Device workspace owned by the library: one growing buffer per (device, STREAM, slot) -- calls on different streams
// never share scratch memory (capi.hip).  Slots keep independent users (host staging in/out, FFT scratch, ...) of one
// call from aliasing.  `s` is the stream the caller launches on.
enum WorkspaceSlot {
    WS_STAGE_IN = 0, WS_STAGE_OUT, WS_SCRATCH_A, WS_SCRATCH_B, WS_STATE, WS_STATE_SNAP, WS_IIR_TABLE, WS_FIR_HIST, WS_COUNTERS, WS_SLOTS
};
int workspace(WorkspaceSlot slot, size_t bytes, void **ptr, hipStream_t s);
// Lets kernel `fn` use up to the CU's whole 160 KiB of dynamic LDS; issued once per kernel and device, not per launch.
int allow_full_lds(const void *fn);
void release_all_workspaces();
bool iir_tables_in_use(int dev);
void release_iir_tables(int dev);  // cached IIR scan tables of device `dev`, which must be current and idle (dsp_gpu_shutdown)

// ---- the host-pointer route (host_path.hip) ------------------------------------------------------------------
// A batch of `units` independent units (transforms / channels) living in host memory: unit u of the input starts at
// in + u * in_unit (in_total bytes in all: the last one may be shorter than in_unit), unit u of the output at
// out + u * out_unit.  launch(d_in, d_out, unit0, count) enqueues on the caller's stream the kernels for units
// [unit0, unit0 + count), whose staged input / output start at d_in / d_out.
// `aux` (may be null): a small host array read before and rewritten after the call as a whole (IIR state); its device
// copy is handed to every launch as d_aux.
struct HostJob {
    const char *in;
    size_t in_unit, in_total;
    char *out;
    size_t out_unit;
    size_t units;
    void *aux;
    size_t aux_bytes;
    std::function<int(const void *d_in, void *d_out, size_t unit0, size_t count, void *d_aux)> launch;
};
int run_host_job(const HostJob &job, hipStream_t s);  // returns when `out` is complete
bool host_pipe_in_use(int dev);
void release_host_pipes(int dev);

// ---- kernel launchers (device pointers, asynchronous on `s`) --------------------------
int launch_fft_batch(const float2 *in, float2 *out, int log2n, size_t batch, int direction, hipStream_t s);
int launch_fft_batch_f64(const double *in, double *out, int log2n, size_t batch, int direction, hipStream_t s);
int launch_fft_batch_noalloc(const float2 *in, float2 *out, int log2n, size_t batch, int direction, hipStream_t s);
size_t fft_noalloc_max_points();
int launch_fft_real_batch(const float *in, size_t n_in, size_t in_stride, float2 *out, int log2n,
                          size_t batch, int direction, int shifted, hipStream_t s);
int launch_fft_columns(const float2 *in, float2 *out, int log2n, size_t cols, int direction, hipStream_t s);
int launch_fft_rows_exchange(const float2 *in, float2 *out, int log2n2, size_t rows, size_t world, int tw_log2n,
                             unsigned long long row0, int direction, hipStream_t s);
int launch_swap_axes(const float2 *in, float2 *out, size_t a, size_t b, size_t c, int tw_log2,
                     unsigned long long row0, int direction, hipStream_t s);
int launch_transpose_blocks(const float2 *in, float2 *out, size_t blocks, size_t rows, size_t cols, hipStream_t s);
// Chunk-wise forms for the pipelined distributed transform (dist_plan.hip):
//   swap_axes_window       as swap_axes, but the innermost rows are windows of `c` elements in rows in_row / out_row long
//                          (the pointers aim at the window); twiddle column of (a, ., c) = a*tw_c + tw_c0 + c
//   transpose_blocks_into  out[col][blk][row_off + row], output rows out_rows long
//   fft_rows_exchange_chunk  rows [row_off, row_off + rows) of the middle phase; `in` = [chunks][world][rows_total][C/chunks]
//                          (column chunk i of every sender's block), out = [world][rows][C]
int launch_swap_axes_window(const float2 *in, float2 *out, size_t a, size_t b, size_t c, size_t in_row, size_t out_row,
                            int tw_log2, unsigned long long row0, size_t tw_c, size_t tw_c0, int direction, hipStream_t s);
int launch_transpose_blocks_into(const float2 *in, float2 *out, size_t blocks, size_t rows, size_t cols, size_t out_rows,
                                 size_t row_off, hipStream_t s);
int launch_fft_rows_exchange_chunk(const float2 *in, float2 *out, int log2n2, size_t rows, size_t world, int tw_log2n,
                                   unsigned long long row0, int direction, hipStream_t s, size_t chunks, size_t rows_total,
                                   size_t row_off);
bool fft_rows_exchange_covers(int log2n2, size_t world, size_t chunks);  // shapes the fused middle phase takes
bool fft_columns_tiled(int log2n, size_t cols);                           // column passes (true) or the transpose route
int launch_fir_direct(const float *x, float *y, size_t channels, size_t n, const float *d_taps,
                      int ntaps, const float *hist, int zero_head, hipStream_t s);
int launch_fir_direct_valu(const float *x, float *y, size_t channels, size_t n, const float *d_taps, int ntaps,
                           hipStream_t s);
int launch_fir_history(const float *x, size_t n, const float *old_hist, float *new_hist, size_t channels,
                       int ntaps, hipStream_t s);
int launch_fir_overlap_save(const float *x, float *y, size_t channels, size_t n, const float2 *d_hspec,
                            int ntaps, const float *hist, int zero_head, int log2n, hipStream_t s);
int launch_fir_os_segment(const float *x, float *y, size_t channels, size_t n, const float2 *d_hspec, int seg_len,
                          size_t seg0, int accumulate, int log2n, hipStream_t s);
int launch_fir_spectrum(const float *d_taps, int ntaps, float2 *d_hspec, int log2n, hipStream_t s);
int launch_moving_average(const float *x, float *y, size_t channels, size_t n, int npoints, hipStream_t s);
int launch_moving_average_f64(const double *x, double *y, size_t channels, size_t n, int npoints, hipStream_t s);
// direct FIR on double samples with double taps (fir_f64_kernels.hip, v_mfma_f64_16x16x4_f64); ntaps <= 4096
int launch_fir_direct_f64(const double *x, double *y, size_t channels, size_t n, const double *d_taps, int ntaps, hipStream_t s);

struct IirCoef {  // a Coefficients<Poles> without its history (coefficients.h:7-15), double as in the reference
    double a[21];  // up to Coefficients<20> (recursive_filters.h:30-31 designs 2..20 poles)
    double b[21];
    double c0, d0;
    int poles;
};
// One coefficient set per channel (device pointers): a, b are [channels][poles + 1] (b[.][0] ignored), c0, d0 [channels].
struct IirPerChannel {
    const double *a, *b, *c0, *d0;
};
// per_channel == nullptr: `co` for every channel; otherwise only co.poles is read from `co` (poles <= 8).
int launch_iir_batch(const float *x, float *y, size_t channels, size_t n, const IirCoef &co, double *state,
                     const IirPerChannel *per_channel, hipStream_t s);
// The same kernel on double samples (iir_batch_ext.hip), and in the reference's own arithmetic -- coefficients.h:106-130 /
// 172-182 operation for operation, no fused multiply-adds, the generic section's accumulator in the sample type.
int launch_iir_batch_f64(const double *x, double *y, size_t channels, size_t n, const IirCoef &co, double *state,
                         const IirPerChannel *per_channel, hipStream_t s);
int launch_iir_batch_ref(const float *x, float *y, size_t channels, size_t n, const IirCoef &co, double *state,
                         const IirPerChannel *per_channel, hipStream_t s);
int launch_iir_batch_ref_f64(const double *x, double *y, size_t channels, size_t n, const IirCoef &co, double *state,
                             const IirPerChannel *per_channel, hipStream_t s);
int launch_iir_scan_f64(const double *x, double *y, size_t channels, size_t n, const IirCoef &co, double *state, hipStream_t s);
int iir_settle_samples(const IirCoef &co, size_t *samples);
int launch_iir_sharded(const float *x, float *y, size_t channels, size_t n, const IirCoef &co, double *state, hipStream_t s);
int launch_iir_scan(const float *x, float *y, size_t channels, size_t n, const IirCoef &co, double *state,
                    hipStream_t s);
// A band-pass (kind 0) / band-reject (kind 1) BiQuad redesigned at every sample from fc[ch * fc_stride + t]
// (iir_swept_kernels.hip); state [channels][4] = {x1, x2, y1, y2}, may be null.
int launch_biquad_swept(const float *x, const float *fc, float *y, size_t channels, size_t n, size_t fc_stride, int kind,
                        double bandwidth, double *state, hipStream_t s);

}  // namespace dspgpu
