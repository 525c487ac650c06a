/*
 * dsp_gpu.h -- C ABI of libdsp_gpu.so: the MI355X (gfx950) implementation of the
 * ejosiah/dsp numeric hot path (radix-2 FFT/IFFT, FIR convolution, IIR recursion).
 *
 * The reference has no FFI / plugin seam: its hot path sits directly
 * behind C++ templates.  Each entry point below names the reference function(s) it
 * stands in for (file:line relative to the reference's dsp/include/dsp/).  The C++
 * drop-in overloads that keep the reference's own signatures are in
 * include/dsp_mi355x/ and call only this header.
 *
 * Conventions
 *   - plain pointers and sizes only; `stream` is a hipStream_t passed as void* (NULL =
 *     the default stream).
 *   - `mem` says where the data pointers live: DSP_GPU_MEM_DEVICE (device pointers, the
 *     call is asynchronous on `stream`) or DSP_GPU_MEM_HOST (host pointers, the library
 *     stages through its own device workspace and the call returns when `out` is ready).
 *     Small parameter arrays (taps, a, b) are ALWAYS host pointers.
 *   - every function returns DSP_GPU_OK (0) or a negative DSP_GPU_E* code; nothing
 *     throws or aborts across this boundary.  dsp_gpu_last_error_string() describes the
 *     last failure on the calling thread.
 *   - the caller owns every buffer it passes.  The library owns per-device workspaces
 *     and cached tap uploads, released by dsp_gpu_shutdown().
 *   - Threads and streams.  Intended multi-GPU use is one host thread (or process) per device with that device
 *     made current (hipSetDevice) before the call.  The library's tables (tap cache, IIR plans) are locked and
 *     handed out pinned, and its growing device workspaces (the ping-pong slab of FFTs above 2^15 points, IIR
 *     tile carries and state snapshots, FIR history scratch, the overlap-save FIR's draw counters, host staging) are kept PER (device, stream): calls
 *     on different streams of one device -- from one host thread or several -- never share scratch memory and may
 *     overlap freely (tests/test_streams_gpu.py).  What remains single-file: (i) calls that share ONE stream must
 *     come from one host thread at a time (the usual rule for a stream); (ii) DSP_GPU_MEM_HOST calls on one device
 *     serialise inside the library (one staging pipeline per device); (iii) dsp_gpu_fft_batch_noalloc's fixed
 *     array is one per device: calls to it that overlap on one GPU race.  A device keeps workspace sets for 8
 *     streams; a ninth takes over the least recently used set after one hipDeviceSynchronize.
 */
#ifndef DSP_GPU_H
#define DSP_GPU_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes */
#define DSP_GPU_OK            0
#define DSP_GPU_EINVAL       (-1) /* bad argument (NULL pointer, size out of range, ...) */
#define DSP_GPU_EHIP         (-2) /* a HIP runtime call failed                         */
#define DSP_GPU_ENOMEM       (-3) /* workspace allocation failed                        */
#define DSP_GPU_EUNSUPPORTED (-4) /* valid request this build does not implement        */

/* direction: the reference's FFTType (fft.h:9).  FORWARD uses e^{+i 2 pi k n / N}
 * (fft.h:22-29,44 -- the conjugate of the numpy/FFTW "forward"); neither direction
 * normalises (fft.h:40-54). */
#define DSP_GPU_FORWARD 0
#define DSP_GPU_INVERSE 1

#define DSP_GPU_MEM_HOST   0
#define DSP_GPU_MEM_DEVICE 1

/* convolution mode (the reference's dsp::Domain, dsp.h:14: Time -> DIRECT, Frequency -> OVERLAP_SAVE) */
#define DSP_GPU_CONV_AUTO         0 /* OVERLAP_SAVE from 64 taps up (channels of >= 256 samples; from 40 taps on channels of
                                      >= 262,144 samples) and above 4096 taps, else DIRECT; overlap-save results carry the FFT's error (1e-5*12), not 1e-6 */
#define DSP_GPU_CONV_DIRECT       1
#define DSP_GPU_CONV_OVERLAP_SAVE 2
#define DSP_GPU_CONV_DIRECT_VALU  3 /* direct form on the vector ALU instead of the matrix cores: the A/B partner
                                      kept for measurement (one-shot calls, <= 1024 taps); AUTO never picks it */

/* IIR execution mode */
#define DSP_GPU_IIR_AUTO  0 /* SCAN below 24,576 channels of >= 2,048 samples and for any channels of >= 16,384 samples (<= 8 poles), else BATCH */
#define DSP_GPU_IIR_BATCH 1 /* one thread per channel, channels in parallel          */
#define DSP_GPU_IIR_SCAN  2 /* block-parallel linear-recurrence scan inside a channel */
#define DSP_GPU_IIR_BATCH_REF 3 /* BATCH in the REFERENCE'S OWN ARITHMETIC, operation for operation and in its order, with no
                                   fused multiply-adds: the generic section (coefficients.h:106-130) accumulates in the SAMPLE
                                   type -- with fp32 samples the accumulator is rounded to float after every `+=`, histories
                                   stay double, output = SampleType(acc*c0 + in*d0) -- and poles == 2 is the reference's
                                   Coefficients<2> = BiQuad (coefficients.h:172-182, all double).  Bit-identical to the
                                   reference for float and double samples; AUTO never picks it (AUTO / BATCH / SCAN compute in
                                   double with fused multiply-adds: <= 2e-6 from the reference's double instantiation, but for
                                   a filter as ill-conditioned as lowPassFilter<4>(0.01) 2.5e-3 from what its float
                                   instantiation returns, which is itself that far from its double one).  The dependency
                                   chain per sample is 2*poles + 1 operations instead of one: slower, all four SIMDs used. */

/* sample types of the entry points that take either (void pointers) */
#define DSP_GPU_F32 0
#define DSP_GPU_F64 1

/*
 * Batched complex FFT.  Replaces dsp::fft0<type>(in, out, log2n) (fft.h:31-55) for
 * `batch` independent transforms of n = 2^log2n points, fp32 instead of the
 * reference's double.  Interleaved (re,im) float pairs; transform b is at
 * in + b*2*n floats.  Natural-order output, unnormalised.  in == out is allowed.
 * 1 <= log2n <= 30 (the reference indexes with int, fft.h:36-37: 2^30 is its own ceiling).
 */
int dsp_gpu_fft_batch(const void *in, void *out, int log2n, size_t batch, int direction, int mem,
                      void *stream);

/*
 * The same transform in the reference's own sample type: dsp::fft0<type>(const std::complex<double>*,
 * std::complex<double>*, int log2n) (fft.h:31-32 declares fft0 for double only).  Interleaved (re,im) double pairs,
 * arithmetic in double throughout (twiddles from sincospi on exact dyadic angles), natural-order output, unnormalised,
 * in == out allowed.  1 <= log2n <= 24: one HBM pass up to 4096 points, two up to 2^18, three above (the transposes
 * folded into the tiles' loads and stores), through a library workspace -- the widening route, not the benchmarked
 * one; DSP_GPU_EINVAL beyond 2^24.
 */
int dsp_gpu_fft_batch_f64(const void *in, void *out, int log2n, size_t batch, int direction, int mem,
                          void *stream);

/*
 * The allocation-free form of dsp_gpu_fft_batch.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot
 * allocate device memory -- code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same
 * arithmetic and kernels as dsp_gpu_fft_batch on DEVICE pointers, asynchronous on `stream`, but the scratch buffer that
 * transforms of more than 2^15 points ping-pong through is a fixed 16 MiB device array inside the library (2^21 complex
 * points, held by every GPU the library runs on until the process ends) instead of the growing workspace, and the call
 * makes no hipMalloc / hipFree.  1 <= log2n <= 21; for log2n >= 16, batch * 2^log2n <= 2^21 points (DSP_GPU_EINVAL
 * beyond that; transforms of up to 2^15 points use no scratch and take any batch -- except under the development
 * switch DSP_GPU_FFT_PLANE=0, which sends 2^15-point transforms down the two-pass route and so under the same limit).  Every call on a GPU shares that
 * GPU's array: calls that overlap on one GPU (two streams, two threads) race.  The first call on a device looks up the
 * array's address (hipGetSymbolAddress) -- make it once outside a graph capture.
 */
int dsp_gpu_fft_batch_noalloc(const void *in, void *out, int log2n, size_t batch, int direction, void *stream);

/*
 * Batched real-input FFT wrapper.  Replaces dsp::fft(first, last, c_out, nf) (fft.h:57-83):
 * each of `batch` rows holds n_in real samples (row stride in_stride floats); a row is
 * zero-padded to n = 2^ceil(log2 nf) (samples beyond n are ignored, as in the reference),
 * transformed with fft0 semantics and written as n interleaved complex values.
 * `shifted` != 0 additionally applies dsp::shift (fft.h:85-94) to each output row.
 */
int dsp_gpu_fft_real_batch(const float *in, size_t n_in, size_t in_stride, void *out, int nf,
                           size_t batch, int direction, int shifted, int mem, void *stream);

/*
 * FIR convolution.  Replaces dsp::filter::SincFilter<>::apply (filter.h:123-140) and
 * dsp::convolve<T, Domain> (convolution.h:8-29; its Frequency branch, which the reference
 * only declares, is DSP_GPU_CONV_OVERLAP_SAVE here).  Channel-major x[ch][n], y[ch][n];
 * y[ch][j] = sum_{i<ntaps} x[ch][j-i]*taps[i] for j >= ntaps-1 and 0 before that
 * (filter.h:133 -- causal, first ntaps-1 outputs zero).  `taps` is a HOST pointer to
 * ntaps floats shared by all channels (e.g. the fp32 cast of dsp::sinc, dsp.h:47-79).
 * 1 <= ntaps <= 4096 (direct).  Overlap-save runs on 4096-point blocks up to 2049 taps (256- ... 2048-point
 * blocks for channels short enough to fit one 4096-point block) and on 16384-point blocks up to 8193 taps; longer filters run partitioned -- 8192-tap segments on
 * 16384-point blocks, one pass each, accumulated -- up to 262,144 taps (AUTO takes overlap-save
 * above 4096 taps whatever n is).  Device buffers x and y must not overlap (DSP_GPU_EINVAL);
 * host buffers are staged and may.
 */
int dsp_gpu_convolve(const float *x, float *y, size_t channels, size_t n, const float *taps,
                     int ntaps, int mode, int mem, void *stream);
/*
 * The same filter on DOUBLE samples with DOUBLE taps: SincFilter<>::apply(SampleBuffer<double>&) (filter.h:123-140), the
 * instantiation the reference's own callers use (eval/windowed_sinc_filter/main.cpp:105, eval/filter_compare/main.cpp:11)
 * with the kernel as dsp::sinc returns it (std::vector<double>, dsp.h:47-79 -- no narrowing to fp32).  Time domain on the
 * double matrix cores (v_mfma_f64_16x16x4_f64), products and sums exact double fmas: the result differs from the
 * reference's loop only by the order of the sum (<= 1e-14 relative L2).  1 <= ntaps <= 4096; modes DSP_GPU_CONV_AUTO and
 * DSP_GPU_CONV_DIRECT (DSP_GPU_CONV_OVERLAP_SAVE: DSP_GPU_EUNSUPPORTED -- the frequency-domain form is the fp32 path's).
 */
int dsp_gpu_convolve_f64(const double *x, double *y, size_t channels, size_t n, const double *taps,
                         int ntaps, int mode, int mem, void *stream);

/*
 * Streaming FIR: the same filter fed in chunks with the input tail carried between calls (the FIR
 * counterpart of dsp_gpu_iir's `state`; the reference's block-wise callers are the audio callbacks of
 * eval/recursive_filters/main.cpp:174-206).  `history` is [ch][ntaps-1] floats, the samples that precede
 * x[0] (oldest first), read before and rewritten after the call; it lives where `mem` says.
 * `position` = samples of this stream consumed by earlier calls (the caller adds n after each call).
 * position == 0 starts a stream: `history` is ignored on input (taken as zeros).  Outputs whose stream
 * index position + j is below ntaps-1 are zero exactly as in the one-shot call (filter.h:126,133), all
 * others are   y[ch][j] = sum_i taps[i] * (history ++ x)[ntaps-1 + j - i].
 * Feeding a signal in chunks of any sizes therefore reproduces dsp_gpu_convolve on the whole signal.
 * ntaps <= 4096 (direct), <= 8193 (overlap-save; the partitioned form has no streaming variant).
 */
int dsp_gpu_convolve_stream(const float *x, float *y, size_t channels, size_t n, const float *taps,
                            int ntaps, float *history, unsigned long long position, int mode, int mem,
                            void *stream);

/*
 * Centred moving average.  Replaces dsp::filter::MovingAverageFilter::apply
 * (filter.h:81-101): with mid = npoints/2,
 *     y[i] = (sum_{k=i-mid}^{i+mid} x[k] - (npoints even ? x[npoints] : 0)) / npoints
 * for mid <= i < n-mid, zero elsewhere -- for odd npoints the plain centred mean; for even
 * npoints (reachable through numPoints(), filter.h:72-74) exactly what the reference's running
 * sum evaluates to.  Windows are summed directly (no running sum along the signal), so the
 * result does not carry the reference's accumulated rounding drift.  n >= npoints >= 1, npoints <= 8192.
 */
int dsp_gpu_moving_average(const float *x, float *y, size_t channels, size_t n, int npoints,
                           int mem, void *stream);
/* MovingAverageFilter::apply(SampleBuffer<double>&) (eval/moving_average_filter_demo/main.cpp:14 filters doubles): the same
 * kernel on double samples, sums in double, the mean by division as in the reference (filter.h:93,97). */
int dsp_gpu_moving_average_f64(const double *x, double *y, size_t channels, size_t n, int npoints,
                               int mem, void *stream);

/*
 * IIR recursion.  Replaces dsp::Coefficients<Poles>::operator()(const SampleBuffer&)
 * (coefficients.h:88-130, and the BiQuad form :172-195):
 *     v[t] = sum_{i=0..poles} a[i] x[t-i] + sum_{i=1..poles} b[i] v[t-i]   (feedback ADDED)
 *     y[t] = v[t]*c0 + x[t]*d0
 * a, b: HOST pointers to poles+1 doubles (b[0] ignored), as produced by the designers in
 * recursive_filters.h.  1 <= poles <= 20 (Coefficients<2..20>, recursive_filters.h:30-31); SCAN mode
 * up to 8 poles (AUTO runs longer sections in BATCH).  Channel-major x[ch][n], y[ch][n].
 * Arithmetic and history are double, as in the reference (coefficients.h:9-12); samples are fp32
 * (dsp_gpu_iir) or double (dsp_gpu_iir_f64: SampleBuffer<double>, the type every in-repo caller of the reference
 * filters -- eval/recursive_filters/main.cpp:35.  SCAN on double samples is the one-lane kernel sharded in TIME: every
 * channel cut into segments that start from states computed over twice dsp_gpu_iir_settle samples of warm-up, exact to
 * 2^-88 of the state -- one channel of 2^20 doubles 67 ms -> 0.16 ms; DSP_GPU_EUNSUPPORTED for filters that do not
 * settle within 8,192 samples, channels of fewer than two segments, and more than 64 channels whose length no segment
 * length divides; AUTO then runs BATCH.  AUTO also takes this form for fp32 sections of more than eight poles).
 * DSP_GPU_IIR_BATCH_REF reproduces the reference bit for bit (see the mode's comment).
 * `state` (may be NULL = start from rest, nothing written back) is [ch][2*poles] DOUBLES,
 * {x[t-1..t-poles], v[t-1..t-poles]} per channel, read before and written after the call:
 * it is the reference's x[]/y[] history (coefficients.h:11-12) made explicit so a long
 * channel can be fed in chunks (chunked == unchunked, bit for bit, in BATCH mode).  Lives
 * where `mem` says.
 */
int dsp_gpu_iir(const float *x, float *y, size_t channels, size_t n, const double *a,
                const double *b, int poles, double c0, double d0, double *state, int mode, int mem,
                void *stream);
int dsp_gpu_iir_f64(const double *x, double *y, size_t channels, size_t n, const double *a,
                    const double *b, int poles, double c0, double d0, double *state, int mode, int mem,
                    void *stream);

/*
 * How long a filter remembers: the number of samples (a multiple of 64) after which the response to ANY initial state
 * {x[t-1..], v[t-1..]} has decayed below 2^-44 of it -- ||A^k||_inf <= 2^-44 for the companion matrix of the feedback
 * coefficients b (coefficients.h:116-118).  Host arithmetic only (no GPU work).  It is what splitting ONE long
 * channel over several GPUs in time needs (dsp_amd/multigpu.py iir_time_sharded): every GPU filters
 * its segment from a zero state, the end states are gathered, and each GPU re-runs only its first `samples` outputs
 * from its predecessor's end state -- exact to double rounding when a segment is at least `samples` long.
 * DSP_GPU_EUNSUPPORTED for filters that do not decay within 262,144 samples (marginally stable ones).
 */
int dsp_gpu_iir_settle(const double *a, const double *b, int poles, size_t *samples);

/*
 * The same recursion with ONE COEFFICIENT SET PER CHANNEL (a filter bank: every channel its own
 * Coefficients<poles> object, e.g. a bank of band-pass filters tuned differently, the block-wise form of
 * eval/signals/wind_generator.h:150-167, which retunes its band-pass as it runs).  a, b: [channels][poles+1]
 * doubles (b[.][0] ignored), c0, d0: [channels] doubles or NULL (= 1 and 0, coefficients.h:11-12).  Unlike
 * dsp_gpu_iir's shared a/b these arrays live where `mem` says, like `state` ([channels][2*poles], in/out,
 * may be NULL).  One lane per channel (the BATCH kernel), 1 <= poles <= 8.
 */
int dsp_gpu_iir_multi(const float *x, float *y, size_t channels, size_t n, const double *a, const double *b,
                      int poles, const double *c0, const double *d0, double *state, int mem, void *stream);
/* The same for either sample type (x, y: float or double arrays by `sample_type` = DSP_GPU_F32 / DSP_GPU_F64) and with
 * the arithmetic chosen by `mode`: DSP_GPU_IIR_AUTO / DSP_GPU_IIR_BATCH (double, fused multiply-adds) or
 * DSP_GPU_IIR_BATCH_REF (the reference's own, bit for bit: coefficients.h:106-130, poles == 2: :172-182). */
int dsp_gpu_iir_multi_ex(const void *x, void *y, int sample_type, size_t channels, size_t n, const double *a,
                         const double *b, int poles, const double *c0, const double *d0, double *state, int mode,
                         int mem, void *stream);

/*
 * A band-pass / band-reject BiQuad RETUNED AT EVERY SAMPLE.  Replaces the loop of eval/signals/wind_generator.h:150-167
 * (Whistling::getSample), which per sample designs dsp::recursive::bandPassFilter(fc, bw) (recursive_filters.h:165-186;
 * kind DSP_GPU_SWEPT_BANDREJECT: bandRejectFilter, :188-209), copies its a and b into a BiQuad that keeps its history,
 * and steps that BiQuad once (coefficients.h:172-182):
 *     R = 1 - 3*bw;  c = cos(2*PI*fc[t]);  K = (1 - 2Rc + R^2) / (2 - 2c)
 *     band-pass  a = {1-K, 2(K-R)c, R^2-K}     band-reject  a = {K, -2Kc, K}      b = {0, 2Rc, -R^2}
 *     v = a0 x[t] + a1 x1 + a2 x2;  v += b1 y1 + b2 y2;  (x2,x1,y2,y1) <- (x1,x[t],y1,v);  y[t] = v
 * Design and recursion run on the device in double, operation for operation as in the reference and without fused
 * multiply-adds; only cos() may differ from a CPU's in its last bit.  x, y: [channels][n] fp32.  fc: centre frequency
 * per sample as a fraction of the sample rate, in (0, 0.5], fp32 (the reference computes it in float,
 * wind_generator.h:157-158): [channels][n] with fc_stride == n, or ONE sweep [n] shared by all channels with
 * fc_stride == 0.  fc[t] == 0 divides by zero exactly as the reference does (K = +inf; the outputs turn infinite, then NaN).  bandwidth: as the reference's
 * second argument (fraction of the sample rate).  `state`: [channels][4] doubles {x1, x2, y1, y2} -- the BiQuad's
 * members (coefficients.h:52-68) -- read before, written after, may be NULL (= from rest); it lets a long run be fed in
 * blocks (blocked == unblocked bit for bit while a voice runs on one lane).  y == x (and y == fc when fc_stride == n) is
 * in place.  One lane per channel: tens of thousands of channels fill the chip.  Up to 4,096 LONG voices whose sweeps
 * run along with their samples (one voice, or fc_stride == n) are sharded in time instead: the pole radius R never
 * changes, so segments started from the state that 2 * 44 ln 2 / -ln R samples of warm-up leave agree with the
 * one-lane run to fp32 rounding (2e-6 relative L2) -- one voice of 2^20 samples 194 ms -> 1.65 ms.  With one sweep per channel the kernel is bound by
 * double-precision arithmetic (~60 instructions per sample), not by its 12 B/sample of traffic; a shared sweep is
 * designed once per sample instead of once per channel and sample, and runs at the rate of dsp_gpu_iir's BATCH kernel.
 */
#define DSP_GPU_SWEPT_BANDPASS 0
#define DSP_GPU_SWEPT_BANDREJECT 1
int dsp_gpu_biquad_swept(const float *x, const float *fc, float *y, size_t channels, size_t n, size_t fc_stride,
                         int kind, double bandwidth, double *state, int mem, void *stream);

/*
 * Local steps of the multi-GPU six-step FFT: one dsp::fft0 (fft.h:31-55) of N = N1*N2 points spread
 * over G GPUs, n = n1*N2 + n2, k = k1 + N1*k2 (BASELINE config 5) -- the building blocks that
 * dsp_gpu_fft_dist (below) sequences around its three all-to-all exchanges.  DEVICE pointers
 * only, asynchronous on `stream`, complex fp32 interleaved, in != out.
 *   dsp_gpu_fft_columns       FFT along axis 0 of a row-major [2^log2n][cols] matrix, natural order,
 *                             4 <= log2n <= 16, any cols >= 1.  Column passes (one or two HBM trips) when cols is a
 *                             multiple of the column tile -- 4096 / 2^log2n for log2n <= 8, else 4096 / 2^floor(log2n/2)
 *                             -- and transpose -> row transforms -> transpose through a library workspace otherwise
 *   dsp_gpu_swap_axes         out[b][a][c] = in[a][b][c]; when tw_log2n > 0 each element is also
 *                             multiplied by W^{(tw_row0 + b) * (a*c_extent + c)}, W = e^{+-2 pi i / 2^tw_log2n}
 *                             (sign by `direction`): the inter-step twiddle fused into the unpack
 *   dsp_gpu_transpose_blocks  out[col][blk][row] = in[blk][row][col]
 *   dsp_gpu_fft_rows_exchange the whole middle phase in two HBM passes: in = the `world` received blocks
 *                             [world][rows][C] (block g holds columns n2 in [g*C, (g+1)*C) of my rows, row r
 *                             being k1 = tw_row0 + r), out = the blocks to send, [world][rows][C] with
 *                             Z[r][k2] = sum_n2 W_N2^{n2 k2} W_N^{n2 k1} in[r][n2], N2 = world*C = 2^log2n2,
 *                             N = 2^tw_log2n.  Equals swap_axes(twiddle) -> fft_batch over rows -> swap_axes.
 *                             DSP_GPU_EUNSUPPORTED for shapes the fused passes do not cover (2^12 <= N2 <= 2^16,
 *                             C >= 2^ceil(log2n2/2)): callers fall back to the three separate steps.
 */
int dsp_gpu_fft_columns(const void *in, void *out, int log2n, size_t cols, int direction, void *stream);
int dsp_gpu_swap_axes(const void *in, void *out, size_t a, size_t b, size_t c, int tw_log2n, size_t tw_row0,
                      int direction, void *stream);
int dsp_gpu_transpose_blocks(const void *in, void *out, size_t blocks, size_t rows, size_t cols, void *stream);
int dsp_gpu_fft_rows_exchange(const void *in, void *out, int log2n2, size_t rows, size_t world, int tw_log2n,
                              size_t tw_row0, int direction, void *stream);

/*
 * One transform spread over several GPUs (BASELINE config 5: a single 2^30-point dsp::fft0, fft.h:31-55; the
 * reference has no counterpart -- it transforms on one core).  One plan per rank (= per process / host thread
 * and GPU).  Rank g of `world` (a power of two) holds the contiguous slice x[g*N/G, (g+1)*N/G) in `in` and
 * receives X[g*N/G, (g+1)*N/G), natural order, in `out`; both are DEVICE buffers of
 * dsp_gpu_fft_dist_local_points(plan) = N/G complex fp32 points, in != out, `in` is not modified.
 * 8 <= log2n <= 32; N1 = 2^ceil(log2n/2) and N2 = N/N1 must both be multiples of `world`, N2/world >= 2.
 * Collective: every rank calls dsp_gpu_fft_dist with the same log2n, direction and flags.  The plan owns up to two work
 * buffers of N/G points, a stream and a few events on the device current at its first use (freed by
 * dsp_gpu_fft_dist_plan_destroy); calls on one plan must not overlap.
 *
 *   dsp_gpu_fft_dist_plan_create       `exchange` performs one all-to-all: block g (block_bytes bytes) of `send` goes
 *                                      to rank g, block g of `recv` comes from rank g, enqueued on `stream` -- which
 *                                      is the caller's stream in the serial schedule and a stream of the plan's own in
 *                                      the pipelined one; device pointers; returns 0 or a negative DSP_GPU_E* code.
 *                                      Once the work it enqueued on `stream` has run, `recv` must be complete AND
 *                                      `send` free to be overwritten (grouped ncclSend/ncclRecv, an MPI all-to-all
 *                                      followed by a stream callback ... behave so).  Every rank's exchange is called
 *                                      the same number of times in the same order: dsp_gpu_fft_dist_exchanges() per
 *                                      transform.  NULL: only dsp_gpu_fft_dist_phase (or world == 1) can be used.
 *   dsp_gpu_fft_dist_plan_create_rccl  the built-in exchange: grouped ncclSend/ncclRecv on the caller's
 *                                      communicator (an ncclComm_t whose rank/size equal rank/world) -- one block
 *                                      per peer over the xGMI mesh.  librccl is looked up at run time (dlopen);
 *                                      DSP_GPU_EUNSUPPORTED when it is not there.
 *   dsp_gpu_fft_dist                   the whole transform: four local phases, three exchanges (skipped when
 *                                      world == 1 and the plan has no exchange).  When the shape allows it (N/G cut
 *                                      into 4 or 2 chunks that still fill the kernels' tiles) the schedule is
 *                                      PIPELINED: pack, first exchange, transform over n1 and second exchange run per
 *                                      chunk of columns, transform over n2, third exchange and final transpose per
 *                                      chunk of rows, the exchanges on the plan's stream so that exchange k of one
 *                                      chunk overlaps the local pass of the next.  Same result bit for bit.
 *   dsp_gpu_fft_dist_phase             one local phase (0 pack, 1 FFT over n1, 2 twiddle + FFT over n2 + pack,
 *                                      3 transpose to natural order) for hosts that run the exchange themselves
 *                                      between phases (dsp_amd/dist_fft.py: torch.distributed all_to_all_single).
 */
typedef struct dsp_gpu_fft_dist_plan dsp_gpu_fft_dist_plan;
typedef int (*dsp_gpu_exchange_fn)(void *user, const void *send, void *recv, size_t block_bytes, int world,
                                   void *stream);
int dsp_gpu_fft_dist_plan_create(dsp_gpu_fft_dist_plan **plan, int log2n, int rank, int world,
                                 dsp_gpu_exchange_fn exchange, void *user);
int dsp_gpu_fft_dist_plan_create_rccl(dsp_gpu_fft_dist_plan **plan, int log2n, int rank, int world, void *nccl_comm);
size_t dsp_gpu_fft_dist_local_points(const dsp_gpu_fft_dist_plan *plan);
int dsp_gpu_fft_dist(dsp_gpu_fft_dist_plan *plan, const void *in, void *out, int direction, void *stream);
/*
 * The same with options.  DSP_GPU_DIST_TRANSPOSED_OUT leaves the spectrum where the transform over n2 produces it: rank g
 * receives out[r][k2] = X[(g*N1/G + r) + N1*k2] (N1/G rows of N2 bins) instead of its natural-order slice, which
 * saves the third all-to-all exchange and the final transpose -- the step SURVEY.md 8(e) marks as skippable when a
 * transposed result will do (a spectrum that is only multiplied pointwise and transformed back, a power spectrum ...).
 * A plan without an exchange (one rank) also skips the pack in front of the first exchange, a plain copy there.
 */
#define DSP_GPU_DIST_TRANSPOSED_OUT 1
#define DSP_GPU_DIST_SERIAL 2 /* the unpipelined schedule: phase, exchange, phase ... on the caller's stream alone */
/* number of times a transform with these flags calls the plan's exchange (0 without one) */
int dsp_gpu_fft_dist_exchanges(const dsp_gpu_fft_dist_plan *plan, int flags);
int dsp_gpu_fft_dist_ex(dsp_gpu_fft_dist_plan *plan, const void *in, void *out, int direction, int flags, void *stream);
int dsp_gpu_fft_dist_phase(dsp_gpu_fft_dist_plan *plan, int phase, const void *in, void *out, int direction,
                           void *stream);
int dsp_gpu_fft_dist_plan_destroy(dsp_gpu_fft_dist_plan *plan);

/* housekeeping */
int dsp_gpu_device_count(void);
const char *dsp_gpu_last_error_string(void);
int dsp_gpu_shutdown(void); /* frees workspaces / cached taps of every device this process used */
const char *dsp_gpu_version(void);

/*
 * Kernel-only timing helper for the bench harness: runs `iters` back-to-back launches of
 * the batched FFT on `stream` between two HIP events recorded on that same stream and
 * returns the average milliseconds per launch in *ms_per_launch (device pointers only).
 */
int dsp_gpu_fft_batch_timed(const void *in, void *out, int log2n, size_t batch, int direction,
                            int iters, float *ms_per_launch, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DSP_GPU_H */
