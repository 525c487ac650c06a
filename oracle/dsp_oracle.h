/*
 * dsp_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-CPU restatement of the numeric hot
 * path of ejosiah/dsp (FFT, FIR, IIR), used as the parity checker by tests/,
 * __graft_entry__.smoke() and the cpu_baseline leg of bench.py.  Nothing in the
 * product path (dsp_amd/, include/) may include, link or call it.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit (double)
 * against the reference's own headers compiled in oracle/_ref (see
 * oracle/ref_shim.cpp, tests/test_oracle_vs_ref.py) and against the committed
 * fixtures in tests/golden/ that were generated from that build.
 *
 * All "ref:" citations are relative to /root/reference/dsp/include/dsp/.
 * The identical symbol set is exported by oracle/_ref/libdsp_ref.so with the
 * prefix ref_ instead of orc_ (that library executes the reference's code).
 */
#ifndef DSP_ORACLE_H
#define DSP_ORACLE_H

#include <stddef.h>

#ifndef ORC_NAME
#define ORC_NAME(x) orc_##x
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- FFT (ref: fft.h) ---------------------------------------------------- */
/* fft.h:31-55  out-of-place radix-2 DIT, complex double interleaved (re,im).
 * inverse=0 -> FFTType::FORWARD (twiddle e^{+i pi/m2}), 1 -> INVERSE. Unnormalised. */
void ORC_NAME(fft0)(const double *in, double *out, int log2n, int inverse);
void ORC_NAME(fft0_batch)(const double *in, double *out, int log2n, int inverse, size_t batch);
/* same algorithm instantiated in float (NOT in the reference; like-for-like CPU baseline) */
void ORC_NAME(fft0_f32)(const float *in, float *out, int log2n, int inverse);
/* fft.h:57-83  real input wrapper: pads n_in samples to 2^ceil(log2 nf); returns padded size */
int ORC_NAME(fft_real)(const double *td, size_t n_in, double *out, int nf, int inverse);
/* fft.h:85-94  fftshift by copy, complex interleaved */
void ORC_NAME(shift)(const double *in, double *out, size_t n);
/* fft.h:11-20 */
int ORC_NAME(bit_reverse)(int x, int n);

/* ---- FIR design + run (ref: dsp.h, filter.h) ------------------------------ */
/* dsp.h:47-79  window: 0=blackman 1=identity 2=hamming; inversion: 0=SpectralInversion
 * 1=SpectralReversal 2=None (enum order of dsp.h:20).  out must hold (length|1). returns taps */
int ORC_NAME(sinc)(double cf, int length, int window, int inversion, double *out);
/* filter.h:123-140  causal direct convolution, first ntaps-1 outputs zero */
void ORC_NAME(fir_f32)(const float *x, size_t n, const double *h, int ntaps, float *y);
void ORC_NAME(fir_f64)(const double *x, size_t n, const double *h, int ntaps, double *y);
/* filter.h:81-101  centred running-sum moving average (npoints used as given) */
void ORC_NAME(moving_average_f32)(const float *x, size_t n, size_t npoints, float *y);
void ORC_NAME(moving_average_f64)(const double *x, size_t n, size_t npoints, double *y);

/* ---- IIR design (ref: recursive_filters.h) -------------------------------- */
/* recursive_filters.h:24-151  highpass=0 -> LowPass. a,b hold poles+1 doubles. returns 0 / -1 */
int ORC_NAME(chebyshev)(int highpass, double ripple, double fc, int poles, double *a, double *b);
/* recursive_filters.h:165-263  kind: 0 bandPass(f,bw) 1 bandReject(f,bw) 2 lowShelf(f,gain)
 * 3 highShelf(f,gain) 4 peaking(f,gain,q).  a[3], b[3], cd[2] = {c0,d0} */
int ORC_NAME(biquad_design)(int kind, double p0, double p1, double p2, double *a, double *b, double *cd);

/* ---- IIR run (ref: coefficients.h) ---------------------------------------- */
/* coefficients.h:88-130  generic N-pole recursion (accumulates in the sample type).
 * state = x[0..poles], y[0..poles] (2*(poles+1) doubles, in/out, may be NULL = zero state) */
void ORC_NAME(iir_f32)(const float *x, size_t n, const double *a, const double *b, int poles,
                       double c0, double d0, double *state, float *y);
void ORC_NAME(iir_f64)(const double *x, size_t n, const double *a, const double *b, int poles,
                       double c0, double d0, double *state, double *y);
/* coefficients.h:172-195  Coefficients<2> specialisation (all-double arithmetic).
 * state = {x1,x2,y1,y2} in/out, may be NULL */
void ORC_NAME(biquad_f32)(const float *x, size_t n, const double *a, const double *b,
                          double c0, double d0, double *state, float *y);
void ORC_NAME(biquad_f64)(const double *x, size_t n, const double *a, const double *b,
                          double c0, double d0, double *state, double *y);
/* eval/signals/wind_generator.h:150-167: the band-pass (kind 0) / band-reject (kind 1) BiQuad redesigned from
 * fc[t] before every step; state {x1,x2,y1,y2} in/out, may be NULL */
void ORC_NAME(biquad_swept_f32)(const float *x, const float *fc, size_t n, int kind, double bw,
                                double *state, float *y);

#ifdef __cplusplus
}
#endif
#endif
