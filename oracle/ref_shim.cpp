/*
 * ref_shim.cpp -- C-callable wrapper around the REAL reference headers.
 *
 * TEST INFRASTRUCTURE ONLY.  This translation unit contains no reference code: it
 * #includes the headers where they lie under /root/reference/dsp/include (passed
 * with -I by oracle/Makefile) and forwards each ref_* entry point to the reference's
 * own templates.  The output, oracle/_ref/libdsp_ref.so, is git-ignored; it is used
 * to (i) prove oracle/dsp_oracle.cpp bit-identical and (ii) generate tests/golden/.
 *
 * The headers define non-inline functions, so they are included in this one TU only,
 * and they rely on <vector>/<cstring>/<stdexcept>/<cassert> being included first
 * (SURVEY.md section 8c).  <dsp/convolution.h> does not compile and is not included.
 */
#include <cassert>
#include <complex>
#include <cstring>
#include <stdexcept>
#include <vector>

#include <dsp/fft.h>
#include <dsp/filter.h>
#include <dsp/recursive_filters.h>

#define ORC_NAME(x) ref_##x
#include "dsp_oracle.h"

namespace {

template <size_t P, typename T>
void run_generic(const T *x, size_t n, const double *a, const double *b, double c0, double d0,
                 double *state, T *y) {
    dsp::Coefficients<P> co{};
    for (size_t i = 0; i <= P; ++i) {
        co.a[i] = a[i];
        co.b[i] = b[i];
        if (state) {
            co.x[i] = state[i];
            co.y[i] = state[P + 1 + i];
        }
    }
    co.c0 = c0;
    co.d0 = d0;
    dsp::SampleBuffer<T> in(x, x + n);
    dsp::SampleBuffer<T> out = co(in);
    for (size_t i = 0; i < n; ++i) y[i] = out[static_cast<int>(i)];
    if (state) {
        for (size_t i = 0; i <= P; ++i) {
            state[i] = co.x[i];
            state[P + 1 + i] = co.y[i];
        }
    }
}

// Coefficients<2> is specialised in the reference (the all-double BiQuad); the generic
// N-pole template cannot be instantiated at P == 2, so generic 2-pole parity is pinned
// through P = 4 with zero upper coefficients where needed.
template <typename T>
void dispatch_generic(const T *x, size_t n, const double *a, const double *b, int poles, double c0,
                      double d0, double *state, T *y) {
    switch (poles) {
    case 4: run_generic<4, T>(x, n, a, b, c0, d0, state, y); break;
    case 6: run_generic<6, T>(x, n, a, b, c0, d0, state, y); break;
    case 8: run_generic<8, T>(x, n, a, b, c0, d0, state, y); break;
    case 10: run_generic<10, T>(x, n, a, b, c0, d0, state, y); break;
    case 12: run_generic<12, T>(x, n, a, b, c0, d0, state, y); break;
    case 20: run_generic<20, T>(x, n, a, b, c0, d0, state, y); break;
    default: break;  // pole count not instantiated here: output left untouched
    }
}

template <typename T>
void run_biquad(const T *x, size_t n, const double *a, const double *b, double c0, double d0,
                double *state, T *y) {
    dsp::BiQuad bq{};
    for (int i = 0; i < 3; ++i) {
        bq.a[i] = a[i];
        bq.b[i] = b[i];
    }
    bq.c0 = c0;
    bq.d0 = d0;
    if (state) {
        bq.x1 = state[0];
        bq.x2 = state[1];
        bq.y1 = state[2];
        bq.y2 = state[3];
    }
    dsp::SampleBuffer<T> in(x, x + n);
    dsp::SampleBuffer<T> out = bq(in);
    for (size_t i = 0; i < n; ++i) y[i] = out[static_cast<int>(i)];
    if (state) {
        state[0] = bq.x1;
        state[1] = bq.x2;
        state[2] = bq.y1;
        state[3] = bq.y2;
    }
}

template <size_t P>
int design_cheby(int highpass, double ripple, double fc, double *a, double *b) {
    auto co = dsp::recursive::chebyshev::computeCoefficients<P>(
        highpass ? dsp::FilterType::HighPass : dsp::FilterType::LowPass, ripple, fc);
    for (size_t i = 0; i <= P; ++i) {
        a[i] = co.a[i];
        b[i] = co.b[i];
    }
    return 0;
}

template <dsp::InversionType Inv>
int make_sinc(double cf, int length, int window, double *out) {
    dsp::Window w = dsp::Windows::blackMan;
    if (window == 1) w = dsp::Windows::identity;
    if (window == 2) w = dsp::Windows::hamming;
    std::vector<double> k = dsp::sinc<Inv>(cf, length, w);
    for (size_t i = 0; i < k.size(); ++i) out[i] = k[i];
    return static_cast<int>(k.size());
}

// SincFilter keeps its kernel private and builds it from (cf, length) only, so arbitrary
// taps are run through a layout-compatible stand-in: the reference's apply() loop
// (filter.h:123-140) is executed unchanged on whatever m_kernel holds.
template <typename T>
void run_fir(const T *x, size_t n, const double *h, int ntaps, T *y) {
    struct Twin {
        double cf;
        std::vector<double> kernel;
        size_t length;
    };
    static_assert(sizeof(Twin) == sizeof(dsp::filter::SincFilter<>), "SincFilter layout changed");
    dsp::filter::SincFilter<> f;
    reinterpret_cast<Twin *>(&f)->kernel.assign(h, h + ntaps);
    dsp::SampleBuffer<T> in(x, x + n);
    dsp::SampleBuffer<T> out = f.apply(in);
    for (size_t i = 0; i < n; ++i) y[i] = out[static_cast<int>(i)];
}

template <typename T>
void run_mavg(const T *x, size_t n, size_t np, T *y) {
    dsp::filter::MovingAverageFilter f;
    f.numPoints(np);
    dsp::SampleBuffer<T> in(x, x + n);
    dsp::SampleBuffer<T> out = f.apply(in);
    for (size_t i = 0; i < n; ++i) y[i] = out[static_cast<int>(i)];
}

}  // namespace

extern "C" {

int ref_bit_reverse(int x, int n) { return dsp::bitReverse(x, n); }

void ref_fft0(const double *in, double *out, int log2n, int inverse) {
    auto *ci = reinterpret_cast<std::complex<double> *>(const_cast<double *>(in));
    auto *co = reinterpret_cast<std::complex<double> *>(out);
    if (inverse) dsp::fft0<dsp::FFTType::INVERSE>(ci, co, log2n);
    else dsp::fft0<dsp::FFTType::FORWARD>(ci, co, log2n);
}

void ref_fft0_batch(const double *in, double *out, int log2n, int inverse, size_t batch) {
    const size_t stride = static_cast<size_t>(2) << log2n;
    for (size_t b = 0; b < batch; ++b) ref_fft0(in + b * stride, out + b * stride, log2n, inverse);
}

// The reference has no float FFT: ref_fft0_f32 is deliberately not defined.

int ref_fft_real(const double *td, size_t n_in, double *out, int nf, int inverse) {
    const int log2n = static_cast<int>(std::ceil(std::log2(nf)));
    const int padded = 1 << log2n;
    auto *co = reinterpret_cast<std::complex<double> *>(out);
    if (inverse) dsp::fft<const double *, std::complex<double> *, dsp::FFTType::INVERSE>(td, td + n_in, co, nf);
    else dsp::fft<const double *, std::complex<double> *, dsp::FFTType::FORWARD>(td, td + n_in, co, nf);
    return padded;
}

void ref_shift(const double *in, double *out, size_t n) {
    auto *ci = reinterpret_cast<const std::complex<double> *>(in);
    std::vector<std::complex<double>> v(ci, ci + n);
    auto r = dsp::shift(v);
    std::memcpy(out, r.data(), n * sizeof(std::complex<double>));
}

int ref_sinc(double cf, int length, int window, int inversion, double *out) {
    switch (inversion) {
    case 0: return make_sinc<dsp::InversionType::SpectralInversion>(cf, length, window, out);
    case 1: return make_sinc<dsp::InversionType::SpectralReversal>(cf, length, window, out);
    default: return make_sinc<dsp::InversionType::None>(cf, length, window, out);
    }
}

void ref_fir_f32(const float *x, size_t n, const double *h, int ntaps, float *y) { run_fir<float>(x, n, h, ntaps, y); }
void ref_fir_f64(const double *x, size_t n, const double *h, int ntaps, double *y) { run_fir<double>(x, n, h, ntaps, y); }
void ref_moving_average_f32(const float *x, size_t n, size_t np, float *y) { run_mavg<float>(x, n, np, y); }
void ref_moving_average_f64(const double *x, size_t n, size_t np, double *y) { run_mavg<double>(x, n, np, y); }

int ref_chebyshev(int highpass, double ripple, double fc, int poles, double *a, double *b) {
    switch (poles) {
    case 2: return design_cheby<2>(highpass, ripple, fc, a, b);
    case 4: return design_cheby<4>(highpass, ripple, fc, a, b);
    case 6: return design_cheby<6>(highpass, ripple, fc, a, b);
    case 8: return design_cheby<8>(highpass, ripple, fc, a, b);
    case 10: return design_cheby<10>(highpass, ripple, fc, a, b);
    case 12: return design_cheby<12>(highpass, ripple, fc, a, b);
    case 20: return design_cheby<20>(highpass, ripple, fc, a, b);
    default: return -1;
    }
}

int ref_biquad_design(int kind, double p0, double p1, double p2, double *a, double *b, double *cd) {
    dsp::BiQuad bq{};
    switch (kind) {
    case 0: bq = dsp::recursive::bandPassFilter(p0, p1); break;
    case 1: bq = dsp::recursive::bandRejectFilter(p0, p1); break;
    case 2: bq = dsp::recursive::lowShelf(p0, p1); break;
    case 3: bq = dsp::recursive::highShelf(p0, p1); break;
    case 4: bq = dsp::recursive::peakingFilter(p0, p1, p2); break;
    default: return -1;
    }
    for (int i = 0; i < 3; ++i) {
        a[i] = bq.a[i];
        b[i] = bq.b[i];
    }
    cd[0] = bq.c0;
    cd[1] = bq.d0;
    return 0;
}

void ref_iir_f32(const float *x, size_t n, const double *a, const double *b, int poles, double c0,
                 double d0, double *state, float *y) {
    dispatch_generic<float>(x, n, a, b, poles, c0, d0, state, y);
}
void ref_iir_f64(const double *x, size_t n, const double *a, const double *b, int poles, double c0,
                 double d0, double *state, double *y) {
    dispatch_generic<double>(x, n, a, b, poles, c0, d0, state, y);
}
void ref_biquad_f32(const float *x, size_t n, const double *a, const double *b, double c0, double d0,
                    double *state, float *y) {
    run_biquad<float>(x, n, a, b, c0, d0, state, y);
}
void ref_biquad_f64(const double *x, size_t n, const double *a, const double *b, double c0, double d0,
                    double *state, double *y) {
    run_biquad<double>(x, n, a, b, c0, d0, state, y);
}

// The retuned band-pass of eval/signals/wind_generator.h:156-164, with the reference's own designer and BiQuad:
// design, overwrite a and b, step.  (The wind-speed synthesis around it needs choc, which is absent; fc comes in.)
void ref_biquad_swept_f32(const float *x, const float *fc, size_t n, int kind, double bw, double *state, float *y) {
    dsp::BiQuad m_bp{};
    if (state) {
        m_bp.x1 = state[0];
        m_bp.x2 = state[1];
        m_bp.y1 = state[2];
        m_bp.y2 = state[3];
    }
    for (size_t i = 0; i < n; ++i) {
        auto bp = kind == 0 ? dsp::recursive::bandPassFilter(fc[i], bw) : dsp::recursive::bandRejectFilter(fc[i], bw);
        m_bp.a = bp.a;
        m_bp.b = bp.b;
        y[i] = m_bp(x[i]);
    }
    if (state) {
        state[0] = m_bp.x1;
        state[1] = m_bp.x2;
        state[2] = m_bp.y1;
        state[3] = m_bp.y2;
    }
}

}  // extern "C"
