/*
 * dsp_oracle.cpp -- CPU restatement of the ejosiah/dsp numeric hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see dsp_oracle.h).  Written from the behaviour of the
 * reference headers, not copied from them: plain loops over raw arrays with the
 * same arithmetic, in the same order, in the same precision, so that the result is
 * bit-identical to the reference when both are built without FMA contraction
 * (-ffp-contract=off; the reference's own build uses no -march flags).
 *
 * "ref:" citations are relative to /root/reference/dsp/include/dsp/.
 */
#include "dsp_oracle.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr double kPi = 3.1415926535897932384626433832795;  // ref: constants.h:4

// ref: fft.h:11-20.  The reference recomputes log2(N) in floating point per call.
int reverse_bits(int x, int n) {
    const int bits = static_cast<int>(std::log2(static_cast<double>(n)));
    int r = 0;
    for (int i = 0; i < bits; ++i) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
}

// Complex product exactly as libgcc's __muldc3 / __mulsc3 evaluates it for finite
// operands: (ac - bd, ad + bc), four products and two sums, no contraction.
template <typename T>
inline void cmul(T ar, T ai, T br, T bi, T &pr, T &pi) {
    const T ac = ar * br, bd = ai * bi, ad = ar * bi, bc = ai * br;
    pr = ac - bd;
    pi = ad + bc;
}

// ref: fft.h:31-55.  Iterative radix-2 decimation in time.  Twiddles advance by the
// recurrence w *= wm inside each stage (fft.h:43-44,52), starting from w = 1, with
// wm = exp(+-i*pi/m2) evaluated in double (also for the float instantiation, which
// then narrows: the float build is ours, not the reference's).
template <typename T>
void radix2_dit(const T *in, T *out, int log2n, bool inverse) {
    const int n = 1 << log2n;
    for (int i = 0; i < n; ++i) {
        const int d = reverse_bits(i, n);
        out[2 * d] = in[2 * i];
        out[2 * d + 1] = in[2 * i + 1];
    }
    for (int s = 1; s <= log2n; ++s) {
        const int span = 1 << s;
        const int half = span >> 1;
        // exp(J * (PI/m2)) with J = (0,+-1): glibc cexp gives exp(0)*cos, exp(0)*sin.
        const double theta = (inverse ? -1.0 : 1.0) * (kPi / half);
        double sn, cs;
        sincos(theta, &sn, &cs);
        const T step_r = static_cast<T>(1.0 * cs), step_i = static_cast<T>(1.0 * sn);
        T wr = 1, wi = 0;
        for (int j = 0; j < half; ++j) {
            for (int k = j; k < n; k += span) {
                T tr, ti;
                cmul(wr, wi, out[2 * (k + half)], out[2 * (k + half) + 1], tr, ti);
                const T ur = out[2 * k], ui = out[2 * k + 1];
                out[2 * k] = ur + tr;
                out[2 * k + 1] = ui + ti;
                out[2 * (k + half)] = ur - tr;
                out[2 * (k + half) + 1] = ui - ti;
            }
            T nr, ni;
            cmul(wr, wi, step_r, step_i, nr, ni);
            wr = nr;
            wi = ni;
        }
    }
}

// ref: filter.h:123-140 (and convolution.h:21-25 with M = taps-1).  Y[j] is re-read and
// re-stored in the sample type on every tap; the product and the sum are double because
// the kernel is std::vector<double>.
template <typename T>
void causal_fir(const T *x, size_t n, const double *h, int ntaps, T *y) {
    for (size_t j = 0; j < n; ++j) y[j] = T(0);
    if (ntaps <= 0) return;
    const size_t order = static_cast<size_t>(ntaps) - 1;
    for (size_t j = order; j < n; ++j) {
        for (size_t i = 0; i <= order; ++i) {
            y[j] = static_cast<T>(static_cast<double>(y[j]) + static_cast<double>(x[j - i]) * h[i]);
        }
    }
}

// ref: filter.h:81-101.  Running sum kept in the sample type; centred window; the
// first mid and last mid outputs stay zero.  Needs n >= npoints (the reference reads
// out of bounds otherwise).
template <typename T>
void centred_mean(const T *x, size_t n, size_t npoints, T *y) {
    for (size_t j = 0; j < n; ++j) y[j] = T(0);
    if (npoints == 0 || n < npoints) return;
    T acc = T(0);
    for (size_t i = 0; i < npoints; ++i) acc += x[i];
    const size_t mid = npoints / 2;
    const T denom = static_cast<T>(npoints);
    y[mid] = acc / denom;
    for (size_t i = mid + 1; i < n - mid; ++i) {
        acc += x[i + mid] - x[i - mid - 1];
        y[i] = acc / denom;
    }
}

// ref: coefficients.h:106-130.  One step of the generic N-pole recursion.  The running
// output is of the SAMPLE type, so every += rounds to it (Q12 in SURVEY.md); histories
// are double.  hx[0..poles], hy[0..poles]; hy[0] is never written by the reference and
// stays at its initial value.
template <typename T>
inline T npole_step(T in, const double *a, const double *b, int poles, double c0, double d0,
                    double *hx, double *hy) {
    T acc = T(0);
    hx[0] = in;
    for (int i = 0; i <= poles; ++i) acc = static_cast<T>(acc + a[i] * hx[i]);
    for (int i = 1; i <= poles; ++i) acc = static_cast<T>(acc + b[i] * hy[i]);
    for (int i = poles; i > 0; --i) hx[i] = hx[i - 1];
    for (int i = poles; i > 0; --i) hy[i] = hy[i - 1];
    hx[0] = 0;
    hy[1] = acc;
    return static_cast<T>(acc * c0 + in * d0);
}

template <typename T>
void npole_run(const T *x, size_t n, const double *a, const double *b, int poles, double c0,
               double d0, double *state, T *y) {
    std::vector<double> local(2 * (static_cast<size_t>(poles) + 1), 0.0);
    double *st = state ? state : local.data();
    double *hx = st, *hy = st + poles + 1;
    for (size_t i = 0; i < n; ++i) y[i] = npole_step<T>(x[i], a, b, poles, c0, d0, hx, hy);
}

// ref: coefficients.h:172-195.  Two-pole specialisation: all arithmetic in double,
// feed-forward sum first, then the feedback pair is summed on its own and added.
template <typename T>
void biquad_run(const T *x, size_t n, const double *a, const double *b, double c0, double d0,
                double *state, T *y) {
    double local[4] = {0, 0, 0, 0};
    double *st = state ? state : local;
    double x1 = st[0], x2 = st[1], y1 = st[2], y2 = st[3];
    for (size_t i = 0; i < n; ++i) {
        const T s = x[i];
        double v = a[0] * s + a[1] * x1 + a[2] * x2;
        v += b[1] * y1 + b[2] * y2;
        x2 = x1;
        x1 = s;
        y2 = y1;
        y1 = v;
        y[i] = static_cast<T>((v * c0) + (s * d0));
    }
    st[0] = x1;
    st[1] = x2;
    st[2] = y1;
    st[3] = y2;
}

// ref: recursive_filters.h:107-151.  One Chebyshev pole pair -> one biquad stage
// (Smith, ch. 20, table 20-5 subroutine): pole on the unit circle, warp to ellipse for
// ripple, bilinear s->z, then LP->LP / LP->HP frequency transform.
struct Stage { double a0, a1, a2, b1, b2; };

Stage chebyshev_stage(bool highpass, double ripple, int npoles, int index, double fc) {
    const double ang = kPi / (npoles * 2) + (index - 1) * kPi / npoles;
    double re = -std::cos(ang);
    double im = std::sin(ang);
    if (ripple != 0) {
        const double es = std::sqrt(std::pow((100.0 / (100.0 - ripple)), 2) - 1);
        const double vx = (1.0 / npoles) * std::log((1.0 / es) + sqrt(1.0 / (es * es) + 1));
        double kx = (1.0 / npoles) * std::log((1.0 / es) + sqrt(1.0 / (es * es) - 1));
        kx = (std::exp(kx) + std::exp(-kx)) * 0.5;
        re *= ((std::exp(vx) - std::exp(-vx)) / 2) / kx;
        im *= ((std::exp(vx) + std::exp(-vx)) / 2) / kx;
    }
    const double t = 2.0 * std::tan(0.5);
    const double w = 2.0 * kPi * fc;
    const double m = re * re + im * im;
    double d = 4.0 - 4.0 * re * t + m * t * t;
    const double x0 = (t * t) / d;
    const double x1 = 2.0 * x0;
    const double x2 = x0;
    const double y1 = (8.0 - 2 * m * t * t) / d;
    const double y2 = (-4.0 - 4.0 * re * t - m * t * t) / d;
    const double k = highpass ? -std::cos(w * 0.5 + 0.5) / std::cos(w * 0.5 - 0.5)
                              : std::sin(-w * 0.5 + 0.5) / std::sin(w * 0.5 + 0.5);
    d = 1.0 + y1 * k - y2 * k * k;
    Stage s;
    s.a0 = (x0 - x1 * k + x2 * k * k) / d;
    s.a1 = (-2.0 * x0 * k + x1 + x1 * k * k - 2 * x2 * k) / d;
    s.a2 = (x0 * k * k - x1 * k + x2) / d;
    s.b1 = (2 * k + y1 + y1 * k * k - 2 * y2 * k) / d;
    s.b2 = (-(k * k) - y1 * k + y2) / d;
    if (highpass) {
        s.a1 = -s.a1;
        s.b1 = -s.b1;
    }
    return s;
}

}  // namespace

extern "C" {

int ORC_NAME(bit_reverse)(int x, int n) { return reverse_bits(x, n); }

void ORC_NAME(fft0)(const double *in, double *out, int log2n, int inverse) {
    radix2_dit<double>(in, out, log2n, inverse != 0);
}

// `batch` back-to-back transforms in one call (timing harness: keeps the per-transform loop out of Python)
void ORC_NAME(fft0_batch)(const double *in, double *out, int log2n, int inverse, size_t batch) {
    const size_t stride = static_cast<size_t>(2) << log2n;
    for (size_t b = 0; b < batch; ++b) radix2_dit<double>(in + b * stride, out + b * stride, log2n, inverse != 0);
}

void ORC_NAME(fft0_f32)(const float *in, float *out, int log2n, int inverse) {
    radix2_dit<float>(in, out, log2n, inverse != 0);
}

// ref: fft.h:57-83.  Real samples become (x,0); zero padded to 2^ceil(log2 nf).  If more
// samples than that are passed the reference still copies all of them but transforms
// only the first 2^log2n.
int ORC_NAME(fft_real)(const double *td, size_t n_in, double *out, int nf, int inverse) {
    const int log2n = static_cast<int>(std::ceil(std::log2(static_cast<double>(nf))));
    const int padded = static_cast<int>(std::pow(2, log2n));
    std::vector<double> buf(2 * (n_in > static_cast<size_t>(padded) ? n_in : static_cast<size_t>(padded)), 0.0);
    for (size_t i = 0; i < n_in; ++i) buf[2 * i] = td[i];
    radix2_dit<double>(buf.data(), out, log2n, inverse != 0);
    return padded;
}

// ref: fft.h:85-94
void ORC_NAME(shift)(const double *in, double *out, size_t n) {
    const size_t mid = n / 2;
    for (size_t i = 0; i < n; ++i) {
        const size_t d = (i + mid) % n;
        out[2 * d] = in[2 * i];
        out[2 * d + 1] = in[2 * i + 1];
    }
}

// ref: dsp.h:23-38 (windows), :40-45 (normalize), :47-79 (sinc)
int ORC_NAME(sinc)(double cf, int length, int window, int inversion, double *out) {
    length |= 1;
    const int order = length - 1;
    for (int i = 0; i <= order; ++i) {
        const double off = double(i) - double(order) / 2;
        double v = 2 * kPi * cf;
        if (off != 0) v = std::sin(v * off) / off;
        const double frac = static_cast<double>(static_cast<size_t>(i)) / static_cast<double>(static_cast<size_t>(order));
        double win = 1.0;
        if (window == 0) win = 0.42 - 0.5 * std::cos(2 * kPi * frac) + 0.08 * std::cos(4 * kPi * frac);
        else if (window == 2) win = 0.54 - 0.46 * std::cos(2.0 * kPi * frac);
        out[i] = v * win;
    }
    double total = 0.0;
    for (int i = 0; i < length; ++i) total = total + out[i];
    for (int i = 0; i < length; ++i) out[i] /= total;
    if (inversion == 0) {  // SpectralInversion
        for (int i = 0; i < length; ++i) out[i] = -out[i];
        out[length / 2] += 1;
    } else if (inversion == 1) {  // SpectralReversal
        for (int i = 1; i < length; i += 2) out[i] = -out[i];
    }
    return length;
}

void ORC_NAME(fir_f32)(const float *x, size_t n, const double *h, int ntaps, float *y) { causal_fir<float>(x, n, h, ntaps, y); }
void ORC_NAME(fir_f64)(const double *x, size_t n, const double *h, int ntaps, double *y) { causal_fir<double>(x, n, h, ntaps, y); }
void ORC_NAME(moving_average_f32)(const float *x, size_t n, size_t np, float *y) { centred_mean<float>(x, n, np, y); }
void ORC_NAME(moving_average_f64)(const double *x, size_t n, size_t np, double *y) { centred_mean<double>(x, n, np, y); }

// ref: recursive_filters.h:24-105.  Stages are multiplied into 22-entry polynomial
// work arrays (index 2 is the constant term), shifted down by two with the feedback
// side negated, then the pass-band gain (DC for LP, Nyquist for HP) is normalised to 1.
int ORC_NAME(chebyshev)(int highpass, double ripple, double fc, int poles, double *a, double *b) {
    if (poles < 2 || poles > 20 || (poles & 1)) return -1;
    if (!(fc >= 0.0 && fc <= 0.5) || !(ripple >= 0.0 && ripple <= 29.0)) return -1;
    double pa[22] = {0}, pb[22] = {0}, qa[22], qb[22];
    pa[2] = 1;
    pb[2] = 1;
    for (int p = 1; p <= poles / 2; ++p) {
        const Stage s = chebyshev_stage(highpass != 0, ripple, poles, p, fc);
        std::memcpy(qa, pa, sizeof qa);
        std::memcpy(qb, pb, sizeof qb);
        for (int i = 2; i < 22; ++i) {
            pa[i] = s.a0 * qa[i] + s.a1 * qa[i - 1] + s.a2 * qa[i - 2];
            pb[i] = qb[i] - s.b1 * qb[i - 1] - s.b2 * qb[i - 2];
        }
    }
    pb[2] = 0;
    for (int i = 0; i < 20; ++i) {
        pa[i] = pa[i + 2];
        pb[i] = -pb[i + 2];
    }
    double sa = 0, sb = 0;
    if (!highpass) {
        for (int i = 0; i < 20; ++i) {
            sa += pa[i];
            sb += pb[i];
        }
    } else {
        for (int i = 0; i < 20; ++i) {
            sa += pa[i] * std::pow(-1, i);
            sb += pb[i] * std::pow(-1, i);
        }
    }
    const double gain = sa / (1 - sb);
    for (int i = 0; i < 20; ++i) pa[i] /= gain;
    for (int i = 0; i <= poles; ++i) {
        a[i] = pa[i];
        b[i] = pb[i];
    }
    return 0;
}

// ref: recursive_filters.h:165-186 (bandPass), :188-209 (bandReject), :211-227 (lowShelf),
// :229-243 (highShelf), :245-263 (peaking)
int ORC_NAME(biquad_design)(int kind, double p0, double p1, double p2, double *a, double *b, double *cd) {
    a[0] = a[1] = a[2] = 0;
    b[0] = b[1] = b[2] = 0;
    cd[0] = 1;
    cd[1] = 0;
    switch (kind) {
    case 0:
    case 1: {
        const double r = 1 - 3 * p1;
        const double c = std::cos(2 * kPi * p0);
        const double k = (1 - 2 * r * c + r * r) / (2 - 2 * c);
        if (kind == 0) {
            a[0] = 1 - k;
            a[1] = 2 * (k - r) * c;
            a[2] = r * r - k;
        } else {
            a[0] = k;
            a[1] = -2 * k * c;
            a[2] = k;
        }
        b[0] = 0;
        b[1] = 2 * r * c;
        b[2] = -(r * r);
        return 0;
    }
    case 2: {  // lowShelf: pow(10.0f, gain/20) promotes to double pow
        const double u = std::pow(10.0f, p1 / 20);
        const double v = 4.0 / (1 + u);
        const double x = v * std::tan(p0 / 2);
        const double y = (1 - x) / (1 + x);
        a[0] = (1 - y) / 2;
        a[1] = a[0];
        a[2] = 0;
        b[1] = y;
        b[2] = 0;
        cd[0] = u - 1;
        cd[1] = 1.0f;
        return 0;
    }
    case 3: {
        const double u = std::pow(10, p1 / 20);
        const double v = (1 + u) / 4;
        const double x = v * std::tan(p0 / 2);
        const double y = (1 - x) / (1 + x);
        a[0] = (1 + y) / 2;
        a[1] = -a[0];
        b[1] = y;
        cd[0] = u - 1;
        cd[1] = 1;
        return 0;
    }
    case 4: {
        const double u = std::pow(10, p1 / 20);
        const double v = 4 / (1 + u);
        const double x = std::tan(p0 / (2 * p2));
        const double vx = v * x;
        const double y = .5 * ((1 - vx) / (1 + vx));
        const double z = (.5 + y) * std::cos(p0);
        a[0] = .5 - y;
        a[2] = -a[0];
        b[1] = 2 * z;
        b[2] = -2 * y;
        cd[0] = u - 1;
        cd[1] = 1;
        return 0;
    }
    default:
        return -1;
    }
}

void ORC_NAME(iir_f32)(const float *x, size_t n, const double *a, const double *b, int poles,
                       double c0, double d0, double *state, float *y) {
    npole_run<float>(x, n, a, b, poles, c0, d0, state, y);
}
void ORC_NAME(iir_f64)(const double *x, size_t n, const double *a, const double *b, int poles,
                       double c0, double d0, double *state, double *y) {
    npole_run<double>(x, n, a, b, poles, c0, d0, state, y);
}
void ORC_NAME(biquad_f32)(const float *x, size_t n, const double *a, const double *b, double c0,
                          double d0, double *state, float *y) {
    biquad_run<float>(x, n, a, b, c0, d0, state, y);
}
void ORC_NAME(biquad_f64)(const double *x, size_t n, const double *a, const double *b, double c0,
                          double d0, double *state, double *y) {
    biquad_run<double>(x, n, a, b, c0, d0, state, y);
}

// ref: eval/signals/wind_generator.h:150-167 (Whistling::getSample).  Per sample: design
// bandPassFilter(fc[t], bw) (kind 0; kind 1: bandRejectFilter) as above, overwrite the BiQuad's a and b,
// keep its history, run one step (coefficients.h:172-182).  fc is float there (:157-158) and widens
// exactly; c0 = 1, d0 = 0 are the BiQuad's defaults (coefficients.h:70-71).  state: {x1,x2,y1,y2}.
void ORC_NAME(biquad_swept_f32)(const float *x, const float *fc, size_t n, int kind, double bw, double *state,
                                float *y) {
    double local[4] = {0, 0, 0, 0};
    double *st = state ? state : local;
    double a[3], b[3], cd[2];
    for (size_t i = 0; i < n; ++i) {
        ORC_NAME(biquad_design)(kind, static_cast<double>(fc[i]), bw, 0.0, a, b, cd);
        biquad_run<float>(x + i, 1, a, b, 1.0, 0.0, st, y + i);
    }
}

}  // extern "C"
