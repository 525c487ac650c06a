// swept_cos.h -- cos(2*PI*fc) for the per-sample band-pass design (iir_swept_kernels.hip), written out so that it
// costs ~30 double-precision instructions instead of the ~90 of the general library cos (whose argument reduction
// carries double-double sums that this argument range never needs).
//
// The reference evaluates std::cos(2 * PI * cf) (recursive_filters.h:169): the product is rounded to double once, then
// the cosine of THAT double is taken.  So does this: x = fl(2*PI*q) for q = cf folded into [0, 0.5] (a no-op for the
// frequencies the designer is meant for), then
//     n = rint(x * 2/pi) in {0, 1, 2};   t = x - n*pi/2  (Cody-Waite, pi/2 in two doubles: the first subtraction is exact)
//     cos(x) = cos(t), -sin(t), -cos(t)  for n = 0, 1, 2, |t| <= pi/4,
// with the classic minimax kernels for sin and cos on [-pi/4, pi/4] (the coefficients of Sun's fdlibm k_sin.c / k_cos.c,
// public domain constants; each kernel is below one ulp of its result).  Lanes are channels, so a per-lane branch would
// diverge: the kernel tests per WAVE whether any lane needs the sine and whether any needs the cosine, evaluates
// those, and picks per lane.  Measured against the host's libm over 2^24 frequencies
// (tests/cpp/swept_cos_check.cpp): the absolute difference never exceeds 2^-53 (half an ulp of 1.0), which is what
// matters here because cos enters the design only in sums with terms of order one.
//
// Plain C++ with explicit fma(): the same text compiles for the device and, in the checker, for the host.
#pragma once
#include <cmath>

#ifdef __HIPCC__
#define DSPGPU_HD __host__ __device__ __forceinline__
#else
#define DSPGPU_HD inline
#endif

namespace dspgpu {

// x = fl(2*PI*fold(cf)) reduced to t in [-pi/4, pi/4] and the quadrant n in {0, 1, 2}.
DSPGPU_HD void swept_cos_reduce(float cf, double &n, double &t) {
    constexpr double kRefPi = 3.1415926535897932384626433832795;  // constants.h:4
    // fold: cos(2 pi cf) has period 1 in cf and is even.  Exact in float; q = cf for 0 <= cf <= 0.5.
    const float folded = fabsf(cf - rintf(cf));
    const double x = 2 * kRefPi * static_cast<double>(folded);  // rounded once, as the reference's argument is
    n = rint(x * 6.36619772367581382433e-01);                   // 2/pi
    t = fma(-n, 1.57079632679489655800e+00, x);                 // pi/2 (head): exact by Sterbenz
    t = fma(-n, 6.12323399573676603587e-17, t);                 // pi/2 (tail)
}

// sin(t) = t + t^3 (S1 + z (S2 + ... z S6)), |t| <= pi/4
DSPGPU_HD double swept_sin_kernel(double t) {
    const double z = t * t;
    double rs = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    rs = fma(z, rs, 2.75573137070700676789e-06);
    rs = fma(z, rs, -1.98412698298579493134e-04);
    rs = fma(z, rs, 8.33333333332248946124e-03);
    return fma(z * t, fma(z, rs, -1.66666666666666324348e-01), t);
}

// cos(t) = w + (((1 - w) - z/2) + z r),  w = 1 - z/2,  r = z (C1 + z (C2 + z C3)) + z^4 (C4 + z (C5 + z C6)), |t| <= pi/4
DSPGPU_HD double swept_cos_kernel(double t) {
    const double z = t * t;
    const double w = z * z;
    const double r1 = z * fma(z, fma(z, 2.48015872894767294178e-05, -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double r2 = fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07);
    const double r = fma(w * w, r2, r1);
    const double hz = 0.5 * z;
    const double hw = 1.0 - hz;
    return hw + fma(z, r, (1.0 - hw) - hz);
}

// cos(x) = cos(t), -sin(t), -cos(t) for n = 0, 1, 2
DSPGPU_HD double swept_cos_pick(double n, double sn, double cs) {
    const double pick = n == 1.0 ? sn : cs;
    return n == 0.0 ? pick : -pick;
}

// The whole of it for one value (the host checker; the kernel calls the parts so that a wave whose 64 channels all sit
// in one quadrant -- audio-rate sweeps stay below fc = 1/8 -- evaluates only the kernel it needs).
DSPGPU_HD double swept_cos(float cf) {
    double n, t;
    swept_cos_reduce(cf, n, t);
    return swept_cos_pick(n, swept_sin_kernel(t), swept_cos_kernel(t));
}

}  // namespace dspgpu
