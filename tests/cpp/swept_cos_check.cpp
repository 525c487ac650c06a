// Host check of dsp_amd/csrc/swept_cos.h against the C library's cos(), the function the reference calls
// (recursive_filters.h:169: std::cos(2 * PI * cf)).  The header is plain C++, so the very text the kernel compiles is
// compiled here with g++.  Prints the largest absolute and relative differences over
//   * 2^24 evenly spaced frequencies in [0, 0.5] (every float of the form k * 2^-25),
//   * the floats next to the points where the argument reduction changes quadrant (1/8, 1/4, 3/8, 1/2) and next to 0,
//   * a sweep of tiny frequencies (the design divides by 2 - 2 cos).
// Usage: swept_cos_check  ->  "max_abs_ulp1 <a> max_rel_ulp <r> worst_cf <cf> samples <count>", abs in units of 2^-53.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../dsp_amd/csrc/swept_cos.h"

int main() {
    const double kRefPi = 3.1415926535897932384626433832795;
    double max_abs = 0, max_rel = 0;
    float worst = 0;
    unsigned long long count = 0;
    auto probe = [&](float cf) {
        const double want = std::cos(2 * kRefPi * static_cast<double>(cf));
        const double got = dspgpu::swept_cos(cf);
        const double abs_err = std::fabs(got - want) / 0x1p-53;
        double ulp = std::fabs(std::nextafter(want, 2.0) - want);
        const double rel_err = ulp > 0 ? std::fabs(got - want) / ulp : 0;
        if (abs_err > max_abs) { max_abs = abs_err; worst = cf; }
        if (rel_err > max_rel && std::fabs(want) > 1e-3) max_rel = rel_err;  // relative accuracy away from the zero crossing
        ++count;
    };
    for (unsigned k = 0; k <= (1u << 24); ++k) probe(static_cast<float>(k) * 0x1p-25f);
    const float edges[] = {0.0f, 0.125f, 0.25f, 0.375f, 0.5f};
    for (float e : edges) {
        float lo = e, hi = e;
        for (int i = 0; i < 4096; ++i) {
            probe(lo);
            probe(hi);
            lo = std::nextafterf(lo, -1.0f);
            hi = std::nextafterf(hi, 1.0f);
        }
    }
    for (float f = 1e-12f; f < 0.01f; f *= 1.0009765625f) probe(f);
    // out of the designer's range: folded by periodicity and symmetry (the reference's cos takes any argument)
    const float wild[] = {0.6f, 0.75f, 1.0f, 1.25f, -0.1f, -0.4f, 3.3f, 100.125f};
    double wild_abs = 0;
    for (float cf : wild) {
        const double want = std::cos(2 * kRefPi * static_cast<double>(cf));
        wild_abs = std::fmax(wild_abs, std::fabs(dspgpu::swept_cos(cf) - want));
    }
    std::printf("max_abs_ulp1 %.4f max_rel_ulp %.4f worst_cf %.9g samples %llu wild_abs %.3g\n", max_abs, max_rel, worst, count,
                wild_abs);
    return 0;
}
