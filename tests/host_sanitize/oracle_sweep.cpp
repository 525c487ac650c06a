// oracle_sweep.cpp -- the CPU restatement (oracle/dsp_oracle.cpp) under AddressSanitizer / UBSan: every entry point on
// the edge shapes the parity tests use (lengths 1, 2, odd, shorter than the filter, even moving-average windows, 2 .. 20
// poles).  Test infrastructure only (tests/host_sanitize/Makefile).
#include <cstdio>
#include <vector>

#include "../../oracle/dsp_oracle.h"

int main() {
    for (int log2n = 1; log2n <= 13; ++log2n) {
        const size_t n = size_t(1) << log2n;
        std::vector<double> in(2 * n, 0.5), out(2 * n);
        std::vector<float> inf(2 * n, 0.5f), outf(2 * n);
        for (int inv = 0; inv < 2; ++inv) {
            orc_fft0(in.data(), out.data(), log2n, inv);
            orc_fft0_f32(inf.data(), outf.data(), log2n, inv);
        }
        for (int x = 0; x < 16 && x < static_cast<int>(n); ++x) (void)orc_bit_reverse(x, static_cast<int>(n));
    }
    for (int nf : {1, 2, 3, 64, 100, 1000, 1024}) {
        std::vector<double> td(77, 1.0), spec(2 * 2048);
        (void)orc_fft_real(td.data(), td.size(), spec.data(), nf, 0);
    }
    for (int len : {1, 2, 3, 10, 11, 200, 511}) {
        std::vector<double> h((len | 1) + 1);
        for (int w = 0; w < 3; ++w)
            for (int inv = 0; inv < 3; ++inv) (void)orc_sinc(0.14, len, w, inv, h.data());
        const int ntaps = len | 1;
        for (size_t n : {size_t(1), size_t(2), size_t(ntaps - 1 > 0 ? ntaps - 1 : 1), size_t(ntaps), size_t(ntaps + 5), size_t(2000)}) {
            std::vector<float> xf(n, 1.f), yf(n);
            std::vector<double> xd(n, 1.0), yd(n);
            orc_fir_f32(xf.data(), n, h.data(), ntaps, yf.data());
            orc_fir_f64(xd.data(), n, h.data(), ntaps, yd.data());
        }
    }
    for (size_t pts : {size_t(1), size_t(2), size_t(5), size_t(6), size_t(511), size_t(512)})
        for (size_t n : {size_t(1), size_t(4), size_t(6), size_t(7), size_t(600), size_t(5000)}) {
            std::vector<float> xf(n, 1.f), yf(n);
            std::vector<double> xd(n, 1.0), yd(n);
            orc_moving_average_f32(xf.data(), n, pts, yf.data());
            orc_moving_average_f64(xd.data(), n, pts, yd.data());
        }
    for (int poles = 2; poles <= 20; poles += 2) {
        std::vector<double> a(poles + 1), b(poles + 1), st(2 * (poles + 1), 0.0);
        for (int hp = 0; hp < 2; ++hp) {
            if (orc_chebyshev(hp, 0.5, 0.2, poles, a.data(), b.data()) != 0) return 3;
            std::vector<float> xf(1000, 1.f), yf(1000);
            std::vector<double> xd(1000, 1.0), yd(1000);
            if (poles == 2) {
                double bst[4] = {0, 0, 0, 0};
                orc_biquad_f32(xf.data(), xf.size(), a.data(), b.data(), 1.0, 0.0, bst, yf.data());
                orc_biquad_f64(xd.data(), xd.size(), a.data(), b.data(), 1.0, 0.0, nullptr, yd.data());
                std::vector<float> fc(1000, 0.05f);
                orc_biquad_swept_f32(xf.data(), fc.data(), xf.size(), 0, 0.01, bst, yf.data());
            }
            orc_iir_f32(xf.data(), xf.size(), a.data(), b.data(), poles, 1.0, 0.0, st.data(), yf.data());
            orc_iir_f64(xd.data(), xd.size(), a.data(), b.data(), poles, 1.0, 0.0, nullptr, yd.data());
        }
    }
    for (int kind = 0; kind < 5; ++kind) {
        double a[3], b[3], cd[2];
        if (orc_biquad_design(kind, 0.1, kind < 2 ? 0.01 : 2.0, 0.7, a, b, cd) != 0) return 4;
    }
    {
        std::vector<double> in(2 * 64, 1.0), out(2 * 64);
        orc_shift(in.data(), out.data(), 64);
        orc_fft0_batch(in.data(), out.data(), 4, 0, 4);
    }
    std::puts("oracle sweep ok");
    return 0;
}
