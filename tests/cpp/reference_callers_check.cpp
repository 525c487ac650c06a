// reference_callers_check.cpp -- the numeric work of the reference's own eval programs, on SampleBuffer<double> as
// they declare it (`using Signal = dsp::SampleBuffer<double>`), compiled against the veneer alone (test
// infrastructure; driven by tests/test_cpp_veneer.py::test_reference_callers_on_gpu, which checks every output file
// against the oracle's double functions).
//
// Call sites reproduced (GUI, audio and threading stripped; the calls into namespace dsp are the callers' own):
//   eval/filter_compare/main.cpp:11-20            SincFilter{0.14, 200} applied to an impulse at 300 of 401
//   eval/recursive_filters/main.cpp:75-101        one of the seven recursive designs applied to a Signal
//   eval/recursive_filters/main.cpp:104-121       1024-point impulse at N/2 -> filter -> dsp::fft -> magnitudes
//   eval/moving_average_filter_demo/main.cpp:57-66  MovingAverageFilter{}; numPoints(size); filter(buffer)
//   eval/windowed_sinc_filter/main.cpp:76,104-108 SincFilter<Inv>{fc, M}; buffer = filter.apply(buffer)
#include <dsp_mi355x/dsp.hpp>

#include <cstdio>
#include <fstream>
#include <functional>

using Signal = dsp::SampleBuffer<double>;

namespace {
void dump(const std::string &path, const double *p, size_t n) {
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char *>(p), static_cast<std::streamsize>(n * sizeof(double)));
}
std::vector<double> slurp(const std::string &path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("cannot read " + path);
    const std::streamsize bytes = f.tellg();
    f.seekg(0);
    std::vector<double> v(static_cast<size_t>(bytes) / sizeof(double));
    f.read(reinterpret_cast<char *>(v.data()), bytes);
    return v;
}

// the seven kinds of eval/recursive_filters/main.cpp:33,75-101, each designed and applied the way applyFilter does it:
// `auto filter = dsp::recursive::<kind>(...); output = filter(input);`
const std::vector<std::pair<const char *, std::function<Signal(const Signal &, float, float, float)>>> kinds = {
    {"low_pass", [](const Signal &in, float f, float, float) { auto filter = dsp::recursive::lowPassFilter<2>(f); return filter(in); }},
    {"high_pass", [](const Signal &in, float f, float, float) { auto filter = dsp::recursive::highPassFilter<2>(f); return filter(in); }},
    {"band_pass", [](const Signal &in, float f, float bw, float) { auto filter = dsp::recursive::bandPassFilter(f, bw); return filter(in); }},
    {"band_reject", [](const Signal &in, float f, float bw, float) { auto filter = dsp::recursive::bandRejectFilter(f, bw); return filter(in); }},
    {"low_shelf", [](const Signal &in, float f, float, float g) { auto filter = dsp::recursive::lowShelf(f, g); return filter(in); }},
    {"high_shelf", [](const Signal &in, float f, float, float g) { auto filter = dsp::recursive::highShelf(f, g); return filter(in); }},
    {"peaking", [](const Signal &in, float f, float bw, float g) { auto filter = dsp::recursive::peakingFilter(f, g, bw); return filter(in); }},
};
}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const std::string dir = argv[1];
    try {
        {  // filter_compare: the sinc kernel seen as an impulse response
            Signal cs{401, 0.0};
            cs[300] = 1;
            dsp::filter::SincFilter sincFilter{0.14, 200};
            auto sKernel = sincFilter.apply(cs);
            dump(dir + "/filter_compare.f64", sKernel.data(), sKernel.size());
        }
        {  // recursive_filters: impulse response and its 1024-point spectrum magnitude for every kind (Settings defaults :45-48)
            const float frequency = 0.25f, bandwidth = 0.006f, gain = 1.0f;
            constexpr int N = 1024;
            for (const auto &kind : kinds) {
                Signal impulse(N);
                impulse[N / 2] = 1;
                Signal impulseResponse = kind.second(impulse, frequency, bandwidth, gain);
                std::vector<std::complex<double>> freq(N);
                dsp::fft(impulseResponse.begin(), impulseResponse.end(), freq.begin(), N);
                dsp::SampleBuffer<double> freqMag;
                for (const auto &c : freq) freqMag.add(std::abs(c));
                dump(dir + "/recursive_" + kind.first + "_response.f64", impulseResponse.data(), impulseResponse.size());
                dump(dir + "/recursive_" + kind.first + "_magnitude.f64", freqMag.data(), freqMag.size());
            }
        }
        {  // the same designs on noise, and in the reference's own arithmetic (bit-exact against the oracle)
            const std::vector<double> noise = slurp(dir + "/noise.f64");
            Signal input(noise.begin(), noise.end());
            auto four = dsp::recursive::lowPassFilter<4>(0.1);
            Signal y = four(input);
            dump(dir + "/noise_lp4.f64", y.data(), y.size());
            auto four_ref = dsp::recursive::lowPassFilter<4>(0.1);
            Signal yr = four_ref(input, DSP_GPU_IIR_BATCH_REF);
            dump(dir + "/noise_lp4_ref.f64", yr.data(), yr.size());
            auto peak = dsp::recursive::peakingFilter(0.25f, 1.0f, 0.006f);
            Signal yp = peak(input, DSP_GPU_IIR_BATCH_REF);
            dump(dir + "/noise_peaking_ref.f64", yp.data(), yp.size());
            // moving_average_filter_demo: default-constructed filter, retuned through numPoints (odd and even sizes)
            dsp::filter::MovingAverageFilter filter{};
            for (int size : {1, 8, 51}) {
                filter.numPoints(static_cast<size_t>(size));
                auto filteredBuffer = filter(input);
                dump(dir + "/mavg_" + std::to_string(size) + ".f64", filteredBuffer.data(), filteredBuffer.size());
            }
            // windowed_sinc_filter: a spectrally inverted (high-pass) kernel applied in place of the buffer
            dsp::filter::SincFilter<dsp::InversionType::SpectralInversion> hp{0.2, 200};
            Signal buffer = input;
            buffer = hp.apply(buffer);
            dump(dir + "/sinc_hp.f64", buffer.data(), buffer.size());
            // block-wise: 256-frame blocks through one filter object equal the one-shot call
            dsp::filter::SincFilter<> lp{0.14, 200};
            Signal streamed;
            for (size_t at = 0; at < input.size(); at += 256) {
                Signal block(input.begin() + static_cast<std::ptrdiff_t>(at),
                             input.begin() + static_cast<std::ptrdiff_t>(std::min(input.size(), at + 256)));
                Signal out = lp.stream(block);
                for (size_t i = 0; i < out.size(); ++i) streamed.add(out[static_cast<int>(i)]);
            }
            dump(dir + "/sinc_lp_streamed.f64", streamed.data(), streamed.size());
            // sample by sample, as the wind generator steps its filters (eval/signals/wind_generator.h:44,73,123,164):
            // Coefficients<P>::operator()(SampleType), coefficients.h:106-130 / 172-182
            {
                auto lop = dsp::recursive::lowPassFilter<4>(0.1);
                auto bp = dsp::recursive::bandPassFilter(0.1, 0.01);
                std::vector<double> s4, s2;
                std::vector<float> s4f;
                auto lopf = dsp::recursive::lowPassFilter<4>(0.1);
                for (size_t i = 0; i < 64; ++i) {
                    s4.push_back(lop(input[static_cast<int>(i)]));
                    s2.push_back(bp(input[static_cast<int>(i)]));
                    s4f.push_back(lopf(static_cast<float>(input[static_cast<int>(i)])));
                }
                dump(dir + "/per_sample_lp4.f64", s4.data(), s4.size());
                dump(dir + "/per_sample_bp.f64", s2.data(), s2.size());
                std::vector<double> widened(s4f.begin(), s4f.end());
                dump(dir + "/per_sample_lp4_float.f64", widened.data(), widened.size());
            }
            // dsp::shift on the spectrum type the reference declares it for (fft.h:85: std::vector<std::complex<double>>)
            std::vector<std::complex<double>> spec(1024);
            dsp::fft(input.begin(), input.begin() + 1000, spec.begin(), 1024);
            std::vector<std::complex<double>> centred = dsp::shift(spec);
            dump(dir + "/spectrum_shifted.c64", reinterpret_cast<const double *>(centred.data()), 2 * centred.size());
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "reference_callers_check: %s\n", e.what());
        return 1;
    }
    return 0;
}
