// Compile-only check (run where /root/reference exists): the veneer's templates accept the
// REFERENCE's own container type, so a maintainer can call them from the reference's headers.
#include <cstring>
#include <stdexcept>
#include <vector>
#define DSP_MI355X_NO_DSP_ALIAS  // the reference's own namespace dsp is in this translation unit
#include <dsp/sample_buffer.h>   // the reference's header, found through -I/root/reference/dsp/include
#include <dsp_mi355x/dsp.hpp>

dsp::SampleBuffer<float> use_all(dsp::SampleBuffer<float> &buf) {
    namespace g = dsp_mi355x;
    g::filter::SincFilter<> lp(0.1, 511);
    auto a = lp.apply(buf);
    auto b = g::filter::MovingAverageFilter(511).apply(a);
    auto c = g::recursive::lowPassFilter<4>(0.1)(b);
    dsp::SampleBuffer<double> kern(11, 0.0);
    auto d = g::convolve<g::Domain::Frequency>(c, kern);
    std::vector<std::complex<double>> spec(1024);
    g::fft(d.begin(), d.end(), spec.begin(), 1024);
    return d;
}

// ... and the double container every in-repo caller of the reference uses (eval/recursive_filters/main.cpp:35)
dsp::SampleBuffer<double> use_all_double(dsp::SampleBuffer<double> &buf) {
    namespace g = dsp_mi355x;
    g::filter::SincFilter<> lp(0.1, 511);
    auto a = lp.apply(buf);
    auto b = g::filter::MovingAverageFilter(511).apply(a);
    auto c = g::recursive::lowPassFilter<4>(0.1)(b);
    auto e = g::recursive::peakingFilter(0.2, 3.0, 0.1)(c, DSP_GPU_IIR_BATCH_REF);
    dsp::SampleBuffer<double> kern(11, 0.0);
    auto d = g::convolve<g::Domain::Time>(e, kern);
    std::vector<std::complex<double>> spec(1024);
    g::fft(d.begin(), d.end(), spec.begin(), 1024);
    return d;
}
