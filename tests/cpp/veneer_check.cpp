// veneer_check.cpp -- exercises include/dsp_mi355x/dsp.hpp the way the reference's eval programs call
// the reference (test infrastructure; driven by tests/test_cpp_veneer.py).
//
//   veneer_check design            prints the host-side designs (no GPU needed)
//   veneer_check run <dir>         reads <dir>/*.f32 written by the test, runs every veneer call on the
//                                  GPU and writes the results next to them
#define DSP_MI355X_AS_DSP
#include <dsp_mi355x/dsp.hpp>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>

// the three HIP runtime calls the distributed-FFT check needs for its device buffers (g++ build, no HIP headers)
extern "C" {
int hipMalloc(void **ptr, size_t bytes);
int hipFree(void *ptr);
int hipMemcpy(void *dst, const void *src, size_t bytes, int kind);
}

namespace {
template <typename T>
std::vector<T> slurp(const std::string &path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("cannot read " + path);
    const std::streamsize bytes = f.tellg();
    f.seekg(0);
    std::vector<T> v(static_cast<size_t>(bytes) / sizeof(T));
    f.read(reinterpret_cast<char *>(v.data()), bytes);
    return v;
}
template <typename T>
void dump(const std::string &path, const T *p, size_t n) {
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char *>(p), static_cast<std::streamsize>(n * sizeof(T)));
}
template <typename C>
void line(const char *name, const C &v) {
    std::printf("%s", name);
    for (double d : v) std::printf(" %.17g", d);
    std::printf("\n");
}
template <size_t P>
void coef(const char *name, const dsp::Coefficients<P> &c) {
    line((std::string(name) + "_a").c_str(), c.a);
    line((std::string(name) + "_b").c_str(), c.b);
    const double cd[2] = {c.c0, c.d0};
    line((std::string(name) + "_cd").c_str(), std::vector<double>(cd, cd + 2));
}

int design() {
    using namespace dsp;
    line("sinc_0p14_10", sinc(0.14, 10));
    line("sinc_0p1_511", sinc(0.1, 511));
    line("sinc_0p14_200", sinc(0.14, 200));
    line("sinc_0p2_64_hamming_inv", sinc<InversionType::SpectralInversion>(0.2, 64, Windows::hamming));
    line("sinc_0p2_64_blackman_rev", sinc<InversionType::SpectralReversal>(0.2, 64));
    line("sinc_0p3_31_identity", sinc(0.3, 31, Windows::identity));
    line("filter_kernel_0p14_200", filter::SincFilter<>(0.14, 200).kernel());
    coef("cheby_lp2", recursive::lowPassFilter<2>(0.1));
    coef("cheby_lp4", recursive::lowPassFilter<4>(0.1));
    coef("cheby_lp6", recursive::lowPassFilter<6>(0.1));
    coef("cheby_hp4", recursive::highPassFilter<4>(0.1));
    coef("cheby_lp4_lo", recursive::lowPassFilter<>(0.01));
    coef("cheby_lp8", recursive::lowPassFilter<8>(0.2));
    coef("bq_bp", recursive::bandPassFilter(0.1, 0.01));
    coef("bq_br", recursive::bandRejectFilter(0.2, 0.02));
    coef("bq_ls", recursive::lowShelf(0.3, 6.0));
    coef("bq_hs", recursive::highShelf(0.5, -3.0));
    coef("bq_pk", recursive::peakingFilter(0.4, 4.0, 0.7));
    return 0;
}

int run(const std::string &dir) {
    using namespace dsp;
    // fft0: 3 transforms of 1024 complex points, forward then inverse of the result
    {
        auto in = slurp<std::complex<float>>(dir + "/fft_in.c64");
        std::vector<std::complex<float>> fwd(in.size()), back(in.size());
        fft0<FFTType::FORWARD>(in.data(), fwd.data(), 10, in.size() / 1024);
        fft0<FFTType::INVERSE>(fwd.data(), back.data(), 10, in.size() / 1024);
        dump(dir + "/fft_fwd.c64", fwd.data(), fwd.size());
        dump(dir + "/fft_back.c64", back.data(), back.size());
    }
    // the reference's own signature (fft.h:31-32): std::complex<double>, same input widened
    {
        auto in32 = slurp<std::complex<float>>(dir + "/fft_in.c64");
        std::vector<std::complex<double>> in(in32.begin(), in32.end()), fwd(in.size());
        fft0<FFTType::FORWARD>(in.data(), fwd.data(), 10, in.size() / 1024);
        dump(dir + "/fft_fwd.c128", fwd.data(), fwd.size());
    }
    // dsp::fft on DOUBLE samples (fft.h:57-83 widens to complex<double>): 1000 samples zero padded to 1024, double route
    {
        auto in32 = slurp<std::complex<float>>(dir + "/fft_in.c64");
        std::vector<double> td(1000);
        for (size_t i = 0; i < td.size(); ++i) td[i] = in32[i].real();
        std::vector<std::complex<double>> freq(1024);
        fft(td.begin(), td.end(), freq.begin(), 1024);
        dump(dir + "/fft_real_double.c128", freq.data(), freq.size());
    }
    // eval/recursive_filters/main.cpp:104-121: impulse at N/2 -> filter -> 1024-point fft
    {
        SampleBuffer<float> impulse(1024, 0.0f);
        impulse[512] = 1.0f;
        auto filt = recursive::lowPassFilter<4>(0.1);
        auto response = filt(impulse, DSP_GPU_IIR_BATCH);
        std::vector<std::complex<double>> freq(1024);
        fft(response.begin(), response.end(), freq.begin(), 1024);
        dump(dir + "/chain_response.f32", response.data(), response.size());
        dump(dir + "/chain_fft.c128", freq.data(), freq.size());
    }
    // SincFilter (filter.h:123-140) on 2 channels, direct, and convolve<Domain::Frequency>
    {
        auto x = slurp<float>(dir + "/fir_in.f32");
        SampleBuffer<float> sig(x.begin(), x.end());
        filter::SincFilter<> f(0.1, 511);
        auto y = f.apply(sig, 2);
        dump(dir + "/fir_direct.f32", y.data(), y.size());
        SampleBuffer<float> one(x.begin(), x.begin() + x.size() / 2);
        const auto k = f.kernel();
        SampleBuffer<double> kern(k.begin(), k.end());
        auto yt = convolve<Domain::Time>(one, kern);
        auto yf = convolve<Domain::Frequency>(one, kern);
        dump(dir + "/conv_time.f32", yt.data(), yt.size());
        dump(dir + "/conv_freq.f32", yf.data(), yf.size());
        // block-wise streaming through the filter object (256-frame blocks like the reference's audio callback)
        std::vector<float> streamed;
        for (size_t pos = 0; pos < one.size(); pos += 256) {
            const size_t len = std::min<size_t>(256, one.size() - pos);
            SampleBuffer<float> blk(x.begin() + pos, x.begin() + pos + len);
            auto yb = f.stream(blk);
            streamed.insert(streamed.end(), yb.begin(), yb.end());
        }
        dump(dir + "/fir_streamed.f32", streamed.data(), streamed.size());
        filter::MovingAverageFilter ma(511);
        auto ym = ma(one);
        dump(dir + "/mavg.f32", ym.data(), ym.size());
        ma.numPoints(8);
        auto y8 = ma.apply(one);
        dump(dir + "/mavg8.f32", y8.data(), y8.size());
    }
    // Coefficients<4>: two chunks with the object's own history carried, then a shelving biquad
    {
        auto x = slurp<float>(dir + "/iir_in.f32");
        auto filt = recursive::lowPassFilter<4>(0.1);
        SampleBuffer<float> a(x.begin(), x.begin() + 1500), b(x.begin() + 1500, x.end());
        auto ya = filt(a), yb = filt(b);
        std::vector<float> y(ya.begin(), ya.end());
        y.insert(y.end(), yb.begin(), yb.end());
        dump(dir + "/iir_lp4.f32", y.data(), y.size());
        std::vector<double> hist(filt.x.begin(), filt.x.end());
        hist.insert(hist.end(), filt.y.begin(), filt.y.end());
        dump(dir + "/iir_lp4_state.f64", hist.data(), hist.size());
        auto shelf = recursive::lowShelf(0.3, 6.0);
        SampleBuffer<float> all(x.begin(), x.end());
        auto ys = shelf(all);
        dump(dir + "/iir_ls.f32", ys.data(), ys.size());
        // the audio-callback form: 256-frame blocks into a ring that wraps (coefficients.h:132-170,197-206)
        {
            CircularBuffer<float, 512> ring;  // sample_buffer.h:175-176: born with 512 slots, operator[] wraps
            if (ring.size() != 512) throw std::runtime_error("CircularBuffer<float, 512> must hold 512 slots");
            ring[-1] = 7.0f;  // (size + idx) % size, sample_buffer.h:67-83
            if (ring[511] != 7.0f || ring[1023] != 7.0f) throw std::runtime_error("circular index does not wrap");
            auto streaming = recursive::lowPassFilter<4>(0.1);
            std::vector<float> collected;
            for (size_t at = 0; at < x.size(); at += 256) {
                const size_t len = std::min<size_t>(256, x.size() - at);
                SampleBuffer<float> block(x.begin() + at, x.begin() + at + len);
                streaming(block, ring);
                collected.insert(collected.end(), ring.begin(), ring.begin() + static_cast<std::ptrdiff_t>(len));
            }
            // the rest of the container: add(buffer) appends elements, view() is a window, Signal / Kernel aliases
            Signal<float> grown;
            grown.add(1.0f);
            grown.add(SampleBuffer<float>(3, 2.0f));
            const auto win = grown.view(1, 4);
            if (grown.size() != 4 || win.size() != 3 || win[0] != 2.0f || win[2] != 2.0f || !grown)
                throw std::runtime_error("SampleBuffer::add / view");
            Kernel<double> none;
            if (none) throw std::runtime_error("an empty buffer must convert to false");
            dump(dir + "/iir_ring.f32", collected.data(), collected.size());
        }
        // batched rows through the same coefficients
        std::vector<float> rows(x.size());
        recursive::lowPassFilter<4>(0.1).apply(x.data(), rows.data(), 4, x.size() / 4);
        dump(dir + "/iir_rows.f32", rows.data(), rows.size());
    }
    // the band-pass retuned every sample (eval/signals/wind_generator.h:156-164), in 256-frame blocks with the BiQuad's
    // own history carried
    {
        auto x = slurp<float>(dir + "/iir_in.f32");
        auto fc = slurp<float>(dir + "/swept_fc.f32");
        BiQuad voice{};
        std::vector<float> y;
        for (size_t at = 0; at < x.size(); at += 256) {
            const size_t len = std::min<size_t>(256, x.size() - at);
            SampleBuffer<float> block(x.begin() + at, x.begin() + at + len), sweep(fc.begin() + at, fc.begin() + at + len);
            auto yb = recursive::bandPassSwept(voice, block, sweep, 60.f / 44100.f);
            y.insert(y.end(), yb.begin(), yb.end());
        }
        dump(dir + "/swept.f32", y.data(), y.size());
        std::vector<double> members{voice.x[1], voice.x[2], voice.y[1], voice.y[2], voice.a[0], voice.a[1], voice.a[2],
                                    voice.b[1], voice.b[2]};
        dump(dir + "/swept_members.f64", members.data(), members.size());
    }
    // DistributedFft: one rank holding the whole 2^16-point signal; a self "exchange" that copies (kind 3 = device
    // to device) stands where RCCL or the caller's all-to-all goes
    {
        auto in = slurp<std::complex<float>>(dir + "/dist_in.c64");
        const size_t bytes = in.size() * sizeof(in[0]);
        void *din = nullptr, *dout = nullptr;
        if (hipMalloc(&din, bytes) || hipMalloc(&dout, bytes)) throw std::runtime_error("hipMalloc failed");
        if (hipMemcpy(din, in.data(), bytes, 1)) throw std::runtime_error("hipMemcpy failed");
        std::vector<std::complex<float>> out(in.size());
        {
            DistributedFft plain(16, 0, 1);
            if (plain.localPoints() != in.size()) throw std::runtime_error("localPoints");
            plain(din, dout);
            if (hipMemcpy(out.data(), dout, bytes, 2)) throw std::runtime_error("hipMemcpy failed");
            dump(dir + "/dist_fwd.c64", out.data(), out.size());
        }
        {
            int calls = 0;
            DistributedFft exchanging(16, 0, 1,
                                      [](void *user, const void *send, void *recv, size_t block, int world, void *) {
                                          ++*static_cast<int *>(user);
                                          return hipMemcpy(recv, send, block * world, 3) ? -2 : 0;
                                      },
                                      &calls);
            exchanging.operator()<FFTType::INVERSE>(din, dout);
            if (calls != 3) throw std::runtime_error("the exchange must run three times");
            if (hipMemcpy(out.data(), dout, bytes, 2)) throw std::runtime_error("hipMemcpy failed");
            dump(dir + "/dist_inv.c64", out.data(), out.size());
        }
        hipFree(din);
        hipFree(dout);
    }
    std::printf("veneer run ok (%s, %d device(s))\n", dsp_gpu_version(), dsp_gpu_device_count());
    return 0;
}
}  // namespace

int main(int argc, char **argv) {
    try {
        if (argc >= 2 && !std::strcmp(argv[1], "design")) return design();
        if (argc >= 3 && !std::strcmp(argv[1], "run")) return run(argv[2]);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "veneer_check: %s\n", e.what());
        return 2;
    }
    std::fprintf(stderr, "usage: veneer_check design | run <dir>\n");
    return 1;
}
