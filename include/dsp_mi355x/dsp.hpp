// dsp_mi355x/dsp.hpp -- C++ drop-in overloads of the ejosiah/dsp hot path on top of the C ABI.
//
// Header-only host layer: every function below keeps the NAME, argument order and result
// convention of the reference function it stands in for (file:line relative to the reference's
// dsp/include/dsp/), for SampleBuffer<float> AND SampleBuffer<double> (sample_buffer.h:20; every in-repo caller of
// the reference filters doubles) and with an optional batch/channel count, and forwards the samples
// to libdsp_gpu.so (include/dsp_gpu.h): float buffers to the fp32 entry points, double buffers to the *_f64 ones.  Nothing is computed on the CPU except the filter
// *designs* (taps, coefficients), which the reference also computes once, in double, in
// microseconds (dsp.h:47-79, recursive_filters.h:24-263).
//
//   reference                                         here (namespace dsp_mi355x)
//   dsp::fft0<type>(in, out, log2n)      fft.h:31     fft0<type>(in, out, log2n, batch = 1)   complex<float> and complex<double>
//   dsp::fft<It,CIt,type>(f, l, out, nf) fft.h:57     fft<type>(f, l, out, nf)
//   dsp::shift(vector<complex>)          fft.h:85     shift(vector<complex<float>>)
//   dsp::sinc<Inv>(cf, length, window)   dsp.h:47     sinc<Inv>(cf, length, window)
//   dsp::convolve<T,Domain>(sig, kern)   convolution.h:8   convolve<Domain>(sig, kern)
//   filter::SincFilter<Inv>::apply       filter.h:123 filter::SincFilter<Inv>::apply(buf [, channels])
//   filter::MovingAverageFilter::apply   filter.h:81  filter::MovingAverageFilter::apply(buf [, channels])
//   Coefficients<P>::operator()(buffer)  coefficients.h:88   Coefficients<P>::operator()(buf)
//   Coefficients<P>::operator()(buffer, CircularBuffer&)  coefficients.h:132,197   operator()(buf, sink)
//   recursive::lowPassFilter<P>(fc) ...  recursive_filters.h:153-263   same names
//   the retuning loop of eval/signals/wind_generator.h:156-164         recursive::bandPassSwept(filter, in, fc, bw)
//
// Buffers: any type with size(), data() and a (size_t) constructor works -- the reference's own
// dsp::SampleBuffer<float> as well as the small SampleBuffer below.  Errors: like the reference
// (assert on argument ranges, one std::runtime_error in convolve) there are no error codes; a
// failing C-ABI call throws std::runtime_error carrying dsp_gpu_last_error_string().
//
// Link with -ldsp_gpu (dsp_amd/lib).  The names live in `namespace dsp` like the reference's (an alias of
// dsp_mi355x, so that a caller of the reference -- tests/cpp/reference_callers_check.cpp -- compiles unchanged); define
// DSP_MI355X_NO_DSP_ALIAS before including when the reference's own headers are in the same translation unit.
#pragma once

#include <array>
#include <cassert>
#include <cmath>
#include <complex>
#include <cstddef>
#include <functional>
#include <iterator>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../dsp_gpu.h"

namespace dsp_mi355x {

constexpr double PI = 3.1415926535897932384626433832795;  // constants.h:4

inline void check(int rc, const char *what) {
    if (rc != DSP_GPU_OK) throw std::runtime_error(std::string(what) + ": " + dsp_gpu_last_error_string());
}

// ---- sample container (sample_buffer.h:14-182) ----------------------------------------------------
// The argument and return type of every filter signature: an owning, contiguous array of floating-point samples.
// Same template parameters, members and aliases as the reference's, so code written against it compiles here:
// Circular buffers wrap their index ((size + idx) % size, sample_buffer.h:67-83 -- what the audio callbacks' ring of
// `Capacity` frames relies on) and are born with `Capacity` samples; plain ones grow with add().  Two slips of the
// reference are not reproduced: add(buffer) copies buffer.size() ELEMENTS (the reference memcpy's that many bytes,
// :57-61), and view() returns the view it declares (the reference's definition constructs a SampleBuffer, :169-173, and
// cannot be instantiated).
template <typename SampleType>
class SampleBufferView {
  public:
    SampleBufferView(const SampleType *first, const SampleType *last) : m_first(first), m_last(last) {}
    size_t size() const noexcept { return static_cast<size_t>(m_last - m_first); }
    const SampleType &operator[](int i) const { return m_first[i]; }
    explicit operator const SampleType *() const noexcept { return m_first; }
    const SampleType *data() const noexcept { return m_first; }
    const SampleType *cbegin() const noexcept { return m_first; }
    const SampleType *cend() const noexcept { return m_last; }
    const SampleType *begin() const noexcept { return m_first; }
    const SampleType *end() const noexcept { return m_last; }

  private:
    const SampleType *m_first, *m_last;
};

template <typename SampleType, bool Circular = false, size_t Capacity = 0>
class SampleBuffer {
    static_assert(std::is_floating_point<SampleType>::value, "SampleType should be floating point type");

  public:
    SampleBuffer() : m_data(Circular ? Capacity : 0) {}
    explicit SampleBuffer(size_t size) : m_data(size) {}
    SampleBuffer(size_t size, SampleType val) : m_data(size, val) {}
    template <typename It, typename = typename std::iterator_traits<It>::iterator_category>
    SampleBuffer(It first, It last) : m_data(first, last) {}
    template <bool C2, size_t Cap2>
    SampleBuffer(const SampleBuffer<SampleType, C2, Cap2> &other) : m_data(other.begin(), other.end()) {}

    void add(SampleType s) {
        static_assert(!Circular, "a circular buffer has a fixed number of slots");
        m_data.push_back(s);
    }
    template <bool C2, size_t Cap2>
    void add(const SampleBuffer<SampleType, C2, Cap2> &other) {
        static_assert(!Circular, "a circular buffer has a fixed number of slots");
        m_data.insert(m_data.end(), other.begin(), other.end());
    }
    size_t size() const noexcept { return m_data.size(); }
    SampleType &operator[](int i) { return m_data[slot(i)]; }
    const SampleType &operator[](int i) const { return m_data[slot(i)]; }
    explicit operator SampleType *() noexcept { return m_data.data(); }
    explicit operator const SampleType *() const noexcept { return m_data.data(); }
    SampleType *data() noexcept { return m_data.data(); }
    const SampleType *data() const noexcept { return m_data.data(); }
    auto begin() noexcept { return m_data.begin(); }
    auto end() noexcept { return m_data.end(); }
    auto begin() const noexcept { return m_data.begin(); }
    auto end() const noexcept { return m_data.end(); }
    auto cbegin() const noexcept { return m_data.cbegin(); }
    auto cend() const noexcept { return m_data.cend(); }
    void resize(size_t n) { m_data.resize(n); }
    void clear() noexcept { m_data.clear(); }
    explicit operator bool() const { return !m_data.empty(); }
    // samples [start, end) of the buffer, not owned
    SampleBufferView<SampleType> view(size_t start, size_t end) const {
        assert(start < m_data.size() && end > start && end <= m_data.size());
        return SampleBufferView<SampleType>(m_data.data() + start, m_data.data() + end);
    }

  private:
    size_t slot(int i) const {
        if constexpr (Circular) {
            const long long n = static_cast<long long>(m_data.size());
            return static_cast<size_t>(((static_cast<long long>(i) % n) + n) % n);  // (size + idx) % size for any idx
        } else {
            return static_cast<size_t>(i);
        }
    }
    std::vector<SampleType> m_data;
};
template <typename SampleType, size_t Capacity>
using CircularBuffer = SampleBuffer<SampleType, true, Capacity>;
template <typename T, bool Circular = false>
using Signal = SampleBuffer<T, Circular>;
template <typename T, bool Circular = false>
using Kernel = SampleBuffer<T, Circular>;

// ---- FFT (fft.h) ------------------------------------------------------------------------------
enum class FFTType { FORWARD, INVERSE };  // fft.h:9; FORWARD = e^{+i 2 pi k n / N}, nothing normalises

// fft.h:31-55 for `batch` contiguous transforms of 2^log2n points, complex fp32, host pointers.
template <FFTType type = FFTType::FORWARD>
void fft0(const std::complex<float> *in, std::complex<float> *out, int log2n, size_t batch = 1) {
    check(dsp_gpu_fft_batch(in, out, log2n, batch, type == FFTType::FORWARD ? DSP_GPU_FORWARD : DSP_GPU_INVERSE,
                            DSP_GPU_MEM_HOST, nullptr),
          "fft0");
}
// The reference's own signature, fft.h:31-32: std::complex<double>, arithmetic in double on the GPU (log2n <= 24).
template <FFTType type = FFTType::FORWARD>
void fft0(const std::complex<double> *in, std::complex<double> *out, int log2n, size_t batch = 1) {
    check(dsp_gpu_fft_batch_f64(in, out, log2n, batch, type == FFTType::FORWARD ? DSP_GPU_FORWARD : DSP_GPU_INVERSE,
                                DSP_GPU_MEM_HOST, nullptr),
          "fft0");
}
// Same on device pointers, asynchronous on `stream` (a hipStream_t).
template <FFTType type = FFTType::FORWARD>
void fft0_device(const void *in, void *out, int log2n, size_t batch, void *stream) {
    check(dsp_gpu_fft_batch(in, out, log2n, batch, type == FFTType::FORWARD ? DSP_GPU_FORWARD : DSP_GPU_INVERSE,
                            DSP_GPU_MEM_DEVICE, stream),
          "fft0_device");
}

// fft.h:57-83: real samples -> complex spectrum of 2^ceil(log2 nf) points, zero padded.
template <FFTType type = FFTType::FORWARD, typename TimeDomainIterator, typename ComplexIterator>
void fft(TimeDomainIterator tdFirst, TimeDomainIterator tdLast, ComplexIterator c_out, int nf) {
    assert(nf >= 2);
    using In = typename std::iterator_traits<TimeDomainIterator>::value_type;
    if constexpr (std::is_same_v<In, double>) {
        // double samples: the reference's own arithmetic (it widens everything to complex<double>, fft.h:60-74)
        const int lg = static_cast<int>(std::ceil(std::log2(nf)));
        if (lg <= 24) {
            std::vector<std::complex<double>> buf(static_cast<size_t>(1) << lg);  // zero padded
            size_t i = 0;
            for (auto it = tdFirst; it != tdLast && i < buf.size(); ++it, ++i) buf[i] = std::complex<double>(*it, 0.0);
            check(dsp_gpu_fft_batch_f64(buf.data(), buf.data(), lg, 1, type == FFTType::FORWARD ? DSP_GPU_FORWARD : DSP_GPU_INVERSE,
                                        DSP_GPU_MEM_HOST, nullptr),
                  "fft");
            using Out = typename std::iterator_traits<ComplexIterator>::value_type;
            for (const auto &c : buf) {
                *c_out = Out(c.real(), c.imag());
                ++c_out;
            }
            return;
        }
    }
    std::vector<float> td;
    for (auto it = tdFirst; it != tdLast; ++it) td.push_back(static_cast<float>(*it));
    const int log2n = static_cast<int>(std::ceil(std::log2(nf)));
    std::vector<std::complex<float>> out(static_cast<size_t>(1) << log2n);
    if (td.empty()) td.push_back(0.0f);
    check(dsp_gpu_fft_real_batch(td.data(), td.size(), td.size(), out.data(), nf, 1,
                                 type == FFTType::FORWARD ? DSP_GPU_FORWARD : DSP_GPU_INVERSE, 0, DSP_GPU_MEM_HOST,
                                 nullptr),
          "fft");
    using Out = typename std::iterator_traits<ComplexIterator>::value_type;
    for (const auto &c : out) {
        *c_out = Out(c.real(), c.imag());
        ++c_out;
    }
}

// Batched form of the real wrapper: `batch` rows of n_in samples -> rows of 2^ceil(log2 nf) bins,
// optionally followed by dsp::shift (fft.h:85-94) on every row.
template <FFTType type = FFTType::FORWARD>
std::vector<std::complex<float>> fft_rows(const float *rows, size_t n_in, size_t batch, int nf, bool shifted = false) {
    const int log2n = static_cast<int>(std::ceil(std::log2(nf)));
    std::vector<std::complex<float>> out(batch << log2n);
    check(dsp_gpu_fft_real_batch(rows, n_in, n_in, out.data(), nf, batch,
                                 type == FFTType::FORWARD ? DSP_GPU_FORWARD : DSP_GPU_INVERSE, shifted ? 1 : 0,
                                 DSP_GPU_MEM_HOST, nullptr),
          "fft_rows");
    return out;
}

// fft.h:85-94 (a copy on the host: the reference's shift has no arithmetic to accelerate; the fused
// device form is the `shifted` flag above).  The reference declares it for std::vector<std::complex<double>>; any
// complex sample type is taken.
template <typename Real>
inline std::vector<std::complex<Real>> shift(std::vector<std::complex<Real>> data) {
    std::vector<std::complex<Real>> out(data.size());
    const size_t n = data.size(), mid = n / 2;
    for (size_t i = 0; i < n; ++i) out[(i + mid) % n] = data[i];
    return out;
}

// One fft0 (fft.h:31-55) of 2^log2n points spread over `world` GPUs, one object per rank (process or host thread
// with its GPU current): rank g passes its contiguous slice of N/world points and receives the same slice of
// the natural-order result (device pointers, in != out).  Either RCCL trades the blocks (`nccl_comm` = the
// caller's ncclComm_t) or a caller-supplied exchange does; with neither only `phase` can be used (world > 1).
class DistributedFft {
  public:
    DistributedFft(int log2n, int rank, int world, void *nccl_comm) {
        check(dsp_gpu_fft_dist_plan_create_rccl(&m_plan, log2n, rank, world, nccl_comm), "DistributedFft");
    }
    DistributedFft(int log2n, int rank, int world, dsp_gpu_exchange_fn exchange = nullptr, void *user = nullptr) {
        check(dsp_gpu_fft_dist_plan_create(&m_plan, log2n, rank, world, exchange, user), "DistributedFft");
    }
    DistributedFft(const DistributedFft &) = delete;
    DistributedFft &operator=(const DistributedFft &) = delete;
    ~DistributedFft() { dsp_gpu_fft_dist_plan_destroy(m_plan); }

    [[nodiscard]] size_t localPoints() const { return dsp_gpu_fft_dist_local_points(m_plan); }

    template <FFTType type = FFTType::FORWARD>
    void operator()(const void *in, void *out, void *stream = nullptr) {
        check(dsp_gpu_fft_dist(m_plan, in, out, type == FFTType::FORWARD ? DSP_GPU_FORWARD : DSP_GPU_INVERSE, stream),
              "DistributedFft");
    }
    // one local phase (0..3) for hosts that run the all-to-all between phases themselves
    template <FFTType type = FFTType::FORWARD>
    void phase(int index, const void *in, void *out, void *stream = nullptr) {
        check(dsp_gpu_fft_dist_phase(m_plan, index, in, out, type == FFTType::FORWARD ? DSP_GPU_FORWARD : DSP_GPU_INVERSE,
                                     stream),
              "DistributedFft::phase");
    }

  private:
    dsp_gpu_fft_dist_plan *m_plan = nullptr;
};

// ---- windowed-sinc design (dsp.h) -----------------------------------------------------------------
enum class Domain : int { Time = 0, Spacial, Frequency };                    // dsp.h:14
enum class FilterType : int { LowPass, HighPass };                           // dsp.h:16
enum class InversionType : int { SpectralInversion, SpectralReversal, None };  // dsp.h:20
using Window = std::function<double(size_t, size_t)>;                        // dsp.h:18

struct Windows {  // dsp.h:23-38
    static double blackMan(size_t i, size_t M) {
        const double f = static_cast<double>(i) / static_cast<double>(M);
        return 0.42 - 0.5 * std::cos(2 * PI * f) + 0.08 * std::cos(4 * PI * f);
    }
    static double identity(size_t, size_t) { return 1.0; }
    static double hamming(size_t i, size_t M) {
        const double f = static_cast<double>(i) / static_cast<double>(M);
        return 0.54 - 0.46 * std::cos(2.0 * PI * f);
    }
};

// dsp.h:47-79: odd length forced, unity DC gain, optional spectral inversion / reversal.
template <InversionType inversionType = InversionType::None>
std::vector<double> sinc(double cf, int length, const Window &window = Windows::blackMan) {
    length |= 1;
    const int order = length - 1;
    std::vector<double> h(static_cast<size_t>(length));
    double total = 0.0;
    for (int i = 0; i < length; ++i) {
        const double off = double(i) - double(order) / 2;
        double v = 2 * PI * cf;
        if (off != 0) v = std::sin(v * off) / off;
        h[i] = v * window(static_cast<size_t>(i), static_cast<size_t>(order));
        total += h[i];
    }
    for (double &v : h) v /= total;
    if (inversionType == InversionType::SpectralInversion) {
        for (double &v : h) v = -v;
        h[length / 2] += 1;
    } else if (inversionType == InversionType::SpectralReversal) {
        for (int i = 1; i < length; i += 2) h[i] = -h[i];
    }
    return h;
}

namespace detail {
// The reference's SampleBuffer::data() const does not compile when instantiated (it returns a
// non-const pointer into a const vector, sample_buffer.h:121-123), so element 0 is addressed instead.
template <typename Buffer>
auto cptr(const Buffer &b) -> decltype(&b[0]) { return b.size() ? &b[0] : nullptr; }
template <typename Buffer>
auto mptr(Buffer &b) -> decltype(&b[0]) { return b.size() ? &b[0] : nullptr; }

inline std::vector<float> narrow(const std::vector<double> &v) { return std::vector<float>(v.begin(), v.end()); }

template <typename Buffer>
using sample_t = std::remove_cv_t<std::remove_reference_t<decltype(std::declval<const Buffer &>()[0])>>;

// double buffers: double taps, time domain on the double matrix cores whatever `mode` says (<= 4096 taps);
// float buffers: the fp32 path with the kernel narrowed to float
template <typename Buffer>
Buffer fir(const Buffer &x, const std::vector<double> &kernel, const std::vector<float> &taps, int mode, size_t channels) {
    Buffer y(x.size());
    if (x.size() == 0 || kernel.empty()) return y;
    assert(channels >= 1 && x.size() % channels == 0);
    if constexpr (std::is_same_v<sample_t<Buffer>, double>) {
        check(dsp_gpu_convolve_f64(cptr(x), mptr(y), channels, x.size() / channels, kernel.data(), static_cast<int>(kernel.size()),
                                   DSP_GPU_CONV_AUTO, DSP_GPU_MEM_HOST, nullptr),
              "convolve");
    } else {
        check(dsp_gpu_convolve(cptr(x), mptr(y), channels, x.size() / channels, taps.data(), static_cast<int>(taps.size()),
                               mode, DSP_GPU_MEM_HOST, nullptr),
              "convolve");
    }
    return y;
}
}  // namespace detail

// convolution.h:8-29.  Domain::Time is the direct loop (with M = kernel.size() - 1, the off-by-one of
// the reference's listing fixed); Domain::Frequency -- which the reference only
// declares (convolution.h:10-12 throws) -- is FFT overlap-save.  Output: signal.size() samples, the
// first kernel.size()-1 zero.
template <Domain domain = Domain::Time, typename Buffer, typename KernelBuffer>
Buffer convolve(const Buffer &signal, const KernelBuffer &kernel) {
    std::vector<double> taps64(kernel.size());
    for (size_t i = 0; i < taps64.size(); ++i) taps64[i] = static_cast<double>(kernel[static_cast<int>(i)]);
    if (domain == Domain::Spacial) throw std::runtime_error("Not yet implemented!");
    return detail::fir(signal, taps64, detail::narrow(taps64),
                       domain == Domain::Frequency ? DSP_GPU_CONV_OVERLAP_SAVE : DSP_GPU_CONV_DIRECT, 1);
}

namespace filter {

enum class Type { FIR, IIR };  // filter.h:12

// filter.h:15-31, 69-101
struct MovingAverageFilter {
    static constexpr Type type = Type::FIR;
    MovingAverageFilter(size_t numPoints = 1) : m_numPoints(numPoints | 1) {}  // filter.h:69-70: forced odd
    void numPoints(size_t n) { m_numPoints = n; }                               // filter.h:72-74: NOT re-forced

    // `channels` rows of sampleBuffer.size()/channels samples each (1 = the reference's call)
    template <typename Buffer>
    Buffer apply(const Buffer &sampleBuffer, size_t channels = 1) const {
        Buffer y(sampleBuffer.size());
        if (sampleBuffer.size() == 0) return y;
        if constexpr (std::is_same_v<detail::sample_t<Buffer>, double>)
            check(dsp_gpu_moving_average_f64(detail::cptr(sampleBuffer), detail::mptr(y), channels, sampleBuffer.size() / channels,
                                             static_cast<int>(m_numPoints), DSP_GPU_MEM_HOST, nullptr),
                  "MovingAverageFilter::apply");
        else
            check(dsp_gpu_moving_average(detail::cptr(sampleBuffer), detail::mptr(y), channels, sampleBuffer.size() / channels,
                                         static_cast<int>(m_numPoints), DSP_GPU_MEM_HOST, nullptr),
                  "MovingAverageFilter::apply");
        return y;
    }
    template <typename Buffer>
    Buffer operator()(const Buffer &b) const { return apply(b); }

  private:
    size_t m_numPoints;
};

// filter.h:33-57, 103-145
template <InversionType inversionType = InversionType::None>
struct SincFilter {
    static constexpr Type type = Type::FIR;
    SincFilter() = default;
    SincFilter(double cf, size_t length) : m_cutoffFrequency(cf), m_length(length) { design(); }
    // The reference's setters recompute the kernel and discard it (filter.h:111-121); here they take effect.
    void cutoffFrequency(double cf) { m_cutoffFrequency = cf; design(); }
    void length(size_t length) { m_length = length; design(); }

    // mode: DSP_GPU_CONV_AUTO / DIRECT / OVERLAP_SAVE; `channels` rows share the kernel
    template <typename Buffer>
    Buffer apply(const Buffer &sampleBuffer, size_t channels = 1, int mode = DSP_GPU_CONV_DIRECT) const {
        // the direct form stops at 4096 taps; longer kernels go through the (partitioned) overlap-save path
        if (mode == DSP_GPU_CONV_DIRECT && m_taps.size() > 4096) mode = DSP_GPU_CONV_OVERLAP_SAVE;
        return detail::fir(sampleBuffer, m_kernel, m_taps, mode, channels);
    }
    std::vector<double> kernel() const { return m_kernel; }

    // Block-wise use (audio callbacks): feed consecutive blocks, the filter object carries the input tail.
    // The concatenated outputs equal apply() on the concatenated input.  reset() starts a new stream.
    template <typename Buffer>
    Buffer stream(const Buffer &block, int mode = DSP_GPU_CONV_DIRECT) {
        Buffer y(block.size());
        if (block.size() == 0 || m_taps.empty()) return y;
        if constexpr (std::is_same_v<detail::sample_t<Buffer>, double>) {
            // double blocks: the one-shot double filter over (carried tail ++ block); the tail lives in the object
            const size_t order = m_kernel.size() - 1;
            m_history64.resize(order, 0.0);
            std::vector<double> in(m_history64), out(order + block.size());
            for (size_t i = 0; i < block.size(); ++i) in.push_back(block[static_cast<int>(i)]);
            check(dsp_gpu_convolve_f64(in.data(), out.data(), 1, in.size(), m_kernel.data(), static_cast<int>(m_kernel.size()),
                                       DSP_GPU_CONV_AUTO, DSP_GPU_MEM_HOST, nullptr),
                  "SincFilter::stream");
            for (size_t i = 0; i < block.size(); ++i)  // outputs before the stream's first full window stay zero (filter.h:126,133)
                y[static_cast<int>(i)] = m_position + i < order ? 0.0 : out[order + i];
            m_history64.assign(in.end() - static_cast<std::ptrdiff_t>(order), in.end());
        } else {
            m_history.resize(m_taps.size() - 1);
            check(dsp_gpu_convolve_stream(detail::cptr(block), detail::mptr(y), 1, block.size(), m_taps.data(),
                                          static_cast<int>(m_taps.size()), m_history.empty() ? nullptr : m_history.data(),
                                          m_position, mode, DSP_GPU_MEM_HOST, nullptr),
                  "SincFilter::stream");
        }
        m_position += block.size();
        return y;
    }
    void reset() {
        m_position = 0;
        m_history64.clear();
    }

  private:
    void design() {
        m_kernel = sinc<inversionType>(m_cutoffFrequency, static_cast<int>(m_length));
        m_taps = detail::narrow(m_kernel);
    }
    double m_cutoffFrequency{0};
    std::vector<double> m_kernel{};
    std::vector<float> m_taps{};
    std::vector<float> m_history{};
    std::vector<double> m_history64{};
    unsigned long long m_position{0};
    size_t m_length{0};
};

}  // namespace filter

// ---- IIR (coefficients.h) ------------------------------------------------------------------------
// coefficients.h:7-30: a, b (b[0] == 0), histories x, y with x[i] = input i steps ago (i >= 1),
// y[i] = feedback value i steps ago, wet/dry c0, d0.  BiQuad = Coefficients<2>.
template <size_t Poles>
struct Coefficients {
    std::array<double, Poles + 1u> a{};
    std::array<double, Poles + 1u> b{};
    std::array<double, Poles + 1u> x{};
    std::array<double, Poles + 1u> y{};
    double c0{1};
    double d0{0};
    static constexpr size_t poles = Poles;

    // coefficients.h:88-104: filter the whole buffer, continuing from (and updating) x[] / y[].  SampleBuffer<float> or
    // SampleBuffer<double>.  mode DSP_GPU_IIR_BATCH_REF reproduces the reference's arithmetic bit for bit (for float
    // buffers the generic section then accumulates in float, coefficients.h:108-118); the default computes in double
    // with fused multiply-adds.
    template <typename Buffer, std::enable_if_t<!std::is_arithmetic_v<Buffer>, int> = 0>
    Buffer operator()(const Buffer &input, int mode = DSP_GPU_IIR_AUTO) {
        assert(b[0] == 0);
        Buffer output(input.size());
        if (input.size() == 0) return output;
        std::array<double, 2 * Poles> st{};
        for (size_t i = 0; i < Poles; ++i) {
            st[i] = x[i + 1];
            st[Poles + i] = y[i + 1];
        }
        if constexpr (std::is_same_v<detail::sample_t<Buffer>, double>)
            check(dsp_gpu_iir_f64(detail::cptr(input), detail::mptr(output), 1, input.size(), a.data(), b.data(), static_cast<int>(Poles),
                                  c0, d0, st.data(), mode, DSP_GPU_MEM_HOST, nullptr),
                  "Coefficients::operator()");
        else
            check(dsp_gpu_iir(detail::cptr(input), detail::mptr(output), 1, input.size(), a.data(), b.data(), static_cast<int>(Poles), c0,
                              d0, st.data(), mode, DSP_GPU_MEM_HOST, nullptr),
                  "Coefficients::operator()");
        for (size_t i = 0; i < Poles; ++i) {
            x[i + 1] = st[i];
            y[i + 1] = st[Poles + i];
        }
        x[0] = 0;  // coefficients.h:126
        return output;
    }

    // coefficients.h:106-130 (generic) / :172-182 (BiQuad): ONE sample, continuing from and updating x[] / y[] -- the call
    // the wind generator makes per sample (eval/signals/wind_generator.h:44,73,123,164).  It goes through the same device
    // entry point as a buffer of one sample, in the reference's own arithmetic (DSP_GPU_IIR_BATCH_REF: bit-identical to the
    // reference's step, float accumulator for float samples), so a caller that steps sample by sample compiles and gets the
    // reference's bits -- at ~40 us a call.  It exists for completeness: hand over blocks (or use bandPassSwept for a
    // filter retuned every sample) wherever the caller's structure allows.
    template <typename SampleType, std::enable_if_t<std::is_floating_point_v<SampleType>, int> = 0>
    SampleType operator()(SampleType iSample) {
        std::array<double, 2 * Poles> st{};
        for (size_t i = 0; i < Poles; ++i) {
            st[i] = x[i + 1];
            st[Poles + i] = y[i + 1];
        }
        SampleType out{};
        if constexpr (std::is_same_v<SampleType, double>)
            check(dsp_gpu_iir_f64(&iSample, &out, 1, 1, a.data(), b.data(), static_cast<int>(Poles), c0, d0, st.data(),
                                  DSP_GPU_IIR_BATCH_REF, DSP_GPU_MEM_HOST, nullptr),
                  "Coefficients::operator()(sample)");
        else {
            const float in = static_cast<float>(iSample);
            float o = 0.0f;
            check(dsp_gpu_iir(&in, &o, 1, 1, a.data(), b.data(), static_cast<int>(Poles), c0, d0, st.data(), DSP_GPU_IIR_BATCH_REF,
                              DSP_GPU_MEM_HOST, nullptr),
                  "Coefficients::operator()(sample)");
            out = static_cast<SampleType>(o);
        }
        for (size_t i = 0; i < Poles; ++i) {
            x[i + 1] = st[i];
            y[i + 1] = st[Poles + i];
        }
        x[0] = 0;  // coefficients.h:126
        return out;
    }

    // coefficients.h:132-170 (generic) and :197-206 (BiQuad): the block-streaming form of the audio callbacks
    // (eval/recursive_filters/main.cpp:174-206) -- filter one block, continuing from the object's history, and
    // write sample i to output[i]; `output` is any indexable sink, the reference's CircularBuffer<T, Capacity>
    // (whose operator[] wraps) included.  Not reproduced from the generic overload: its function-static input
    // history shared by every filter object (:142), feedback taken from the already wet/dry-mixed ring (:162-165)
    // and the X[i] index slip (:165); this is the recursion of :106-130 block by block, as the BiQuad overload is.
    template <typename Buffer, typename Sink, std::enable_if_t<!std::is_arithmetic_v<Sink>, int> = 0>
    void operator()(const Buffer &input, Sink &output, int mode = DSP_GPU_IIR_AUTO) {
        const Buffer block = (*this)(input, mode);
        for (size_t i = 0; i < input.size(); ++i) output[static_cast<int>(i)] = block[static_cast<int>(i)];
    }

    // `channels` independent rows filtered with the same coefficients; state[ch][2*Poles] as in the
    // C ABI (may be nullptr: every row starts from rest).  The object's own x[] / y[] are untouched.
    void apply(const float *in, float *out, size_t channels, size_t n, double *state = nullptr,
               int mode = DSP_GPU_IIR_AUTO) const {
        check(dsp_gpu_iir(in, out, channels, n, a.data(), b.data(), static_cast<int>(Poles), c0, d0, state, mode,
                          DSP_GPU_MEM_HOST, nullptr),
              "Coefficients::apply");
    }
    void apply(const double *in, double *out, size_t channels, size_t n, double *state = nullptr,
               int mode = DSP_GPU_IIR_AUTO) const {
        check(dsp_gpu_iir_f64(in, out, channels, n, a.data(), b.data(), static_cast<int>(Poles), c0, d0, state, mode,
                              DSP_GPU_MEM_HOST, nullptr),
              "Coefficients::apply");
    }
};
using BiQuad = Coefficients<2>;
using TwoPoleCoefficients = Coefficients<2>;
using FourPoleCoefficients = Coefficients<4>;
using SixPoleCoefficients = Coefficients<6>;

namespace recursive {

namespace chebyshev {
struct Stage { double a0, a1, a2, b1, b2; };

// recursive_filters.h:107-151: one pole pair of the Chebyshev prototype -> one second-order stage
inline Stage computePoles(FilterType filter, double passBandRipple, int numPoles, int pole, double cutoffFrequency) {
    const double ang = PI / (numPoles * 2) + (pole - 1) * PI / numPoles;
    double re = -std::cos(ang), im = std::sin(ang);
    if (passBandRipple != 0) {
        const double es = std::sqrt(std::pow(100.0 / (100.0 - passBandRipple), 2) - 1);
        const double vx = (1.0 / numPoles) * std::log((1.0 / es) + std::sqrt(1.0 / (es * es) + 1));
        double kx = (1.0 / numPoles) * std::log((1.0 / es) + std::sqrt(1.0 / (es * es) - 1));
        kx = (std::exp(kx) + std::exp(-kx)) * 0.5;
        re *= ((std::exp(vx) - std::exp(-vx)) / 2) / kx;
        im *= ((std::exp(vx) + std::exp(-vx)) / 2) / kx;
    }
    const double t = 2.0 * std::tan(0.5), w = 2.0 * PI * cutoffFrequency, m = re * re + im * im;
    double d = 4.0 - 4.0 * re * t + m * t * t;
    const double x0 = (t * t) / d, x1 = 2.0 * x0, x2 = x0;
    const double y1 = (8.0 - 2 * m * t * t) / d, y2 = (-4.0 - 4.0 * re * t - m * t * t) / d;
    const bool hp = filter == FilterType::HighPass;
    const double k = hp ? -std::cos(w * 0.5 + 0.5) / std::cos(w * 0.5 - 0.5)
                        : std::sin(-w * 0.5 + 0.5) / std::sin(w * 0.5 + 0.5);
    d = 1.0 + y1 * k - y2 * k * k;
    Stage s{(x0 - x1 * k + x2 * k * k) / d, (-2.0 * x0 * k + x1 + x1 * k * k - 2 * x2 * k) / d,
            (x0 * k * k - x1 * k + x2) / d, (2 * k + y1 + y1 * k * k - 2 * y2 * k) / d, (-(k * k) - y1 * k + y2) / d};
    if (hp) {
        s.a1 = -s.a1;
        s.b1 = -s.b1;
    }
    return s;
}

// recursive_filters.h:24-105: multiply the stages out into one direct-form section, unity pass-band gain
template <size_t Poles>
Coefficients<Poles> computeCoefficients(FilterType filter, double passBandRipple, double cutoffFrequency) {
    static_assert(Poles % 2 == 0 && Poles >= 2 && Poles <= 20, "even, 2..20 poles");
    assert(cutoffFrequency >= 0 && cutoffFrequency <= 0.5);
    assert(passBandRipple >= 0 && passBandRipple <= 29);
    std::array<double, 22> pa{}, pb{};
    pa[2] = pb[2] = 1;
    for (size_t p = 1; p <= Poles / 2; ++p) {
        const Stage s = computePoles(filter, passBandRipple, static_cast<int>(Poles), static_cast<int>(p), cutoffFrequency);
        const auto qa = pa, qb = pb;
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
    for (int i = 0; i < 20; ++i) {
        const double sign = filter == FilterType::HighPass ? std::pow(-1, i) : 1.0;
        sa += pa[i] * sign;
        sb += pb[i] * sign;
    }
    const double gain = sa / (1 - sb);
    Coefficients<Poles> c;
    for (size_t i = 0; i <= Poles; ++i) {
        c.a[i] = pa[i] / gain;
        c.b[i] = pb[i];
    }
    return c;
}
}  // namespace chebyshev

template <size_t Poles = 4>
Coefficients<Poles> lowPassFilter(double cutoffFrequency) {  // recursive_filters.h:153-157
    return chebyshev::computeCoefficients<Poles>(FilterType::LowPass, 0.5, cutoffFrequency);
}
template <size_t Poles = 4>
Coefficients<Poles> highPassFilter(double cutoffFrequency) {  // recursive_filters.h:159-163
    return chebyshev::computeCoefficients<Poles>(FilterType::HighPass, 0.5, cutoffFrequency);
}

namespace detail {
inline BiQuad narrowBand(double f, double bw, bool reject) {  // recursive_filters.h:165-209
    const double r = 1 - 3 * bw, c = std::cos(2 * PI * f);
    const double k = (1 - 2 * r * c + r * r) / (2 - 2 * c);
    BiQuad q;
    if (reject) q.a = {k, -2 * k * c, k};
    else q.a = {1 - k, 2 * (k - r) * c, r * r - k};
    q.b = {0.0, 2 * r * c, -(r * r)};
    return q;
}
}  // namespace detail
inline BiQuad bandPassFilter(double centerFrequency, double bandwidth) { return detail::narrowBand(centerFrequency, bandwidth, false); }
inline BiQuad bandRejectFilter(double centerFrequency, double bandwidth) { return detail::narrowBand(centerFrequency, bandwidth, true); }

// The band-pass (or band-reject) RETUNED AT EVERY SAMPLE, as eval/signals/wind_generator.h:156-164 does it one sample at
// a time: `bp = bandPassFilter(fc[t], bw); filter.a = bp.a; filter.b = bp.b; out[t] = filter(in[t]);` for a whole block
// in one call, design included, on the GPU.  `filter` keeps its history across calls exactly as m_bp does there, and
// leaves with the a and b of the last sample's design; its c0 / d0 are not applied (the designers leave them 1 and 0).
template <typename Buffer, typename Sweep>
Buffer bandPassSwept(BiQuad &filter, const Buffer &input, const Sweep &centerFrequency, double bandwidth, bool reject = false) {
    assert(centerFrequency.size() == input.size());
    Buffer output(input.size());
    if (input.size() == 0) return output;
    std::vector<float> fc(input.size());
    for (size_t i = 0; i < fc.size(); ++i) fc[i] = static_cast<float>(centerFrequency[static_cast<int>(i)]);
    std::array<double, 4> st{filter.x[1], filter.x[2], filter.y[1], filter.y[2]};
    // (the wind generator's samples are float, wind_generator.h:150-167; other buffer types ride through a float copy)
    std::vector<float> xin(input.size()), yout(input.size());
    for (size_t i = 0; i < xin.size(); ++i) xin[i] = static_cast<float>(input[static_cast<int>(i)]);
    check(dsp_gpu_biquad_swept(xin.data(), fc.data(), yout.data(), 1, input.size(),
                               input.size(), reject ? DSP_GPU_SWEPT_BANDREJECT : DSP_GPU_SWEPT_BANDPASS, bandwidth, st.data(),
                               DSP_GPU_MEM_HOST, nullptr),
          "bandPassSwept");
    for (size_t i = 0; i < yout.size(); ++i) output[static_cast<int>(i)] = yout[i];
    filter.x[1] = st[0];
    filter.x[2] = st[1];
    filter.y[1] = st[2];
    filter.y[2] = st[3];
    const BiQuad last = detail::narrowBand(static_cast<double>(fc.back()), bandwidth, reject);
    filter.a = last.a;
    filter.b = last.b;
    return output;
}
// `channels` voices at once: in, fc, out [channels][n] (fc_stride = n) or one sweep fc[n] for all (fc_stride = 0);
// state [channels][4] = {x1, x2, y1, y2} per voice, may be nullptr.
inline void bandPassSwept(const float *in, const float *fc, float *out, size_t channels, size_t n, size_t fc_stride,
                          double bandwidth, double *state = nullptr, bool reject = false) {
    check(dsp_gpu_biquad_swept(in, fc, out, channels, n, fc_stride, reject ? DSP_GPU_SWEPT_BANDREJECT : DSP_GPU_SWEPT_BANDPASS,
                               bandwidth, state, DSP_GPU_MEM_HOST, nullptr),
          "bandPassSwept");
}

inline BiQuad lowShelf(double frequency, double gain) {  // recursive_filters.h:211-227
    const double u = std::pow(10.0, gain / 20), v = 4.0 / (1 + u);
    const double xx = v * std::tan(frequency / 2), yy = (1 - xx) / (1 + xx);
    BiQuad q;
    q.a = {(1 - yy) / 2, (1 - yy) / 2, 0.0};
    q.b = {0.0, yy, 0.0};
    q.c0 = u - 1;
    q.d0 = 1;
    return q;
}
inline BiQuad highShelf(double frequency, double gain) {  // recursive_filters.h:229-243
    const double u = std::pow(10, gain / 20), v = (1 + u) / 4;
    const double xx = v * std::tan(frequency / 2), yy = (1 - xx) / (1 + xx);
    BiQuad q;
    q.a = {(1 + yy) / 2, -(1 + yy) / 2, 0.0};
    q.b = {0.0, yy, 0.0};
    q.c0 = u - 1;
    q.d0 = 1;
    return q;
}
inline BiQuad peakingFilter(double frequency, double gain, double q_) {  // recursive_filters.h:245-263
    const double u = std::pow(10, gain / 20), v = 4 / (1 + u);
    const double xx = std::tan(frequency / (2 * q_)), vx = v * xx;
    const double yy = .5 * ((1 - vx) / (1 + vx)), z = (.5 + yy) * std::cos(frequency);
    BiQuad q;
    q.a = {.5 - yy, 0.0, -(.5 - yy)};
    q.b = {0.0, 2 * z, -2 * yy};
    q.c0 = u - 1;
    q.d0 = 1;
    return q;
}

}  // namespace recursive
}  // namespace dsp_mi355x

#ifndef DSP_MI355X_NO_DSP_ALIAS
namespace dsp = dsp_mi355x;  // (DSP_MI355X_AS_DSP, which used to ask for this, is still accepted and no longer needed)
#endif
