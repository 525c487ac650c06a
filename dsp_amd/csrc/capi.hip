// capi.hip -- the extern "C" surface of libdsp_gpu.so (declared in include/dsp_gpu.h).
//
// Validates arguments, stages host buffers when mem == DSP_GPU_MEM_HOST, keeps the
// per-device workspaces and the cache of uploaded tap sets, and forwards to the kernel
// launchers.  Nothing here computes samples on the CPU: if HIP is unusable the calls
// fail with DSP_GPU_EHIP.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace dspgpu {

char *last_error_buffer() {
    static thread_local char buf[kLastErrorBytes] = "";
    return buf;
}

namespace {

constexpr int kMaxDevices = 64;

struct Buffer {
    void *ptr = nullptr;
    size_t bytes = 0;
};

constexpr size_t kTapCacheEntries = 256;      // a partitioned filter caches one entry per segment
constexpr int kSegmentLog2N = 14;             // partitioned long filters use 16384-point blocks ...
constexpr int kSegmentTaps = 8192;            // ... of 8192 taps per overlap-save pass (hop 8193)
constexpr int kMaxPartitionedTaps = 1 << 18;  // 32 passes

// One uploaded tap set (+ lazily built overlap-save spectra).  The cache is a fixed pool of slots with LRU
// replacement: a new tap set TAKES OVER the least recently used idle slot together with its device buffers (nothing is
// freed, so nothing synchronises the device), after waiting on the host for that slot's own last use only.
// A slot is handed out as a lease: pinned (never a victim) until the call that holds it has launched its kernels and
// recorded `last_use` on its stream.
struct TapSlot {
    std::vector<float> taps;   // host copy: the key, and the source of the (asynchronous) upload
    unsigned long long hash = 0;
    float *d_taps = nullptr;
    size_t cap_taps = 0;       // floats d_taps holds
    float2 *d_spectrum[15] = {};   // [log2n]: buffers, kept across take-overs
    bool spectrum_valid[15] = {};  // spectrum of the CURRENT taps zero padded to 2^log2n points (8 ... 12 and 14 in use)
    hipEvent_t ready = nullptr;     // taps + spectra built (recorded on the building stream)
    hipEvent_t last_use = nullptr;  // recorded by every call after its kernels
    bool used = false;
    bool used_once = false;  // last_use has been recorded at least once
    int pins = 0;
    unsigned long long stamp = 0;  // LRU clock
};

struct TapLease {  // what a call gets: pointers by value, valid until release_taps()
    int dev = -1;
    int slot = -1;
    const float *d_taps = nullptr;
    const float2 *d_spectrum = nullptr;
};

// The growing device workspaces, ONE SET PER (device, stream): two streams (or two host threads, each on its own
// stream) never share scratch memory, so calls on different streams of one device may overlap.  A device keeps about
// kStreamSets sets; a further stream takes over the least recently used UNPINNED set together with its buffers after one
// hipDeviceSynchronize (rare: more than kStreamSets distinct streams in rotation).  A set is pinned by the host thread
// that last asked for it (a call's workspace() requests all name the same device and stream) until that thread asks for
// another set or exits: a call that is still running on the host -- a DSP_GPU_MEM_HOST call pipelining its chunks --
// never has its buffers taken from under it.  When every set is pinned the table grows instead.
constexpr size_t kStreamSets = 8;
struct StreamSet {
    hipStream_t stream = nullptr;
    bool used = false;
    int pins = 0;
    unsigned long long stamp = 0;
    Buffer ws[WS_SLOTS];
};

struct DeviceState {
    std::vector<std::unique_ptr<StreamSet>> sets;  // the sets never move (threads hold pointers to the one they pin)
    unsigned long long set_clock = 0;
    unsigned long long epoch = 0;  // bumped by dsp_gpu_shutdown: pins taken before it name sets that no longer exist
    std::vector<TapSlot> taps;  // sized once (kTapCacheEntries): slots never move
    unsigned long long clock = 0;
};

std::mutex g_mutex;
DeviceState g_dev[kMaxDevices];

int current_device(int *dev) {
    DSPGPU_HIP(hipGetDevice(dev));
    if (*dev < 0 || *dev >= kMaxDevices) return fail(DSP_GPU_EUNSUPPORTED, "device index %d out of range", *dev);
    return DSP_GPU_OK;
}

unsigned long long tap_hash(const float *taps, int ntaps) {  // FNV-1a over the bytes
    unsigned long long h = 1469598103934665603ull ^ static_cast<unsigned long long>(ntaps);
    const unsigned char *p = reinterpret_cast<const unsigned char *>(taps);
    for (size_t i = 0; i < sizeof(float) * static_cast<size_t>(ntaps); ++i) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}

// Find or upload a tap set on the current device; spectrum_log2n == 0: no spectrum wanted.  On success the slot is
// pinned: the caller launches its kernels on `s` and then calls release_taps(lease, s).
int acquire_taps(const float *taps, int ntaps, int spectrum_log2n, hipStream_t s, TapLease *out) {
    if (spectrum_log2n < 0 || spectrum_log2n > 14) return fail(DSP_GPU_EINVAL, "tap spectrum of 2^%d points", spectrum_log2n);
    int dev = 0;
    if (int rc = current_device(&dev)) return rc;
    const unsigned long long h = tap_hash(taps, ntaps);
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceState &st = g_dev[dev];
    if (st.taps.empty()) st.taps.resize(kTapCacheEntries);
    TapSlot *hit = nullptr, *victim = nullptr;
    for (TapSlot &e : st.taps) {
        if (e.used && e.hash == h && static_cast<int>(e.taps.size()) == ntaps &&
            std::memcmp(e.taps.data(), taps, sizeof(float) * ntaps) == 0) {
            hit = &e;
            break;
        }
        if (e.pins == 0 && (!victim || (!e.used && victim->used) || (e.used == victim->used && e.stamp < victim->stamp))) victim = &e;
    }
    bool fresh = false;
    if (!hit) {
        if (!victim) return fail(DSP_GPU_ENOMEM, "tap cache: all %zu slots are in use by running calls", kTapCacheEntries);
        hit = victim;
        if (!hit->ready) {
            DSPGPU_HIP(hipEventCreateWithFlags(&hit->ready, hipEventDisableTiming));
            DSPGPU_HIP(hipEventCreateWithFlags(&hit->last_use, hipEventDisableTiming));
        } else if (hit->used) {
            DSPGPU_HIP(hipEventSynchronize(hit->last_use));  // this slot's own last kernels only; an LRU victim is long idle
        }
        hit->used = false;
        if (hit->cap_taps < static_cast<size_t>(ntaps)) {  // grow (rare): frees only a buffer nothing uses any more
            if (hit->d_taps) (void)hipFree(hit->d_taps);
            hit->d_taps = nullptr;
            hit->cap_taps = 0;
            const size_t cap = std::max<size_t>(static_cast<size_t>(ntaps), 1024);
            if (hipMalloc(&hit->d_taps, sizeof(float) * cap) != hipSuccess)
                return fail(DSP_GPU_ENOMEM, "hipMalloc of %d taps failed", ntaps);
            hit->cap_taps = cap;
        }
        hit->taps.assign(taps, taps + ntaps);
        hit->hash = h;
        for (bool &v : hit->spectrum_valid) v = false;
        // asynchronous upload from the slot's own host copy (stable until the slot is taken over again)
        DSPGPU_HIP(hipMemcpyAsync(hit->d_taps, hit->taps.data(), sizeof(float) * ntaps, hipMemcpyHostToDevice, s));
        hit->used = true;
        fresh = true;
    }
    if (spectrum_log2n && !hit->spectrum_valid[spectrum_log2n]) {
        if (!fresh) DSPGPU_HIP(hipStreamWaitEvent(s, hit->ready, 0));  // the taps may have been uploaded on another stream
        if (!hit->d_spectrum[spectrum_log2n] &&
            hipMalloc(&hit->d_spectrum[spectrum_log2n], sizeof(float2) << spectrum_log2n) != hipSuccess)
            return fail(DSP_GPU_ENOMEM, "hipMalloc of the tap spectrum failed");
        if (int rc = launch_fir_spectrum(hit->d_taps, ntaps, hit->d_spectrum[spectrum_log2n], spectrum_log2n, s)) return rc;
        hit->spectrum_valid[spectrum_log2n] = true;
        fresh = true;
    }
    if (fresh) DSPGPU_HIP(hipEventRecord(hit->ready, s));          // later users on other streams wait for this
    else DSPGPU_HIP(hipStreamWaitEvent(s, hit->ready, 0));         // same stream: a no-op in effect
    hit->stamp = ++st.clock;
    ++hit->pins;
    out->dev = dev;
    out->slot = static_cast<int>(hit - st.taps.data());
    out->d_taps = hit->d_taps;
    out->d_spectrum = spectrum_log2n ? hit->d_spectrum[spectrum_log2n] : nullptr;
    return DSP_GPU_OK;
}

// The call has launched everything that reads the lease's buffers on `s`.
void release_taps(const TapLease &lease, hipStream_t s) {
    if (lease.slot < 0) return;
    std::lock_guard<std::mutex> lock(g_mutex);
    TapSlot &e = g_dev[lease.dev].taps[lease.slot];
    // `last_use` is one event re-recorded by every call on whatever stream it used: make `s` wait for the previous
    // record first, so that the new record covers every earlier use on every stream (a take-over waits on it alone)
    if (e.used_once) (void)hipStreamWaitEvent(s, e.last_use, 0);
    (void)hipEventRecord(e.last_use, s);
    e.used_once = true;
    --e.pins;
}

struct LeaseGuard {  // releases on every return path
    TapLease lease;
    hipStream_t s;
    ~LeaseGuard() { release_taps(lease, s); }
};

// Block length of one overlap-save pass.  4096 points hold 2049 taps, 16384 points 8193.  A channel short enough to
// fit ONE 4096-point block is cut into the 256- ... 4096-point blocks that cost the fewest butterflies
// (many short channels -- frames of 1,024 samples, streamed 256-sample audio blocks -- are the case this serves).
int os_block_log2n(int ntaps, size_t n, bool with_history) {
    if (ntaps > 2049) return kSegmentLog2N;
    const long long order = ntaps - 1;
    const long long valid = with_history ? static_cast<long long>(n) : std::max<long long>(static_cast<long long>(n) - order, 1);
    if (valid > 4096 - order) return 12;
    int best = 12;
    long long best_cost = 4096 * 12;
    for (int l = 11; l >= 8; --l) {
        const long long N = 1ll << l, L = N - order;
        if (L < 1) break;
        const long long cost = (valid + L - 1) / L * N * l;
        if (cost < best_cost) best = l, best_cost = cost;
    }
    return best;
}

struct HostStage {  // mem == HOST helper: device mirrors of the caller's buffers
    void *d_in = nullptr;
    void *d_out = nullptr;
};

int stage_in(const void *host, size_t in_bytes, size_t out_bytes, hipStream_t s, HostStage *hs) {
    if (int rc = workspace(WS_STAGE_IN, in_bytes, &hs->d_in, s)) return rc;
    if (int rc = workspace(WS_STAGE_OUT, out_bytes, &hs->d_out, s)) return rc;
    DSPGPU_HIP(hipMemcpyAsync(hs->d_in, host, in_bytes, hipMemcpyHostToDevice, s));
    return DSP_GPU_OK;
}

int stage_out(void *host, size_t out_bytes, hipStream_t s, const HostStage &hs) {
    DSPGPU_HIP(hipMemcpyAsync(host, hs.d_out, out_bytes, hipMemcpyDeviceToHost, s));
    DSPGPU_HIP(hipStreamSynchronize(s));
    return DSP_GPU_OK;
}

bool bad_mem(int mem) { return mem != DSP_GPU_MEM_HOST && mem != DSP_GPU_MEM_DEVICE; }

// The pipelined host route (host_path.hip): `f(d_in, d_out, unit0, count)` runs the device-pointer form on a chunk.
template <class F>
int host_route(const void *in, size_t in_unit, size_t in_total, void *out, size_t out_unit, size_t units, hipStream_t s, F f,
               void *aux = nullptr, size_t aux_bytes = 0) {
    HostJob job{static_cast<const char *>(in), in_unit, in_total, static_cast<char *>(out), out_unit, units, aux, aux_bytes,
                [f](const void *din, void *dout, size_t u0, size_t count, void *daux) { return f(din, dout, u0, count, daux); }};
    return run_host_job(job, s);
}

}  // namespace

namespace {
struct ThreadPin {  // the set this host thread pins (at most one), released when the thread moves on or exits
    StreamSet *set = nullptr;
    int dev = -1;
    unsigned long long epoch = 0;
    void drop_locked() {
        if (set && g_dev[dev].epoch == epoch) --set->pins;
        set = nullptr;
    }
    ~ThreadPin() {
        if (!set) return;
        std::lock_guard<std::mutex> lock(g_mutex);
        drop_locked();
    }
};
thread_local ThreadPin t_pin;
}  // namespace

int workspace(WorkspaceSlot slot, size_t bytes, void **ptr, hipStream_t s) {
    int dev = 0;
    if (int rc = current_device(&dev)) return rc;
    // DSP_GPU_SHARED_WORKSPACES=1: one set per device whatever the stream, as until round 2 (what
    // tests/test_streams_gpu.py guards against; kept to show that the test can fail)
    static const bool shared = [] { const char *e = std::getenv("DSP_GPU_SHARED_WORKSPACES"); return e && e[0] == '1'; }();
    if (shared) s = nullptr;
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceState &st = g_dev[dev];
    StreamSet *set = nullptr, *victim = nullptr;
    for (auto &up : st.sets) {
        StreamSet &c = *up;
        if (c.used && c.stream == s) {
            set = &c;
            break;
        }
        const bool mine = t_pin.set == &c && t_pin.dev == dev && t_pin.epoch == st.epoch && c.pins == 1;
        if (c.pins != 0 && !mine) continue;  // in use by a call of another thread
        if (!victim || (!c.used && victim->used) || (c.used == victim->used && c.stamp < victim->stamp)) victim = &c;
    }
    if (!set) {
        if (st.sets.size() < kStreamSets || !victim) {  // room left, or every set is pinned: a new one
            st.sets.push_back(std::make_unique<StreamSet>());
            set = st.sets.back().get();
        } else {
            set = victim;
            if (set->used) DSPGPU_HIP(hipDeviceSynchronize());  // the old stream's launches may still use these buffers
        }
        set->used = true;
        set->stream = s;
    }
    if (t_pin.set != set || t_pin.dev != dev || t_pin.epoch != st.epoch) {
        t_pin.drop_locked();
        t_pin.set = set;
        t_pin.dev = dev;
        t_pin.epoch = st.epoch;
        ++set->pins;
    }
    set->stamp = ++st.set_clock;
    Buffer &b = set->ws[slot];
    if (b.bytes < bytes) {
        if (b.ptr) {
            DSPGPU_HIP(hipDeviceSynchronize());  // earlier launches may still use the old buffer
            (void)hipFree(b.ptr);
            b.ptr = nullptr;
            b.bytes = 0;
        }
        const size_t want = (bytes + 255) & ~static_cast<size_t>(255);
        if (hipMalloc(&b.ptr, want) != hipSuccess) {
            b.ptr = nullptr;
            return fail(DSP_GPU_ENOMEM, "workspace allocation of %zu bytes failed", want);
        }
        b.bytes = want;
    }
    *ptr = b.ptr;
    return DSP_GPU_OK;
}

int allow_full_lds(const void *fn) {
    static std::mutex m;
    static std::unordered_map<const void *, unsigned long long> done;  // kernel -> devices it has been set on
    int dev = 0;
    if (int rc = current_device(&dev)) return rc;
    const unsigned long long bit = 1ull << dev;
    std::lock_guard<std::mutex> lock(m);
    unsigned long long &mask = done[fn];
    if (mask & bit) return DSP_GPU_OK;
    DSPGPU_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    mask |= bit;
    return DSP_GPU_OK;
}

void release_all_workspaces() {
    int saved = 0, count = 0;
    if (hipGetDevice(&saved) != hipSuccess || hipGetDeviceCount(&count) != hipSuccess) return;
    for (int d = 0; d < count && d < kMaxDevices; ++d) {
        bool used = iir_tables_in_use(d) || host_pipe_in_use(d);
        {
            std::lock_guard<std::mutex> lock(g_mutex);
            DeviceState &st = g_dev[d];
            used = used || !st.taps.empty();
            used = used || !st.sets.empty();
        }
        if (!used) continue;
        if (hipSetDevice(d) != hipSuccess) continue;
        (void)hipDeviceSynchronize();
        release_iir_tables(d);
        release_host_pipes(d);
        std::lock_guard<std::mutex> lock(g_mutex);
        DeviceState &st = g_dev[d];
        for (auto &c : st.sets)
            for (Buffer &b : c->ws)
                if (b.ptr) (void)hipFree(b.ptr);
        st.sets.clear();
        ++st.epoch;  // threads that still pin one of those sets must not touch it
        for (TapSlot &e : st.taps) {
            if (e.d_taps) (void)hipFree(e.d_taps);
            for (float2 *sp : e.d_spectrum)
                if (sp) (void)hipFree(sp);
            if (e.ready) (void)hipEventDestroy(e.ready);
            if (e.last_use) (void)hipEventDestroy(e.last_use);
        }
        st.taps.clear();
    }
    (void)hipSetDevice(saved);
}

}  // namespace dspgpu

using namespace dspgpu;

extern "C" {

int dsp_gpu_fft_batch(const void *in, void *out, int log2n, size_t batch, int direction, int mem, void *stream) {
    if (!in || !out) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_batch: null buffer");
    if (log2n < 1 || log2n > 30) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_batch: log2n %d outside [1,30]", log2n);
    if (direction != DSP_GPU_FORWARD && direction != DSP_GPU_INVERSE)
        return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_batch: direction %d", direction);
    if (bad_mem(mem)) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_batch: mem %d", mem);
    if (batch == 0) return DSP_GPU_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t bytes = (batch << log2n) * sizeof(float2);
    if (mem == DSP_GPU_MEM_DEVICE)
        return launch_fft_batch(static_cast<const float2 *>(in), static_cast<float2 *>(out), log2n, batch, direction, s);
    const size_t unit = sizeof(float2) << log2n;
    return host_route(in, unit, bytes, out, unit, batch, s, [=](const void *din, void *dout, size_t, size_t count, void *) {
        return launch_fft_batch(static_cast<const float2 *>(din), static_cast<float2 *>(dout), log2n, count, direction, s);
    });
}

int dsp_gpu_fft_batch_f64(const void *in, void *out, int log2n, size_t batch, int direction, int mem, void *stream) {
    if (!in || !out) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_batch_f64: null buffer");
    if (log2n < 1 || log2n > 24) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_batch_f64: log2n %d outside [1,24]", log2n);
    if (direction != DSP_GPU_FORWARD && direction != DSP_GPU_INVERSE)
        return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_batch_f64: direction %d", direction);
    if (bad_mem(mem)) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_batch_f64: mem %d", mem);
    if (batch == 0) return DSP_GPU_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mem == DSP_GPU_MEM_DEVICE)
        return launch_fft_batch_f64(static_cast<const double *>(in), static_cast<double *>(out), log2n, batch, direction, s);
    const size_t unit = (2 * sizeof(double)) << log2n;
    return host_route(in, unit, unit * batch, out, unit, batch, s, [=](const void *din, void *dout, size_t, size_t count, void *) {
        return launch_fft_batch_f64(static_cast<const double *>(din), static_cast<double *>(dout), log2n, count, direction, s);
    });
}

int dsp_gpu_fft_batch_noalloc(const void *in, void *out, int log2n, size_t batch, int direction, void *stream) {
    if (!in || !out) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_batch_noalloc: null buffer");
    if (log2n < 1 || log2n > 21) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_batch_noalloc: log2n %d outside [1,21]", log2n);
    if (direction != DSP_GPU_FORWARD && direction != DSP_GPU_INVERSE)
        return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_batch_noalloc: direction %d", direction);
    return launch_fft_batch_noalloc(static_cast<const float2 *>(in), static_cast<float2 *>(out), log2n, batch, direction,
                                    static_cast<hipStream_t>(stream));
}

int dsp_gpu_fft_batch_timed(const void *in, void *out, int log2n, size_t batch, int direction, int iters,
                            float *ms_per_launch, void *stream) {
    if (!in || !out || !ms_per_launch || iters < 1) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_batch_timed: bad argument");
    if (log2n < 1 || log2n > 30) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_batch_timed: log2n %d", log2n);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipEvent_t e0, e1;
    DSPGPU_HIP(hipEventCreate(&e0));
    DSPGPU_HIP(hipEventCreate(&e1));
    int rc = DSP_GPU_OK;
    hipError_t he = hipEventRecord(e0, s);
    for (int i = 0; i < iters && rc == DSP_GPU_OK; ++i)
        rc = launch_fft_batch(static_cast<const float2 *>(in), static_cast<float2 *>(out), log2n, batch, direction, s);
    if (he == hipSuccess) he = hipEventRecord(e1, s);
    if (he == hipSuccess) he = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != DSP_GPU_OK) return rc;
    if (he != hipSuccess) return fail(DSP_GPU_EHIP, "event timing failed: %s", hipGetErrorString(he));
    *ms_per_launch = ms / static_cast<float>(iters);
    return DSP_GPU_OK;
}

int dsp_gpu_fft_real_batch(const float *in, size_t n_in, size_t in_stride, void *out, int nf, size_t batch,
                           int direction, int shifted, int mem, void *stream) {
    if (!in || !out) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_real_batch: null buffer");
    if (nf < 2) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_real_batch: nf %d < 2", nf);
    if (in_stride < n_in) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_real_batch: in_stride < n_in");
    if (direction != DSP_GPU_FORWARD && direction != DSP_GPU_INVERSE)
        return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_real_batch: direction %d", direction);
    if (bad_mem(mem)) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_real_batch: mem %d", mem);
    // fft.h:62-63: log2n = ceil(log2(nf))
    int log2n = 0;
    while ((static_cast<size_t>(1) << log2n) < static_cast<size_t>(nf)) ++log2n;
    if (log2n < 1 || log2n > 30) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_real_batch: padded size 2^%d unsupported", log2n);
    if (batch == 0) return DSP_GPU_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mem == DSP_GPU_MEM_DEVICE)
        return launch_fft_real_batch(in, n_in, in_stride, static_cast<float2 *>(out), log2n, batch, direction, shifted, s);
    const size_t in_bytes = ((batch - 1) * in_stride + n_in) * sizeof(float);
    const size_t out_bytes = (batch << log2n) * sizeof(float2);
    (void)out_bytes;
    return host_route(in, in_stride * sizeof(float), in_bytes, out, sizeof(float2) << log2n, batch, s,
                      [=](const void *din, void *dout, size_t, size_t count, void *) {
                          return launch_fft_real_batch(static_cast<const float *>(din), n_in, in_stride,
                                                       static_cast<float2 *>(dout), log2n, count, direction, shifted, s);
                      });
}

// shared by the one-shot and the streaming form; `history` == nullptr: one-shot (filter.h:123-140)
// Device buffers that overlap would have one workgroup overwrite samples another still has to read (FIR history,
// window and warm-up samples reach back across tile borders); host buffers are staged and may alias or overlap in any
// way (run_host_job stages the whole input before it writes anything back unless the two are the same array with equal
// unit sizes, the one case its pipeline handles chunk by chunk).
static bool device_overlap(const void *a, const void *b, size_t bytes, int mem) {
    if (mem != DSP_GPU_MEM_DEVICE) return false;
    const char *p = static_cast<const char *>(a), *q = static_cast<const char *>(b);
    return p < q + bytes && q < p + bytes;
}

static int convolve_impl(const char *who, const float *x, float *y, size_t channels, size_t n, const float *taps,
                         int ntaps, float *history, unsigned long long position, int mode, int mem, void *stream) {
    if (!x || !y || !taps) return fail(DSP_GPU_EINVAL, "%s: null buffer", who);
    if (mode != DSP_GPU_CONV_AUTO && mode != DSP_GPU_CONV_DIRECT && mode != DSP_GPU_CONV_OVERLAP_SAVE &&
        mode != DSP_GPU_CONV_DIRECT_VALU)
        return fail(DSP_GPU_EINVAL, "%s: mode %d", who, mode);
    if (mode == DSP_GPU_CONV_DIRECT_VALU && history)
        return fail(DSP_GPU_EUNSUPPORTED, "%s: the VALU direct form has no streaming variant", who);
    // a 4096-point block holds 2049 taps, a 16384-point block 8193; longer filters run partitioned (8192-tap segments
    // on 16384-point blocks, one overlap-save pass each, accumulated) in the one-shot form
    const bool os_wanted = mode == DSP_GPU_CONV_OVERLAP_SAVE || (mode == DSP_GPU_CONV_AUTO && (ntaps > 4096 || (ntaps >= 64 && n >= 256) || (ntaps >= 40 && n >= 262144)));  // scripts/fir_crossover.py, fir_short_channels.py
                                                                   // (long channels: the wave kernel at hop 3840 runs 1.6 ms on 1,024 x 2^20 at any
                                                                   // of these lengths, the direct kernel 1.60 at 31 taps and 1.93 at 63)
    const bool partitioned = ntaps > kSegmentTaps + 1 && !history && os_wanted;
    // (one-shot calls on several channels of 1,536 ... 4,096 samples: 4096-point blocks, because launch_os then runs the channels
    // as one signal -- 87,381 x 3,072 samples, 511 taps: 0.46 of HBM on two 2048-point blocks per channel, 0.58 that way)
    const bool as_one_signal = !history && channels >= 2 && n >= 1536 && n <= 4096 && ntaps <= 2049 && !getenv("DSP_GPU_FIR_NO_FLATTEN");
    const int os_log2n = as_one_signal ? 12 : os_block_log2n(ntaps, n, history != nullptr && position != 0);
    const int max_taps = partitioned ? kMaxPartitionedTaps : (os_wanted ? kSegmentTaps + 1 : 4096);
    if (ntaps < 1 || ntaps > max_taps)
        return fail(DSP_GPU_EINVAL, "%s: ntaps %d outside [1,%d] for this mode", who, ntaps, max_taps);
    if (bad_mem(mem)) return fail(DSP_GPU_EINVAL, "%s: mem %d", who, mem);
    if (channels == 0 || n == 0) return DSP_GPU_OK;
    if (device_overlap(x, y, channels * n * sizeof(float), mem))
        return fail(DSP_GPU_EINVAL, "%s: x and y overlap (the filter is not in-place on device buffers)", who);
    if (mode == DSP_GPU_CONV_AUTO)  // long filters on long signals amortise the FFT blocks (crossover: DESIGN.md 4.4)
        mode = os_wanted ? DSP_GPU_CONV_OVERLAP_SAVE : DSP_GPU_CONV_DIRECT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t bytes = channels * n * sizeof(float);
    if (!history && n < static_cast<size_t>(ntaps)) {
        // one-shot call on channels no longer than the filter's order: no output has a full window behind it, all stay zero
        // (filter.h:126,133) -- not worth 511 taps of arithmetic per sample
        if (mem == DSP_GPU_MEM_HOST) std::memset(y, 0, bytes);
        else DSPGPU_HIP(hipMemsetAsync(y, 0, bytes, s));
        return DSP_GPU_OK;
    }
    if (mem == DSP_GPU_MEM_HOST && !history) {
        // one-shot filters: channels are independent, so the batch is pipelined in chunks of channels (mode is resolved
        // above: every chunk filters the same way)
        const size_t row = n * sizeof(float);
        return host_route(x, row, bytes, y, row, channels, s, [=](const void *din, void *dout, size_t, size_t count, void *) {
            return convolve_impl(who, static_cast<const float *>(din), static_cast<float *>(dout), count, n, taps, ntaps, nullptr, 0,
                                 mode, DSP_GPU_MEM_DEVICE, stream);
        });
    }
    if (partitioned) {
        const float *dx = x;
        float *dy = y;
        HostStage hs;
        if (mem == DSP_GPU_MEM_HOST) {
            if (int rc = stage_in(x, bytes, bytes, s, &hs)) return rc;
            dx = static_cast<const float *>(hs.d_in);
            dy = static_cast<float *>(hs.d_out);
        }
        for (int seg0 = 0; seg0 < ntaps; seg0 += kSegmentTaps) {
            const int len = std::min(kSegmentTaps, ntaps - seg0);
            LeaseGuard seg{TapLease{}, s};
            if (int rc = acquire_taps(taps + seg0, len, kSegmentLog2N, s, &seg.lease)) return rc;
            if (int rc = launch_fir_os_segment(dx, dy, channels, n, seg.lease.d_spectrum, len, static_cast<size_t>(seg0),
                                               seg0 > 0, kSegmentLog2N, s))
                return rc;
        }
        // filter.h:126,133: outputs without a full window behind them stay zero
        const size_t head = std::min<size_t>(n, static_cast<size_t>(ntaps - 1));
        if (head) DSPGPU_HIP(hipMemset2DAsync(dy, n * sizeof(float), 0, head * sizeof(float), channels, s));
        return mem == DSP_GPU_MEM_HOST ? stage_out(y, bytes, s, hs) : DSP_GPU_OK;
    }
    LeaseGuard te{TapLease{}, s};
    if (int rc = acquire_taps(taps, ntaps, mode == DSP_GPU_CONV_OVERLAP_SAVE ? os_log2n : 0, s, &te.lease)) return rc;
    const size_t hist_bytes = channels * static_cast<size_t>(ntaps - 1) * sizeof(float);
    const float *dx = x;
    float *dy = y;
    float *dhist = history;
    HostStage hs;
    if (mem == DSP_GPU_MEM_HOST) {
        if (int rc = stage_in(x, bytes, bytes, s, &hs)) return rc;
        dx = static_cast<const float *>(hs.d_in);
        dy = static_cast<float *>(hs.d_out);
        if (history && hist_bytes) {
            void *p = nullptr;
            if (int rc = workspace(WS_STATE, hist_bytes, &p, s)) return rc;
            dhist = static_cast<float *>(p);
            DSPGPU_HIP(hipMemcpyAsync(dhist, history, hist_bytes, hipMemcpyHostToDevice, s));
        }
    }
    // the kernels read the old history only when this is not the first chunk of a stream; outputs whose
    // stream index is below ntaps-1 stay zero wherever the chunk boundaries fall (filter.h:126,133)
    const float *old_hist = (history && position && hist_bytes) ? dhist : nullptr;
    const unsigned long long order = static_cast<unsigned long long>(ntaps - 1);
    const int zero_head = position >= order ? 0 : static_cast<int>(order - position);
    int rc = mode == DSP_GPU_CONV_OVERLAP_SAVE
                 ? launch_fir_overlap_save(dx, dy, channels, n, te.lease.d_spectrum, ntaps, old_hist, zero_head, os_log2n, s)
             : mode == DSP_GPU_CONV_DIRECT_VALU ? launch_fir_direct_valu(dx, dy, channels, n, te.lease.d_taps, ntaps, s)
                                                : launch_fir_direct(dx, dy, channels, n, te.lease.d_taps, ntaps, old_hist, zero_head, s);
    if (rc) return rc;
    if (history && hist_bytes) {  // new history = last ntaps-1 samples of (old history ++ x), via a scratch copy
        void *p = nullptr;
        if (int rc2 = workspace(WS_FIR_HIST, hist_bytes, &p, s)) return rc2;
        if (int rc2 = launch_fir_history(dx, n, old_hist, static_cast<float *>(p), channels, ntaps, s)) return rc2;
        DSPGPU_HIP(hipMemcpyAsync(dhist, p, hist_bytes, hipMemcpyDeviceToDevice, s));
        if (mem == DSP_GPU_MEM_HOST) DSPGPU_HIP(hipMemcpyAsync(history, dhist, hist_bytes, hipMemcpyDeviceToHost, s));
    }
    return mem == DSP_GPU_MEM_HOST ? stage_out(y, bytes, s, hs) : DSP_GPU_OK;
}

int dsp_gpu_convolve(const float *x, float *y, size_t channels, size_t n, const float *taps, int ntaps, int mode,
                     int mem, void *stream) {
    return convolve_impl("dsp_gpu_convolve", x, y, channels, n, taps, ntaps, nullptr, 0, mode, mem, stream);
}

int dsp_gpu_convolve_stream(const float *x, float *y, size_t channels, size_t n, const float *taps, int ntaps,
                            float *history, unsigned long long position, int mode, int mem, void *stream) {
    if (!history && ntaps > 1) return fail(DSP_GPU_EINVAL, "dsp_gpu_convolve_stream: null history");
    return convolve_impl("dsp_gpu_convolve_stream", x, y, channels, n, taps, ntaps, history, position, mode, mem,
                         stream);
}

}  // extern "C"

namespace {
template <typename T>
int moving_average_impl(const char *who, const T *x, T *y, size_t channels, size_t n, int npoints, int mem, void *stream) {
    if (!x || !y) return fail(DSP_GPU_EINVAL, "%s: null buffer", who);
    if (npoints < 1) return fail(DSP_GPU_EINVAL, "%s: npoints %d < 1", who, npoints);
    if (bad_mem(mem)) return fail(DSP_GPU_EINVAL, "%s: mem %d", who, mem);
    if (channels == 0 || n == 0) return DSP_GPU_OK;
    if (n < static_cast<size_t>(npoints))
        return fail(DSP_GPU_EINVAL, "%s: n %zu < npoints %d (the reference reads out of bounds)", who, n, npoints);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t bytes = channels * n * sizeof(T);
    if (device_overlap(x, y, bytes, mem)) return fail(DSP_GPU_EINVAL, "%s: x and y overlap (not in-place on device buffers)", who);
    auto run = [=](const T *dx, T *dy, size_t count) {
        if constexpr (sizeof(T) == 8) return launch_moving_average_f64(dx, dy, count, n, npoints, s);
        else return launch_moving_average(dx, dy, count, n, npoints, s);
    };
    if (mem == DSP_GPU_MEM_DEVICE) return run(x, y, channels);
    const size_t row = n * sizeof(T);
    return host_route(x, row, bytes, y, row, channels, s, [=](const void *din, void *dout, size_t, size_t count, void *) {
        return run(static_cast<const T *>(din), static_cast<T *>(dout), count);
    });
}
}  // namespace

extern "C" {

int dsp_gpu_moving_average(const float *x, float *y, size_t channels, size_t n, int npoints, int mem, void *stream) {
    return moving_average_impl<float>("dsp_gpu_moving_average", x, y, channels, n, npoints, mem, stream);
}

int dsp_gpu_moving_average_f64(const double *x, double *y, size_t channels, size_t n, int npoints, int mem, void *stream) {
    return moving_average_impl<double>("dsp_gpu_moving_average_f64", x, y, channels, n, npoints, mem, stream);
}

// SincFilter<>::apply(SampleBuffer<double>&), filter.h:123-140: double samples, double taps (what the reference keeps:
// filter.h:57), double arithmetic on the double matrix cores.  The tap cache holds the bytes of the double taps.
int dsp_gpu_convolve_f64(const double *x, double *y, size_t channels, size_t n, const double *taps, int ntaps, int mode, int mem,
                         void *stream) {
    const char *who = "dsp_gpu_convolve_f64";
    if (!x || !y || !taps) return fail(DSP_GPU_EINVAL, "%s: null buffer", who);
    if (mode != DSP_GPU_CONV_AUTO && mode != DSP_GPU_CONV_DIRECT && mode != DSP_GPU_CONV_OVERLAP_SAVE)
        return fail(DSP_GPU_EINVAL, "%s: mode %d", who, mode);
    if (mode == DSP_GPU_CONV_OVERLAP_SAVE)
        return fail(DSP_GPU_EUNSUPPORTED, "%s: double samples are filtered in the time domain (DIRECT; AUTO picks it)", who);
    if (ntaps < 1 || ntaps > 4096) return fail(DSP_GPU_EINVAL, "%s: ntaps %d outside [1,4096]", who, ntaps);
    if (bad_mem(mem)) return fail(DSP_GPU_EINVAL, "%s: mem %d", who, mem);
    if (channels == 0 || n == 0) return DSP_GPU_OK;
    const size_t bytes = channels * n * sizeof(double);
    if (device_overlap(x, y, bytes, mem)) return fail(DSP_GPU_EINVAL, "%s: x and y overlap (the filter is not in-place on device buffers)", who);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n < static_cast<size_t>(ntaps)) {  // no output has a full window behind it: all stay zero (filter.h:126,133)
        if (mem == DSP_GPU_MEM_HOST) std::memset(y, 0, bytes);
        else DSPGPU_HIP(hipMemsetAsync(y, 0, bytes, s));
        return DSP_GPU_OK;
    }
    auto run = [=](const double *dx, double *dy, size_t count) -> int {
        LeaseGuard te{TapLease{}, s};
        if (int rc = acquire_taps(reinterpret_cast<const float *>(taps), 2 * ntaps, 0, s, &te.lease)) return rc;
        return launch_fir_direct_f64(dx, dy, count, n, reinterpret_cast<const double *>(te.lease.d_taps), ntaps, s);
    };
    if (mem == DSP_GPU_MEM_DEVICE) return run(x, y, channels);
    const size_t row = n * sizeof(double);
    return host_route(x, row, bytes, y, row, channels, s, [=](const void *din, void *dout, size_t, size_t count, void *) {
        return run(static_cast<const double *>(din), static_cast<double *>(dout), count);
    });
}

}  // extern "C"

namespace {

// dsp_gpu_iir / dsp_gpu_iir_f64: T = float or double samples
template <typename T>
int iir_impl(const char *who, const T *x, T *y, size_t channels, size_t n, const double *a, const double *b, int poles,
             double c0, double d0, double *state, int mode, int mem, void *stream) {
    constexpr bool kF64 = sizeof(T) == 8;
    if (!x || !y || !a || !b) return fail(DSP_GPU_EINVAL, "%s: null buffer", who);
    if (poles < 1 || poles > 20) return fail(DSP_GPU_EINVAL, "%s: poles %d outside [1,20]", who, poles);
    if (mode != DSP_GPU_IIR_AUTO && mode != DSP_GPU_IIR_BATCH && mode != DSP_GPU_IIR_SCAN && mode != DSP_GPU_IIR_BATCH_REF)
        return fail(DSP_GPU_EINVAL, "%s: mode %d", who, mode);
    if (bad_mem(mem)) return fail(DSP_GPU_EINVAL, "%s: mem %d", who, mem);
    if (channels == 0 || n == 0) return DSP_GPU_OK;
    IirCoef co{};
    co.poles = poles;
    for (int i = 0; i <= poles; ++i) {
        co.a[i] = a[i];
        co.b[i] = i == 0 ? 0.0 : b[i];  // coefficients.h:93 requires b[0] == 0
        if (!std::isfinite(co.a[i]) || !std::isfinite(co.b[i])) return fail(DSP_GPU_EINVAL, "%s: non-finite coefficient", who);
    }
    co.c0 = c0;
    co.d0 = d0;
    if (!std::isfinite(c0) || !std::isfinite(d0)) return fail(DSP_GPU_EINVAL, "%s: non-finite c0/d0", who);
    const bool auto_mode = mode == DSP_GPU_IIR_AUTO;
    if (auto_mode)  // one lane per channel needs tens of thousands of channels to fill the chip; measured crossover on
                    // 2^28 samples in total (scripts/iir_auto_threshold.py): 16,384 ch scan 0.53 / batch 1.08 ms,
                    // 32,768 ch 0.67 / 0.58 ms.  Rows of 16,384 samples and more go to the scan at any channel count: a
                    // wave of the batch kernel then walks 64 rows that lie 64 KB or more apart, which the memory system
                    // serves at 0.44-0.59 of 8 TB/s against the scan's 0.62-0.69 (65,536 x 16,384 / 53,687 x 20,000 /
                    // 32,768 x 32,768, profiles/r03_iir_batch_row_lengths.txt).  AUTO never picks the reference arithmetic.
        mode = (((channels < 24576 && n >= 2048) || n >= 16384) && poles <= 8) ? DSP_GPU_IIR_SCAN : DSP_GPU_IIR_BATCH;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t bytes = channels * n * sizeof(T);
    if (device_overlap(x, y, bytes, mem)) {
        // the sequential kernel reads every tile before it writes it, so y == x is fine there; the scan's warm-up
        // samples are not (AUTO then takes the sequential kernel)
        if (x != y || (mode == DSP_GPU_IIR_SCAN && !auto_mode))
            return fail(DSP_GPU_EINVAL, "%s: x and y overlap (only y == x in BATCH mode is in-place)", who);
        if (mode == DSP_GPU_IIR_SCAN) mode = DSP_GPU_IIR_BATCH;
    }
    const size_t state_bytes = channels * 2 * static_cast<size_t>(poles) * sizeof(double);
    auto run = [=](const T *dx, T *dy, size_t count, double *dstate) {
        if constexpr (kF64) {
            if (mode == DSP_GPU_IIR_BATCH_REF) return launch_iir_batch_ref_f64(dx, dy, count, n, co, dstate, nullptr, s);
            if (auto_mode && mode == DSP_GPU_IIR_BATCH && poles > 8 && dx != dy && ((count < 24576 && n >= 2048) || n >= 16384)) {
                // (AUTO asks for the scan up to eight poles only, the fp32 kernel's limit: the time-sharded form has none)
                const int rc = launch_iir_scan_f64(dx, dy, count, n, co, dstate, s);
                if (rc != DSP_GPU_EUNSUPPORTED) return rc;
            }
            int rc = mode == DSP_GPU_IIR_SCAN ? launch_iir_scan_f64(dx, dy, count, n, co, dstate, s)
                                              : launch_iir_batch_f64(dx, dy, count, n, co, dstate, nullptr, s);
            if (rc == DSP_GPU_EUNSUPPORTED && auto_mode && mode == DSP_GPU_IIR_SCAN)
                rc = launch_iir_batch_f64(dx, dy, count, n, co, dstate, nullptr, s);
            return rc;
        } else {
            if (mode == DSP_GPU_IIR_BATCH_REF) return launch_iir_batch_ref(dx, dy, count, n, co, dstate, nullptr, s);
            if (auto_mode && mode == DSP_GPU_IIR_BATCH && poles > 8 && dx != dy && ((count < 24576 && n >= 2048) || n >= 16384)) {
                // more poles than the scan has a kernel for, few long channels: the one-lane kernel sharded in time
                const int rc = launch_iir_sharded(dx, dy, count, n, co, dstate, s);
                if (rc != DSP_GPU_EUNSUPPORTED) return rc;
            }
            int rc = mode == DSP_GPU_IIR_SCAN ? launch_iir_scan(dx, dy, count, n, co, dstate, s)
                                              : launch_iir_batch(dx, dy, count, n, co, dstate, nullptr, s);
            if (rc == DSP_GPU_EUNSUPPORTED && auto_mode && mode == DSP_GPU_IIR_SCAN)  // ill-conditioned for the scan
                rc = launch_iir_batch(dx, dy, count, n, co, dstate, nullptr, s);
            return rc;
        }
    };
    if (mem == DSP_GPU_MEM_DEVICE) return run(x, y, channels, state);
    // host route: channels are independent (the mode was chosen above for the whole batch, so every chunk runs the same
    // kernels); the small state array rides along whole
    const size_t row = n * sizeof(T);
    return host_route(
        x, row, bytes, y, row, channels, s,
        [=](const void *din, void *dout, size_t u0, size_t count, void *daux) {
            double *dstate = static_cast<double *>(daux);
            return run(static_cast<const T *>(din), static_cast<T *>(dout), count,
                       dstate ? dstate + u0 * 2 * static_cast<size_t>(poles) : nullptr);
        },
        state, state_bytes);
}

}  // namespace

extern "C" {

int dsp_gpu_iir(const float *x, float *y, size_t channels, size_t n, const double *a, const double *b, int poles,
                double c0, double d0, double *state, int mode, int mem, void *stream) {
    return iir_impl<float>("dsp_gpu_iir", x, y, channels, n, a, b, poles, c0, d0, state, mode, mem, stream);
}

int dsp_gpu_iir_f64(const double *x, double *y, size_t channels, size_t n, const double *a, const double *b, int poles,
                    double c0, double d0, double *state, int mode, int mem, void *stream) {
    return iir_impl<double>("dsp_gpu_iir_f64", x, y, channels, n, a, b, poles, c0, d0, state, mode, mem, stream);
}

int dsp_gpu_iir_settle(const double *a, const double *b, int poles, size_t *samples) {
    if (!a || !b || !samples) return fail(DSP_GPU_EINVAL, "dsp_gpu_iir_settle: null argument");
    if (poles < 1 || poles > 20) return fail(DSP_GPU_EINVAL, "dsp_gpu_iir_settle: poles %d outside [1,20]", poles);
    IirCoef co{};
    co.poles = poles;
    for (int i = 0; i <= poles; ++i) {
        co.a[i] = a[i];
        co.b[i] = i == 0 ? 0.0 : b[i];
        if (!std::isfinite(co.a[i]) || !std::isfinite(co.b[i])) return fail(DSP_GPU_EINVAL, "dsp_gpu_iir_settle: non-finite coefficient");
    }
    co.c0 = 1.0;
    co.d0 = 0.0;
    return iir_settle_samples(co, samples);
}

int dsp_gpu_iir_multi_ex(const void *xv, void *yv, int sample_type, size_t channels, size_t n, const double *a, const double *b,
                         int poles, const double *c0, const double *d0, double *state, int mode, int mem, void *stream) {
    if (!xv || !yv || !a || !b) return fail(DSP_GPU_EINVAL, "dsp_gpu_iir_multi: null buffer");
    if (poles < 1 || poles > 8) return fail(DSP_GPU_EINVAL, "dsp_gpu_iir_multi: poles %d outside [1,8]", poles);
    if (sample_type != DSP_GPU_F32 && sample_type != DSP_GPU_F64) return fail(DSP_GPU_EINVAL, "dsp_gpu_iir_multi: sample type %d", sample_type);
    if (mode != DSP_GPU_IIR_AUTO && mode != DSP_GPU_IIR_BATCH && mode != DSP_GPU_IIR_BATCH_REF)
        return fail(DSP_GPU_EINVAL, "dsp_gpu_iir_multi: mode %d (one lane per channel: AUTO, BATCH or BATCH_REF)", mode);
    if (bad_mem(mem)) return fail(DSP_GPU_EINVAL, "dsp_gpu_iir_multi: mem %d", mem);
    if (channels == 0 || n == 0) return DSP_GPU_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    IirCoef co{};
    co.poles = poles;
    const bool f64 = sample_type == DSP_GPU_F64, ref = mode == DSP_GPU_IIR_BATCH_REF;
    const size_t bytes = channels * n * (f64 ? sizeof(double) : sizeof(float));
    const size_t state_bytes = channels * 2 * static_cast<size_t>(poles) * sizeof(double);
    const size_t ab = channels * static_cast<size_t>(poles + 1);  // doubles in a (and in b)
    const void *dx = xv;
    void *dy = yv;
    double *dstate = state;
    IirPerChannel pc{a, b, c0, d0};
    HostStage hs;
    if (mem == DSP_GPU_MEM_HOST) {
        for (size_t i = 0; i < ab; ++i)
            if (!std::isfinite(a[i]) || !std::isfinite(b[i])) return fail(DSP_GPU_EINVAL, "dsp_gpu_iir_multi: non-finite coefficient");
        if (int rc = stage_in(xv, bytes, bytes, s, &hs)) return rc;
        dx = hs.d_in;
        dy = hs.d_out;
        void *p = nullptr;
        if (int rc = workspace(WS_SCRATCH_A, (2 * ab + 2 * channels) * sizeof(double), &p, s)) return rc;
        double *dc = static_cast<double *>(p);
        DSPGPU_HIP(hipMemcpyAsync(dc, a, ab * sizeof(double), hipMemcpyHostToDevice, s));
        DSPGPU_HIP(hipMemcpyAsync(dc + ab, b, ab * sizeof(double), hipMemcpyHostToDevice, s));
        pc.a = dc;
        pc.b = dc + ab;
        if (c0) {
            DSPGPU_HIP(hipMemcpyAsync(dc + 2 * ab, c0, channels * sizeof(double), hipMemcpyHostToDevice, s));
            pc.c0 = dc + 2 * ab;
        }
        if (d0) {
            DSPGPU_HIP(hipMemcpyAsync(dc + 2 * ab + channels, d0, channels * sizeof(double), hipMemcpyHostToDevice, s));
            pc.d0 = dc + 2 * ab + channels;
        }
        if (state) {
            if (int rc = workspace(WS_STATE, state_bytes, &p, s)) return rc;
            dstate = static_cast<double *>(p);
            DSPGPU_HIP(hipMemcpyAsync(dstate, state, state_bytes, hipMemcpyHostToDevice, s));
        }
    }
    int rc;
    if (f64) {
        const double *x64 = static_cast<const double *>(dx);
        double *y64 = static_cast<double *>(dy);
        rc = ref ? launch_iir_batch_ref_f64(x64, y64, channels, n, co, dstate, &pc, s)
                 : launch_iir_batch_f64(x64, y64, channels, n, co, dstate, &pc, s);
    } else {
        const float *x32 = static_cast<const float *>(dx);
        float *y32 = static_cast<float *>(dy);
        rc = ref ? launch_iir_batch_ref(x32, y32, channels, n, co, dstate, &pc, s)
                 : launch_iir_batch(x32, y32, channels, n, co, dstate, &pc, s);
    }
    if (rc) return rc;
    if (mem == DSP_GPU_MEM_HOST) {
        if (state) DSPGPU_HIP(hipMemcpyAsync(state, dstate, state_bytes, hipMemcpyDeviceToHost, s));
        return stage_out(yv, bytes, s, hs);
    }
    return DSP_GPU_OK;
}

int dsp_gpu_iir_multi(const float *x, float *y, size_t channels, size_t n, const double *a, const double *b, int poles,
                      const double *c0, const double *d0, double *state, int mem, void *stream) {
    return dsp_gpu_iir_multi_ex(x, y, DSP_GPU_F32, channels, n, a, b, poles, c0, d0, state, DSP_GPU_IIR_BATCH, mem, stream);
}

int dsp_gpu_biquad_swept(const float *x, const float *fc, float *y, size_t channels, size_t n, size_t fc_stride, int kind,
                         double bandwidth, double *state, int mem, void *stream) {
    if (!x || !fc || !y) return fail(DSP_GPU_EINVAL, "dsp_gpu_biquad_swept: null buffer");
    if (kind != DSP_GPU_SWEPT_BANDPASS && kind != DSP_GPU_SWEPT_BANDREJECT)
        return fail(DSP_GPU_EINVAL, "dsp_gpu_biquad_swept: kind %d", kind);
    if (!std::isfinite(bandwidth)) return fail(DSP_GPU_EINVAL, "dsp_gpu_biquad_swept: non-finite bandwidth");
    if (bad_mem(mem)) return fail(DSP_GPU_EINVAL, "dsp_gpu_biquad_swept: mem %d", mem);
    if (channels == 0 || n == 0) return DSP_GPU_OK;
    if (fc_stride != 0 && fc_stride != n)
        return fail(DSP_GPU_EINVAL, "dsp_gpu_biquad_swept: fc_stride %zu is neither 0 (one sweep for all channels) nor n", fc_stride);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t bytes = channels * n * sizeof(float);
    const size_t fc_bytes = (fc_stride ? channels : 1) * n * sizeof(float);
    const size_t state_bytes = channels * 4 * sizeof(double);
    // every tile is read (x and fc) before it is written, so y == x is in place; so is y == fc when each channel has
    // its own sweep.  Anything else that overlaps is refused.
    if (device_overlap(x, y, bytes, mem) && x != y) return fail(DSP_GPU_EINVAL, "dsp_gpu_biquad_swept: x and y overlap");
    if (mem == DSP_GPU_MEM_DEVICE) {  // [fc, fc + fc_bytes) against [y, y + bytes): the two ranges differ in length
        const char *f0 = reinterpret_cast<const char *>(fc), *y0 = reinterpret_cast<const char *>(y);
        if (f0 < y0 + bytes && y0 < f0 + fc_bytes && (f0 != y0 || fc_stride == 0))
            return fail(DSP_GPU_EINVAL, "dsp_gpu_biquad_swept: fc and y overlap");
    }
    if (mem == DSP_GPU_MEM_DEVICE) return launch_biquad_swept(x, fc, y, channels, n, fc_stride, kind, bandwidth, state, s);
    HostStage hs;
    if (int rc = stage_in(x, bytes, bytes, s, &hs)) return rc;
    void *p = nullptr;
    if (int rc = workspace(WS_SCRATCH_A, fc_bytes, &p, s)) return rc;
    float *dfc = static_cast<float *>(p);
    DSPGPU_HIP(hipMemcpyAsync(dfc, fc, fc_bytes, hipMemcpyHostToDevice, s));
    double *dstate = nullptr;
    if (state) {
        if (int rc = workspace(WS_STATE, state_bytes, &p, s)) return rc;
        dstate = static_cast<double *>(p);
        DSPGPU_HIP(hipMemcpyAsync(dstate, state, state_bytes, hipMemcpyHostToDevice, s));
    }
    if (int rc = launch_biquad_swept(static_cast<const float *>(hs.d_in), dfc, static_cast<float *>(hs.d_out), channels, n,
                                     fc_stride, kind, bandwidth, dstate, s))
        return rc;
    if (state) DSPGPU_HIP(hipMemcpyAsync(state, dstate, state_bytes, hipMemcpyDeviceToHost, s));
    return stage_out(y, bytes, s, hs);
}

int dsp_gpu_fft_columns(const void *in, void *out, int log2n, size_t cols, int direction, void *stream) {
    if (!in || !out) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_columns: null buffer");
    if (direction != DSP_GPU_FORWARD && direction != DSP_GPU_INVERSE)
        return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_columns: direction %d", direction);
    if (cols == 0) return DSP_GPU_OK;
    return launch_fft_columns(static_cast<const float2 *>(in), static_cast<float2 *>(out), log2n, cols, direction,
                              static_cast<hipStream_t>(stream));
}

int dsp_gpu_swap_axes(const void *in, void *out, size_t a, size_t b, size_t c, int tw_log2n, size_t tw_row0,
                      int direction, void *stream) {
    if (!in || !out || in == out) return fail(DSP_GPU_EINVAL, "dsp_gpu_swap_axes: null or aliased buffer");
    if (tw_log2n < 0 || tw_log2n > 40) return fail(DSP_GPU_EINVAL, "dsp_gpu_swap_axes: tw_log2n %d", tw_log2n);
    if (direction != DSP_GPU_FORWARD && direction != DSP_GPU_INVERSE)
        return fail(DSP_GPU_EINVAL, "dsp_gpu_swap_axes: direction %d", direction);
    return launch_swap_axes(static_cast<const float2 *>(in), static_cast<float2 *>(out), a, b, c, tw_log2n, tw_row0,
                            direction, static_cast<hipStream_t>(stream));
}

int dsp_gpu_fft_rows_exchange(const void *in, void *out, int log2n2, size_t rows, size_t world, int tw_log2n,
                              size_t tw_row0, int direction, void *stream) {
    if (!in || !out || in == out) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_rows_exchange: null or aliased buffer");
    if (log2n2 < 1 || log2n2 > 24 || tw_log2n < log2n2 || tw_log2n > 40)
        return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_rows_exchange: log2n2 %d, tw_log2n %d", log2n2, tw_log2n);
    if (direction != DSP_GPU_FORWARD && direction != DSP_GPU_INVERSE)
        return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_rows_exchange: direction %d", direction);
    if (rows == 0) return DSP_GPU_OK;
    return launch_fft_rows_exchange(static_cast<const float2 *>(in), static_cast<float2 *>(out), log2n2, rows, world,
                                    tw_log2n, tw_row0, direction, static_cast<hipStream_t>(stream));
}

int dsp_gpu_transpose_blocks(const void *in, void *out, size_t blocks, size_t rows, size_t cols, void *stream) {
    if (!in || !out || in == out) return fail(DSP_GPU_EINVAL, "dsp_gpu_transpose_blocks: null or aliased buffer");
    return launch_transpose_blocks(static_cast<const float2 *>(in), static_cast<float2 *>(out), blocks, rows, cols,
                                   static_cast<hipStream_t>(stream));
}

int dsp_gpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *dsp_gpu_last_error_string(void) { return last_error_buffer(); }

int dsp_gpu_shutdown(void) {
    release_all_workspaces();
    return DSP_GPU_OK;
}

const char *dsp_gpu_version(void) { return "dsp_gpu 0.1 (gfx950)"; }

}  // extern "C"
