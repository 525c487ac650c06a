// host_path.hip -- the DSP_GPU_MEM_HOST route: caller buffers in host memory (what every reference signature takes:
// SampleBuffer is a std::vector, sample_buffer.h:14-133), device work in between.
//
// The batch is cut into chunks of whole units (transforms / channels, which are independent) and pipelined:
//     host copy-in (i+1)  ||  H2D (i+1)  ||  kernels (i)  ||  D2H (i-1)  ||  host copy-out (i-2)
// Pageable caller memory cannot be DMA'd directly; it is copied by a few host threads into / out of a small ring of
// page-locked slots (three each way), so that the link runs both directions at once at its page-locked rate instead of
// one direction at a time through the runtime's own bounce buffers.  Caller buffers that are already page-locked
// (hipHostMalloc / hipHostRegister / torch pin_memory) skip the ring.  Small calls take the same code with one chunk:
// one page-locked bounce each way.  Device side: the library's two staging workspaces hold the whole batch, so kernels
// see the same contiguous arrays as on the device-pointer route.  No arithmetic happens on the host.
#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"

namespace dspgpu {
namespace {

constexpr int kSlots = 3;
constexpr size_t kChunkBytes = 8u << 20;      // target size of a chunk (each way)
constexpr size_t kMaxSlotBytes = 32u << 20;   // larger chunks (a single huge unit) bypass the ring
constexpr size_t kPipelineFrom = 4u << 20;    // total bytes (in + out) from which a batch is cut up
constexpr size_t kSmallCall = 1u << 20;       // total bytes up to which a call takes the single-bounce route
constexpr int kMaxDev = 64;

// ---- a handful of persistent copy threads -----------------------------------------------------------------------
struct CopyTask {
    char *dst;
    const char *src;
    size_t bytes;
};

class CopyPool {
public:
    static CopyPool &get() {
        static CopyPool p;
        return p;
    }
    // copies every task, cut into pieces, on the pool's threads plus the caller; returns when all are done
    void run(const CopyTask *tasks, int ntasks) {
        size_t total = 0;
        for (int i = 0; i < ntasks; ++i) total += tasks[i].bytes;
        if (total == 0) return;
        if (total < (1u << 20) || workers_.empty()) {  // not worth waking anybody
            for (int i = 0; i < ntasks; ++i)
                if (tasks[i].bytes) std::memcpy(tasks[i].dst, tasks[i].src, tasks[i].bytes);
            return;
        }
        // one parallel copy at a time; a caller that finds the pool busy (another device's host thread is inside) copies
        // its own pieces inline rather than queueing behind it
        std::unique_lock<std::mutex> call(call_mutex_, std::try_to_lock);
        if (!call.owns_lock()) {
            for (int i = 0; i < ntasks; ++i)
                if (tasks[i].bytes) std::memcpy(tasks[i].dst, tasks[i].src, tasks[i].bytes);
            return;
        }
        unsigned long long gen;
        {
            std::lock_guard<std::mutex> lock(m_);
            pieces_.clear();
            const size_t piece = std::max<size_t>(1u << 20, (total / (4 * (workers_.size() + 1)) + 4095) & ~static_cast<size_t>(4095));
            for (int i = 0; i < ntasks; ++i)
                for (size_t off = 0; off < tasks[i].bytes; off += piece)
                    pieces_.push_back(CopyTask{tasks[i].dst + off, tasks[i].src + off, std::min(piece, tasks[i].bytes - off)});
            next_ = 0;
            left_ = pieces_.size();
            gen = ++generation_;
        }
        cv_.notify_all();
        work(gen);  // the caller copies too
        std::unique_lock<std::mutex> lock(m_);
        done_cv_.wait(lock, [&] { return left_ == 0; });
    }

private:
    CopyPool() {
        int n = 6;
        if (const char *e = getenv("DSP_GPU_HOST_THREADS")) n = std::atoi(e) - 1;
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw && static_cast<unsigned>(n) + 1 > hw) n = static_cast<int>(hw) - 1;
        for (int i = 0; i < std::max(n, 0); ++i) workers_.emplace_back([this] { loop(); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lock(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread &t : workers_) t.join();
    }
    // Claims pieces of generation `gen` until none is left.  The piece list is only ever read under m_ and only for the
    // generation the thread was woken for: a worker that is late (run() returns as soon as the LAST piece is copied, not
    // when every worker has noticed) finds the generation changed and leaves the next call's list alone.  Pieces are
    // >= 1 MiB, so the lock per claim costs nothing.
    void work(unsigned long long gen) {
        for (;;) {
            CopyTask t;
            {
                std::lock_guard<std::mutex> lock(m_);
                if (gen != generation_ || next_ >= pieces_.size()) return;
                t = pieces_[next_++];
            }
            std::memcpy(t.dst, t.src, t.bytes);
            std::lock_guard<std::mutex> lock(m_);
            if (--left_ == 0) done_cv_.notify_all();
        }
    }
    void loop() {
        unsigned long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lock(m_);
                cv_.wait(lock, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
            }
            work(seen);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_, call_mutex_;
    std::condition_variable cv_, done_cv_;
    std::vector<CopyTask> pieces_;
    size_t next_ = 0;
    size_t left_ = 0;
    unsigned long long generation_ = 0;
    bool stop_ = false;
};

// ---- per-device pipeline state -------------------------------------------------------------------------------
struct Pipe {
    hipStream_t h2d = nullptr, d2h = nullptr;
    hipEvent_t ev_h2d[kSlots] = {}, ev_k[kSlots] = {}, ev_d2h[kSlots] = {};
    char *ring_in[kSlots] = {}, *ring_out[kSlots] = {};
    size_t cap_in = 0, cap_out = 0;
    bool ready = false;
};
// One per device, held for the whole of a host-route call on that device (such calls share the device's staging
// workspaces and ring slots); host threads driving DIFFERENT devices -- the header's intended multi-GPU use -- do not
// serialise on each other.
std::mutex g_pipe_mutex[kMaxDev];
Pipe g_pipe[kMaxDev];

int pipe_init(Pipe &p) {
    if (p.ready) return DSP_GPU_OK;
    DSPGPU_HIP(hipStreamCreateWithFlags(&p.h2d, hipStreamNonBlocking));
    DSPGPU_HIP(hipStreamCreateWithFlags(&p.d2h, hipStreamNonBlocking));
    for (int i = 0; i < kSlots; ++i) {
        DSPGPU_HIP(hipEventCreateWithFlags(&p.ev_h2d[i], hipEventDisableTiming));
        DSPGPU_HIP(hipEventCreateWithFlags(&p.ev_k[i], hipEventDisableTiming));
        DSPGPU_HIP(hipEventCreateWithFlags(&p.ev_d2h[i], hipEventDisableTiming));
    }
    p.ready = true;
    return DSP_GPU_OK;
}

int ring_reserve(char *(&ring)[kSlots], size_t &cap, size_t bytes) {
    if (cap >= bytes) return DSP_GPU_OK;
    size_t want = 64u << 10;
    while (want < bytes) want <<= 1;
    for (int i = 0; i < kSlots; ++i) {
        if (ring[i]) (void)hipHostFree(ring[i]);
        ring[i] = nullptr;
    }
    cap = 0;
    for (int i = 0; i < kSlots; ++i)
        if (hipHostMalloc(reinterpret_cast<void **>(&ring[i]), want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return fail(DSP_GPU_ENOMEM, "page-locked staging slot of %zu bytes could not be allocated", want);
        }
    cap = want;
    return DSP_GPU_OK;
}

bool page_locked(const void *p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();  // plain pageable memory is "invalid value" to the runtime
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

}  // namespace

void release_host_pipes(int dev) {  // with device `dev` current and idle (dsp_gpu_shutdown)
    if (dev < 0 || dev >= kMaxDev) return;
    std::lock_guard<std::mutex> lock(g_pipe_mutex[dev]);
    Pipe &p = g_pipe[dev];
    if (!p.ready) return;
    for (int i = 0; i < kSlots; ++i) {
        if (p.ring_in[i]) (void)hipHostFree(p.ring_in[i]);
        if (p.ring_out[i]) (void)hipHostFree(p.ring_out[i]);
        (void)hipEventDestroy(p.ev_h2d[i]);
        (void)hipEventDestroy(p.ev_k[i]);
        (void)hipEventDestroy(p.ev_d2h[i]);
    }
    (void)hipStreamDestroy(p.h2d);
    (void)hipStreamDestroy(p.d2h);
    p = Pipe{};
}

bool host_pipe_in_use(int dev) {
    if (dev < 0 || dev >= kMaxDev) return false;
    std::lock_guard<std::mutex> lock(g_pipe_mutex[dev]);
    return g_pipe[dev].ready;
}

int run_host_job(const HostJob &job, hipStream_t s) {
    int dev = 0;
    DSPGPU_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDev) return fail(DSP_GPU_EUNSUPPORTED, "device index %d out of range", dev);
    const size_t out_total = job.out_unit * job.units;
    const size_t aux_bytes = job.aux ? job.aux_bytes : 0;
    auto up16 = [](size_t v) { return (v + 15) & ~static_cast<size_t>(15); };

    std::lock_guard<std::mutex> lock(g_pipe_mutex[dev]);
    Pipe &p = g_pipe[dev];
    if (int rc = pipe_init(p)) return rc;

    if (job.in_total + out_total + aux_bytes <= kSmallCall) {
        // Small calls (the reference's demo sizes: a 1024-point frame, a 256-sample audio block) are all latency: one
        // page-locked bounce and ONE transfer each way on the caller's own stream, no second stream, no event, no pointer
        // query.  Device layout [out | aux | in]: the way up carries [aux | in], the way back [out | aux].
        const size_t o_aux = up16(out_total), o_in = o_aux + up16(aux_bytes), all = o_in + up16(job.in_total);
        void *dv = nullptr;
        if (int rc = workspace(WS_STAGE_OUT, std::max<size_t>(all, 16), &dv, s)) return rc;
        char *d = static_cast<char *>(dv);
        if (int rc = ring_reserve(p.ring_in, p.cap_in, all)) return rc;
        char *h = p.ring_in[0];
        if (aux_bytes) std::memcpy(h + o_aux, job.aux, aux_bytes);
        std::memcpy(h + o_in, job.in, job.in_total);
        if (all > o_aux) DSPGPU_HIP(hipMemcpyAsync(d + o_aux, h + o_aux, all - o_aux, hipMemcpyHostToDevice, s));
        if (int rc = job.launch(d + o_in, d, 0, job.units, aux_bytes ? d + o_aux : nullptr)) {
            (void)hipStreamSynchronize(s);
            return rc;
        }
        const size_t back = aux_bytes ? o_aux + aux_bytes : out_total;
        if (back) DSPGPU_HIP(hipMemcpyAsync(h, d, back, hipMemcpyDeviceToHost, s));
        DSPGPU_HIP(hipStreamSynchronize(s));
        std::memcpy(job.out, h, out_total);
        if (aux_bytes) std::memcpy(job.aux, h + o_aux, aux_bytes);
        return DSP_GPU_OK;
    }

    void *din_v = nullptr, *dout_v = nullptr, *daux = nullptr;
    if (int rc = workspace(WS_STAGE_IN, std::max<size_t>(job.in_total, 16), &din_v, s)) return rc;
    if (int rc = workspace(WS_STAGE_OUT, std::max<size_t>(out_total, 16), &dout_v, s)) return rc;
    char *din = static_cast<char *>(din_v), *dout = static_cast<char *>(dout_v);
    if (aux_bytes) {
        if (int rc = workspace(WS_STATE, aux_bytes, &daux, s)) return rc;
        DSPGPU_HIP(hipMemcpyAsync(daux, job.aux, aux_bytes, hipMemcpyHostToDevice, s));
    }

    // chunks of whole units
    size_t chunks = 1;
    if (job.units >= 2 && job.in_total + out_total >= kPipelineFrom)
        chunks = std::min<size_t>(job.units, std::max<size_t>(2, (std::max(job.in_total, out_total) + kChunkBytes - 1) / kChunkBytes));
    // Caller buffers that overlap: the pipeline writes chunk i-2 of `out` while later chunks of `in` are still unread.
    // That is safe when the two are the SAME array with equal unit sizes (chunk k's output lands on chunk k's input,
    // already staged); for any other overlap (y = x + k, or in == out with in_unit != out_unit: the real-input FFT)
    // everything is staged in before anything is written back, as the unpipelined route did.
    {
        const char *i0 = job.in, *i1 = job.in + job.in_total, *o0 = job.out, *o1 = job.out + out_total;
        const bool overlap = i0 < o1 && o0 < i1;
        if (overlap && !(i0 == o0 && job.in_unit == job.out_unit)) chunks = 1;
    }
    auto unit_of = [&](size_t k) { return job.units * k / chunks; };
    auto in_off = [&](size_t u) { return std::min(u * job.in_unit, job.in_total); };  // the last unit may be short
    size_t max_in = 0, max_out = 0;
    for (size_t k = 0; k < chunks; ++k) {
        max_in = std::max(max_in, in_off(unit_of(k + 1)) - in_off(unit_of(k)));
        max_out = std::max(max_out, (unit_of(k + 1) - unit_of(k)) * job.out_unit);
    }
    // the ring is for pageable memory cut into modest chunks; page-locked caller buffers and single huge units go direct
    const bool ring_in = !page_locked(job.in) && max_in <= kMaxSlotBytes;
    const bool ring_out = !page_locked(job.out) && max_out <= kMaxSlotBytes;
    if (ring_in)
        if (int rc = ring_reserve(p.ring_in, p.cap_in, max_in)) return rc;
    if (ring_out)
        if (int rc = ring_reserve(p.ring_out, p.cap_out, max_out)) return rc;
    CopyPool &pool = CopyPool::get();

    // what came before on the caller's stream may still read the staging buffers (an earlier asynchronous use is not
    // possible through this API, which returns only when its output is ready -- but keep the order explicit)
    DSPGPU_HIP(hipEventRecord(p.ev_k[0], s));
    DSPGPU_HIP(hipStreamWaitEvent(p.h2d, p.ev_k[0], 0));

    for (size_t i = 0; i < chunks + 2; ++i) {
        // stage B: kernels of chunk i-1 on the caller's stream, then its way back.  Enqueued first, so that the GPU is
        // busy with them while the host copies below.
        if (i >= 1 && i - 1 < chunks) {
            const size_t kb = i - 1, u0 = unit_of(kb), u1 = unit_of(kb + 1);
            DSPGPU_HIP(hipStreamWaitEvent(s, p.ev_h2d[kb % kSlots], 0));
            if (int rc = job.launch(din + in_off(u0), dout + u0 * job.out_unit, u0, u1 - u0, daux)) {
                (void)hipStreamSynchronize(p.h2d);
                (void)hipStreamSynchronize(p.d2h);
                (void)hipStreamSynchronize(s);
                return rc;
            }
            DSPGPU_HIP(hipEventRecord(p.ev_k[kb % kSlots], s));
            DSPGPU_HIP(hipStreamWaitEvent(p.d2h, p.ev_k[kb % kSlots], 0));
            // its slot last held chunk i-4, which the host copied out two iterations ago
            char *dst = ring_out ? p.ring_out[kb % kSlots] : job.out + u0 * job.out_unit;
            const size_t bytes = (u1 - u0) * job.out_unit;
            if (bytes) DSPGPU_HIP(hipMemcpyAsync(dst, dout + u0 * job.out_unit, bytes, hipMemcpyDeviceToHost, p.d2h));
            DSPGPU_HIP(hipEventRecord(p.ev_d2h[kb % kSlots], p.d2h));
        }
        CopyTask tasks[2];
        int ntasks = 0;
        // stage C: chunk i-2 has arrived in its slot (or in the caller's page-locked buffer)
        if (i >= 2) {
            const size_t kc = i - 2;
            DSPGPU_HIP(hipEventSynchronize(p.ev_d2h[kc % kSlots]));
            const size_t u0 = unit_of(kc), u1 = unit_of(kc + 1);
            if (ring_out) tasks[ntasks++] = CopyTask{job.out + u0 * job.out_unit, p.ring_out[kc % kSlots], (u1 - u0) * job.out_unit};
        }
        // stage A: chunk i goes into its slot once the DMA that last read the slot (chunk i-3) is done
        const bool have_a = i < chunks;
        size_t a_off = 0, a_bytes = 0;
        if (have_a) {
            a_off = in_off(unit_of(i));
            a_bytes = in_off(unit_of(i + 1)) - a_off;
            if (ring_in) {
                if (i >= static_cast<size_t>(kSlots)) DSPGPU_HIP(hipEventSynchronize(p.ev_h2d[i % kSlots]));
                tasks[ntasks++] = CopyTask{p.ring_in[i % kSlots], job.in + a_off, a_bytes};
            }
        }
        pool.run(tasks, ntasks);  // copy-out of chunk i-2 and copy-in of chunk i share the threads
        if (have_a) {
            const char *src = ring_in ? p.ring_in[i % kSlots] : job.in + a_off;
            if (a_bytes) DSPGPU_HIP(hipMemcpyAsync(din + a_off, src, a_bytes, hipMemcpyHostToDevice, p.h2d));
            DSPGPU_HIP(hipEventRecord(p.ev_h2d[i % kSlots], p.h2d));
        }
    }
    if (aux_bytes) DSPGPU_HIP(hipMemcpyAsync(job.aux, daux, aux_bytes, hipMemcpyDeviceToHost, s));  // behind the last kernels
    DSPGPU_HIP(hipStreamSynchronize(s));  // the caller's stream is idle again when the call returns, as before
    return DSP_GPU_OK;
}

}  // namespace dspgpu
