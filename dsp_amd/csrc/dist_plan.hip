// dist_plan.hip -- plan handle of the one transform that spans GPUs (BASELINE config 5: a single 2^30-point fft0,
// reference fft.h:31-55, spread over G ranks).  Host code only: it sequences the local HIP passes of
// fft_kernels.hip / dist_kernels.hip around three all-to-all exchanges.
//
//   N = N1*N2, n = n1*N2 + n2, k = k1 + N1*k2, rank g holds x[g*N/G, (g+1)*N/G) and receives X[g*N/G, (g+1)*N/G):
//     phase 0  pack          [R][G][C] -> [G][R][C]           (R = N1/G rows, C = N2/G columns)
//     phase 1  FFT over n1   [N1][C], rows k1 leave grouped by destination rank
//     phase 2  unpack * W_N^{k1 n2} -> FFT over n2 -> pack    (two fused column passes; three plain ones otherwise)
//     phase 3  transpose     [G][R][C] -> [C][N1] = X[k1 + N1*k2], natural order
//   with an exchange (block g of every rank -> rank g) between consecutive phases.
//
// The exchange is the caller's (a callback: torch.distributed, MPI, peer copies ...) or RCCL's grouped
// ncclSend/ncclRecv over the xGMI mesh -- one block per peer, no ring.  RCCL is looked up at run time
// (dlopen), so the library itself depends on libamdhip64 only.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.h"

using namespace dspgpu;

struct dsp_gpu_fft_dist_plan {
    int log2n, rank, world;
    int log2n1, log2n2;
    size_t n1, n2, rows, cols, local;  // local = N/G points held by this rank
    dsp_gpu_exchange_fn exchange;
    void *user;
    void *comm;  // ncclComm_t when the built-in exchange is used
    int device;
    float2 *work;  // N/G points, allocated on first use on `device`
    // pipelined schedule (see run_pipelined): column / row chunks per phase, a second work buffer of four chunk slots,
    // the stream the exchanges are enqueued on and the events that order it against the caller's stream
    int chunks;
    float2 *slots;
    hipStream_t comm_stream;
    hipEvent_t *events;
    int nevents;
};

namespace {

struct Rccl {
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    bool ok = false;
};

const Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // the copy already in the process (the caller made its communicator with it) wins: same soname
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        r.group_start = reinterpret_cast<decltype(r.group_start)>(dlsym(h, "ncclGroupStart"));
        r.group_end = reinterpret_cast<decltype(r.group_end)>(dlsym(h, "ncclGroupEnd"));
        r.send = reinterpret_cast<decltype(r.send)>(dlsym(h, "ncclSend"));
        r.recv = reinterpret_cast<decltype(r.recv)>(dlsym(h, "ncclRecv"));
        r.error_string = reinterpret_cast<decltype(r.error_string)>(dlsym(h, "ncclGetErrorString"));
        r.ok = r.group_start && r.group_end && r.send && r.recv && r.error_string;
    });
    return r;
}

// Block g of `send` goes to rank g, block g of `recv` comes from rank g; every pair of GPUs trades one block.
int rccl_exchange(void *user, const void *send, void *recv, size_t block_bytes, int world, void *stream) {
    const Rccl &r = rccl();
    auto comm = static_cast<ncclComm_t>(user);
    auto s = static_cast<hipStream_t>(stream);
    const char *src = static_cast<const char *>(send);
    char *dst = static_cast<char *>(recv);
    ncclResult_t e = r.group_start();
    for (int g = 0; g < world && e == ncclSuccess; ++g) {
        e = r.send(src + static_cast<size_t>(g) * block_bytes, block_bytes, ncclUint8, g, comm, s);
        if (e == ncclSuccess) e = r.recv(dst + static_cast<size_t>(g) * block_bytes, block_bytes, ncclUint8, g, comm, s);
    }
    const ncclResult_t e2 = r.group_end();
    if (e == ncclSuccess) e = e2;
    if (e != ncclSuccess) return fail(DSP_GPU_EHIP, "dsp_gpu_fft_dist: RCCL exchange failed: %s", r.error_string(e));
    return DSP_GPU_OK;
}

int make_plan(dsp_gpu_fft_dist_plan **out, int log2n, int rank, int world, dsp_gpu_exchange_fn fn, void *user, void *comm) {
    if (!out) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_dist_plan_create: null plan pointer");
    *out = nullptr;
    if (world < 1 || (world & (world - 1)) || rank < 0 || rank >= world)
        return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_dist_plan_create: rank %d of %d (world must be a power of two)", rank, world);
    const int l1 = (log2n + 1) / 2, l2 = log2n - l1;
    if (log2n < 8 || l1 > 16 || l2 > 30)
        return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_dist_plan_create: log2n %d outside [8,32]", log2n);
    const size_t n1 = static_cast<size_t>(1) << l1, n2 = static_cast<size_t>(1) << l2;
    if (n1 % world || n2 % world || n2 / world < 2)
        return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_dist_plan_create: 2^%d points cannot be spread over %d ranks", log2n, world);
    auto *p = new (std::nothrow) dsp_gpu_fft_dist_plan{};
    if (!p) return fail(DSP_GPU_ENOMEM, "dsp_gpu_fft_dist_plan_create: out of host memory");
    p->log2n = log2n, p->rank = rank, p->world = world, p->log2n1 = l1, p->log2n2 = l2;
    p->n1 = n1, p->n2 = n2, p->rows = n1 / world, p->cols = n2 / world, p->local = p->rows * n2;
    p->exchange = fn, p->user = user, p->comm = comm, p->device = -1, p->work = nullptr;
    p->slots = nullptr, p->comm_stream = nullptr, p->events = nullptr, p->nevents = 0;
    // Chunks of the pipelined schedule: as many as the kernels' tiles allow, four at most (fewer, larger exchanges
    // keep the links busy; more only shorten the pipeline's fill and drain, which four already bring to a quarter).
    p->chunks = 1;
    if (fn) {
        for (int k = 4; k >= 2; k >>= 1) {
            const size_t cc = p->cols / k;
            if (p->cols % k || p->rows % k || cc % 2 || !fft_columns_tiled(l1, cc) || !fft_rows_exchange_covers(l2, world, k)) continue;
            p->chunks = k;
            break;
        }
    }
    *out = p;
    return DSP_GPU_OK;
}

int work_buffer(dsp_gpu_fft_dist_plan *p, float2 **w) {
    int dev = 0;
    DSPGPU_HIP(hipGetDevice(&dev));
    if (p->work && p->device != dev)
        return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_dist: plan was first used on device %d, current device is %d", p->device, dev);
    if (!p->work) {
        void *v = nullptr;
        if (hipMalloc(&v, p->local * sizeof(float2)) != hipSuccess) {
            (void)hipGetLastError();
            return fail(DSP_GPU_ENOMEM, "dsp_gpu_fft_dist: cannot allocate the %zu-byte work buffer", p->local * sizeof(float2));
        }
        p->work = static_cast<float2 *>(v), p->device = dev;
    }
    *w = p->work;
    return DSP_GPU_OK;
}

// second work buffer, comm stream and events of the pipelined schedule (after work_buffer: the device is checked there)
int pipeline_state(dsp_gpu_fft_dist_plan *p) {
    const int K = p->chunks;
    if (!p->slots) {
        void *v = nullptr;
        const size_t bytes = 4 * (p->local / K) * sizeof(float2);
        if (hipMalloc(&v, bytes) != hipSuccess) {
            (void)hipGetLastError();
            return fail(DSP_GPU_ENOMEM, "dsp_gpu_fft_dist: cannot allocate the %zu-byte chunk buffer", bytes);
        }
        p->slots = static_cast<float2 *>(v);
    }
    if (!p->comm_stream) DSPGPU_HIP(hipStreamCreateWithFlags(&p->comm_stream, hipStreamNonBlocking));
    const int n = 7 * K;
    if (p->events && p->nevents < n) {  // an earlier call failed part-way: start over rather than record null events
        for (int i = 0; i < p->nevents; ++i) (void)hipEventDestroy(p->events[i]);
        delete[] p->events;
        p->events = nullptr;
        p->nevents = 0;
    }
    if (!p->events) {
        p->events = new (std::nothrow) hipEvent_t[n]();
        if (!p->events) return fail(DSP_GPU_ENOMEM, "dsp_gpu_fft_dist: out of host memory");
        for (p->nevents = 0; p->nevents < n; ++p->nevents)
            DSPGPU_HIP(hipEventCreateWithFlags(&p->events[p->nevents], hipEventDisableTiming));
    }
    return DSP_GPU_OK;
}

int check_direction(const char *who, int direction) {
    if (direction != DSP_GPU_FORWARD && direction != DSP_GPU_INVERSE) return fail(DSP_GPU_EINVAL, "%s: direction %d", who, direction);
    return DSP_GPU_OK;
}

// One local phase, in != out.  `tmp` (N/G points, not `out`; may be `in` if that may be overwritten) is touched only
// by phase 2 when the fused passes do not cover the shape.
int run_phase(const dsp_gpu_fft_dist_plan *p, int phase, const float2 *in, float2 *out, float2 *tmp, int direction, hipStream_t s) {
    const size_t G = p->world, R = p->rows, C = p->cols;
    switch (phase) {
    case 0: return launch_swap_axes(in, out, R, G, C, 0, 0, direction, s);
    case 1: return launch_fft_columns(in, out, p->log2n1, C, direction, s);
    case 2: {
        const int rc = launch_fft_rows_exchange(in, out, p->log2n2, R, G, p->log2n, static_cast<unsigned long long>(p->rank) * R,
                                                direction, s);
        if (rc != DSP_GPU_EUNSUPPORTED) return rc;
        if (int e = launch_swap_axes(in, out, G, R, C, p->log2n, static_cast<unsigned long long>(p->rank) * R, direction, s)) return e;
        if (int e = launch_fft_batch(out, tmp, p->log2n2, R, direction, s)) return e;
        return launch_swap_axes(tmp, out, R, G, C, 0, 0, direction, s);
    }
    case 3: return launch_transpose_blocks(in, out, G, R, C, s);
    }
    return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_dist_phase: phase %d outside [0,3]", phase);
}

// The transform with its exchanges overlapped ("pipelining column-FFT chunks with sends").  Local passes run
// on the caller's stream `s`, exchanges on the plan's own stream `cs`, events between them; everything the call enqueues
// is complete when `s` has drained.  K = plan->chunks.
//   segment 1, per column chunk i (C/K of my columns):   pack(i) -> exchange 1(i) -> FFT over n1 (i) -> exchange 2(i)
//       P = `out` (a free buffer until the very end), Q = work:  pack: in -> P_i, ex 1: P_i -> Q_i, FFT: Q_i -> P_i, ex 2: P_i -> Q_i
//       issue order on s: pack(0) pack(1) fft(0) pack(2) fft(1) ...;  on cs: ex1(0) ex1(1) ex2(0) ex1(2) ex2(1) ...
//   join: the transform over n2 needs every column of a row, i.e. all of Q = [K][G][R][C/K]
//   segment 2, per row chunk j (R/K of my rows):          FFT over n2 (j) -> exchange 3(j) -> transpose(j) into `out`
//       FFT: Q -> A[j%2], ex 3: A[j%2] -> B[j%2], transpose: B[j%2] -> out   (A, B: the four chunk slots)
//   TRANSPOSED_OUT: segment 2 has no exchange: unpack * twiddle Q -> out, row transforms in place.
// An exchange must leave `send` reusable and `recv` complete once the work it enqueued on its stream has run (what
// grouped ncclSend / ncclRecv do).
int run_pipelined(dsp_gpu_fft_dist_plan *plan, const float2 *in, float2 *out, float2 *w, int direction, bool transposed, hipStream_t s) {
    if (int rc = pipeline_state(plan)) return rc;
    const size_t G = plan->world, R = plan->rows, C = plan->cols, K = plan->chunks;
    const size_t Cc = C / K, Rc = R / K, chunk = plan->local / K;
    const size_t block_bytes = chunk / G * sizeof(float2);
    const unsigned long long row0 = static_cast<unsigned long long>(plan->rank) * R;
    hipStream_t cs = plan->comm_stream;
    hipEvent_t *e_pack = plan->events, *e_ex1 = e_pack + K, *e_fft1 = e_ex1 + K, *e_ex2 = e_fft1 + K, *e_fft2 = e_ex2 + K,
               *e_ex3 = e_fft2 + K, *e_tr = e_ex3 + K;
    auto exchange = [&](const float2 *send, float2 *recv) -> int {
        int rc = plan->exchange(plan->user, send, recv, block_bytes, plan->world, cs);
        if (rc > 0) rc = fail(DSP_GPU_EHIP, "dsp_gpu_fft_dist: the exchange callback returned %d", rc);
        return rc;
    };
    for (size_t t = 0; t <= K; ++t) {
        if (t < K) {  // pack: P_t[g][r][cc] = in[r][g*C + t*Cc + cc]
            if (int rc = launch_swap_axes_window(in + t * Cc, out + t * chunk, R, G, Cc, C, Cc, 0, 0, 0, 0, direction, s)) return rc;
            DSPGPU_HIP(hipEventRecord(e_pack[t], s));
            DSPGPU_HIP(hipStreamWaitEvent(cs, e_pack[t], 0));
            if (int rc = exchange(out + t * chunk, w + t * chunk)) return rc;
            DSPGPU_HIP(hipEventRecord(e_ex1[t], cs));
        }
        if (t >= 1) {  // Q_i = [N1][Cc], the rows from every sender stacked: transform over n1, rows leave grouped by destination
            const size_t i = t - 1;
            DSPGPU_HIP(hipStreamWaitEvent(s, e_ex1[i], 0));
            if (int rc = launch_fft_columns(w + i * chunk, out + i * chunk, plan->log2n1, Cc, direction, s)) return rc;
            DSPGPU_HIP(hipEventRecord(e_fft1[i], s));
            DSPGPU_HIP(hipStreamWaitEvent(cs, e_fft1[i], 0));
            if (int rc = exchange(out + i * chunk, w + i * chunk)) return rc;
            DSPGPU_HIP(hipEventRecord(e_ex2[i], cs));
        }
    }
    DSPGPU_HIP(hipStreamWaitEvent(s, e_ex2[K - 1], 0));  // cs runs in order: every exchange before it is done as well
    if (transposed) {
        for (size_t i = 0; i < K; ++i)  // Q_i[g][r][cc] -> out[r][g*C + i*Cc + cc], times W_N^{k1 n2}
            if (int rc = launch_swap_axes_window(w + i * chunk, out + i * Cc, G, R, Cc, Cc, C, plan->log2n, row0, C, i * Cc, direction, s))
                return rc;
        return launch_fft_batch(out, out, plan->log2n2, R, direction, s);
    }
    float2 *A[2] = {plan->slots, plan->slots + chunk}, *B[2] = {plan->slots + 2 * chunk, plan->slots + 3 * chunk};
    for (size_t t = 0; t <= K; ++t) {
        if (t < K) {
            // A[t%2] was last sent by exchange 3 of chunk t-2, whose transpose (issued on s a tick ago) waited for it
            if (int rc = launch_fft_rows_exchange_chunk(w, A[t & 1], plan->log2n2, Rc, G, plan->log2n, row0, direction, s, K, R, t * Rc)) return rc;
            DSPGPU_HIP(hipEventRecord(e_fft2[t], s));
            DSPGPU_HIP(hipStreamWaitEvent(cs, e_fft2[t], 0));
            if (t >= 2) DSPGPU_HIP(hipStreamWaitEvent(cs, e_tr[t - 2], 0));  // B[t%2] has been read
            if (int rc = exchange(A[t & 1], B[t & 1])) return rc;
            DSPGPU_HIP(hipEventRecord(e_ex3[t], cs));
        }
        if (t >= 1) {  // B[g][rr][c] = X[(g*R + j*Rc + rr) + N1*(rank*C + c)] -> out[c][g*R + j*Rc + rr]
            const size_t j = t - 1;
            DSPGPU_HIP(hipStreamWaitEvent(s, e_ex3[j], 0));
            if (int rc = launch_transpose_blocks_into(B[j & 1], out, G, Rc, C, R, j * Rc, s)) return rc;
            DSPGPU_HIP(hipEventRecord(e_tr[j], s));
        }
    }
    return DSP_GPU_OK;
}

}  // namespace

extern "C" {

int dsp_gpu_fft_dist_plan_create(dsp_gpu_fft_dist_plan **plan, int log2n, int rank, int world, dsp_gpu_exchange_fn exchange,
                                 void *user) {
    return make_plan(plan, log2n, rank, world, exchange, user, nullptr);
}

int dsp_gpu_fft_dist_plan_create_rccl(dsp_gpu_fft_dist_plan **plan, int log2n, int rank, int world, void *nccl_comm) {
    if (!nccl_comm) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_dist_plan_create_rccl: null communicator");
    if (!rccl().ok) return fail(DSP_GPU_EUNSUPPORTED, "dsp_gpu_fft_dist_plan_create_rccl: librccl.so not found (%s)", dlerror());
    return make_plan(plan, log2n, rank, world, rccl_exchange, nccl_comm, nccl_comm);
}

size_t dsp_gpu_fft_dist_local_points(const dsp_gpu_fft_dist_plan *plan) { return plan ? plan->local : 0; }

int dsp_gpu_fft_dist_phase(dsp_gpu_fft_dist_plan *plan, int phase, const void *in, void *out, int direction, void *stream) {
    if (!plan || !in || !out || in == out) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_dist_phase: null plan or null / aliased buffer");
    if (int rc = check_direction("dsp_gpu_fft_dist_phase", direction)) return rc;
    if (phase < 0 || phase > 3) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_dist_phase: phase %d outside [0,3]", phase);
    float2 *tmp = nullptr;
    if (phase == 2)
        if (int rc = work_buffer(plan, &tmp)) return rc;
    return run_phase(plan, phase, static_cast<const float2 *>(in), static_cast<float2 *>(out), tmp, direction,
                     static_cast<hipStream_t>(stream));
}

int dsp_gpu_fft_dist_ex(dsp_gpu_fft_dist_plan *plan, const void *in, void *out, int direction, int flags, void *stream) {
    if (!plan || !in || !out || in == out) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_dist: null plan or null / aliased buffer");
    if (int rc = check_direction("dsp_gpu_fft_dist", direction)) return rc;
    if (flags & ~(DSP_GPU_DIST_TRANSPOSED_OUT | DSP_GPU_DIST_SERIAL)) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_dist: flags %d", flags);
    if (plan->world > 1 && !plan->exchange) return fail(DSP_GPU_EINVAL, "dsp_gpu_fft_dist: the plan has no exchange (use the phase calls)");
    float2 *w = nullptr;
    if (int rc = work_buffer(plan, &w)) return rc;
    auto s = static_cast<hipStream_t>(stream);
    float2 *o = static_cast<float2 *>(out);
    const size_t block_bytes = plan->local / plan->world * sizeof(float2);
    const bool exchanging = plan->exchange != nullptr;  // at world 1 a given exchange still runs (a self copy)
    const bool transposed = (flags & DSP_GPU_DIST_TRANSPOSED_OUT) != 0;
    if (exchanging && plan->chunks > 1 && !(flags & DSP_GPU_DIST_SERIAL))
        return run_pipelined(plan, static_cast<const float2 *>(in), o, w, direction, transposed, s);
    // The serial schedule.  The steps: local phases with an exchange behind each but the last.  Without an exchange (world 1) the pack of
    // phase 0 is a plain copy and is left out.  TRANSPOSED_OUT ends after the transform over n2: no third exchange, no
    // final transpose -- the rank keeps X[k1 + N1 k2] for ITS rows k1 = rank * N1/G + r as out[r][k2].
    enum { PACK, FFT_N1, FFT_N2, NATURAL, ROWS_UNPACK, ROWS_FFT, EXCHANGE };
    int steps[8], nsteps = 0;
    if (exchanging) steps[nsteps++] = PACK, steps[nsteps++] = EXCHANGE;
    steps[nsteps++] = FFT_N1;
    if (exchanging) steps[nsteps++] = EXCHANGE;
    if (transposed) {
        steps[nsteps++] = ROWS_UNPACK;
        steps[nsteps++] = ROWS_FFT;  // in place when its input is `out`, else a hop
    } else {
        steps[nsteps++] = FFT_N2;
        if (exchanging) steps[nsteps++] = EXCHANGE;
        steps[nsteps++] = NATURAL;
    }
    // hops alternate between `out` and the work buffer; the first destination is chosen so that the last one is `out`
    // (`in` is never written).  ROWS_FFT hops only if it does not already sit in `out`.
    int hops = 0;
    for (int i = 0; i < nsteps; ++i) hops += steps[i] != ROWS_FFT;
    // after the hops before ROWS_FFT the data is in `out` iff their number has the parity that starts in `out`
    const bool start_in_out = transposed ? true : (hops % 2 == 1);  // transposed: ROWS_UNPACK must land in w or o, see below
    const float2 *cur = static_cast<const float2 *>(in);
    float2 *next = nullptr;
    if (transposed) {
        // hops before ROWS_FFT (incl. ROWS_UNPACK): if even the unpack lands in `w` and ROWS_FFT hops w -> out; if odd it
        // lands in `out` and ROWS_FFT runs in place.  Either first destination works; start in `out`.
        next = o;
    } else {
        next = start_in_out ? o : w;
    }
    const size_t G = plan->world, R = plan->rows, C = plan->cols;
    for (int i = 0; i < nsteps; ++i) {
        int rc = DSP_GPU_OK;
        switch (steps[i]) {
        case PACK: rc = run_phase(plan, 0, cur, next, nullptr, direction, s); break;
        case FFT_N1: rc = run_phase(plan, 1, cur, next, nullptr, direction, s); break;
        case FFT_N2:
            // phase 2's three-pass fallback needs a spare buffer: its input, which its first hop has consumed by then
            // (never the caller's `in`: a phase-1 result precedes it)
            rc = run_phase(plan, 2, cur, next, const_cast<float2 *>(cur), direction, s);
            break;
        case NATURAL: rc = run_phase(plan, 3, cur, next, nullptr, direction, s); break;
        case ROWS_UNPACK:  // [G][R][C] -> [R][N2], times W_N^{k1 n2}
            rc = launch_swap_axes(cur, next, G, R, C, plan->log2n, static_cast<unsigned long long>(plan->rank) * R, direction, s);
            break;
        case ROWS_FFT:
            rc = launch_fft_batch(cur, o, plan->log2n2, R, direction, s);  // cur == o: in place
            next = o;
            break;
        case EXCHANGE:
            rc = plan->exchange(plan->user, cur, next, block_bytes, plan->world, stream);
            if (rc > 0) rc = fail(DSP_GPU_EHIP, "dsp_gpu_fft_dist: the exchange callback returned %d", rc);
            break;
        }
        if (rc) return rc;
        cur = next;
        next = cur == o ? w : o;
    }
    return DSP_GPU_OK;
}

int dsp_gpu_fft_dist_exchanges(const dsp_gpu_fft_dist_plan *plan, int flags) {
    if (!plan || !plan->exchange) return 0;
    const int per = (flags & DSP_GPU_DIST_TRANSPOSED_OUT) ? 2 : 3;
    return per * ((flags & DSP_GPU_DIST_SERIAL) ? 1 : plan->chunks);
}

int dsp_gpu_fft_dist(dsp_gpu_fft_dist_plan *plan, const void *in, void *out, int direction, void *stream) {
    return dsp_gpu_fft_dist_ex(plan, in, out, direction, 0, stream);
}

int dsp_gpu_fft_dist_plan_destroy(dsp_gpu_fft_dist_plan *plan) {
    if (!plan) return DSP_GPU_OK;
    if (plan->work) {
        int dev = 0;
        const bool switched = hipGetDevice(&dev) == hipSuccess && dev != plan->device && hipSetDevice(plan->device) == hipSuccess;
        if (plan->comm_stream) (void)hipStreamSynchronize(plan->comm_stream);
        (void)hipFree(plan->work);
        if (plan->slots) (void)hipFree(plan->slots);
        for (int i = 0; i < plan->nevents; ++i) (void)hipEventDestroy(plan->events[i]);
        if (plan->comm_stream) (void)hipStreamDestroy(plan->comm_stream);
        if (switched) (void)hipSetDevice(dev);
    }
    delete[] plan->events;
    delete plan;
    return DSP_GPU_OK;
}

}  // extern "C"
