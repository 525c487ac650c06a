// copy_patterns.hip -- the copy microbenchmarks of rounds 1-2 in one file (dev tool, not product).
// What read+write streaming rate does this MI355X give to which access pattern?  Each section is one generation of the
// experiment, kept as it was run (its findings are quoted in DESIGN.md 4 and 8):
//   fft_tile      the access patterns the FFT row kernel could use
//   streaming     best read+write streaming pattern (non-temporal, XCD map)
//   row_segments  the IIR-batch pattern: cost of short row segments
//   in_flight     bytes in flight per CU against the streaming rate (profiles/r02_copy_ceiling.txt)
//   width         access width of loads / stores at equal bytes per thread
//   hipcc -O3 --offload-arch=gfx950 -o scripts/copy_patterns scripts/copy_patterns.hip && scripts/copy_patterns <section>|all
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

// ===== section `fft_tile` (was scripts/copy_patterns.hip) ========================================
namespace fft_tile {
// Microbenchmark (dev tool, not product): HBM ceilings of the access patterns the FFT kernel could use.

// pattern 1: as fft_lds_kernel: tile of 4096 float2 per WG, thread t moves elements t + 256 m (8 B each)
__global__ __launch_bounds__(256) void copy_f2(const float2* in, float2* out) {
    const float2* s = in + (size_t)blockIdx.x * 4096; float2* d = out + (size_t)blockIdx.x * 4096;
    float2 r[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) r[m] = s[threadIdx.x + 256 * m];
#pragma unroll
    for (int m = 0; m < 16; ++m) d[threadIdx.x + 256 * m] = r[m];
}
// pattern 2: 16-byte accesses, lane pairs: even lane rows m<8, odd lane rows m>=8 (two 512 B runs per wave instr)
__global__ __launch_bounds__(256) void copy_f4_pair(const float4* in, float4* out) {
    const float4* s = in + (size_t)blockIdx.x * 2048; float4* d = out + (size_t)blockIdx.x * 2048;
    const int t = threadIdx.x, p = t >> 1, odd = t & 1;
    float4 r[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) r[m] = s[p + 128 * (m + 8 * odd)];
#pragma unroll
    for (int m = 0; m < 8; ++m) d[p + 128 * (m + 8 * odd)] = r[m];
}
// pattern 3: 16-byte accesses, thread t moves float4 t + 256 m (1 KiB contiguous per wave instr)
__global__ __launch_bounds__(256) void copy_f4(const float4* in, float4* out) {
    const float4* s = in + (size_t)blockIdx.x * 2048; float4* d = out + (size_t)blockIdx.x * 2048;
    float4 r[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) r[m] = s[threadIdx.x + 256 * m];
#pragma unroll
    for (int m = 0; m < 8; ++m) d[threadIdx.x + 256 * m] = r[m];
}
// pattern 4: grid-stride float4 copy, 2048 blocks
__global__ __launch_bounds__(256) void copy_gs(const float4* in, float4* out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
// pattern 5: nontemporal float4
__global__ __launch_bounds__(256) void copy_f4_nt(const float4* in, float4* out) {
    const float4* s = in + (size_t)blockIdx.x * 2048; float4* d = out + (size_t)blockIdx.x * 2048;
    float4 r[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) { const float* q = (const float*)(s + threadIdx.x + 256 * m);
        r[m] = make_float4(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q+1), __builtin_nontemporal_load(q+2), __builtin_nontemporal_load(q+3)); }
#pragma unroll
    for (int m = 0; m < 8; ++m) { float* q = (float*)(d + threadIdx.x + 256 * m);
        __builtin_nontemporal_store(r[m].x, q); __builtin_nontemporal_store(r[m].y, q+1); __builtin_nontemporal_store(r[m].z, q+2); __builtin_nontemporal_store(r[m].w, q+3); }
}
int run() {
    const size_t tiles = 65536, bytes = tiles * 4096 * 8;
    void *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-14s %.4f ms  %.3f TB/s\n", name, ms, 2.0 * bytes / ms / 1e9);
    };
    for (int rep = 0; rep < 2; ++rep) {
    run("f2 strided", [&]{ hipLaunchKernelGGL(copy_f2, dim3(tiles), dim3(256), 0, 0, (const float2*)a, (float2*)b); });
    run("f4 pair", [&]{ hipLaunchKernelGGL(copy_f4_pair, dim3(tiles), dim3(256), 0, 0, (const float4*)a, (float4*)b); });
    run("f4 tile", [&]{ hipLaunchKernelGGL(copy_f4, dim3(tiles), dim3(256), 0, 0, (const float4*)a, (float4*)b); });
    run("f4 gridstride", [&]{ hipLaunchKernelGGL(copy_gs, dim3(2048), dim3(256), 0, 0, (const float4*)a, (float4*)b, bytes / 16); });
    run("f4 nt", [&]{ hipLaunchKernelGGL(copy_f4_nt, dim3(tiles), dim3(256), 0, 0, (const float4*)a, (float4*)b); });
    }
    return 0;
}
#undef RUN
#undef RUNT
}  // namespace fft_tile

// ===== section `streaming` (was scripts/copy_patterns2.hip) ========================================
namespace streaming {
// Microbenchmark round 2 (dev tool): looking for the best read+write streaming pattern on MI355X.
template <typename T> __device__ inline T ntload(const T* p);
template <> __device__ inline float2 ntload(const float2* p) { const float* q=(const float*)p; return make_float2(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q+1)); }
__device__ inline void ntstore(float2 v, float2* p) { float* q=(float*)p; __builtin_nontemporal_store(v.x,q); __builtin_nontemporal_store(v.y,q+1); }

template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_f2(const float2* in, float2* out) {
    const float2* s = in + (size_t)blockIdx.x * 4096; float2* d = out + (size_t)blockIdx.x * 4096;
    float2 r[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) r[m] = NTL ? ntload(s + threadIdx.x + 256 * m) : s[threadIdx.x + 256 * m];
#pragma unroll
    for (int m = 0; m < 16; ++m) { if (NTS) ntstore(r[m], d + threadIdx.x + 256 * m); else d[threadIdx.x + 256 * m] = r[m]; }
}
// persistent: grid = CUs*k, each WG loops over tiles
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_f2_persist(const float2* in, float2* out, unsigned tiles) {
    for (unsigned t = blockIdx.x; t < tiles; t += gridDim.x) {
        const float2* s = in + (size_t)t * 4096; float2* d = out + (size_t)t * 4096;
        float2 r[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) r[m] = NTL ? ntload(s + threadIdx.x + 256 * m) : s[threadIdx.x + 256 * m];
#pragma unroll
        for (int m = 0; m < 16; ++m) { if (NTS) ntstore(r[m], d + threadIdx.x + 256 * m); else d[threadIdx.x + 256 * m] = r[m]; }
    }
}
// XCD-contiguous mapping: block b -> tile (b % 8) * (tiles/8) + b / 8, so each XCD streams one contiguous eighth
__global__ __launch_bounds__(256) void copy_f2_xcd(const float2* in, float2* out, unsigned tiles) {
    const unsigned t = (blockIdx.x & 7) * (tiles >> 3) + (blockIdx.x >> 3);
    const float2* s = in + (size_t)t * 4096; float2* d = out + (size_t)t * 4096;
    float2 r[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) r[m] = s[threadIdx.x + 256 * m];
#pragma unroll
    for (int m = 0; m < 16; ++m) d[threadIdx.x + 256 * m] = r[m];
}
__global__ __launch_bounds__(256) void read_f2(const float2* in, float2* out) {
    const float2* s = in + (size_t)blockIdx.x * 4096;
    float2 acc = make_float2(0,0);
#pragma unroll
    for (int m = 0; m < 16; ++m) { float2 v = s[threadIdx.x + 256 * m]; acc.x += v.x; acc.y += v.y; }
    if (acc.x == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void write_f2(float2* out) {
    float2* d = out + (size_t)blockIdx.x * 4096;
#pragma unroll
    for (int m = 0; m < 16; ++m) d[threadIdx.x + 256 * m] = make_float2(1.f, 2.f);
}
int run() {
    const size_t tiles = 65536, bytes = tiles * 4096 * 8;
    void *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, double nbytes, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        (void)hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-22s %.4f ms  %.3f TB/s\n", name, ms, nbytes / ms / 1e9);
    };
    const float2* A = (const float2*)a; float2* B = (float2*)b;
    run("f2", 2.0*bytes, [&]{ hipLaunchKernelGGL((copy_f2<false,false>), dim3(tiles), dim3(256), 0, 0, A, B); });
    run("f2 ntl", 2.0*bytes, [&]{ hipLaunchKernelGGL((copy_f2<true,false>), dim3(tiles), dim3(256), 0, 0, A, B); });
    run("f2 nts", 2.0*bytes, [&]{ hipLaunchKernelGGL((copy_f2<false,true>), dim3(tiles), dim3(256), 0, 0, A, B); });
    run("f2 ntl+nts", 2.0*bytes, [&]{ hipLaunchKernelGGL((copy_f2<true,true>), dim3(tiles), dim3(256), 0, 0, A, B); });
    for (int k : {2, 4, 8}) { char nm[64]; snprintf(nm, 64, "f2 persist x%d", k);
        run(nm, 2.0*bytes, [&]{ hipLaunchKernelGGL((copy_f2_persist<false,false>), dim3(256*k), dim3(256), 0, 0, A, B, (unsigned)tiles); }); }
    run("f2 persist x4 nt", 2.0*bytes, [&]{ hipLaunchKernelGGL((copy_f2_persist<true,true>), dim3(1024), dim3(256), 0, 0, A, B, (unsigned)tiles); });
    run("f2 xcd-contig", 2.0*bytes, [&]{ hipLaunchKernelGGL(copy_f2_xcd, dim3(tiles), dim3(256), 0, 0, A, B, (unsigned)tiles); });
    run("read only", 1.0*bytes, [&]{ hipLaunchKernelGGL(read_f2, dim3(tiles), dim3(256), 0, 0, A, B); });
    run("write only", 1.0*bytes, [&]{ hipLaunchKernelGGL(write_f2, dim3(tiles), dim3(256), 0, 0, B); });
    run("in-place f2", 2.0*bytes, [&]{ hipLaunchKernelGGL((copy_f2<false,false>), dim3(tiles), dim3(256), 0, 0, (const float2*)b, B); });
    return 0;
}
#undef RUN
#undef RUNT
}  // namespace streaming

// ===== section `row_segments` (was scripts/copy_patterns3.hip) ========================================
namespace row_segments {
// Microbenchmark round 3 (dev tool): the IIR-batch access pattern.  A [channels][n] fp32 matrix, one wave per 64
// channels walking time in tiles of SEG samples per row: how much HBM bandwidth does the row-segment length cost?
__device__ inline float4 ld(const float* p) { typedef float v4 __attribute__((ext_vector_type(4))); v4 v = __builtin_nontemporal_load((const v4*)p); return make_float4(v.x,v.y,v.z,v.w); }
__device__ inline void st(float* p, float4 a) { typedef float v4 __attribute__((ext_vector_type(4))); v4 v = {a.x,a.y,a.z,a.w}; __builtin_nontemporal_store(v, (v4*)p); }

template <int SEG>
__global__ __launch_bounds__(64) void copy_rows(const float* __restrict__ x, float* __restrict__ y, size_t n, size_t chunk) {
    extern __shared__ float dummy[];
    const int lane = threadIdx.x;
    const size_t group = (size_t)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    constexpr int LPR = SEG / 4 < 64 ? SEG / 4 : 64;   // lanes per row per instruction
    constexpr int RPI = 64 / LPR;                       // rows per instruction
    constexpr int IPR = SEG / 4 / LPR;                  // instructions per row (SEG > 256)
    constexpr int ITERS = 64 / RPI * IPR;               // instructions per tile
    const size_t ch0 = group * 64;
    if (lane == 0 && n == 1) dummy[0] = 0;
    for (size_t t0 = 0; t0 < n; t0 += SEG) {
        for (int base = 0; base < ITERS; base += 16) {
            float4 r[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int it = base + k;
                const size_t row = ch0 + (it / IPR) * RPI + lane / LPR;
                const size_t col = t0 + (size_t)(it % IPR) * 256 + (lane % LPR) * 4;
                r[k] = ld(x + row * n + col);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int it = base + k;
                const size_t row = ch0 + (it / IPR) * RPI + lane / LPR;
                const size_t col = t0 + (size_t)(it % IPR) * 256 + (lane % LPR) * 4;
                st(y + row * n + col, r[k]);
            }
        }
    }
}
int run() {
    const size_t channels = 262144, n = 4096, bytes = channels * n * 4;
    void *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, int lds, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        (void)hipEventRecord(e0); for (int i = 0; i < 10; ++i) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-12s lds %6d  %.4f ms  %.3f TB/s\n", name, lds, ms, 2.0 * bytes / ms / 1e9);
    };
    const float* A = (const float*)a; float* B = (float*)b;
    const size_t groups = channels / 64, chunk = groups / 8;
    for (int lds : {0, 17000, 34000}) {
        run("seg 64", lds, [&]{ hipLaunchKernelGGL((copy_rows<64>), dim3(groups), dim3(64), lds, 0, A, B, n, chunk); });
        run("seg 128", lds, [&]{ hipLaunchKernelGGL((copy_rows<128>), dim3(groups), dim3(64), lds, 0, A, B, n, chunk); });
        run("seg 256", lds, [&]{ hipLaunchKernelGGL((copy_rows<256>), dim3(groups), dim3(64), lds, 0, A, B, n, chunk); });
        run("seg 512", lds, [&]{ hipLaunchKernelGGL((copy_rows<512>), dim3(groups), dim3(64), lds, 0, A, B, n, chunk); });
        run("seg 1024", lds, [&]{ hipLaunchKernelGGL((copy_rows<1024>), dim3(groups), dim3(64), lds, 0, A, B, n, chunk); });
    }
    return 0;
}
#undef RUN
#undef RUNT
}  // namespace row_segments

// ===== section `in_flight` (was scripts/copy_patterns4.hip) ========================================
namespace in_flight {
// Microbenchmark (dev tool): is there a read+write streaming pattern on this MI355X that beats the 5.45-5.66 TB/s the
// kernels' skeletons reach?  MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy.
//   hipcc -O3 --offload-arch=gfx950 -o scripts/copy_patterns4 scripts/copy_patterns4.hip
typedef float f4 __attribute__((ext_vector_type(4)));

// V float4 per thread, all loads first; tile = 256 * V float4, XCD-contiguous tile map, NT on/off
template <int V, bool NT, bool XCD>
__global__ __launch_bounds__(256) void copy4(const f4 *in, f4 *out, unsigned tiles) {
    const unsigned t = XCD ? (blockIdx.x & 7) * (tiles >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const f4 *s = in + (size_t)t * 256 * V + threadIdx.x;
    f4 *d = out + (size_t)t * 256 * V + threadIdx.x;
    f4 r[V];
#pragma unroll
    for (int m = 0; m < V; ++m) r[m] = NT ? __builtin_nontemporal_load(s + 256 * m) : s[256 * m];
#pragma unroll
    for (int m = 0; m < V; ++m) { if (NT) __builtin_nontemporal_store(r[m], d + 256 * m); else d[256 * m] = r[m]; }
}
// persistent grid-stride, W workgroups per CU
template <int V, bool NT>
__global__ __launch_bounds__(256) void copy4_persist(const f4 *in, f4 *out, unsigned tiles) {
    const unsigned per_xcd = gridDim.x >> 3, chunk = tiles >> 3;
    for (unsigned i = blockIdx.x >> 3; i < chunk; i += per_xcd) {
        const unsigned t = (blockIdx.x & 7) * chunk + i;
        const f4 *s = in + (size_t)t * 256 * V + threadIdx.x;
        f4 *d = out + (size_t)t * 256 * V + threadIdx.x;
        f4 r[V];
#pragma unroll
        for (int m = 0; m < V; ++m) r[m] = NT ? __builtin_nontemporal_load(s + 256 * m) : s[256 * m];
#pragma unroll
        for (int m = 0; m < V; ++m) { if (NT) __builtin_nontemporal_store(r[m], d + 256 * m); else d[256 * m] = r[m]; }
    }
}
// each thread owns a contiguous 16 * V bytes?  no: each WAVE owns a contiguous 1 KB * V run (lane-contiguous float4 per row)
template <int V, bool NT>
__global__ __launch_bounds__(256) void copy4_waverun(const f4 *in, f4 *out, unsigned tiles) {
    const unsigned t = (blockIdx.x & 7) * (tiles >> 3) + (blockIdx.x >> 3);
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const f4 *s = in + (size_t)t * 256 * V + wave * 64 * V + lane;
    f4 *d = out + (size_t)t * 256 * V + wave * 64 * V + lane;
    f4 r[V];
#pragma unroll
    for (int m = 0; m < V; ++m) r[m] = NT ? __builtin_nontemporal_load(s + 64 * m) : s[64 * m];
#pragma unroll
    for (int m = 0; m < V; ++m) { if (NT) __builtin_nontemporal_store(r[m], d + 64 * m); else d[64 * m] = r[m]; }
}

// THREADS per workgroup as a parameter: separates "bytes per thread" from "bytes per workgroup"
template <int THREADS, int V>
__global__ __launch_bounds__(THREADS) void copy4_t(const f4 *in, f4 *out, unsigned tiles) {
    const unsigned t = (blockIdx.x & 7) * (tiles >> 3) + (blockIdx.x >> 3);
    const f4 *s = in + (size_t)t * THREADS * V + threadIdx.x;
    f4 *d = out + (size_t)t * THREADS * V + threadIdx.x;
    f4 r[V];
#pragma unroll
    for (int m = 0; m < V; ++m) r[m] = __builtin_nontemporal_load(s + THREADS * m);
#pragma unroll
    for (int m = 0; m < V; ++m) __builtin_nontemporal_store(r[m], d + THREADS * m);
}
// loads through LDS with a barrier before the stores (what every transform kernel does)
template <int V>
__global__ __launch_bounds__(256) void copy4_lds(const f4 *in, f4 *out, unsigned tiles) {
    __shared__ f4 img[256 * V];
    const unsigned t = (blockIdx.x & 7) * (tiles >> 3) + (blockIdx.x >> 3);
    const f4 *s = in + (size_t)t * 256 * V + threadIdx.x;
    f4 *d = out + (size_t)t * 256 * V + threadIdx.x;
    f4 r[V];
#pragma unroll
    for (int m = 0; m < V; ++m) r[m] = __builtin_nontemporal_load(s + 256 * m);
#pragma unroll
    for (int m = 0; m < V; ++m) img[threadIdx.x + 256 * m] = r[m];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < V; ++m) r[m] = img[(threadIdx.x + 256 * m) ^ 1];
#pragma unroll
    for (int m = 0; m < V; ++m) __builtin_nontemporal_store(r[m], d + 256 * m);
}

// 8 float4 per thread, loaded in batches of B with a wait after each batch (fewer bytes in flight per wave), then stored
template <int B>
__global__ __launch_bounds__(256) void copy4_batched(const f4 *in, f4 *out, unsigned tiles) {
    constexpr int V = 8;
    const unsigned t = (blockIdx.x & 7) * (tiles >> 3) + (blockIdx.x >> 3);
    const f4 *s = in + (size_t)t * 256 * V + threadIdx.x;
    f4 *d = out + (size_t)t * 256 * V + threadIdx.x;
    f4 r[V];
#pragma unroll
    for (int m0 = 0; m0 < V; m0 += B) {
#pragma unroll
        for (int m = m0; m < m0 + B; ++m) r[m] = __builtin_nontemporal_load(s + 256 * m);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int m = 0; m < V; ++m) __builtin_nontemporal_store(r[m], d + 256 * m);
}

int run() {
    const size_t bytes = (size_t)2 << 30;  // 2 GiB in, 2 GiB out
    f4 *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t n4 = bytes / 16;
    auto time = [&](const char *name, auto launch) -> int {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
        printf("%-40s %.4f ms  %.3f TB/s (read + write)\n", name, ms, 2.0 * bytes / ms / 1e9);
        return 0;
    };
#define RUN(name, K, T) time(name, [&] { hipLaunchKernelGGL(K, dim3(T), dim3(256), 0, 0, in, out, (unsigned)(T)); })
    time("warm-up", [&] { hipLaunchKernelGGL((copy4<4, true, true>), dim3(n4 / 1024), dim3(256), 0, 0, in, out, (unsigned)(n4 / 1024)); });
    RUN("f4 x1  nt xcd", (copy4<1, true, true>), n4 / 256);
    RUN("f4 x2  nt xcd", (copy4<2, true, true>), n4 / 512);
    RUN("f4 x4  nt xcd", (copy4<4, true, true>), n4 / 1024);
    RUN("f4 x8  nt xcd", (copy4<8, true, true>), n4 / 2048);
    RUN("f4 x16 nt xcd", (copy4<16, true, true>), n4 / 4096);
    RUN("f4 x4  plain xcd", (copy4<4, false, true>), n4 / 1024);
    RUN("f4 x4  nt linear", (copy4<4, true, false>), n4 / 1024);
    RUN("f4 x4  plain linear", (copy4<4, false, false>), n4 / 1024);
    RUN("f4 x8  nt waverun", (copy4_waverun<8, true>), n4 / 2048);
    RUN("f4 x16 nt waverun", (copy4_waverun<16, true>), n4 / 4096);
#define RUNT(name, TH, V) time(name, [&] { hipLaunchKernelGGL((copy4_t<TH, V>), dim3(n4 / (TH * V)), dim3(TH), 0, 0, in, out, (unsigned)(n4 / (TH * V))); })
    RUNT("64 thr x 8   (8 KB/wg, 128 B/thr)", 64, 8);
    RUNT("64 thr x 1   (1 KB/wg, 16 B/thr)", 64, 1);
    RUNT("128 thr x 2  (4 KB/wg, 32 B/thr)", 128, 2);
    RUNT("512 thr x 1  (8 KB/wg, 16 B/thr)", 512, 1);
    RUNT("1024 thr x 1 (16 KB/wg, 16 B/thr)", 1024, 1);
    RUNT("1024 thr x 2 (32 KB/wg, 32 B/thr)", 1024, 2);
    RUNT("512 thr x 4  (32 KB/wg, 64 B/thr)", 512, 4);
    RUNT("256 thr x 8  (32 KB/wg, 128 B/thr)", 256, 8);
    RUN("lds x8 (32 KB/wg, barrier)", (copy4_lds<8>), n4 / 2048);
    RUN("lds x2 (8 KB/wg, barrier)", (copy4_lds<2>), n4 / 512);
    // the same 32 KB-per-workgroup copy with its occupancy limited by a dynamic LDS allocation it never touches
    for (int kb : {20, 40, 54, 80, 160}) {
        char nm[64]; snprintf(nm, 64, "256 thr x 8, %d KB LDS (%d wg/cu)", kb, 160 / kb);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(copy4<8, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        time(nm, [&] { hipLaunchKernelGGL((copy4<8, true, true>), dim3(n4 / 2048), dim3(256), kb * 1024, 0, in, out, (unsigned)(n4 / 2048)); });
    }
    for (int kb : {20, 40, 80, 160}) {
        char nm[64]; snprintf(nm, 64, "256 thr x 16, %d KB LDS (%d wg/cu)", kb, 160 / kb);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(copy4<16, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        time(nm, [&] { hipLaunchKernelGGL((copy4<16, true, true>), dim3(n4 / 4096), dim3(256), kb * 1024, 0, in, out, (unsigned)(n4 / 4096)); });
    }
    for (int kb : {10, 20, 40, 80}) {
        char nm[64]; snprintf(nm, 64, "256 thr x 1, %d KB LDS (%d wg/cu)", kb, 160 / kb);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(copy4<1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        time(nm, [&] { hipLaunchKernelGGL((copy4<1, true, true>), dim3(n4 / 256), dim3(256), kb * 1024, 0, in, out, (unsigned)(n4 / 256)); });
    }
    for (int kb : {20, 40}) {
        char nm[64];
        snprintf(nm, 64, "x8 in batches of 4, %d wg/cu", 160 / kb);
        time(nm, [&] { hipLaunchKernelGGL((copy4_batched<4>), dim3(n4 / 2048), dim3(256), kb * 1024, 0, in, out, (unsigned)(n4 / 2048)); });
        snprintf(nm, 64, "x8 in batches of 2, %d wg/cu", 160 / kb);
        time(nm, [&] { hipLaunchKernelGGL((copy4_batched<2>), dim3(n4 / 2048), dim3(256), kb * 1024, 0, in, out, (unsigned)(n4 / 2048)); });
        snprintf(nm, 64, "x8 in batches of 1, %d wg/cu", 160 / kb);
        time(nm, [&] { hipLaunchKernelGGL((copy4_batched<1>), dim3(n4 / 2048), dim3(256), kb * 1024, 0, in, out, (unsigned)(n4 / 2048)); });
    }
    for (int w : {2, 4, 8}) {
        char nm[64]; snprintf(nm, 64, "f4 x4 nt persistent %d wg/cu", w);
        time(nm, [&] { hipLaunchKernelGGL((copy4_persist<4, true>), dim3(256 * w), dim3(256), 0, 0, in, out, (unsigned)(n4 / 1024)); });
        snprintf(nm, 64, "f4 x8 nt persistent %d wg/cu", w);
        time(nm, [&] { hipLaunchKernelGGL((copy4_persist<8, true>), dim3(256 * w), dim3(256), 0, 0, in, out, (unsigned)(n4 / 2048)); });
    }
    time("hipMemcpyAsync D2D", [&] { (void)hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0); });
    // smaller footprint: 256 MiB in + 256 MiB out (fits the 256 MB of infinity cache?)
    {
        const size_t small = (size_t)256 << 20, m4 = small / 16;
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((copy4<4, true, true>), dim3(m4 / 1024), dim3(256), 0, 0, in, out, (unsigned)(m4 / 1024));
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((copy4<4, true, true>), dim3(m4 / 1024), dim3(256), 0, 0, in, out, (unsigned)(m4 / 1024));
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 50;
        printf("%-40s %.4f ms  %.3f TB/s (read + write)\n", "f4 x4 nt xcd, 256 MiB + 256 MiB", ms, 2.0 * small / ms / 1e9);
    }
    return 0;
}
#undef RUN
#undef RUNT
}  // namespace in_flight

// ===== section `width` (was scripts/copy_patterns5.hip) ========================================
namespace width {
// Microbenchmark (dev tool): does the access width of the stores / loads matter at equal bytes per thread?
//   hipcc -O3 --offload-arch=gfx950 -o scripts/copy_patterns5 scripts/copy_patterns5.hip
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// 64 B per thread, tile of 16 KB per workgroup: loads of LW floats, stores of SW floats (through registers)
template <int LW, int SW>
__global__ __launch_bounds__(256) void copyw(const float *in, float *out, unsigned tiles) {
    const unsigned t = (blockIdx.x & 7) * (tiles >> 3) + (blockIdx.x >> 3);
    const float *s = in + (size_t)t * 4096;
    float *d = out + (size_t)t * 4096;
    float r[16];
    if constexpr (LW == 4) {
#pragma unroll
        for (int m = 0; m < 4; ++m) { f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(s) + threadIdx.x + 256 * m); r[4*m]=v.x; r[4*m+1]=v.y; r[4*m+2]=v.z; r[4*m+3]=v.w; }
    } else {
#pragma unroll
        for (int m = 0; m < 16; ++m) r[m] = __builtin_nontemporal_load(s + threadIdx.x + 256 * m);
    }
    if constexpr (SW == 4) {
#pragma unroll
        for (int m = 0; m < 4; ++m) { f4 v = {r[4*m], r[4*m+1], r[4*m+2], r[4*m+3]}; __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(d) + threadIdx.x + 256 * m); }
    } else if constexpr (SW == 2) {
#pragma unroll
        for (int m = 0; m < 8; ++m) { f2 v = {r[2*m], r[2*m+1]}; __builtin_nontemporal_store(v, reinterpret_cast<f2 *>(d) + threadIdx.x + 256 * m); }
    } else {
#pragma unroll
        for (int m = 0; m < 16; ++m) __builtin_nontemporal_store(r[m], d + threadIdx.x + 256 * m);
    }
}

int run() {
    const size_t bytes = (size_t)2 << 30;
    float *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned tiles = bytes / 16384;
    auto time = [&](const char *name, auto launch) -> int {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
        printf("%-44s %.4f ms  %.3f TB/s (read + write)\n", name, ms, 2.0 * bytes / ms / 1e9);
        return 0;
    };
#define RUN(name, K) time(name, [&] { hipLaunchKernelGGL(K, dim3(tiles), dim3(256), 0, 0, in, out, tiles); })
    RUN("warm-up", (copyw<4, 4>));
    RUN("16-byte loads, 16-byte stores", (copyw<4, 4>));
    RUN("16-byte loads,  8-byte stores", (copyw<4, 2>));
    RUN("16-byte loads,  4-byte stores", (copyw<4, 1>));
    RUN(" 4-byte loads, 16-byte stores", (copyw<1, 4>));
    RUN(" 4-byte loads,  4-byte stores", (copyw<1, 1>));
    return 0;
}
#undef RUN
#undef RUNT
}  // namespace width

int main(int argc, char **argv) {
    const char *which = argc > 1 ? argv[1] : "all";
    const bool all = !strcmp(which, "all");
    int rc = 0, ran = 0;
    if (all || !strcmp(which, "fft_tile")) { printf("== fft_tile: the access patterns the FFT row kernel could use\n"); rc |= fft_tile::run(); ++ran; }
    if (all || !strcmp(which, "streaming")) { printf("== streaming: best read+write streaming pattern (non-temporal, XCD map)\n"); rc |= streaming::run(); ++ran; }
    if (all || !strcmp(which, "row_segments")) { printf("== row_segments: the IIR-batch pattern: cost of short row segments\n"); rc |= row_segments::run(); ++ran; }
    if (all || !strcmp(which, "in_flight")) { printf("== in_flight: bytes in flight per CU against the streaming rate (profiles/r02_copy_ceiling.txt)\n"); rc |= in_flight::run(); ++ran; }
    if (all || !strcmp(which, "width")) { printf("== width: access width of loads / stores at equal bytes per thread\n"); rc |= width::run(); ++ran; }
    if (!ran) { printf("sections: fft_tile streaming row_segments in_flight width all\n"); return 2; }
    return rc;
}
