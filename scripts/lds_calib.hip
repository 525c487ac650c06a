// lds_calib.hip -- what SQ_LDS_BANK_CONFLICT reads on access patterns whose conflict degree is known (dev tool).
// Each kernel hammers LDS with one instruction form and one address pattern; run under
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -- ./scripts/lds_calib
// and divide BANK_CONFLICT by IDX_ACTIVE per kernel.
//   hipcc -O3 --offload-arch=gfx950 -o scripts/lds_calib scripts/lds_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename T, int STRIDE, bool WRITE>  // lane l touches element l * STRIDE (+ a per-iteration offset)
__global__ __launch_bounds__(256) void lds_kernel(float *sink, int iters) {
    __shared__ __attribute__((aligned(16))) float buf[16384];
    T *b = reinterpret_cast<T *>(buf);
    constexpr int kElems = 16384 * 4 / sizeof(T);
    const int lane = threadIdx.x;
    for (int i = lane; i < 16384; i += 256) buf[i] = static_cast<float>(i);
    __syncthreads();
    T acc{};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = (lane * STRIDE + it + 64 * u) & (kElems - 1);
            if (WRITE) b[idx] = acc + static_cast<float>(u);
            else acc += b[idx];
        }
    }
    if (lane == 300) sink[0] = reinterpret_cast<float *>(&acc)[0];
    if (!WRITE) asm volatile("" ::"v"(acc));
}

#define RUN(T, S, W, name)                                                                 \
    hipLaunchKernelGGL((lds_kernel<T, S, W>), dim3(1024), dim3(256), 0, 0, sink, 200);     \
    hipDeviceSynchronize();                                                                \
    printf("%s\n", name);

int main() {
    float *sink;
    hipMalloc(&sink, 256);
    RUN(float, 1, false, "read_b32 stride 1 (clean)");
    RUN(float, 2, false, "read_b32 stride 2 (2-way)");
    RUN(float, 32, false, "read_b32 stride 32 (32-way)");
    RUN(f2, 1, false, "read_b64 stride 1 (clean)");
    RUN(f2, 17, false, "read_b64 stride 17 elements (the FFT image's padded radix-16 gather)");
    RUN(f4, 1, false, "read_b128 stride 1 (clean)");
    RUN(f4, 17, false, "read_b128 stride 17 elements = 68 dwords (the IIR scan's row walk)");
    RUN(float, 1, true, "write_b32 stride 1 (clean)");
    RUN(f2, 1, true, "write_b64 stride 1 (clean)");
    RUN(f2, 17, true, "write_b64 stride 17 elements (the FFT image's padded radix-16 scatter)");
    RUN(f4, 1, true, "write_b128 stride 1 (clean)");
    RUN(f4, 17, true, "write_b128 stride 17 elements = 68 dwords");
    return 0;
}
