// mfma_f64_layout.hip -- where v_mfma_f64_16x16x4_f64 keeps its operands and results (dev tool: the layout the
// double-sample direct FIR kernel assumes is asserted here on the hardware).
//   hipcc -O2 --offload-arch=gfx950 -o scripts/mfma_f64_layout scripts/mfma_f64_layout.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void probe(int *table) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            d4 c = {0, 0, 0, 0};
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
            for (int r = 0; r < 4; ++r)
                if (c[r] != 0.0) table[la * 64 + lb] = lane * 4 + r;
        }
}

int main() {
    int *d;
    hipMalloc(&d, 4096 * sizeof(int));
    hipMemset(d, 0xff, 4096 * sizeof(int));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    std::vector<int> t(4096);
    hipMemcpy(t.data(), d, 4096 * sizeof(int), hipMemcpyDeviceToHost);
    // hypothesis: A lane l = A[i = l % 16][k = l / 16], B lane l = B[k = l / 16][j = l % 16],
    //             D[i][j] in lane j + 16 * (i % 4), register i / 4   (the fp32 16x16x4 form has lane j + 16 * (i / 4), register i % 4)
    int bad = 0, nonzero = 0;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const int got = t[la * 64 + lb];
            int want = -1;
            if (la / 16 == lb / 16) {
                const int i = la % 16, j = lb % 16;
                want = (j + 16 * (i % 4)) * 4 + (i / 4);
            }
            if (got >= 0) ++nonzero;
            if (got != want) {
                if (bad < 24) printf("la %d lb %d: got lane %d reg %d, hypothesis lane %d reg %d\n", la, lb, got >> 2, got & 3, want >> 2, want & 3);
                ++bad;
            }
        }
    printf("nonzero %d (expect 1024), mismatches against the hypothesis: %d\n", nonzero, bad);
    return bad != 0;
}
