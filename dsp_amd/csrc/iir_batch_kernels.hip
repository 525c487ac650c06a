// iir_batch_kernels.hip -- the one-lane-per-channel IIR recursion on fp32 samples in the fast arithmetic (C4a): the
// instantiations of iir_batch_core.h that dsp_gpu_iir BATCH / AUTO and dsp_gpu_iir_multi run.  (A translation unit of
// its own so that the library builds in parallel: the unrolled tiles of twelve orders take minutes to compile.)
#include "iir_batch_core.h"

namespace dspgpu {

int launch_iir_batch(const float *x, float *y, size_t channels, size_t n, const IirCoef &co, double *state,
                     const IirPerChannel *pc, hipStream_t s) {
    using namespace iirb;
    const bool al = is_aligned<float>(x, y, n);
    if (pc && co.poles > 8) return fail(DSP_GPU_EUNSUPPORTED, "per-channel IIR coefficients: poles %d > 8", co.poles);
#define DSP_IIR_BATCH_CASE(P)                                                                                         \
    return al ? batch_dispatch<P, true, float, IIR_ARITH_FMA>(x, y, channels, n, co, state, pc, false, s)             \
              : batch_dispatch<P, false, float, IIR_ARITH_FMA>(x, y, channels, n, co, state, pc, false, s)
    switch (co.poles) {  // orders without a kernel of their own run on the next one up (zero top coefficients)
    case 1: DSP_IIR_BATCH_CASE(1);
    case 2: DSP_IIR_BATCH_CASE(2);
    case 3: DSP_IIR_BATCH_CASE(3);
    case 4: DSP_IIR_BATCH_CASE(4);
    case 5: DSP_IIR_BATCH_CASE(5);
    case 6: DSP_IIR_BATCH_CASE(6);
    case 7: DSP_IIR_BATCH_CASE(7);
    case 8: DSP_IIR_BATCH_CASE(8);
    case 9: case 10: DSP_IIR_BATCH_CASE(10);
    case 11: case 12: DSP_IIR_BATCH_CASE(12);
    case 13: case 14: case 15: case 16: DSP_IIR_BATCH_CASE(16);
    case 17: case 18: case 19: case 20: DSP_IIR_BATCH_CASE(20);
    default: return fail(DSP_GPU_EINVAL, "poles %d outside [1,20]", co.poles);
    }
#undef DSP_IIR_BATCH_CASE
}

}  // namespace dspgpu
