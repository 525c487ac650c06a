// iir_batch_ext.hip -- the one-lane-per-channel IIR recursion for the sample types and arithmetics beside the
// benchmarked one (iir_batch_kernels.hip: fp32 samples, fast arithmetic):
//   double samples, fast arithmetic       dsp_gpu_iir_f64 BATCH / AUTO -- Coefficients<P>::operator()(SampleBuffer<double>),
//                                         the type every in-repo caller of the reference uses (coefficients.h:88-104)
//   float / double samples, REF arithmetic  DSP_GPU_IIR_BATCH_REF: coefficients.h:106-130 / 172-182 operation for operation
// Alignment is a kernel argument here (one instantiation per order instead of two): these are the widening routes.
#include "iir_batch_core.h"

namespace dspgpu {

namespace {
template <typename T, int ARITH>
int launch_any(const T *x, T *y, size_t channels, size_t n, const IirCoef &co, double *state, const IirPerChannel *pc,
               hipStream_t s) {
    using namespace iirb;
    const bool al = is_aligned<T>(x, y, n);
    if (pc && co.poles > 8) return fail(DSP_GPU_EUNSUPPORTED, "per-channel IIR coefficients: poles %d > 8", co.poles);
#define DSP_IIR_BATCH_CASE(P) return batch_dispatch<P, false, T, ARITH>(x, y, channels, n, co, state, pc, al, s)
    switch (co.poles) {  // orders without a kernel of their own run on the next one up (zero top coefficients: in the
                         // REF arithmetic the extra terms add +0.0, which changes no value)
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
}  // namespace

int launch_iir_batch_f64(const double *x, double *y, size_t channels, size_t n, const IirCoef &co, double *state,
                         const IirPerChannel *pc, hipStream_t s) {
    return launch_any<double, iirb::IIR_ARITH_FMA>(x, y, channels, n, co, state, pc, s);
}

int launch_iir_batch_ref(const float *x, float *y, size_t channels, size_t n, const IirCoef &co, double *state,
                         const IirPerChannel *pc, hipStream_t s) {
    return launch_any<float, iirb::IIR_ARITH_REF>(x, y, channels, n, co, state, pc, s);
}

int launch_iir_batch_ref_f64(const double *x, double *y, size_t channels, size_t n, const IirCoef &co, double *state,
                             const IirPerChannel *pc, hipStream_t s) {
    return launch_any<double, iirb::IIR_ARITH_REF>(x, y, channels, n, co, state, pc, s);
}

// The block-parallel scan stages its zero-state outputs as fp32 rows (iir_kernels.hip); a double-sample form of it is
// not built: AUTO runs double samples one lane per channel.
int launch_iir_scan_f64(const double *, double *, size_t, size_t, const IirCoef &, double *, hipStream_t) {
    return fail(DSP_GPU_EUNSUPPORTED, "IIR scan: double samples run in DSP_GPU_IIR_BATCH (AUTO picks it)");
}

}  // namespace dspgpu
