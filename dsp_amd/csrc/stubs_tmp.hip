// temporary: FIR / IIR launchers until their kernels land
#include "common.h"
namespace dspgpu {
int launch_fir_direct(const float *, float *, size_t, size_t, const float *, int, hipStream_t) { return fail(DSP_GPU_EUNSUPPORTED, "fir_direct not built"); }
int launch_fir_overlap_save(const float *, float *, size_t, size_t, const float2 *, int, hipStream_t) { return fail(DSP_GPU_EUNSUPPORTED, "fir_os not built"); }
int launch_fir_spectrum(const float *, int, float2 *, hipStream_t) { return fail(DSP_GPU_EUNSUPPORTED, "fir_spectrum not built"); }
int launch_moving_average(const float *, float *, size_t, size_t, int, hipStream_t) { return fail(DSP_GPU_EUNSUPPORTED, "mavg not built"); }
int launch_iir_batch(const float *, float *, size_t, size_t, const IirCoef &, float *, hipStream_t) { return fail(DSP_GPU_EUNSUPPORTED, "iir_batch not built"); }
int launch_iir_scan(const float *, float *, size_t, size_t, const IirCoef &, float *, hipStream_t) { return fail(DSP_GPU_EUNSUPPORTED, "iir_scan not built"); }
}
