"""dsp_amd -- MI355X (gfx950) implementation of the ejosiah/dsp numeric hot path.

csrc/   hand-written HIP kernels + the C ABI (include/dsp_gpu.h) -> lib/libdsp_gpu.so
capi    ctypes binding of that ABI
ops     host-side mirror of the reference interface (fft0, fft, convolve, moving_average, iir)
"""
from . import capi, ops  # noqa: F401
from .ops import convolve, fft, fft0, iir, moving_average  # noqa: F401
