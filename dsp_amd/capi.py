"""ctypes binding of libdsp_gpu.so (the C ABI declared in include/dsp_gpu.h).

This is host-side plumbing only: it loads the in-tree HIP library and passes raw pointers.
There is NO CPU fallback: if the library is missing, or a call fails, a DspGpuError is
raised.  torch is used by callers for device memory and streams, never for the arithmetic.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdsp_gpu.so")

OK, EINVAL, EHIP, ENOMEM, EUNSUPPORTED = 0, -1, -2, -3, -4
FORWARD, INVERSE = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
CONV_AUTO, CONV_DIRECT, CONV_OVERLAP_SAVE, CONV_DIRECT_VALU = 0, 1, 2, 3
IIR_AUTO, IIR_BATCH, IIR_SCAN, IIR_BATCH_REF = 0, 1, 2, 3
F32, F64 = 0, 1
DIST_TRANSPOSED_OUT = 1
DIST_SERIAL = 2

# every symbol include/dsp_gpu.h declares (tests check the library exports all of them)
EXPORTS = (
    "dsp_gpu_fft_batch",
    "dsp_gpu_fft_batch_f64",
    "dsp_gpu_fft_batch_noalloc",
    "dsp_gpu_fft_real_batch",
    "dsp_gpu_convolve",
    "dsp_gpu_convolve_f64",
    "dsp_gpu_convolve_stream",
    "dsp_gpu_moving_average",
    "dsp_gpu_moving_average_f64",
    "dsp_gpu_iir",
    "dsp_gpu_iir_f64",
    "dsp_gpu_iir_multi",
    "dsp_gpu_iir_multi_ex",
    "dsp_gpu_iir_settle",
    "dsp_gpu_biquad_swept",
    "dsp_gpu_device_count",
    "dsp_gpu_last_error_string",
    "dsp_gpu_shutdown",
    "dsp_gpu_version",
    "dsp_gpu_fft_batch_timed",
    "dsp_gpu_fft_columns",
    "dsp_gpu_swap_axes",
    "dsp_gpu_transpose_blocks",
    "dsp_gpu_fft_rows_exchange",
    "dsp_gpu_fft_dist_plan_create",
    "dsp_gpu_fft_dist_plan_create_rccl",
    "dsp_gpu_fft_dist_local_points",
    "dsp_gpu_fft_dist",
    "dsp_gpu_fft_dist_ex",
    "dsp_gpu_fft_dist_exchanges",
    "dsp_gpu_fft_dist_phase",
    "dsp_gpu_fft_dist_plan_destroy",
)

# dsp_gpu_exchange_fn: (user, send, recv, block_bytes, world, stream) -> status
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)


class DspGpuError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"dsp_gpu error {code}: {message}")
        self.code = code


_lib = None


def lib() -> C.CDLL:
    """Load libdsp_gpu.so (once).  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DspGpuError(EUNSUPPORTED, f"{LIB_PATH} not built (run __graft_entry__.build() / make -C dsp_amd/csrc)")
        L = C.CDLL(LIB_PATH)
        vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
        L.dsp_gpu_fft_batch.argtypes = [vp, vp, i, sz, i, i, vp]
        L.dsp_gpu_fft_batch_f64.argtypes = [vp, vp, i, sz, i, i, vp]
        L.dsp_gpu_iir_settle.argtypes = [vp, vp, i, C.POINTER(sz)]
        L.dsp_gpu_iir_settle.restype = i
        L.dsp_gpu_fft_batch_f64.restype = i
        L.dsp_gpu_fft_batch_noalloc.argtypes = [vp, vp, i, sz, i, vp]
        L.dsp_gpu_fft_batch_noalloc.restype = i
        L.dsp_gpu_fft_batch_timed.argtypes = [vp, vp, i, sz, i, i, C.POINTER(C.c_float), vp]
        L.dsp_gpu_fft_real_batch.argtypes = [vp, sz, sz, vp, i, sz, i, i, i, vp]
        L.dsp_gpu_convolve.argtypes = [vp, vp, sz, sz, vp, i, i, i, vp]
        L.dsp_gpu_convolve_stream.argtypes = [vp, vp, sz, sz, vp, i, vp, C.c_ulonglong, i, i, vp]
        L.dsp_gpu_convolve_stream.restype = i
        L.dsp_gpu_moving_average.argtypes = [vp, vp, sz, sz, i, i, vp]
        L.dsp_gpu_moving_average_f64.argtypes = [vp, vp, sz, sz, i, i, vp]
        L.dsp_gpu_moving_average_f64.restype = i
        L.dsp_gpu_convolve_f64.argtypes = [vp, vp, sz, sz, vp, i, i, i, vp]
        L.dsp_gpu_convolve_f64.restype = i
        L.dsp_gpu_iir.argtypes = [vp, vp, sz, sz, vp, vp, i, C.c_double, C.c_double, vp, i, i, vp]
        L.dsp_gpu_iir_multi.argtypes = [vp, vp, sz, sz, vp, vp, i, vp, vp, vp, i, vp]
        L.dsp_gpu_iir_multi.restype = i
        L.dsp_gpu_iir_f64.argtypes = [vp, vp, sz, sz, vp, vp, i, C.c_double, C.c_double, vp, i, i, vp]
        L.dsp_gpu_iir_f64.restype = i
        L.dsp_gpu_iir_multi_ex.argtypes = [vp, vp, i, sz, sz, vp, vp, i, vp, vp, vp, i, i, vp]
        L.dsp_gpu_iir_multi_ex.restype = i
        L.dsp_gpu_biquad_swept.argtypes = [vp, vp, vp, sz, sz, sz, i, C.c_double, vp, i, vp]
        L.dsp_gpu_biquad_swept.restype = i
        L.dsp_gpu_fft_columns.argtypes = [vp, vp, i, sz, i, vp]
        L.dsp_gpu_swap_axes.argtypes = [vp, vp, sz, sz, sz, i, sz, i, vp]
        L.dsp_gpu_transpose_blocks.argtypes = [vp, vp, sz, sz, sz, vp]
        L.dsp_gpu_fft_rows_exchange.argtypes = [vp, vp, i, sz, sz, i, sz, i, vp]
        for name in ("dsp_gpu_fft_columns", "dsp_gpu_swap_axes", "dsp_gpu_transpose_blocks", "dsp_gpu_fft_rows_exchange"):
            getattr(L, name).restype = i
        for name in ("dsp_gpu_fft_batch", "dsp_gpu_fft_batch_timed", "dsp_gpu_fft_real_batch", "dsp_gpu_convolve",
                     "dsp_gpu_moving_average", "dsp_gpu_iir", "dsp_gpu_device_count", "dsp_gpu_shutdown"):
            getattr(L, name).restype = i
        L.dsp_gpu_fft_dist_plan_create.argtypes = [C.POINTER(vp), i, i, i, EXCHANGE_FN, vp]
        L.dsp_gpu_fft_dist_plan_create_rccl.argtypes = [C.POINTER(vp), i, i, i, vp]
        L.dsp_gpu_fft_dist_local_points.argtypes = [vp]
        L.dsp_gpu_fft_dist_local_points.restype = sz
        L.dsp_gpu_fft_dist.argtypes = [vp, vp, vp, i, vp]
        L.dsp_gpu_fft_dist_ex.argtypes = [vp, vp, vp, i, i, vp]
        L.dsp_gpu_fft_dist_ex.restype = i
        L.dsp_gpu_fft_dist_exchanges.argtypes = [vp, i]
        L.dsp_gpu_fft_dist_exchanges.restype = i
        L.dsp_gpu_fft_dist_phase.argtypes = [vp, i, vp, vp, i, vp]
        L.dsp_gpu_fft_dist_plan_destroy.argtypes = [vp]
        for name in ("dsp_gpu_fft_dist_plan_create", "dsp_gpu_fft_dist_plan_create_rccl", "dsp_gpu_fft_dist",
                     "dsp_gpu_fft_dist_phase", "dsp_gpu_fft_dist_plan_destroy"):
            getattr(L, name).restype = i
        L.dsp_gpu_last_error_string.restype = C.c_char_p
        L.dsp_gpu_version.restype = C.c_char_p
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != OK:
        raise DspGpuError(rc, lib().dsp_gpu_last_error_string().decode(errors="replace"))
