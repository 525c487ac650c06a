"""Host-side mirror of the reference's hot-path interface, on top of the C ABI.

Names and argument meaning follow the reference (file:line relative to its dsp/include/dsp/):

    fft0(x, inverse)                  fft.h:31-55      complex [.., 2^k] -> complex, unnormalised
    fft(x_real, nf, inverse, shift)   fft.h:57-94      real rows, zero padded to 2^ceil(log2 nf)
    convolve(x, taps, mode)           convolution.h:8-29 / filter.h:123-140
    moving_average(x, npoints)        filter.h:81-101
    iir(x, a, b, c0, d0, state, mode) coefficients.h:88-130,172-195

Inputs are torch CUDA tensors (device pointers, asynchronous on the current torch stream)
or numpy arrays (host pointers, staged by the library).  fp32 only.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

try:  # torch is plumbing for device memory / streams only
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_torch(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


def _stream_of(x) -> int:
    if _is_torch(x) and x.is_cuda:
        return torch.cuda.current_stream(x.device).cuda_stream
    return 0


def _prep(x, dtype_np, dtype_torch_name):
    """-> (contiguous array/tensor, data pointer, mem flag)"""
    if _is_torch(x):
        want = getattr(torch, dtype_torch_name)
        if not x.is_cuda:
            raise capi.DspGpuError(capi.EINVAL, "torch tensors must live on the GPU (pass numpy for host memory)")
        if x.dtype != want:
            raise capi.DspGpuError(capi.EINVAL, f"expected {want}, got {x.dtype}")
        x = x.contiguous()
        return x, x.data_ptr(), capi.MEM_DEVICE
    x = np.ascontiguousarray(x, dtype=dtype_np)
    return x, x.ctypes.data, capi.MEM_HOST


def _empty_like(x, shape, dtype_np, dtype_torch_name):
    if _is_torch(x):
        return torch.empty(shape, dtype=getattr(torch, dtype_torch_name), device=x.device)
    return np.empty(shape, dtype=dtype_np)


def _ptr(a):
    return a.data_ptr() if _is_torch(a) else a.ctypes.data


def _out(out, x, shape, dtype_np, dtype_torch_name):
    """The output buffer: a fresh one, or the caller's after the checks the kernels rely on (they write `count`
    elements through a raw pointer: same kind of container and device as x, dtype, size, contiguity)."""
    if out is None:
        return _empty_like(x, shape, dtype_np, dtype_torch_name)
    count = int(np.prod(shape))
    if _is_torch(out) != _is_torch(x):
        raise capi.DspGpuError(capi.EINVAL, "out must live where x lives (torch tensor with torch, numpy with numpy)")
    if _is_torch(out):
        ok = (out.device == x.device and out.dtype == getattr(torch, dtype_torch_name) and out.numel() == count
              and out.is_contiguous())
    else:
        ok = out.dtype == dtype_np and out.size == count and out.flags.c_contiguous and out.flags.writeable
    if not ok:
        raise capi.DspGpuError(capi.EINVAL, f"out must be a contiguous {dtype_torch_name} buffer of {count} elements on x's device")
    return out


class _on_device:
    """Runs the C call with x's GPU current: the library takes its workspaces and tables from hipGetDevice()."""

    def __init__(self, x):
        self._ctx = torch.cuda.device(x.device) if _is_torch(x) and x.is_cuda else None

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()

    def __exit__(self, *exc):
        if self._ctx is not None:
            return self._ctx.__exit__(*exc)
        return False


def fft0(x, inverse: bool = False, out=None):
    """Batched complex FFT over the last axis (length 2^k, k in 1..30).  fft.h:31-55."""
    x, px, mem = _prep(x, np.complex64, "complex64")
    n = x.shape[-1]
    log2n = int(n).bit_length() - 1
    if n < 2 or (1 << log2n) != n:
        raise capi.DspGpuError(capi.EINVAL, f"transform length {n} is not a power of two >= 2")
    batch = int(np.prod(x.shape[:-1])) if x.ndim > 1 else 1
    out = _out(out, x, tuple(x.shape), np.complex64, "complex64")
    with _on_device(x):
        capi.check(capi.lib().dsp_gpu_fft_batch(px, _ptr(out), log2n, batch, capi.INVERSE if inverse else capi.FORWARD,
                                                mem, _stream_of(x)))
    return out


def fft0_f64(x, inverse: bool = False, out=None):
    """The same in the reference's own sample type, std::complex<double> (fft.h:31-32): complex128 in and out, double
    arithmetic, length 2^k with k in 1..24."""
    x, px, mem = _prep(x, np.complex128, "complex128")
    n = x.shape[-1]
    log2n = int(n).bit_length() - 1
    if n < 2 or (1 << log2n) != n:
        raise capi.DspGpuError(capi.EINVAL, f"transform length {n} is not a power of two >= 2")
    batch = int(np.prod(x.shape[:-1])) if x.ndim > 1 else 1
    out = _out(out, x, tuple(x.shape), np.complex128, "complex128")
    with _on_device(x):
        capi.check(capi.lib().dsp_gpu_fft_batch_f64(px, _ptr(out), log2n, batch, capi.INVERSE if inverse else capi.FORWARD,
                                                    mem, _stream_of(x)))
    return out


def fft(x, nf: int, inverse: bool = False, shift: bool = False, out=None):
    """Real-input wrapper dsp::fft (fft.h:57-83), optionally followed by dsp::shift (fft.h:85-94)."""
    x, px, mem = _prep(x, np.float32, "float32")
    n_in = x.shape[-1]
    batch = int(np.prod(x.shape[:-1])) if x.ndim > 1 else 1
    log2n = max(1, int(np.ceil(np.log2(nf))))
    out = _out(out, x, tuple(x.shape[:-1]) + (1 << log2n,), np.complex64, "complex64")
    with _on_device(x):
        capi.check(capi.lib().dsp_gpu_fft_real_batch(px, n_in, n_in, _ptr(out), int(nf), batch,
                                                     capi.INVERSE if inverse else capi.FORWARD, int(shift), mem,
                                                     _stream_of(x)))
    return out


_MODES = {"auto": capi.CONV_AUTO, "direct": capi.CONV_DIRECT, "time": capi.CONV_DIRECT,
          "overlap_save": capi.CONV_OVERLAP_SAVE, "frequency": capi.CONV_OVERLAP_SAVE,
          "direct_valu": capi.CONV_DIRECT_VALU}


def convolve(x, taps, mode: str = "auto", out=None):
    """Causal FIR, first len(taps)-1 outputs zero.  x: [n] or [channels, n] fp32.  filter.h:123-140."""
    x, px, mem = _prep(x, np.float32, "float32")
    taps = np.ascontiguousarray(taps, dtype=np.float32)  # the designers produce double; the device runs fp32
    n = x.shape[-1]
    channels = int(np.prod(x.shape[:-1])) if x.ndim > 1 else 1
    y = _out(out, x, tuple(x.shape), np.float32, "float32")
    with _on_device(x):
        capi.check(capi.lib().dsp_gpu_convolve(px, _ptr(y), channels, n, taps.ctypes.data, taps.size, _MODES[mode], mem,
                                               _stream_of(x)))
    return y


def convolve_f64(x, taps, mode: str = "auto", out=None):
    """The same filter on double samples with double taps: SincFilter<>::apply(SampleBuffer<double>&), filter.h:123-140
    (time domain on the double matrix cores; mode "auto" or "direct")."""
    x, px, mem = _prep(x, np.float64, "float64")
    taps = np.ascontiguousarray(taps, dtype=np.float64)
    n = x.shape[-1]
    channels = int(np.prod(x.shape[:-1])) if x.ndim > 1 else 1
    y = _out(out, x, tuple(x.shape), np.float64, "float64")
    with _on_device(x):
        capi.check(capi.lib().dsp_gpu_convolve_f64(px, _ptr(y), channels, n, taps.ctypes.data, taps.size, _MODES[mode], mem,
                                                   _stream_of(x)))
    return y


def convolve_stream(x, taps, history, position: int, mode: str = "direct", out=None):
    """One chunk of a streamed FIR.  history: [channels, len(taps)-1] fp32 where x lives, updated in place;
    position = samples of the stream consumed so far (0 starts it; the first len(taps)-1 outputs of the
    stream are zero as in the one-shot call)."""
    x, px, mem = _prep(x, np.float32, "float32")
    taps = np.ascontiguousarray(taps, dtype=np.float32)
    n = x.shape[-1]
    channels = int(np.prod(x.shape[:-1])) if x.ndim > 1 else 1
    if _is_torch(history) != _is_torch(x):
        raise capi.DspGpuError(capi.EINVAL, "history must live where x lives")
    want = channels * (taps.size - 1)
    ok = (history.is_contiguous() and history.dtype == torch.float32 and history.numel() == want) if _is_torch(history) \
        else (history.flags.c_contiguous and history.dtype == np.float32 and history.size == want)
    if not ok:
        raise capi.DspGpuError(capi.EINVAL, "history must be contiguous fp32 [channels, ntaps-1]")
    y = _out(out, x, tuple(x.shape), np.float32, "float32")
    with _on_device(x):
        capi.check(capi.lib().dsp_gpu_convolve_stream(px, _ptr(y), channels, n, taps.ctypes.data, taps.size,
                                                      _ptr(history) if want else None, int(position), _MODES[mode], mem,
                                                      _stream_of(x)))
    return y


def moving_average(x, npoints: int, out=None):
    """Centred moving average, zero edges.  filter.h:81-101."""
    x, px, mem = _prep(x, np.float32, "float32")
    n = x.shape[-1]
    channels = int(np.prod(x.shape[:-1])) if x.ndim > 1 else 1
    y = _out(out, x, tuple(x.shape), np.float32, "float32")
    with _on_device(x):
        capi.check(capi.lib().dsp_gpu_moving_average(px, _ptr(y), channels, n, int(npoints), mem, _stream_of(x)))
    return y


def moving_average_f64(x, npoints: int, out=None):
    """MovingAverageFilter::apply on double samples (filter.h:81-101)."""
    x, px, mem = _prep(x, np.float64, "float64")
    n = x.shape[-1]
    channels = int(np.prod(x.shape[:-1])) if x.ndim > 1 else 1
    y = _out(out, x, tuple(x.shape), np.float64, "float64")
    with _on_device(x):
        capi.check(capi.lib().dsp_gpu_moving_average_f64(px, _ptr(y), channels, n, int(npoints), mem, _stream_of(x)))
    return y


_IIR_MODES = {"auto": capi.IIR_AUTO, "batch": capi.IIR_BATCH, "scan": capi.IIR_SCAN, "batch_ref": capi.IIR_BATCH_REF}


def _iir_any(x, f64, a, b, c0, d0, state, mode, out):
    np_t, torch_t = (np.float64, "float64") if f64 else (np.float32, "float32")
    x, px, mem = _prep(x, np_t, torch_t)
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    if a.size != b.size:
        raise capi.DspGpuError(capi.EINVAL, "a and b must have poles+1 entries each")
    poles = a.size - 1
    n = x.shape[-1]
    channels = int(np.prod(x.shape[:-1])) if x.ndim > 1 else 1
    y = _out(out, x, tuple(x.shape), np_t, torch_t)
    pstate = None
    if state is not None:
        if _is_torch(state) != _is_torch(x):
            raise capi.DspGpuError(capi.EINVAL, "state must live where x lives")
        if _is_torch(state):
            if not (state.is_contiguous() and state.dtype == torch.float64 and state.numel() == channels * 2 * poles):
                raise capi.DspGpuError(capi.EINVAL, "state must be contiguous float64 [channels, 2*poles]")
        elif not (state.flags.c_contiguous and state.dtype == np.float64 and state.size == channels * 2 * poles):
            raise capi.DspGpuError(capi.EINVAL, "state must be contiguous float64 [channels, 2*poles]")
        pstate = _ptr(state)
    fn = capi.lib().dsp_gpu_iir_f64 if f64 else capi.lib().dsp_gpu_iir
    with _on_device(x):
        capi.check(fn(px, _ptr(y), channels, n, a.ctypes.data, b.ctypes.data, poles, float(c0), float(d0), pstate,
                      _IIR_MODES[mode], mem, _stream_of(x)))
    return y


def iir(x, a, b, c0: float = 1.0, d0: float = 0.0, state=None, mode: str = "auto", out=None):
    """IIR recursion with feedback ADDED (coefficients.h:106-130), fp32 samples.

    state: [channels, 2*poles] float64 {x[t-1..t-poles], v[t-1..t-poles]}, updated in place
    (the reference's x[]/y[] history, coefficients.h:11-12).  mode "batch_ref": the reference's own arithmetic,
    bit for bit (float accumulator for the generic section, the all-double BiQuad for two poles)."""
    return _iir_any(x, False, a, b, c0, d0, state, mode, out)


def iir_f64(x, a, b, c0: float = 1.0, d0: float = 0.0, state=None, mode: str = "auto", out=None):
    """The same on double samples: Coefficients<P>::operator()(const SampleBuffer<double>&), the instantiation every
    in-repo caller of the reference uses (eval/recursive_filters/main.cpp:35).  One lane per channel ("scan" is
    unsupported, "auto" runs "batch")."""
    return _iir_any(x, True, a, b, c0, d0, state, mode, out)


def iir_settle(a, b) -> int:
    """Samples after which the response to any initial state has decayed below 2^-44 of it (dsp_gpu_iir_settle);
    host arithmetic only."""
    import ctypes as C
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    if a.size != b.size:
        raise capi.DspGpuError(capi.EINVAL, "a and b must have poles+1 entries each")
    out = C.c_size_t(0)
    capi.check(capi.lib().dsp_gpu_iir_settle(a.ctypes.data, b.ctypes.data, a.size - 1, C.byref(out)))
    return int(out.value)


def iir_multi(x, a, b, c0=None, d0=None, state=None, out=None, mode: str = "batch", f64: bool = False):
    """Filter bank: channel k runs its own Coefficients (a[k], b[k], c0[k], d0[k]); a, b [channels, poles+1]
    float64, c0/d0 [channels] float64 or None (1 and 0), all living where x lives.  Block-wise form of a
    retuned filter (eval/signals/wind_generator.h:150-167): call per block with new coefficients, carry `state`.
    f64: double samples; mode "batch_ref": the reference's own arithmetic (dsp_gpu_iir_multi_ex)."""
    np_t, torch_t = (np.float64, "float64") if f64 else (np.float32, "float32")
    x, px, mem = _prep(x, np_t, torch_t)
    n = x.shape[-1]
    channels = int(np.prod(x.shape[:-1])) if x.ndim > 1 else 1
    poles = a.shape[-1] - 1

    def chk(t, count, what):
        if t is None:
            return None
        if _is_torch(t) != _is_torch(x):
            raise capi.DspGpuError(capi.EINVAL, f"{what} must live where x lives")
        ok = (t.is_contiguous() and t.dtype == torch.float64 and t.numel() == count) if _is_torch(t) \
            else (t.flags.c_contiguous and t.dtype == np.float64 and t.size == count)
        if not ok:
            raise capi.DspGpuError(capi.EINVAL, f"{what} must be contiguous float64 with {count} entries")
        return _ptr(t)

    pa, pb = chk(a, channels * (poles + 1), "a"), chk(b, channels * (poles + 1), "b")
    pc0, pd0 = chk(c0, channels, "c0"), chk(d0, channels, "d0")
    pstate = chk(state, channels * 2 * poles, "state")
    y = _out(out, x, tuple(x.shape), np_t, torch_t)
    with _on_device(x):
        if not f64 and mode == "batch":
            capi.check(capi.lib().dsp_gpu_iir_multi(px, _ptr(y), channels, n, pa, pb, poles, pc0, pd0, pstate, mem,
                                                    _stream_of(x)))
        else:
            capi.check(capi.lib().dsp_gpu_iir_multi_ex(px, _ptr(y), capi.F64 if f64 else capi.F32, channels, n, pa, pb,
                                                       poles, pc0, pd0, pstate, _IIR_MODES[mode], mem, _stream_of(x)))
    return y


_SWEPT_KINDS = {"bandpass": 0, "bandreject": 1}


def biquad_swept(x, fc, bandwidth: float, kind: str = "bandpass", state=None, out=None):
    """A band-pass / band-reject BiQuad retuned at every sample (dsp_gpu_biquad_swept): per sample the design of
    recursive_filters.h:165-209 at centre frequency fc[t], then one BiQuad step (coefficients.h:172-182) -- the
    loop of eval/signals/wind_generator.h:150-167.  x float32 [..., n]; fc float32 of the same shape (a sweep per
    channel) or [n] (one sweep for all channels), living where x lives; state float64 [channels, 4] = {x1, x2, y1,
    y2}, carried across calls."""
    x, px, mem = _prep(x, np.float32, "float32")
    n = x.shape[-1]
    channels = int(np.prod(x.shape[:-1])) if x.ndim > 1 else 1
    if _is_torch(fc) != _is_torch(x):
        raise capi.DspGpuError(capi.EINVAL, "fc must live where x lives")
    if _is_torch(fc) and fc.device != x.device:
        raise capi.DspGpuError(capi.EINVAL, "fc must be on x's device")
    fc, pfc, _ = _prep(fc, np.float32, "float32")
    if tuple(fc.shape) == tuple(x.shape):
        stride = n
    elif tuple(fc.shape) == (n,):
        stride = 0
    else:
        raise capi.DspGpuError(capi.EINVAL, "fc must have x's shape or be one sweep of n samples")
    if kind not in _SWEPT_KINDS:
        raise capi.DspGpuError(capi.EINVAL, f"kind {kind!r}")
    pstate = None
    if state is not None:
        if _is_torch(state) != _is_torch(x):
            raise capi.DspGpuError(capi.EINVAL, "state must live where x lives")
        ok = (state.is_contiguous() and state.dtype == torch.float64 and state.numel() == channels * 4) \
            if _is_torch(state) else (state.flags.c_contiguous and state.dtype == np.float64 and state.size == channels * 4)
        if not ok:
            raise capi.DspGpuError(capi.EINVAL, "state must be contiguous float64 [channels, 4]")
        pstate = _ptr(state)
    y = _out(out, x, tuple(x.shape), np.float32, "float32")
    with _on_device(x):
        capi.check(capi.lib().dsp_gpu_biquad_swept(px, pfc, _ptr(y), channels, n, stride, _SWEPT_KINDS[kind],
                                                   float(bandwidth), pstate, mem, _stream_of(x)))
    return y


def fft_kernel_ms(x, out, inverse: bool, iters: int) -> float:
    """Average milliseconds per batched-FFT launch, timed with HIP events on x's stream."""
    n = x.shape[-1]
    log2n = int(n).bit_length() - 1
    batch = x.numel() // n
    ms = C.c_float(0.0)
    capi.check(capi.lib().dsp_gpu_fft_batch_timed(x.data_ptr(), out.data_ptr(), log2n, batch,
                                                  capi.INVERSE if inverse else capi.FORWARD, int(iters), C.byref(ms),
                                                  _stream_of(x)))
    return float(ms.value)
