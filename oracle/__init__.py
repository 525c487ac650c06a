"""ctypes front-end of the CPU oracle libraries (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product path (dsp_amd/) never does.

    from oracle import load
    orc = load("oracle")   # oracle/libdsp_oracle.so      -- our restatement, strict IEEE
    ref = load("ref")      # oracle/_ref/libdsp_ref.so    -- the reference's own headers (may be absent)
    fast = load("fast")    # oracle/libdsp_oracle_fast.so -- restatement at -O3/AVX2 for CPU timing

Every method takes/returns numpy arrays; complex data is complex128 / complex64.
The functions mirror oracle/dsp_oracle.h, which cites the reference file:line of each.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATHS = {
    "oracle": ("libdsp_oracle.so", "orc_"),
    "fast": ("libdsp_oracle_fast.so", "orc_"),
    "ref": (os.path.join("_ref", "libdsp_ref.so"), "ref_"),
}

WINDOW = {"blackman": 0, "identity": 1, "hamming": 2}
INVERSION = {"inversion": 0, "reversal": 1, "none": 2}
BIQUAD_KIND = {"bandpass": 0, "bandreject": 1, "lowshelf": 2, "highshelf": 3, "peaking": 4}

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)


def _d(a):
    return a.ctypes.data_as(_dp)


def _f(a):
    return a.ctypes.data_as(_fp)


def available(kind: str) -> bool:
    return os.path.exists(os.path.join(_HERE, _PATHS[kind][0]))


class OracleLib:
    def __init__(self, kind: str):
        rel, self.prefix = _PATHS[kind]
        self.kind = kind
        self.path = os.path.join(_HERE, rel)
        self.lib = C.CDLL(self.path)

    def _fn(self, name, restype=None):
        fn = getattr(self.lib, self.prefix + name)
        fn.restype = restype
        return fn

    # ---- FFT -------------------------------------------------------------
    def bit_reverse(self, x: int, n: int) -> int:
        return self._fn("bit_reverse", C.c_int)(C.c_int(x), C.c_int(n))

    def fft0(self, x, inverse=False):
        """fft.h:31-55 on a complex128 vector of length 2^k (or [batch, 2^k])."""
        x = np.ascontiguousarray(x, dtype=np.complex128)
        n = x.shape[-1]
        log2n = int(n).bit_length() - 1
        assert 1 << log2n == n
        out = np.empty_like(x)
        xi, oi = x.reshape(-1, n), out.reshape(-1, n)
        self._fn("fft0_batch")(_d(xi.view(np.float64)), _d(oi.view(np.float64)), C.c_int(log2n), C.c_int(int(inverse)),
                               C.c_size_t(xi.shape[0]))
        return out

    def fft0_f32(self, x, inverse=False):
        x = np.ascontiguousarray(x, dtype=np.complex64)
        n = x.shape[-1]
        log2n = int(n).bit_length() - 1
        assert 1 << log2n == n
        out = np.empty_like(x)
        fn = self._fn("fft0_f32")
        xi, oi = x.reshape(-1, n), out.reshape(-1, n)
        for r in range(xi.shape[0]):
            fn(_f(xi[r].view(np.float32)), _f(oi[r].view(np.float32)), C.c_int(log2n), C.c_int(int(inverse)))
        return out

    def fft_real(self, td, nf: int, inverse=False):
        """fft.h:57-83: real samples, zero padded to 2^ceil(log2 nf)."""
        td = np.ascontiguousarray(td, dtype=np.float64)
        log2n = int(np.ceil(np.log2(nf)))
        out = np.empty(1 << log2n, dtype=np.complex128)
        got = self._fn("fft_real", C.c_int)(_d(td), C.c_size_t(td.size), _d(out.view(np.float64)),
                                            C.c_int(nf), C.c_int(int(inverse)))
        assert got == out.size
        return out

    def shift(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex128)
        out = np.empty_like(x)
        self._fn("shift")(_d(x.view(np.float64)), _d(out.view(np.float64)), C.c_size_t(x.size))
        return out

    # ---- FIR -------------------------------------------------------------
    def sinc(self, cf: float, length: int, window="blackman", inversion="none"):
        out = np.empty(length | 1, dtype=np.float64)
        n = self._fn("sinc", C.c_int)(C.c_double(cf), C.c_int(length), C.c_int(WINDOW[window]),
                                      C.c_int(INVERSION[inversion]), _d(out))
        assert n == out.size
        return out

    def fir(self, x, h):
        """filter.h:123-140; dtype of x (float32/float64) selects SampleType. x: [n] or [ch, n]."""
        h = np.ascontiguousarray(h, dtype=np.float64)
        x = np.ascontiguousarray(x)
        f32 = x.dtype == np.float32
        if not f32:
            x = x.astype(np.float64, copy=False)
        y = np.empty_like(x)
        fn = self._fn("fir_f32" if f32 else "fir_f64")
        p = _f if f32 else _d
        xi, yi = x.reshape(-1, x.shape[-1]), y.reshape(-1, x.shape[-1])
        for r in range(xi.shape[0]):
            fn(p(xi[r]), C.c_size_t(xi.shape[1]), _d(h), C.c_int(h.size), p(yi[r]))
        return y

    def moving_average(self, x, npoints: int):
        x = np.ascontiguousarray(x)
        f32 = x.dtype == np.float32
        if not f32:
            x = x.astype(np.float64, copy=False)
        y = np.empty_like(x)
        fn = self._fn("moving_average_f32" if f32 else "moving_average_f64")
        p = _f if f32 else _d
        xi, yi = x.reshape(-1, x.shape[-1]), y.reshape(-1, x.shape[-1])
        for r in range(xi.shape[0]):
            fn(p(xi[r]), C.c_size_t(xi.shape[1]), C.c_size_t(npoints), p(yi[r]))
        return y

    # ---- IIR design ------------------------------------------------------
    def chebyshev(self, fc: float, poles: int = 4, highpass=False, ripple=0.5):
        """recursive_filters.h:24-163 (lowPassFilter/highPassFilter use ripple 0.5)."""
        a = np.zeros(poles + 1)
        b = np.zeros(poles + 1)
        rc = self._fn("chebyshev", C.c_int)(C.c_int(int(highpass)), C.c_double(ripple), C.c_double(fc),
                                            C.c_int(poles), _d(a), _d(b))
        if rc != 0:
            raise ValueError("chebyshev design rejected arguments")
        return a, b

    def biquad_design(self, kind: str, p0: float, p1: float, p2: float = 0.0):
        a, b, cd = np.zeros(3), np.zeros(3), np.zeros(2)
        rc = self._fn("biquad_design", C.c_int)(C.c_int(BIQUAD_KIND[kind]), C.c_double(p0), C.c_double(p1),
                                                C.c_double(p2), _d(a), _d(b), _d(cd))
        if rc != 0:
            raise ValueError(kind)
        return a, b, float(cd[0]), float(cd[1])

    # ---- IIR run ---------------------------------------------------------
    def iir(self, x, a, b, c0=1.0, d0=0.0, state=None):
        """coefficients.h:88-130 generic N-pole recursion. state: 2*(poles+1) doubles, updated in place."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        poles = a.size - 1
        x = np.ascontiguousarray(x)
        f32 = x.dtype == np.float32
        if not f32:
            x = x.astype(np.float64, copy=False)
        y = np.empty_like(x)
        fn = self._fn("iir_f32" if f32 else "iir_f64")
        p = _f if f32 else _d
        xi, yi = x.reshape(-1, x.shape[-1]), y.reshape(-1, x.shape[-1])
        if state is not None:
            assert state.dtype == np.float64 and state.size == xi.shape[0] * 2 * (poles + 1)
            st = state.reshape(xi.shape[0], 2 * (poles + 1))
        for r in range(xi.shape[0]):
            sp = _d(st[r]) if state is not None else None
            fn(p(xi[r]), C.c_size_t(xi.shape[1]), _d(a), _d(b), C.c_int(poles), C.c_double(c0), C.c_double(d0),
               sp, p(yi[r]))
        return y

    def biquad(self, x, a, b, c0=1.0, d0=0.0, state=None):
        """coefficients.h:172-195 BiQuad. state: {x1,x2,y1,y2} doubles per channel, updated in place."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        assert a.size == 3 and b.size == 3
        x = np.ascontiguousarray(x)
        f32 = x.dtype == np.float32
        if not f32:
            x = x.astype(np.float64, copy=False)
        y = np.empty_like(x)
        fn = self._fn("biquad_f32" if f32 else "biquad_f64")
        p = _f if f32 else _d
        xi, yi = x.reshape(-1, x.shape[-1]), y.reshape(-1, x.shape[-1])
        if state is not None:
            st = state.reshape(xi.shape[0], 4)
        for r in range(xi.shape[0]):
            sp = _d(st[r]) if state is not None else None
            fn(p(xi[r]), C.c_size_t(xi.shape[1]), _d(a), _d(b), C.c_double(c0), C.c_double(d0), sp, p(yi[r]))
        return y

    def biquad_swept(self, x, fc, bw: float, kind="bandpass", state=None):
        """eval/signals/wind_generator.h:150-167: the BiQuad redesigned from fc[t] before every step.
        x float32 [n] or [ch, n]; fc float32 of x's shape or [n]; state [ch, 4] doubles, updated in place."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        fc = np.ascontiguousarray(fc, dtype=np.float32)
        y = np.empty_like(x)
        xi, yi = x.reshape(-1, x.shape[-1]), y.reshape(-1, x.shape[-1])
        fi = np.broadcast_to(fc, x.shape).reshape(xi.shape)
        if state is not None:
            assert state.dtype == np.float64 and state.size == xi.shape[0] * 4
            st = state.reshape(xi.shape[0], 4)
        fn = self._fn("biquad_swept_f32")
        for r in range(xi.shape[0]):
            sp = _d(st[r]) if state is not None else None
            fn(_f(xi[r]), _f(np.ascontiguousarray(fi[r])), C.c_size_t(xi.shape[1]), C.c_int(BIQUAD_KIND[kind]),
               C.c_double(bw), sp, _f(yi[r]))
        return y


_cache: dict[str, OracleLib] = {}


def load(kind: str = "oracle") -> OracleLib:
    if kind not in _cache:
        _cache[kind] = OracleLib(kind)
    return _cache[kind]
