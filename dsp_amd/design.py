"""Host-side filter design: the reference's designers, kept on the CPU in double.

The design functions of the reference (dsp.h:47-79, recursive_filters.h:24-263) run in
microseconds and only produce the small tap / coefficient arrays the kernels consume
(SURVEY.md 8a rows a9, a14, a15), so they stay on the host.  Same names and argument meaning
as the reference; `math` (libm) is used so the values match the reference's to the last bit.
All citations are relative to the reference's dsp/include/dsp/.
"""
from __future__ import annotations

import math

import numpy as np

PI = 3.1415926535897932384626433832795  # constants.h:4


# ---- windows (dsp.h:23-38) -------------------------------------------------------------------
def blackman(i: int, m: int) -> float:
    f = float(i) / float(m)
    return 0.42 - 0.5 * math.cos(2 * PI * f) + 0.08 * math.cos(4 * PI * f)


def hamming(i: int, m: int) -> float:
    f = float(i) / float(m)
    return 0.54 - 0.46 * math.cos(2.0 * PI * f)


def identity(i: int, m: int) -> float:
    return 1.0


WINDOWS = {"blackman": blackman, "hamming": hamming, "identity": identity}


def sinc(cf: float, length: int, window="blackman", inversion: str = "none") -> np.ndarray:
    """dsp::sinc<InversionType>(cf, length, window) (dsp.h:47-79).

    Low-pass windowed-sinc taps with unity DC gain; `length` is forced odd (dsp.h:60).
    inversion: "none" | "inversion" (spectral inversion, dsp.h:66-71) | "reversal" (dsp.h:72-76).
    """
    win = WINDOWS[window] if isinstance(window, str) else window
    length |= 1
    m = length - 1
    h = [0.0] * length
    for i in range(length):
        off = float(i) - float(m) / 2
        v = 2 * PI * cf
        if off != 0:
            v = math.sin(v * off) / off
        h[i] = v * win(i, m)
    total = 0.0
    for v in h:
        total = total + v
    h = [v / total for v in h]
    if inversion == "inversion":
        h = [-v for v in h]
        h[length // 2] += 1
    elif inversion == "reversal":
        h = [-v if (i & 1) else v for i, v in enumerate(h)]
    elif inversion != "none":
        raise ValueError(inversion)
    return np.array(h, dtype=np.float64)


# ---- Chebyshev (recursive_filters.h:24-163) --------------------------------------------------------
def _chebyshev_stage(highpass: bool, ripple: float, npoles: int, index: int, fc: float):
    """computePoles (recursive_filters.h:107-151): one pole pair -> one biquad stage."""
    ang = PI / (npoles * 2) + (index - 1) * PI / npoles
    re = -math.cos(ang)
    im = math.sin(ang)
    if ripple != 0:
        es = math.sqrt(math.pow(100.0 / (100.0 - ripple), 2) - 1)
        vx = (1.0 / npoles) * math.log((1.0 / es) + math.sqrt(1.0 / (es * es) + 1))
        kx = (1.0 / npoles) * math.log((1.0 / es) + math.sqrt(1.0 / (es * es) - 1))
        kx = (math.exp(kx) + math.exp(-kx)) * 0.5
        re *= ((math.exp(vx) - math.exp(-vx)) / 2) / kx
        im *= ((math.exp(vx) + math.exp(-vx)) / 2) / kx
    t = 2.0 * math.tan(0.5)
    w = 2.0 * PI * fc
    m = re * re + im * im
    d = 4.0 - 4.0 * re * t + m * t * t
    x0 = (t * t) / d
    x1 = 2.0 * x0
    x2 = x0
    y1 = (8.0 - 2 * m * t * t) / d
    y2 = (-4.0 - 4.0 * re * t - m * t * t) / d
    if highpass:
        k = -math.cos(w * 0.5 + 0.5) / math.cos(w * 0.5 - 0.5)
    else:
        k = math.sin(-w * 0.5 + 0.5) / math.sin(w * 0.5 + 0.5)
    d = 1.0 + y1 * k - y2 * k * k
    a0 = (x0 - x1 * k + x2 * k * k) / d
    a1 = (-2.0 * x0 * k + x1 + x1 * k * k - 2 * x2 * k) / d
    a2 = (x0 * k * k - x1 * k + x2) / d
    b1 = (2 * k + y1 + y1 * k * k - 2 * y2 * k) / d
    b2 = (-(k * k) - y1 * k + y2) / d
    if highpass:
        a1, b1 = -a1, -b1
    return a0, a1, a2, b1, b2


def chebyshev(fc: float, poles: int = 4, highpass: bool = False, ripple: float = 0.5):
    """recursive::lowPassFilter<P>(fc) / highPassFilter<P>(fc) (recursive_filters.h:153-163) via
    chebyshev::computeCoefficients (:24-105).  Returns (a[poles+1], b[poles+1]) with b[0] = 0;
    the recursion ADDS the feedback terms (coefficients.h:116-118)."""
    if poles < 2 or poles > 20 or poles % 2:
        raise ValueError("poles must be even, 2..20 (recursive_filters.h:30-31)")
    if not (0.0 <= fc <= 0.5) or not (0.0 <= ripple <= 29.0):
        raise ValueError("fc in [0, 0.5], ripple in [0, 29] (recursive_filters.h:27-29)")
    pa = [0.0] * 22
    pb = [0.0] * 22
    pa[2] = 1.0
    pb[2] = 1.0
    for p in range(1, poles // 2 + 1):
        a0, a1, a2, b1, b2 = _chebyshev_stage(highpass, ripple, poles, p, fc)
        qa, qb = pa[:], pb[:]
        for i in range(2, 22):
            pa[i] = a0 * qa[i] + a1 * qa[i - 1] + a2 * qa[i - 2]
            pb[i] = qb[i] - b1 * qb[i - 1] - b2 * qb[i - 2]
    pb[2] = 0.0
    for i in range(20):
        pa[i] = pa[i + 2]
        pb[i] = -pb[i + 2]
    sa = sb = 0.0
    for i in range(20):
        sgn = math.pow(-1, i) if highpass else 1.0
        sa += pa[i] * sgn
        sb += pb[i] * sgn
    gain = sa / (1 - sb)
    a = np.array([pa[i] / gain for i in range(poles + 1)], dtype=np.float64)
    b = np.array([pb[i] for i in range(poles + 1)], dtype=np.float64)
    return a, b


def low_pass(fc: float, poles: int = 4):
    return chebyshev(fc, poles, False, 0.5)


def high_pass(fc: float, poles: int = 4):
    return chebyshev(fc, poles, True, 0.5)


# ---- biquads (recursive_filters.h:165-263): return (a[3], b[3], c0, d0) ------------------------------
def _narrow(f: float, bw: float):
    r = 1 - 3 * bw
    c = math.cos(2 * PI * f)
    k = (1 - 2 * r * c + r * r) / (2 - 2 * c)
    return r, c, k


def band_pass(f: float, bw: float):
    r, c, k = _narrow(f, bw)
    return np.array([1 - k, 2 * (k - r) * c, r * r - k]), np.array([0.0, 2 * r * c, -(r * r)]), 1.0, 0.0


def band_reject(f: float, bw: float):
    r, c, k = _narrow(f, bw)
    return np.array([k, -2 * k * c, k]), np.array([0.0, 2 * r * c, -(r * r)]), 1.0, 0.0


def low_shelf(f: float, gain_db: float):
    u = math.pow(10.0, gain_db / 20)
    v = 4.0 / (1 + u)
    x = v * math.tan(f / 2)
    y = (1 - x) / (1 + x)
    a0 = (1 - y) / 2
    return np.array([a0, a0, 0.0]), np.array([0.0, y, 0.0]), u - 1, 1.0


def high_shelf(f: float, gain_db: float):
    u = math.pow(10, gain_db / 20)
    v = (1 + u) / 4
    x = v * math.tan(f / 2)
    y = (1 - x) / (1 + x)
    a0 = (1 + y) / 2
    return np.array([a0, -a0, 0.0]), np.array([0.0, y, 0.0]), u - 1, 1.0


def peaking(f: float, gain_db: float, q: float):
    u = math.pow(10, gain_db / 20)
    v = 4 / (1 + u)
    x = math.tan(f / (2 * q))
    vx = v * x
    y = .5 * ((1 - vx) / (1 + vx))
    z = (.5 + y) * math.cos(f)
    a0 = .5 - y
    return np.array([a0, 0.0, -a0]), np.array([0.0, 2 * z, -2 * y]), u - 1, 1.0
