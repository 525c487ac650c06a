"""The restatement (oracle/) against the reference's own headers (oracle/_ref) on fresh seeds.

Runs only where /root/reference was available to build oracle/_ref (this container); skipped on
the GPU box.  Everything must agree bit-for-bit.
"""
import numpy as np
import pytest


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


@pytest.mark.parametrize("log2n", [1, 2, 3, 5, 7, 10, 12, 14])
def test_fft0(orc, ref, log2n):
    rng = np.random.default_rng(100 + log2n)
    n = 1 << log2n
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    for inv in (False, True):
        assert same(orc.fft0(x, inv), ref.fft0(x, inv))


@pytest.mark.parametrize("n_in,nf", [(5, 8), (64, 64), (100, 128), (300, 300), (1000, 4096), (70, 64)])
def test_fft_real(orc, ref, n_in, nf):
    rng = np.random.default_rng(n_in * 31 + nf)
    x = rng.standard_normal(n_in)
    for inv in (False, True):
        assert same(orc.fft_real(x, nf, inv), ref.fft_real(x, nf, inv))


def test_shift_and_bitrev(orc, ref):
    rng = np.random.default_rng(5)
    for n in (2, 8, 64, 1024):
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        assert same(orc.shift(x), ref.shift(x))
        assert [orc.bit_reverse(i, n) for i in range(n)] == [ref.bit_reverse(i, n) for i in range(n)]


@pytest.mark.parametrize("cf,length,window,inv", [
    (0.1, 511, "blackman", "none"), (0.14, 200, "blackman", "none"), (0.25, 33, "hamming", "inversion"),
    (0.05, 1001, "blackman", "reversal"), (0.4, 7, "identity", "none"), (0.01, 2048, "hamming", "none")])
def test_sinc(orc, ref, cf, length, window, inv):
    assert same(orc.sinc(cf, length, window, inv), ref.sinc(cf, length, window, inv))


@pytest.mark.parametrize("n,taps", [(64, 1), (100, 11), (2000, 201), (5000, 511), (20, 21)])
def test_fir(orc, ref, n, taps):
    rng = np.random.default_rng(n + taps)
    h = orc.sinc(0.2, taps)
    x = rng.standard_normal(n)
    assert same(orc.fir(x, h), ref.fir(x, h))
    assert same(orc.fir(x.astype(np.float32), h), ref.fir(x.astype(np.float32), h))


@pytest.mark.parametrize("n,pts", [(16, 5), (1000, 511), (1000, 8), (511, 511), (64, 1)])
def test_moving_average(orc, ref, n, pts):
    rng = np.random.default_rng(n + pts)
    x = rng.standard_normal(n)
    assert same(orc.moving_average(x, pts), ref.moving_average(x, pts))
    assert same(orc.moving_average(x.astype(np.float32), pts), ref.moving_average(x.astype(np.float32), pts))


@pytest.mark.parametrize("poles", [2, 4, 6, 8, 12, 20])
@pytest.mark.parametrize("hp", [False, True])
def test_chebyshev_design(orc, ref, poles, hp):
    for fc in (0.01, 0.1, 0.25, 0.45):
        a0, b0 = orc.chebyshev(fc, poles, hp)
        a1, b1 = ref.chebyshev(fc, poles, hp)
        assert same(a0, a1) and same(b0, b1)


@pytest.mark.parametrize("poles", [4, 6, 8, 12])  # Coefficients<2> is the BiQuad specialisation: test_biquad
def test_iir_run(orc, ref, poles):
    rng = np.random.default_rng(poles)
    a, b = ref.chebyshev(0.1, poles)
    x = rng.standard_normal(3000)
    for xx in (x, x.astype(np.float32)):
        s0, s1 = np.zeros(2 * (poles + 1)), np.zeros(2 * (poles + 1))
        y0 = np.concatenate([orc.iir(xx[:1234], a, b, state=s0), orc.iir(xx[1234:], a, b, state=s0)])
        y1 = np.concatenate([ref.iir(xx[:1234], a, b, state=s1), ref.iir(xx[1234:], a, b, state=s1)])
        assert same(y0, y1) and same(s0, s1)


@pytest.mark.parametrize("kind,args", [("bandpass", (0.1, 0.01)), ("bandreject", (0.3, 0.05)), ("lowshelf", (0.2, 3.0)),
                                       ("highshelf", (0.7, -6.0)), ("peaking", (0.9, 2.0, 1.5))])
def test_biquad(orc, ref, kind, args):
    d0, d1 = orc.biquad_design(kind, *args), ref.biquad_design(kind, *args)
    assert same(d0[0], d1[0]) and same(d0[1], d1[1]) and d0[2:] == d1[2:]
    rng = np.random.default_rng(11)
    x = rng.standard_normal(2000)
    for xx in (x, x.astype(np.float32)):
        s0, s1 = np.zeros(4), np.zeros(4)
        y0 = np.concatenate([orc.biquad(xx[:777], *d0, state=s0), orc.biquad(xx[777:], *d0, state=s0)])
        y1 = np.concatenate([ref.biquad(xx[:777], *d1, state=s1), ref.biquad(xx[777:], *d1, state=s1)])
        assert same(y0, y1) and same(s0, s1)


@pytest.mark.parametrize("kind", ["bandpass", "bandreject"])
def test_biquad_swept(orc, ref, kind):
    # the retuned band-pass of eval/signals/wind_generator.h:150-167, against the reference's own designer + BiQuad
    rng = np.random.default_rng(21)
    x = rng.standard_normal((3, 2500)).astype(np.float32)
    t = np.arange(2500)
    fc = (0.02 + rng.uniform(0.0, 0.4, size=(3, 1)) * (0.5 + 0.5 * np.sin(rng.uniform(0.001, 0.05, size=(3, 1)) * t))).astype(np.float32)
    s0, s1 = rng.standard_normal((3, 4)), None
    s1 = s0.copy()
    y0 = np.concatenate([orc.biquad_swept(x[:, :701], fc[:, :701], 0.004, kind, state=s0),
                         orc.biquad_swept(x[:, 701:], fc[:, 701:], 0.004, kind, state=s0)], axis=1)
    y1 = np.concatenate([ref.biquad_swept(x[:, :701], fc[:, :701], 0.004, kind, state=s1),
                         ref.biquad_swept(x[:, 701:], fc[:, 701:], 0.004, kind, state=s1)], axis=1)
    assert same(y0, y1) and same(s0, s1)
    assert same(orc.biquad_swept(x, fc[0], 0.004, kind), ref.biquad_swept(x, fc[0], 0.004, kind))  # one sweep for all
