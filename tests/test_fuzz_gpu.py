"""Seeded random shapes through every entry point against the oracle: ragged channel counts, lengths that straddle
tile borders, tap / window / pole counts across the kernels' internal switches.  Sizes stay small enough for the
CPU oracle to answer in seconds."""
import os

import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu
SEEDS = int(os.environ.get("DSP_FUZZ_SEEDS", "6"))  # a longer soak: DSP_FUZZ_SEEDS=60 python -m pytest tests/test_fuzz_gpu.py


@pytest.fixture(scope="module")
def ops(dev):
    from dsp_amd import ops as o
    return o


def _stable_section(rng, poles):
    """Random stable direct form: poles inside |z| < 0.9 in conjugate pairs (plus one real pole for odd orders)."""
    roots = []
    for _ in range(poles // 2):
        r, th = rng.uniform(0.2, 0.9), rng.uniform(0.1, 3.0)
        roots += [r * np.exp(1j * th), r * np.exp(-1j * th)]
    if poles % 2:
        roots.append(rng.uniform(-0.9, 0.9))
    den = np.real(np.poly(roots))            # 1 + d1 z^-1 + ...
    b = np.concatenate([[0.0], -den[1:]])    # the reference ADDS the feedback terms
    a = rng.standard_normal(poles + 1) * 0.5
    return a, b


@pytest.mark.parametrize("seed", range(SEEDS))
def test_fft_random_shapes(ops, dev, orc, seed):
    import torch
    rng = np.random.default_rng(1000 + seed)
    for _ in range(8):
        log2n = int(rng.integers(1, 15))
        n, batch = 1 << log2n, int(rng.integers(1, 40))
        x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
        inv = bool(rng.integers(0, 2))
        got = ops.fft0(torch.from_numpy(x).to(dev), inverse=inv).cpu().numpy()
        want = orc.fft0(x.astype(np.complex128), inv)
        assert rel_l2(got, want) <= 1e-5 * max(log2n, 1), (log2n, batch, inv)
        # real wrapper with a ragged input length and zero padding
        n_in = int(rng.integers(1, n + 1))
        xr = rng.standard_normal((batch, n_in)).astype(np.float32)
        nf = int(rng.integers(max(n // 2 + 1, 2), n + 1)) if n > 2 else n
        got = ops.fft(torch.from_numpy(xr).to(dev), nf, inverse=inv).cpu().numpy()
        for r in (0, batch - 1):
            want = orc.fft_real(xr[r].astype(np.float64), nf, inv)
            assert got.shape[1] == want.size and rel_l2(got[r], want) <= 1e-5 * max(log2n, 1) + 1e-7, (n_in, nf)


@pytest.mark.parametrize("seed", range(SEEDS))
def test_fir_random_shapes(ops, dev, orc, seed):
    import torch
    rng = np.random.default_rng(2000 + seed)
    for _ in range(6):
        ntaps = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 255, 511, 600, int(rng.integers(1, 700))]))
        channels, n = int(rng.integers(1, 9)), int(rng.integers(1, 30000))
        h = rng.standard_normal(ntaps) / np.sqrt(ntaps)
        x = rng.standard_normal((channels, n)).astype(np.float32)
        want = orc.fir(x.astype(np.float64), h.astype(np.float32).astype(np.float64))
        xd = torch.from_numpy(x).to(dev)
        for mode, tol in (("direct", 1e-6), ("direct_valu", 1e-6), ("overlap_save", 1.2e-4), ("auto", 1.2e-4)):
            got = ops.convolve(xd, h, mode=mode).cpu().numpy()
            assert rel_l2(got, want) <= tol, (mode, ntaps, channels, n)
        npts = int(rng.integers(1, min(n, 3000) + 1))
        got = ops.moving_average(xd, npts).cpu().numpy()
        assert rel_l2(got, orc.moving_average(x.astype(np.float64), npts)) <= 2e-6, (npts, n)


@pytest.mark.parametrize("seed", range(max(SEEDS * 2 // 3, 1)))
def test_fir_long_filters_and_random_chunking(ops, dev, orc, seed):
    """Tap counts around the block-length switches (2049 | 2050: 4096- to 16384-point blocks; 8193 | 8194: one pass to
    partitioned) one-shot, and streamed in random chunks where the streaming form exists (<= 8193 taps)."""
    import torch
    from dsp_amd import capi
    rng = np.random.default_rng(2500 + seed)
    for _ in range(3):
        ntaps = int(rng.choice([2048, 2049, 2050, 4096, 4097, 8192, 8193, 8194, int(rng.integers(2050, 8194)),
                                int(rng.integers(8194, 30000))]))
        channels, n = int(rng.integers(1, 4)), int(rng.integers(1, 60000))
        h = (rng.standard_normal(ntaps) / np.sqrt(ntaps)).astype(np.float32)
        x = rng.standard_normal((channels, n)).astype(np.float32)
        want = orc.fir(x.astype(np.float64), h.astype(np.float64))
        xd = torch.from_numpy(x).to(dev)
        for mode in ("overlap_save", "auto"):
            got = ops.convolve(xd, h, mode=mode).cpu().numpy()
            assert rel_l2(got, want) <= 1.4e-4 or not want.any(), (mode, ntaps, channels, n)
            assert not got[:, : ntaps - 1].any()
        if ntaps > 8193:
            with pytest.raises(capi.DspGpuError):
                ops.convolve_stream(xd, h, torch.zeros(channels, ntaps - 1, device=dev), 0, mode="overlap_save")
            continue
        hist = torch.zeros(channels, ntaps - 1, device=dev)
        pos, outs = 0, []
        while pos < n:
            step = int(min(n - pos, rng.choice([1, 7, 100, 4096, 9000, int(rng.integers(1, 30000))])))
            outs.append(ops.convolve_stream(xd[:, pos:pos + step].contiguous(), h, hist, pos, mode="overlap_save"))
            pos += step
        got = torch.cat(outs, dim=1).cpu().numpy()
        assert rel_l2(got, want) <= 1.4e-4 or not want.any(), ("stream", ntaps, channels, n)


@pytest.mark.parametrize("seed", range(SEEDS))
def test_iir_random_shapes(ops, dev, orc, seed):
    import torch
    rng = np.random.default_rng(3000 + seed)
    for _ in range(6):
        poles = int(rng.integers(1, 9))
        a, b = _stable_section(rng, poles)
        c0, d0 = (1.0, 0.0) if rng.integers(0, 2) else (float(rng.uniform(-2, 2)), float(rng.uniform(-1, 1)))
        channels, n = int(rng.integers(1, 200)), int(rng.integers(1, 12000))
        x = rng.standard_normal((channels, n)).astype(np.float32)
        st0 = rng.standard_normal((channels, 2 * poles)) * 0.3
        st0[:, :poles] = st0[:, :poles].astype(np.float32)
        so = np.zeros((channels, 2 * (poles + 1)))
        so[:, 1:poles + 1] = st0[:, :poles]
        so[:, poles + 2:] = st0[:, poles:]
        want = orc.iir(x.astype(np.float64), a, b, c0, d0, state=so)
        for mode in ("batch", "scan", "auto"):
            st = torch.from_numpy(st0.copy()).to(dev)
            got = ops.iir(torch.from_numpy(x).to(dev), a, b, c0, d0, state=st, mode=mode).cpu().numpy()
            assert rel_l2(got, want) <= 2e-6, (mode, poles, channels, n)
            g = st.cpu().numpy()
            assert np.allclose(g[:, :poles], so[:, 1:poles + 1])
            assert np.allclose(g[:, poles:], so[:, poles + 2:], rtol=1e-6, atol=1e-9), (mode, poles, channels, n)


@pytest.mark.parametrize("seed", range(SEEDS))
def test_double_sample_filters_random_shapes(ops, dev, orc, seed):
    """SampleBuffer<double> forms (dsp_gpu_convolve_f64 / moving_average_f64 / iir_f64): ragged shapes against the oracle's
    double functions at double tolerances (sums in another order: a few ulp; the recursion amplifies them a little)."""
    import torch
    rng = np.random.default_rng(4000 + seed)
    for _ in range(5):
        channels, n = int(rng.integers(1, 40)), int(rng.integers(1, 9000))
        x = rng.standard_normal((channels, n))
        xd = torch.from_numpy(x).to(dev)
        ntaps = int(rng.integers(1, 600))
        h = rng.standard_normal(ntaps) / np.sqrt(ntaps)
        want = orc.fir(x, h)
        got = ops.convolve_f64(xd, h, "direct").cpu().numpy()
        assert rel_l2(got, want) <= 1e-13 or not want.any(), ("fir", channels, n, ntaps)
        assert not got[:, : min(ntaps - 1, n)].any()
        pts = int(rng.integers(1, min(1200, n) + 1))  # n < npoints is refused (the reference reads out of bounds there)
        want = orc.moving_average(x, pts)
        got = ops.moving_average_f64(xd, pts).cpu().numpy()
        assert rel_l2(got, want) <= 1e-11 or not want.any(), ("mavg", channels, n, pts)
        poles = int(rng.integers(1, 9))
        a, b = _stable_section(rng, poles)
        want = orc.iir(x, a, b)
        for mode in ("batch", "auto"):
            got = ops.iir_f64(xd, a, b, mode=mode).cpu().numpy()
            assert rel_l2(got, want) <= 1e-11 or not want.any(), ("iir", mode, poles, channels, n)
        # the reference's own arithmetic: bit for bit (two poles = Coefficients<2>, the BiQuad specialisation: coefficients.h:172-182)
        want_ref = orc.biquad(x, a, b) if poles == 2 else want
        assert np.array_equal(ops.iir_f64(xd, a, b, mode="batch_ref").cpu().numpy(), want_ref), ("iir ref", poles, channels, n)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(SEEDS))
def test_moving_average_long_channels_random_windows(ops, dev, orc, seed):
    """Channels long enough for interior tiles (the path with the chunk-prefix differences formed per chunk and carry bit) with
    windows of every size and parity up to the limit, lengths of every alignment, both sample types."""
    import torch
    rng = np.random.default_rng(5000 + seed)
    for _ in range(4):
        channels, n = int(rng.integers(1, 4)), int(rng.integers(5000, 60000))
        pts = int(rng.integers(1, min(8192, n) + 1))
        if rng.integers(0, 2):
            pts = int(rng.integers(1, 700))
        x = rng.standard_normal((channels, n)) + float(rng.integers(0, 2))  # sometimes a DC offset: long prefix sums
        want = orc.moving_average(x, pts)
        got = ops.moving_average_f64(torch.from_numpy(x).to(dev), pts).cpu().numpy()
        assert rel_l2(got, want) <= 1e-11, ("double", channels, n, pts)
        x32 = x.astype(np.float32)
        got = ops.moving_average(torch.from_numpy(x32).to(dev), pts).cpu().numpy()
        assert rel_l2(got, orc.moving_average(x32.astype(np.float64), pts)) <= 2e-6, ("float", channels, n, pts)
