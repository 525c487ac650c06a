"""GPU parity of FIR convolution (dsp_gpu_convolve) and the moving average through the C ABI.

Tolerances (BASELINE.json north_star), relative L2 against the reference run on the same
float-exact inputs in double: direct <= 1e-6; overlap-save is FFT-class, <= 1e-5*log2(4096).
"""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu

DIRECT_TOL = 1e-6
OS_TOL = 1e-5 * 12


@pytest.fixture(scope="module")
def ops(dev):
    from dsp_amd import ops as _ops
    return _ops


def test_kat_alignment(ops, dev, golden):
    """SincFilter{0.14,10}.apply(delta[n-15]): zeros for j<10, kernel at j=15..25."""
    import torch
    x = golden["fir_kat_in"]
    h = golden["sinc_0p14_10"]
    for mode in ("direct", "overlap_save"):
        got = ops.convolve(torch.from_numpy(x).to(dev), h, mode).cpu().numpy()
        assert not got[:10].any()
        assert np.allclose(got, golden["fir_kat_out"], atol=2e-6 if mode == "overlap_save" else 1e-7)


@pytest.mark.parametrize("mode,tol", [("direct", DIRECT_TOL), ("overlap_save", OS_TOL), ("auto", OS_TOL)])
def test_golden_noise_511(ops, dev, golden, mode, tol):
    import torch
    x = golden["fir_noise_in"]
    got = ops.convolve(torch.from_numpy(x).to(dev), golden["sinc_0p1_511"], mode).cpu().numpy()
    assert not got[:510].any()  # filter.h:133: first M outputs stay zero
    assert rel_l2(got, golden["fir_noise_511_f64"]) <= tol
    # the reference's own float instantiation rounds at every tap; we must be at least as close to exact
    assert rel_l2(got, golden["fir_noise_511_f32"].astype(np.float64)) <= tol + 5e-7
    assert np.array_equal(got, ops.convolve(x, golden["sinc_0p1_511"], mode))  # host-pointer path


def test_golden_highpass_and_tones(ops, dev, golden):
    import torch
    x = golden["fir_noise_in"]
    got = ops.convolve(torch.from_numpy(x).to(dev), golden["sinc_0p2_64_hamming_inv"], "direct").cpu().numpy()
    want = golden["fir_noise_hp64_f32"].astype(np.float64)
    assert rel_l2(got, want) <= 2e-6  # vs the float reference (its own rounding included)
    tone = golden["fir_tone_in"].astype(np.float32)
    for mode, tol in (("direct", 2e-6), ("overlap_save", OS_TOL)):
        got = ops.convolve(torch.from_numpy(tone).to(dev), golden["sinc_0p14_200"], mode).cpu().numpy()
        assert rel_l2(got, golden["fir_tone_out_f64"]) <= tol


@pytest.mark.parametrize("channels,n,ntaps", [(1, 64, 1), (3, 100, 11), (2, 2000, 201), (5, 8192, 511), (2, 8193, 511),
                                              (1, 20, 21), (1, 10, 21), (3, 30001, 64), (2, 16384 + 7, 4096),
                                              (7, 4099, 33), (1, 70000, 1023), (4, 12288, 2049)])
def test_direct_vs_oracle(ops, dev, orc, channels, n, ntaps):
    import torch
    rng = np.random.default_rng(n * 7 + ntaps)
    h = rng.standard_normal(ntaps) / np.sqrt(ntaps) if ntaps in (1, 64, 33) else orc.sinc(0.2, ntaps)[:ntaps]
    x = rng.standard_normal((channels, n)).astype(np.float32)
    got = ops.convolve(torch.from_numpy(x).to(dev), h, "direct").cpu().numpy()
    want = orc.fir(x.astype(np.float64), h.astype(np.float32).astype(np.float64))  # the device holds fp32 taps
    assert got.shape == want.shape
    if n >= ntaps:
        assert rel_l2(got, want) <= DIRECT_TOL
    assert not got[:, : min(n, ntaps - 1)].any()
    # taps rounded to fp32 stay within the budget of the double-tap reference too
    if n >= ntaps and ntaps > 1:
        assert rel_l2(got, orc.fir(x.astype(np.float64), h)) <= DIRECT_TOL


@pytest.mark.parametrize("channels,n,ntaps", [(1, 4096, 511), (3, 100, 11), (2, 20000, 511), (2, 3586 * 2 + 510, 511),
                                              (2, 3586 * 2 + 511, 511), (1, 20, 21), (3, 30001, 64), (2, 50000, 2049),
                                              (5, 9000, 2), (1, 70000, 1023)])
def test_overlap_save_vs_oracle(ops, dev, orc, channels, n, ntaps):
    import torch
    rng = np.random.default_rng(n * 3 + ntaps)
    h = orc.sinc(0.15, ntaps)[:ntaps]
    x = rng.standard_normal((channels, n)).astype(np.float32)
    got = ops.convolve(torch.from_numpy(x).to(dev), h, "overlap_save").cpu().numpy()
    want = orc.fir(x.astype(np.float64), h)
    if n >= ntaps:
        assert rel_l2(got, want) <= OS_TOL
    assert not got[:, : min(n, ntaps - 1)].any()


def test_output_guard(ops, dev, orc):
    """Nothing is written outside [ch][n] for ragged sizes."""
    import torch
    rng = np.random.default_rng(3)
    h = orc.sinc(0.1, 101)
    for mode in ("direct", "overlap_save"):
        x = rng.standard_normal((3, 5001)).astype(np.float32)
        guard = torch.full((5, 5001), 9.0, device=dev)
        ops.convolve(torch.from_numpy(x).to(dev), h, mode, out=guard[1:4])
        g = guard.cpu().numpy()
        assert (g[0] == 9).all() and (g[4] == 9).all()
        assert rel_l2(g[1:4], orc.fir(x.astype(np.float64), h)) <= OS_TOL


def test_c3_full_size_properties(ops, dev):
    """BASELINE config 3 at full size (1,024 ch x 2^20, 511 taps): direct vs overlap-save agree, linearity,
    impulse response reproduces the taps, DC gain 1."""
    import torch
    from dsp_amd import design
    h = design.sinc(0.1, 511)
    g = torch.Generator(device=dev).manual_seed(4321)
    C, N = 1024, 1 << 20
    x = torch.randn(C, N, device=dev, generator=g)
    yd = ops.convolve(x, h, "direct")
    yo = ops.convolve(x, h, "overlap_save")
    den = torch.linalg.vector_norm(yd.double())
    assert float(torch.linalg.vector_norm((yd - yo).double()) / den) <= OS_TOL
    assert float(yd[:, :510].abs().max()) == 0.0 and float(yo[:, :510].abs().max()) == 0.0
    # linearity on a slice
    a, b = x[:8], x[8:16]
    lhs = ops.convolve(1.5 * a - 0.25 * b, h, "direct")
    rhs = 1.5 * yd[:8] - 0.25 * yd[8:16]
    assert float(torch.linalg.vector_norm((lhs - rhs).double()) / torch.linalg.vector_norm(rhs.double())) <= 5e-7
    # impulse -> taps (causal alignment), constant -> DC gain 1 after the transient
    imp = torch.zeros(2, N, device=dev)
    imp[0, 1000] = 1.0
    imp[1, :] = 1.0
    for mode, tol in (("direct", 1e-7), ("overlap_save", 5e-6)):
        r = ops.convolve(imp, h, mode).cpu().numpy()
        assert np.abs(r[0, 1000:1511] - h.astype(np.float32)).max() <= tol
        assert np.abs(r[0, 1511:]).max() <= tol and np.abs(r[0, :1000]).max() <= tol
        assert np.abs(r[1, 510:] - 1.0).max() <= 20 * tol + 2e-6


# ---- moving average ---------------------------------------------------------------------------
def test_mavg_kat(ops, dev, golden):
    import torch
    got = ops.moving_average(torch.arange(16, dtype=torch.float32, device=dev), 5).cpu().numpy()
    assert np.array_equal(got, golden["mavg_kat_out"])


@pytest.mark.parametrize("channels,n,pts", [(1, 8192, 511), (1, 8192, 8), (3, 1000, 511), (2, 511, 511), (2, 64, 1),
                                            (2, 30000, 4096), (2, 30000, 4097), (1, 30000, 8192), (3, 9001, 100), (1, 8, 8), (1, 9, 8),
                                            (2, 5000, 2), (2, 5000, 3)])
def test_mavg_vs_oracle(ops, dev, orc, channels, n, pts):
    import torch
    rng = np.random.default_rng(n + pts)
    x = rng.standard_normal((channels, n)).astype(np.float32)
    got = ops.moving_average(torch.from_numpy(x).to(dev), pts).cpu().numpy()
    want = orc.moving_average(x.astype(np.float64), pts)  # even lengths included: numPoints() does not re-force odd
    assert rel_l2(got, want) <= 1e-6, (n, pts)
    mid = pts // 2
    assert not got[:, :mid].any()
    if n - mid > mid + 1:
        assert not got[:, n - mid:].any()


@pytest.mark.parametrize("channels,n,pts", [(2, 4096 * 40 + 1236, 511), (3, 1 << 18, 8), (1, 4096 * 30, 2043), (2, 4096 * 25 + 4, 100),
                                            (1, 4096 * 9 + 2048, 1), (5, 4096 * 21, 1024), (1, 1 << 20, 511)])
def test_mavg_sequential_kernel(ops, dev, orc, channels, n, pts, monkeypatch):
    """Channels long enough for moving_average_seq_kernel (runs of consecutive interior tiles, the shared samples carried in
    LDS; the first and last tiles of a channel on the one-tile kernel): runs of 16 tiles and shorter last runs, odd and even
    windows, a window of one sample, the largest window it takes, ragged channel ends."""
    import torch
    monkeypatch.setenv("DSP_GPU_MAVG_SEQ", "1")  # opt-in kernel (fewer bytes, slower: DESIGN.md 4.4)
    rng = np.random.default_rng(n + pts)
    x = rng.standard_normal((channels, n)).astype(np.float32)
    guard = torch.full((channels + 2, n), 7.0, device=dev)
    ops.moving_average(torch.from_numpy(x).to(dev), pts, out=guard[1:-1])
    g = guard.cpu().numpy()
    assert (g[0] == 7).all() and (g[-1] == 7).all()
    want = orc.moving_average(x.astype(np.float64), pts)
    assert rel_l2(g[1:-1], want) <= 1e-6, (n, pts)
    # tile by tile too: a carry that went wrong shows in one tile of a run, which an L2 norm over a megasample would forgive
    for lo in range(0, n - 4096, 4096):
        assert rel_l2(g[1:-1, lo:lo + 4096], want[:, lo:lo + 4096]) <= 2e-6 or not want[:, lo:lo + 4096].any(), lo
    mid = pts // 2
    assert not g[1:-1, :mid].any() and not g[1:-1, n - mid:].any()


def test_mavg_golden(ops, dev, golden):
    import torch
    x = golden["fir_noise_in"]
    xd = torch.from_numpy(x).to(dev)
    assert rel_l2(ops.moving_average(xd, 511).cpu().numpy(), golden["mavg_noise_511_f64"]) <= 1e-6
    # the reference's float running sum drifts; its double instantiation is the yardstick,
    # the float one only has to be within its own drift
    assert rel_l2(ops.moving_average(xd, 511).cpu().numpy(), golden["mavg_noise_511_f32"].astype(np.float64)) <= 1e-4
    got8 = ops.moving_average(xd, 8).cpu().numpy()
    assert rel_l2(got8, golden["mavg_noise_even8_f32"].astype(np.float64)) <= 1e-5
    assert np.array_equal(got8, ops.moving_average(x, 8))


def test_errors(ops, dev):
    import torch
    from dsp_amd import capi
    x = torch.zeros(2, 100, device=dev)
    with pytest.raises(capi.DspGpuError):
        ops.convolve(x, np.zeros((1 << 18) + 1), "overlap_save")  # beyond the partitioned limit
    with pytest.raises(capi.DspGpuError):  # no streaming variant of the partitioned form (more than 8193 taps)
        ops.convolve_stream(x, np.zeros(9000), torch.zeros(2, 8999, device=dev), 0, mode="overlap_save")
    with pytest.raises(capi.DspGpuError):
        ops.convolve(x, np.zeros(5000), "direct")
    with pytest.raises(capi.DspGpuError):
        ops.moving_average(x, 101)


# ---- streaming FIR (carried input tail) ---------------------------------------------------------------
@pytest.mark.parametrize("mode,tol", [("direct", DIRECT_TOL), ("overlap_save", OS_TOL)])
@pytest.mark.parametrize("ntaps,chunks", [(511, [4096, 4096, 4096]), (511, [100, 7000, 3, 511, 510, 5000]),
                                          (33, [1, 1, 30, 2000]), (2049, [3000, 50, 9000]), (1, [10, 20]),
                                          (3000, [5000, 100, 9000, 20000])])
def test_streaming_equals_one_shot(ops, dev, orc, mode, tol, ntaps, chunks):
    """Chunks of any size with the input tail carried == the one-shot call on the whole signal
    (the FIR counterpart of the IIR state; block-wise callers: eval/recursive_filters/main.cpp:174-206)."""
    import torch
    rng = np.random.default_rng(ntaps + len(chunks))
    h = orc.sinc(0.2, ntaps)[:ntaps] if ntaps > 1 else np.array([0.75])
    n = sum(chunks)
    x = rng.standard_normal((3, n)).astype(np.float32)
    want = orc.fir(x.astype(np.float64), h)
    xd = torch.from_numpy(x).to(dev)
    hist = torch.full((3, ntaps - 1), 123.0, device=dev)  # garbage: must be ignored on the first chunk
    outs, pos = [], 0
    for i, c in enumerate(chunks):
        outs.append(ops.convolve_stream(xd[:, pos:pos + c].contiguous(), h, hist, pos, mode=mode))
        pos += c
    got = torch.cat(outs, dim=1).cpu().numpy()
    assert rel_l2(got, want) <= tol
    assert not got[:, : min(n, ntaps - 1)].any()
    if ntaps > 1:  # the carried tail is the last ntaps-1 input samples
        tail = np.concatenate([np.zeros((3, max(0, ntaps - 1 - n)), np.float32), x[:, max(0, n - (ntaps - 1)):]], axis=1)
        assert np.array_equal(hist.cpu().numpy(), tail)
    # host-pointer path
    hh = np.zeros((3, ntaps - 1), dtype=np.float32)
    outs, pos = [], 0
    for i, c in enumerate(chunks):
        outs.append(ops.convolve_stream(np.ascontiguousarray(x[:, pos:pos + c]), h, hh, pos, mode=mode))
        pos += c
    assert np.array_equal(np.concatenate(outs, axis=1), got)


# ---- the VALU direct form kept as the matrix-core kernel's A/B partner ---------------------------------
@pytest.mark.parametrize("ntaps,channels,n", [(511, 3, 20000), (1, 2, 100), (20, 1, 4096), (33, 5, 4097), (1024, 2, 9000), (101, 2, 50)])
def test_direct_valu_matches(ops, dev, orc, ntaps, channels, n):
    """Same results as the MFMA kernel to the direct-FIR tolerance (north_star: <= 1e-6 against the reference)."""
    import torch
    rng = np.random.default_rng(ntaps * 7 + n)
    h = orc.sinc(0.15, ntaps)[:ntaps] if ntaps > 1 else np.array([1.25])
    x = rng.standard_normal((channels, n)).astype(np.float32)
    want = orc.fir(x.astype(np.float64), h)
    guard = torch.full((channels + 2, n), 7.0, device=dev)
    ops.convolve(torch.from_numpy(x).to(dev), h, mode="direct_valu", out=guard[1:-1])
    g = guard.cpu().numpy()
    assert (g[0] == 7).all() and (g[-1] == 7).all()
    assert rel_l2(g[1:-1], want) <= DIRECT_TOL
    assert not g[1:-1][:, : min(n, ntaps - 1)].any()
    assert rel_l2(ops.convolve(x, h, mode="direct_valu"), want) <= DIRECT_TOL  # host pointers


def test_more_channels_than_one_grid(ops, dev, orc):
    """70,000 short channels: the channel axis rides grid.y (16 bits), so the launchers cut it into slices."""
    import torch
    rng = np.random.default_rng(70)
    channels, n = 70000, 96
    x = rng.standard_normal((channels, n)).astype(np.float32)
    h = orc.sinc(0.2, 9)[:9]
    xd = torch.from_numpy(x).to(dev)
    pick = [0, 1, 65534, 65535, 65536, 69999]
    y = ops.convolve(xd, h, mode="direct").cpu().numpy()
    assert rel_l2(y[pick], orc.fir(x[pick].astype(np.float64), h)) <= DIRECT_TOL
    m = ops.moving_average(xd, 7).cpu().numpy()
    assert rel_l2(m[pick], orc.moving_average(x[pick].astype(np.float64), 7)) <= 1e-6
    a, b = orc.chebyshev(0.2, 2)
    st = torch.zeros(channels, 4, dtype=torch.float64, device=dev)
    s1 = ops.iir(xd, a, b, state=st, mode="scan").cpu().numpy()
    assert rel_l2(s1[pick], orc.iir(x[pick].astype(np.float64), a, b)) <= 2e-6
    assert torch.equal(st, _state_after(ops, xd, a, b))


def _state_after(ops, xd, a, b):
    import torch
    st = torch.zeros(xd.shape[0], 4, dtype=torch.float64, device=xd.device)
    ops.iir(xd, a, b, state=st, mode="batch")
    return st


# ---- filters longer than one overlap-save block: partitioned ------------------------------------------
@pytest.mark.parametrize("ntaps,channels,n", [(2050, 2, 20000), (4097, 2, 16000), (6000, 1, 25000), (8193, 1, 40000),
                                              (8194, 1, 40000), (10001, 2, 12000), (20000, 1, 50000), (5000, 1, 3000),
                                              (3000, 2, 2500)])
def test_partitioned_long_filters(ops, dev, orc, ntaps, channels, n):
    """More taps than a 4096-point block holds (2049): 16384-point blocks up to 8193 taps, beyond that partitioned
    into 8192-tap segments, one overlap-save pass each, accumulated; the first ntaps-1 outputs stay zero as ever
    (filter.h:126,133).  Against the reference's direct loop."""
    import torch
    from dsp_amd import capi
    rng = np.random.default_rng(ntaps)
    h = (rng.standard_normal(ntaps) * np.exp(-np.arange(ntaps) / (ntaps / 4))).astype(np.float32)  # a decaying response
    x = rng.standard_normal((channels, n)).astype(np.float32)
    want = orc.fir(x.astype(np.float64), h.astype(np.float64))
    xd = torch.from_numpy(x).to(dev)
    guard = torch.full((channels + 2, n), 3.0, device=dev)
    ops.convolve(xd, h, mode="overlap_save", out=guard[1:-1])
    g = guard.cpu().numpy()
    assert (g[0] == 3).all() and (g[-1] == 3).all()
    assert rel_l2(g[1:-1], want) <= OS_TOL
    assert not g[1:-1][:, : min(n, ntaps - 1)].any()
    assert np.array_equal(ops.convolve(x, h, mode="overlap_save"), g[1:-1])       # host pointers
    if ntaps <= 4096:
        assert rel_l2(ops.convolve(xd, h, mode="direct").cpu().numpy(), want) <= DIRECT_TOL
        auto = ops.convolve(xd, h, mode="auto").cpu().numpy()                       # AUTO: partitioned from 4096 samples up
        assert rel_l2(auto, want) <= OS_TOL
    else:
        assert np.array_equal(ops.convolve(xd, h, mode="auto").cpu().numpy(), g[1:-1])
        with pytest.raises(capi.DspGpuError):
            ops.convolve(xd, h, mode="direct")


@pytest.mark.gpu
@pytest.mark.parametrize("n,ntaps,channels", [(256, 31, 19), (256, 255, 5), (256, 256, 5), (200, 100, 3), (512, 127, 11), (300, 63, 7), (1024, 511, 7), (1024, 1023, 3), (1024, 1024, 2), (2048, 511, 5),
                                              (3000, 511, 9), (3000, 63, 1), (1500, 1025, 4), (514 + 510, 511, 33),
                                              (515 + 510, 511, 33)])
def test_overlap_save_short_channels(ops, dev, orc, n, ntaps, channels):
    """Channels that fit one 4096-point block run on 1024- / 2048-point blocks, several transforms to a workgroup, pairs
    taken across channels (odd channel counts leave transforms of the last workgroup idle)."""
    import torch
    rng = np.random.default_rng(n * 7 + ntaps)
    h = (rng.standard_normal(ntaps) / np.sqrt(ntaps)).astype(np.float32)
    x = rng.standard_normal((channels, n)).astype(np.float32)
    want = orc.fir(x.astype(np.float64), h.astype(np.float64))
    guard = torch.full((channels + 2, n), 3.0, device=dev)
    ops.convolve(torch.from_numpy(x).to(dev), h, mode="overlap_save", out=guard[1:-1])
    g = guard.cpu().numpy()
    assert (g[0] == 3).all() and (g[-1] == 3).all()
    assert rel_l2(g[1:-1], want) <= OS_TOL
    assert not g[1:-1][:, : ntaps - 1].any()
    # the same signal streamed in 256-sample blocks (an audio callback): one small block per call
    hist = torch.zeros(channels, ntaps - 1, device=dev)
    xd = torch.from_numpy(x).to(dev)
    outs = [ops.convolve_stream(xd[:, p:p + 256].contiguous(), h, hist, p, mode="overlap_save") for p in range(0, n, 256)]
    assert rel_l2(torch.cat(outs, dim=1).cpu().numpy(), want) <= OS_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("n,ntaps,channels", [(1024, 511, 7), (1000, 63, 9), (100, 31, 20), (2048, 511, 3), (2049, 200, 5),
                                              (4096, 1023, 3), (4097, 511, 3), (1024, 4096, 5), (3000, 2500, 2), (1, 1, 70000)])
def test_direct_short_channels(ops, dev, orc, n, ntaps, channels):
    """Channels of at most 4096 samples go several to a workgroup in the direct (MFMA) kernel, each with its own history
    in front; one-shot and streamed in 256-sample blocks."""
    import torch
    rng = np.random.default_rng(n * 13 + ntaps)
    h = (rng.standard_normal(ntaps) / np.sqrt(ntaps)).astype(np.float32)
    x = rng.standard_normal((channels, n)).astype(np.float32)
    want = orc.fir(x.astype(np.float64), h.astype(np.float64))
    guard = torch.full((channels + 2, n), 3.0, device=dev)
    xd = torch.from_numpy(x).to(dev)
    ops.convolve(xd, h, mode="direct", out=guard[1:-1])
    g = guard.cpu().numpy()
    assert (g[0] == 3).all() and (g[-1] == 3).all()
    assert rel_l2(g[1:-1], want) <= DIRECT_TOL or not want.any()
    assert not g[1:-1][:, : ntaps - 1].any()
    if channels <= 100:
        hist = torch.zeros(channels, ntaps - 1, device=dev)
        outs = [ops.convolve_stream(xd[:, p:p + 256].contiguous(), h, hist, p, mode="direct") for p in range(0, n, 256)]
        assert rel_l2(torch.cat(outs, dim=1).cpu().numpy(), want) <= DIRECT_TOL or not want.any()


@pytest.mark.gpu
@pytest.mark.parametrize("channels,n,ntaps", [(6, 40000, 511), (3, 65536, 255), (2, 100000, 1025), (5, 30720, 64),
                                              (1, 3584 * 9 + 4096, 511), (4, 20480, 2049), (7, 12288, 300), (2, 50000, 1800),
                                              (3, 40000, 700), (2, 60000, 1200), (2, 60000, 1500), (2, 60000, 1700)])
def test_overlap_save_wave_kernel(ops, dev, orc, channels, n, ntaps, monkeypatch):
    """The one-wave-per-block-pair kernel (long signals by default) forced onto small inputs: against the oracle and
    against the workgroup kernel, incl. pairs that straddle channels, first / last blocks (edge pairs) and hops
    that do not round to a multiple of 256 (those calls fall back to the workgroup kernel by themselves)."""
    import torch
    rng = np.random.default_rng(ntaps + n)
    h = rng.standard_normal(ntaps) / np.sqrt(ntaps)
    x = rng.standard_normal((channels, n)).astype(np.float32)
    xd = torch.from_numpy(x).to(dev)
    want = orc.fir(x.astype(np.float64), h)
    monkeypatch.setenv("DSP_GPU_OS_WAVE_MIN_TILES", "1")
    guard = torch.full((channels + 2, n), 9.0, device=dev)
    ops.convolve(xd, h, "overlap_save", out=guard[1:-1])
    g = guard.cpu().numpy()
    assert (g[0] == 9).all() and (g[-1] == 9).all()
    assert rel_l2(g[1:-1], want) <= OS_TOL
    assert (g[1:-1, :ntaps - 1] == 0).all()
    # the two forms of the wave kernel (round 2: one wave per SIMD, prefetch into AGPRs; round 3, the default: two waves
    # per SIMD, QUAD transpose, runs of consecutive pairs that keep the shared samples in registers) give the same bits --
    # with runs of one pair (what inputs this small get) and with runs of 2, 3 and 7 pairs (kept rows in use)
    monkeypatch.setenv("DSP_GPU_OS_WAVE_FORM", "agpr")
    old_form = ops.convolve(xd, h, "overlap_save").cpu().numpy()
    monkeypatch.delenv("DSP_GPU_OS_WAVE_FORM")
    assert np.array_equal(g[1:-1], old_form)
    for run in (2, 3, 7):
        monkeypatch.setenv("DSP_GPU_OS_WAVE_RUN", str(run))
        assert np.array_equal(ops.convolve(xd, h, "overlap_save").cpu().numpy(), old_form), run
    monkeypatch.delenv("DSP_GPU_OS_WAVE_RUN")
    # the default above DRAWS its pairs one at a time from a counter per XCD; the same with 2, 3 and 5 consecutive pairs per
    # draw (kept rows in use inside a draw), twice in a row (the counters are cleared in front of every launch), and with the
    # pairs dealt out in advance instead (DSP_GPU_OS_WAVE_DRAW=0)
    for draw in (2, 3, 5, 0, 1):
        monkeypatch.setenv("DSP_GPU_OS_WAVE_DRAW", str(draw))
        for _ in range(2):
            assert np.array_equal(ops.convolve(xd, h, "overlap_save").cpu().numpy(), old_form), draw
    monkeypatch.delenv("DSP_GPU_OS_WAVE_DRAW")
    monkeypatch.setenv("DSP_GPU_OS_WAVE_MIN_TILES", "999999999")
    other = ops.convolve(xd, h, "overlap_save").cpu().numpy()
    assert rel_l2(g[1:-1], other) <= 5e-6


@pytest.mark.gpu
def test_tap_cache_many_sets_without_device_sync(ops, dev, orc):
    """300 distinct tap sets streamed through dsp_gpu_convolve_stream (more than the 256 cache slots: the least
    recently used slots are taken over).  While they run a second stream is kept busy for much longer: were any
    call to synchronise the whole device (the old flush-all did), it would return only after that stream had
    drained.  Results are checked for the first, a middle and the last set, and for a re-used early one."""
    import time
    import torch
    rng = np.random.default_rng(300)
    ntaps, channels, n = 63, 4, 2048
    sets = [rng.standard_normal(ntaps) / 8 for _ in range(300)]
    x = rng.standard_normal((channels, n)).astype(np.float32)
    xd = torch.from_numpy(x).to(dev)
    hist = torch.zeros(channels, ntaps - 1, device=dev)
    ops.convolve_stream(xd, sets[0], hist, 0, "overlap_save")  # warm up: kernels loaded
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    a = torch.randn(8192, 8192, device=dev)
    done = torch.cuda.Event()
    with torch.cuda.stream(side):
        for _ in range(150):  # >= 1 s of matrix products
            a = (a @ a) * 1e-4
        done.record(side)
    t0 = time.perf_counter()
    outs = {}
    for k, h in enumerate(sets):
        hist.zero_()
        y = ops.convolve_stream(xd, h, hist, 0, "overlap_save" if k % 2 else "direct")
        if k in (0, 150, 299):
            outs[k] = y
    y_again = ops.convolve_stream(xd, sets[3], torch.zeros_like(hist), 0, "direct")  # evicted by now: uploaded again
    host_s = time.perf_counter() - t0
    still_busy = not done.query()
    torch.cuda.synchronize()
    assert still_busy, f"the side stream drained while 300 tap sets were issued ({host_s:.3f} s): a device-wide sync?"
    for k, y in outs.items():
        assert rel_l2(y.cpu().numpy(), orc.fir(x.astype(np.float64), sets[k])) <= OS_TOL
    assert rel_l2(y_again.cpu().numpy(), orc.fir(x.astype(np.float64), sets[3])) <= DIRECT_TOL


@pytest.mark.gpu
def test_out_argument_is_validated(ops, dev):
    import torch
    from dsp_amd import capi
    x = torch.zeros(4, 1024, device=dev)
    h = np.ones(5)
    for bad in (torch.empty(4, 1000, device=dev), torch.empty(4, 1024, device=dev, dtype=torch.float64),
                torch.empty(4, 2048, device=dev)[:, ::2], np.empty((4, 1024), np.float32)):
        with pytest.raises(capi.DspGpuError):
            ops.convolve(x, h, "direct", out=bad)
        with pytest.raises(capi.DspGpuError):
            ops.moving_average(x, 5, out=bad)
        with pytest.raises(capi.DspGpuError):
            ops.iir(x, [0.5, 0.0], [0.0, 0.5], out=bad)
    with pytest.raises(capi.DspGpuError):
        ops.fft0(torch.zeros(4, 1024, dtype=torch.complex64, device=dev), out=torch.empty(4, 512, dtype=torch.complex64, device=dev))


@pytest.mark.gpu
def test_moving_average_tile_edge_shapes(ops, dev, orc):
    """Channel lengths that are exact multiples of a tile (256 * 2^k outputs) or a few samples off, windows of 1 ... 257 points of
    both parities (the window's length decides which chunk-prefix difference an output reads), both sample types."""
    import torch
    rng = np.random.default_rng(1)
    for n in (256, 512, 1024, 2048, 4096, 8192, 12288, 4100, 8190, 16384 + 4):
        for pts in (1, 2, 3, 4, 15, 16, 17, 31, 32, 33, 255, 256, 257):
            if pts > n:
                continue
            x = rng.standard_normal((3, n)) + 1.0
            want = orc.moving_average(x, pts)
            got = ops.moving_average_f64(torch.from_numpy(x).to(dev), pts).cpu().numpy()
            assert rel_l2(got, want) <= 1e-11, ("double", n, pts)
            x32 = x.astype(np.float32)
            got = ops.moving_average(torch.from_numpy(x32).to(dev), pts).cpu().numpy()
            assert rel_l2(got, orc.moving_average(x32.astype(np.float64), pts)) <= 2e-6, ("float", n, pts)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [17, 64, 100, 333, 1000, 1024, 1535, 2048, 4500, 6144])
def test_moving_average_many_short_channels(ops, dev, orc, n, monkeypatch):
    """Many short channels run as one signal of channels * n samples in full tiles (fir_kernels.hip, PERIODIC): tiles that
    start and end inside channels, channel lengths that are not multiples of four, windows of both parities up to the channel
    length, exact zeros at every channel's edges, nothing written around the batch; against the oracle and against the
    one-workgroup-per-channel path (DSP_GPU_MAVG_NO_FLATTEN=1), both sample types."""
    import torch
    rng = np.random.default_rng(n)
    channels = 3 * 4096 // n + 5
    x = rng.standard_normal((channels, n)) + 0.5
    for pts in sorted({1, 2, 3, 8, 15, 16, min(255, n), min(256, n - 1), n - 2, n - 1, n}):
        if pts < 1:
            continue
        want = orc.moving_average(x, pts)
        mid = pts // 2
        for f64 in (True, False):
            xs = x if f64 else x.astype(np.float32)
            fn = ops.moving_average_f64 if f64 else ops.moving_average
            guard = torch.full((channels + 2, n), 7.0, device=dev, dtype=torch.float64 if f64 else torch.float32)
            fn(torch.from_numpy(xs).to(dev), pts, out=guard[1:-1])
            g = guard.cpu().numpy()
            assert (g[0] == 7).all() and (g[-1] == 7).all(), (n, pts, f64)
            got = g[1:-1]
            ref = want if f64 else orc.moving_average(xs.astype(np.float64), pts)
            assert rel_l2(got, ref) <= (1e-11 if f64 else 2e-6), (n, pts, f64)
            if 2 * mid + 1 <= n:
                assert not got[:, :mid].any() and not got[:, n - mid:].any(), (n, pts, f64)
            monkeypatch.setenv("DSP_GPU_MAVG_NO_FLATTEN", "1")
            other = fn(torch.from_numpy(xs).to(dev), pts).cpu().numpy()
            monkeypatch.delenv("DSP_GPU_MAVG_NO_FLATTEN")
            assert rel_l2(got, other) <= (1e-12 if f64 else 5e-7), (n, pts, f64)  # (prefix sums with another origin)



@pytest.mark.gpu
def test_channels_shorter_than_the_filter_are_zero(ops, dev):
    """One-shot calls on channels of fewer samples than taps: no output has a full window behind it, all stay zero
    (filter.h:126,133) -- the library answers with a memset; every mode, both sample types, device and host pointers, the
    output buffer pre-filled, nothing written around it."""
    import torch
    rng = np.random.default_rng(5)
    h = rng.standard_normal(511)
    for n in (1, 64, 510):
        x = rng.standard_normal((9, n)).astype(np.float32)
        for mode in ("direct", "overlap_save", "auto"):
            guard = torch.full((11, n), 3.0, device=dev)
            ops.convolve(torch.from_numpy(x).to(dev), h, mode, out=guard[1:-1])
            g = guard.cpu().numpy()
            assert (g[0] == 3).all() and (g[-1] == 3).all() and not g[1:-1].any(), (n, mode)
            assert not ops.convolve(x, h, mode).any(), (n, mode)  # host pointers
        guard = torch.full((11, n), 3.0, device=dev, dtype=torch.float64)
        ops.convolve_f64(torch.from_numpy(x.astype(np.float64)).to(dev), h, out=guard[1:-1])
        g = guard.cpu().numpy()
        assert (g[0] == 3).all() and (g[-1] == 3).all() and not g[1:-1].any(), n
        assert not ops.convolve_f64(x.astype(np.float64), h).any(), n
    # one sample more and the last output is the first with a full window
    x = rng.standard_normal((2, 511)).astype(np.float32)
    got = ops.convolve(torch.from_numpy(x).to(dev), h, "direct").cpu().numpy()
    assert not got[:, :510].any()
    want = (x.astype(np.float64)[:, ::-1] * h.astype(np.float32).astype(np.float64)).sum(axis=1)
    assert np.allclose(got[:, 510], want, rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("channels,n,ntaps", [(700, 64, 7), (333, 100, 31), (64, 256, 63), (50, 1000, 511), (9, 10000, 129),
                                              (3, 65535, 33), (5, 4099, 4096), (40, 511, 511)])
def test_direct_many_channels_as_one_signal(ops, dev, orc, channels, n, ntaps, monkeypatch):
    """One-shot direct FIR on many channels runs them as one signal (fir_kernels.hip, PERIODIC): the first ntaps - 1 outputs of
    every channel are exact zeros, nothing leaks across channel borders (each channel also filtered on its own), totals that
    are not multiples of four, nothing written around the batch; against the oracle and the per-channel path."""
    import torch
    rng = np.random.default_rng(channels + n + ntaps)
    h = rng.standard_normal(ntaps) / np.sqrt(ntaps)
    x = (rng.standard_normal((channels, n)) * 3 + 1).astype(np.float32)
    want = orc.fir(x.astype(np.float64), h.astype(np.float32).astype(np.float64))
    guard = torch.full((channels + 2, n), 9.0, device=dev)
    ops.convolve(torch.from_numpy(x).to(dev), h, "direct", out=guard[1:-1])
    g = guard.cpu().numpy()
    assert (g[0] == 9).all() and (g[-1] == 9).all()
    got = g[1:-1]
    assert rel_l2(got, want) <= DIRECT_TOL
    assert not got[:, : ntaps - 1].any()
    for c in (0, channels // 2, channels - 1):  # a channel alone: no neighbour to leak from
        alone = ops.convolve(torch.from_numpy(x[c:c + 1]).to(dev), h, "direct").cpu().numpy()
        assert rel_l2(got[c:c + 1], alone) <= 2 * DIRECT_TOL, c  # (two fp32 sums in different orders)
    monkeypatch.setenv("DSP_GPU_FIR_NO_FLATTEN", "1")
    other = ops.convolve(torch.from_numpy(x).to(dev), h, "direct").cpu().numpy()
    assert rel_l2(got, other) <= 2 * DIRECT_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("channels,n,ntaps", [(40, 6000, 511), (9, 10001, 129), (3, 65535, 33), (5, 4099, 2049), (17, 8192, 1023),
                                              (50, 3072, 511), (33, 1536, 200), (21, 2049, 1025), (12, 4096, 2049)])
def test_overlap_save_many_channels_as_one_signal(ops, dev, orc, channels, n, ntaps, monkeypatch):
    """One-shot overlap-save on several channels of 1,536 ... 262,143 samples runs them as one signal and then zeroes the first
    ntaps - 1 outputs of every channel (fir_kernels.hip, launch_os): exact zeros there, nothing leaked across channel borders,
    nothing written around the batch; against the oracle and the per-channel path (DSP_GPU_FIR_NO_FLATTEN=1)."""
    import torch
    rng = np.random.default_rng(channels * n + ntaps)
    h = rng.standard_normal(ntaps) / np.sqrt(ntaps)
    x = (rng.standard_normal((channels, n)) * 3 + 1).astype(np.float32)
    want = orc.fir(x.astype(np.float64), h)
    guard = torch.full((channels + 2, n), 9.0, device=dev)
    ops.convolve(torch.from_numpy(x).to(dev), h, "overlap_save", out=guard[1:-1])
    g = guard.cpu().numpy()
    assert (g[0] == 9).all() and (g[-1] == 9).all()
    got = g[1:-1]
    assert rel_l2(got, want) <= OS_TOL
    assert not got[:, : ntaps - 1].any()
    for c in (0, channels // 2, channels - 1):
        assert rel_l2(got[c], want[c]) <= OS_TOL, c
    monkeypatch.setenv("DSP_GPU_FIR_NO_FLATTEN", "1")
    other = ops.convolve(torch.from_numpy(x).to(dev), h, "overlap_save").cpu().numpy()
    assert rel_l2(got, other) <= 1e-5
