"""GPU parity of the IIR recursion (dsp_gpu_iir, batch and scan modes) through the C ABI.

Tolerance (BASELINE.json north_star): relative L2 <= 1e-4 against the reference (scan-reassociated);
the kernels compute in double like the reference, so the observed error is far smaller.
"""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-4
TIGHT = 2e-6  # what double arithmetic + fp32 output should reach against the double reference


@pytest.fixture(scope="module")
def ops(dev):
    from dsp_amd import ops as _ops
    return _ops


CHEBY = ["lp2", "lp4", "lp6", "hp4", "lp4_lo", "lp8"]


@pytest.mark.parametrize("tag", CHEBY)
@pytest.mark.parametrize("mode", ["batch", "scan"])
def test_golden_chebyshev(ops, dev, golden, tag, mode):
    import torch
    x = golden["iir_noise_in"]
    a, b = golden[f"cheby_{tag}_a"], golden[f"cheby_{tag}_b"]
    got = ops.iir(torch.from_numpy(x).to(dev), a, b, mode=mode).cpu().numpy()
    # The fast modes compute in double with fused multiply-adds: parity is against the reference's DOUBLE instantiation
    # on the same float-exact samples, to fp32 output rounding.  (fc = 0.01, `lp4_lo`: four poles clustered at z = 1; the
    # scan carries its states in double-double there.)
    assert rel_l2(got, golden[f"cheby_{tag}_y64"]) <= TIGHT
    # Against the reference's FLOAT instantiation (`_y32`, whose accumulator is rounded to float at every `+=`,
    # SURVEY.md Q12) the bar is north_star's 1e-4 with no allowance; the mode that reproduces that instantiation is
    # DSP_GPU_IIR_BATCH_REF, bit for bit for every fixture, lp4_lo included.  (The fast modes are inside 1e-4 of `_y32`
    # wherever the float instantiation is itself inside 1e-4 of the double one -- every fixture but lp4_lo, where the
    # reference's float result is 2.5e-3 from its own double result -- and that is asserted for them too.)
    ref = ops.iir(torch.from_numpy(x).to(dev), a, b, mode="batch_ref").cpu().numpy()
    assert np.array_equal(ref, golden[f"cheby_{tag}_y32"])
    if tag != "lp4_lo":
        assert rel_l2(got, golden[f"cheby_{tag}_y32"].astype(np.float64)) <= TOL
    assert np.array_equal(got, ops.iir(x, a, b, mode=mode))  # host-pointer path


@pytest.mark.parametrize("tag", ["bp", "br", "ls", "hs", "pk"])
@pytest.mark.parametrize("mode", ["batch", "scan"])
def test_golden_biquads(ops, dev, golden, tag, mode):
    """recursive_filters.h:165-263; the shelves and the peaking filter exercise c0/d0."""
    import torch
    x = golden["iir_noise_in"]
    a, b = golden[f"bq_{tag}_a"], golden[f"bq_{tag}_b"]
    c0, d0 = golden[f"bq_{tag}_cd"]
    got = ops.iir(torch.from_numpy(x).to(dev), a, b, c0, d0, mode=mode).cpu().numpy()
    assert rel_l2(got, golden[f"bq_{tag}_y64"]) <= TIGHT
    assert rel_l2(got, golden[f"bq_{tag}_y32"].astype(np.float64)) <= TOL


def test_impulse_kat(ops, dev, golden):
    import torch
    a, b = golden["cheby_lp4_a"], golden["cheby_lp4_b"]
    imp = torch.zeros(32, device=dev)
    imp[0] = 1
    got = ops.iir(imp, a, b, mode="batch").cpu().numpy()
    # SURVEY.md section 4 KAT (first six values), then the whole response in double by hand
    assert np.allclose(got[:6], [0.00278076, 0.01880912, 0.05998964, 0.12282682, 0.18527392, 0.22180125], rtol=1e-6)
    v = np.zeros(32 + 4)
    for t in range(32):
        v[t + 4] = (a[t] if t < 5 else 0.0) + sum(b[i] * v[t + 4 - i] for i in range(1, 5))
    assert np.allclose(got, v[4:], rtol=2e-7, atol=1e-9)
    # the fixture is the reference's float instantiation, which rounds every fed-back output to float
    assert np.allclose(got, golden["cheby_lp4_impulse32"], rtol=1e-5, atol=5e-7)


@pytest.mark.parametrize("mode", ["batch", "scan"])
def test_carried_state(ops, dev, orc, golden, mode):
    """Chunked == unchunked (coefficients.h:11-12 history made explicit), and the final history matches
    the reference's x[]/y[] arrays."""
    import torch
    a, b = golden["cheby_lp4_a"], golden["cheby_lp4_b"]
    x = golden["iir_noise_in"]
    xd = torch.from_numpy(x).to(dev)
    st = torch.zeros(1, 8, dtype=torch.float64, device=dev)
    y = torch.cat([ops.iir(xd[:1500], a, b, state=st, mode=mode), ops.iir(xd[1500:], a, b, state=st, mode=mode)])
    whole = ops.iir(xd, a, b, mode=mode)
    if mode == "batch":
        assert torch.equal(y, whole)
    assert rel_l2(y.cpu().numpy(), golden["cheby_lp4_y64"]) <= TIGHT
    # reference state layout: x[0..4], y[0..4] with x[0] = 0 scratch, y[1..4] = v[t-1..t-4]
    ref_state = golden["cheby_lp4_state_end"]
    got = st.cpu().numpy()[0]
    assert np.allclose(got[:4], ref_state[1:5], rtol=0, atol=0)          # x history: exact float values
    assert np.allclose(got[4:], ref_state[6:10], rtol=1e-5, atol=1e-7)   # v history (reference rounds v to float)
    # host-pointer path carries state too
    sth = np.zeros((1, 8))
    yh = np.concatenate([ops.iir(x[:1500], a, b, state=sth, mode=mode), ops.iir(x[1500:], a, b, state=sth, mode=mode)])
    assert np.array_equal(yh, y.cpu().numpy())
    assert np.array_equal(sth, st.cpu().numpy())


@pytest.mark.parametrize("channels,n,poles", [(1, 1, 4), (1, 3, 4), (64, 64, 4), (65, 63, 2), (130, 257, 6), (3, 4099, 8),
                                              (1000, 129, 4), (5, 1, 2), (2, 70, 1), (2, 1000, 3), (2, 500, 5), (2, 500, 7)])
def test_batch_ragged(ops, dev, orc, channels, n, poles):
    import torch
    rng = np.random.default_rng(channels * 31 + n)
    if poles % 2 == 0:
        a, b = orc.chebyshev(0.12, poles)
    else:  # odd orders: not produced by the reference's designers but legal Coefficients<P>
        a = rng.standard_normal(poles + 1) * 0.3
        b = np.concatenate([[0.0], rng.uniform(-0.4, 0.4, poles) / poles])
    x = rng.standard_normal((channels, n)).astype(np.float32)
    st0 = rng.standard_normal((channels, 2 * poles))
    st0[:, :poles] = st0[:, :poles].astype(np.float32)  # x history holds float samples
    # oracle state layout: x[0..P] (x[0] scratch), y[0..P] (y[0] unused)
    so = np.zeros((channels, 2 * (poles + 1)))
    so[:, 1:poles + 1] = st0[:, :poles]
    so[:, poles + 2:] = st0[:, poles:]
    want = orc.iir(x.astype(np.float64), a, b, state=so)
    for mode in ("batch", "scan"):
        st = torch.from_numpy(st0.copy()).to(dev)
        guard = torch.full((channels + 2, n), 5.0, device=dev)
        ops.iir(torch.from_numpy(x).to(dev), a, b, state=st, mode=mode, out=guard[1:-1])
        g = guard.cpu().numpy()
        assert (g[0] == 5).all() and (g[-1] == 5).all()
        assert rel_l2(g[1:-1], want) <= TIGHT, mode
        got_st = st.cpu().numpy()
        assert np.allclose(got_st[:, :poles], so[:, 1:poles + 1])
        assert np.allclose(got_st[:, poles:], so[:, poles + 2:], rtol=1e-5, atol=1e-8)  # 6/8-pole direct forms amplify rounding order


@pytest.mark.parametrize("n", [8192, 8193, 20000, 1 << 18, (1 << 18) + 5, 100])
@pytest.mark.parametrize("tag", ["lp4", "lp4_lo", "bp"])
def test_scan_long_channel(ops, dev, orc, golden, n, tag):
    """Block-parallel scan against the sequential reference: lp4 takes the fused warm-up path,
    lp4_lo (fc 0.01, slow decay) and the narrow band-pass take the exact reduce/carry/apply path."""
    import torch
    rng = np.random.default_rng(n)
    if tag == "bp":
        a, b = golden["bq_bp_a"], golden["bq_bp_b"]
    else:
        a, b = golden[f"cheby_{tag}_a"], golden[f"cheby_{tag}_b"]
    x = rng.standard_normal((2, n)).astype(np.float32)
    run = orc.biquad if len(a) == 3 else orc.iir
    want = run(x.astype(np.float64), a, b)
    st = torch.zeros(2, 2 * (len(a) - 1), dtype=torch.float64, device=dev)
    got = ops.iir(torch.from_numpy(x).to(dev), a, b, state=st, mode="scan").cpu().numpy()
    assert rel_l2(got, want) <= TIGHT
    # batch mode computes the same recursion sequentially: the scan must agree with it
    st_b = torch.zeros_like(st)
    got_b = ops.iir(torch.from_numpy(x).to(dev), a, b, state=st_b, mode="batch").cpu().numpy()
    assert rel_l2(got, got_b.astype(np.float64)) <= 1e-6
    assert np.allclose(st.cpu().numpy(), st_b.cpu().numpy(), rtol=1e-6, atol=1e-9)


def test_scan_nearly_unstable(ops, dev, orc):
    """fc = 0.001 (44 Hz at 44.1 kHz): poles at radius 0.9985, companion powers ~1e10 over a tile.  Double
    carries would be garbage; the scan switches to double-double carries and still matches the sequential
    reference.  n > one tile, slow decay: exercises the exact reduce / carry / apply path in wide mode."""
    import torch
    a, b = orc.chebyshev(0.001, 4)
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, 100000)).astype(np.float32)
    want = orc.iir(x.astype(np.float64), a, b)
    xd = torch.from_numpy(x).to(dev)
    st = torch.zeros(2, 8, dtype=torch.float64, device=dev)
    got = ops.iir(xd, a, b, state=st, mode="scan").cpu().numpy()
    assert rel_l2(got, want) <= 1e-5
    st_b = torch.zeros_like(st)
    got_b = ops.iir(xd, a, b, state=st_b, mode="batch").cpu().numpy()
    assert rel_l2(got, got_b.astype(np.float64)) <= 1e-5
    assert np.allclose(st.cpu().numpy(), st_b.cpu().numpy(), rtol=1e-4, atol=1e-9)
    # 8-pole, low cutoff, long channel
    a8, b8 = orc.chebyshev(0.02, 8)
    x8 = rng.standard_normal((1, 60000)).astype(np.float32)
    got8 = ops.iir(torch.from_numpy(x8).to(dev), a8, b8, mode="scan").cpu().numpy()
    assert rel_l2(got8, orc.iir(x8.astype(np.float64), a8, b8)) <= TOL


def test_scan_refuses_unstable_filters(ops, dev):
    """A pole outside the unit circle: no carry arithmetic can hold the budget; SCAN says so, AUTO falls back."""
    import torch
    from dsp_amd import capi
    a, b = np.array([1.0, 0.0, 0.0]), np.array([0.0, 2.5, -1.0])
    x = torch.randn(1, 70000, device=dev)
    with pytest.raises(capi.DspGpuError) as e:
        ops.iir(x, a, b, mode="scan")
    assert e.value.code == capi.EUNSUPPORTED and "ill-conditioned" in str(e.value)
    y = ops.iir(x[:, :64].contiguous(), a, b, mode="auto")  # few short channels: AUTO is BATCH anyway
    assert y.shape == (1, 64)
    ops.iir(x, a, b, mode="auto")  # long channel: AUTO tries SCAN, is refused, runs BATCH (output overflows like the reference's)


def test_c4_full_size_properties(ops, dev, golden):
    """BASELINE config 4 at full size: 262,144 channels (batch) and one 2^28-sample channel (scan)."""
    import torch
    a, b = golden["cheby_lp4_a"], golden["cheby_lp4_b"]
    g = torch.Generator(device=dev).manual_seed(777)
    C, N = 262144, 4096
    x = torch.randn(C, N, device=dev, generator=g)
    y = ops.iir(x, a, b, mode="batch")
    # the same data seen as ONE long channel by the scan, and as channels by the batch kernel
    # must agree wherever the channel boundary does not matter: compare channel 0 .. 63 individually
    ys = ops.iir(x[:64].contiguous(), a, b, mode="scan")
    assert float(torch.linalg.vector_norm((y[:64] - ys).double()) / torch.linalg.vector_norm(ys.double())) <= 1e-6
    # DC gain 1 (low-pass normalised at DC, recursive_filters.h:67-86) and linearity
    ones = torch.ones(4, N, device=dev)
    assert float((ops.iir(ones, a, b, mode="batch")[:, -1] - 1).abs().max()) < 1e-5
    lhs = ops.iir(2.0 * x[:32] - 0.5 * x[32:64], a, b, mode="batch")
    rhs = 2.0 * y[:32] - 0.5 * y[32:64]
    assert float(torch.linalg.vector_norm((lhs - rhs).double()) / torch.linalg.vector_norm(rhs.double())) <= 1e-6
    del y, ys, lhs, rhs
    # scan: 2^28 samples, the channel-major buffer reinterpreted as one channel
    flat = x.view(1, -1)[:, : 1 << 28]
    z = ops.iir(flat, a, b, mode="scan")
    # chunked batch run of the first 2^20 samples with carried state is the sequential truth
    st = torch.zeros(1, 8, dtype=torch.float64, device=dev)
    seq = ops.iir(flat[:, : 1 << 20].contiguous(), a, b, state=st, mode="batch")
    assert float(torch.linalg.vector_norm((z[:, : 1 << 20] - seq).double()) / torch.linalg.vector_norm(seq.double())) <= 1e-6
    # and a window deep inside the signal, restarted from rest 4096 samples earlier (history decays below fp32)
    o = (1 << 27) + 12345
    seg = ops.iir(flat[:, o - 4096: o + 65536].contiguous(), a, b, mode="batch")
    assert float(torch.linalg.vector_norm((z[:, o: o + 65536] - seg[:, 4096:]).double())
                 / torch.linalg.vector_norm(seg[:, 4096:].double())) <= 1e-6


def test_errors(ops, dev):
    import torch
    from dsp_amd import capi
    x = torch.zeros(2, 100, device=dev)
    with pytest.raises(capi.DspGpuError):
        ops.iir(x, np.zeros(22), np.zeros(22))  # 21 poles
    with pytest.raises(capi.DspGpuError):
        ops.iir(x, np.zeros(3), np.zeros(3), state=torch.zeros(2, 4, device=dev))  # state must be float64
    with pytest.raises(capi.DspGpuError):
        ops.iir(x, np.array([1.0, np.nan, 0]), np.zeros(3))


# ---- sections beyond 8 poles (Coefficients<10..20>, recursive_filters.h:30-31) ------------------------
@pytest.mark.parametrize("poles,fc", [(10, 0.25), (12, 0.25), (16, 0.25), (20, None), (9, None), (13, None), (19, None)])
def test_batch_high_order(ops, dev, orc, poles, fc):
    """10..20-pole direct forms run one lane per channel; orders without a kernel of their own ride the next
    one up with zero top coefficients.  Chebyshev designs at fc = 0.25, where the reference's own double
    design is still well conditioned up to 16 poles (its 20-pole direct form diverges there, in the reference as
    here); 20 poles and the odd orders are random stable sections."""
    import torch
    rng = np.random.default_rng(poles)
    if fc is not None:
        a, b = orc.chebyshev(fc, poles)
    else:
        a = rng.standard_normal(poles + 1) * 0.3
        b = np.concatenate([[0.0], rng.uniform(-0.4, 0.4, poles) / poles])
    channels, n = 70, 1500
    x = rng.standard_normal((channels, n)).astype(np.float32)
    st0 = rng.standard_normal((channels, 2 * poles)) * 0.1
    st0[:, :poles] = st0[:, :poles].astype(np.float32)
    so = np.zeros((channels, 2 * (poles + 1)))
    so[:, 1:poles + 1] = st0[:, :poles]
    so[:, poles + 2:] = st0[:, poles:]
    want = orc.iir(x.astype(np.float64), a, b, state=so)
    st = torch.from_numpy(st0.copy()).to(dev)
    xd = torch.from_numpy(x).to(dev)
    y = torch.cat([ops.iir(xd[:, :700].contiguous(), a, b, state=st, mode="batch"),
                   ops.iir(xd[:, 700:].contiguous(), a, b, state=st, mode="batch")], dim=1)
    assert rel_l2(y.cpu().numpy(), want) <= TOL
    got_st = st.cpu().numpy()
    assert np.allclose(got_st[:, :poles], so[:, 1:poles + 1])
    assert np.allclose(got_st[:, poles:], so[:, poles + 2:], rtol=1e-4, atol=1e-6)
    # AUTO takes the one-lane kernel for these orders -- sharded in time on a channel this long (late round 3: equal to the
    # sequential run to fp32 rounding, no longer bit for bit); SCAN says why it cannot
    long = torch.from_numpy(rng.standard_normal((1, 70000)).astype(np.float32)).to(dev)
    ya = ops.iir(long, a, b, mode="auto")
    yb = ops.iir(long, a, b, mode="batch")
    assert float(torch.linalg.vector_norm((ya - yb).double()) / torch.linalg.vector_norm(yb.double())) <= 1e-7
    from dsp_amd import capi
    with pytest.raises(capi.DspGpuError) as e:
        ops.iir(long, a, b, mode="scan")
    assert e.value.code == capi.EUNSUPPORTED


# ---- one coefficient set per channel (filter bank) ----------------------------------------------------
@pytest.mark.parametrize("channels,n", [(1, 100), (67, 777), (300, 4096)])
def test_iir_multi_filter_bank(ops, dev, orc, channels, n):
    """Every channel its own band-pass / shelf (the block-wise form of the retuned filter of
    eval/signals/wind_generator.h:150-167): each channel must equal the single-filter oracle run with its
    coefficients; retuning between blocks carries the state like the reference's m_bp.a = bp.a assignment."""
    import torch
    from dsp_amd import design
    rng = np.random.default_rng(channels + n)
    x = rng.standard_normal((channels, n)).astype(np.float32)

    def bank(seed):
        r = np.random.default_rng(seed)
        A, B, C0, D0 = [], [], [], []
        for k in range(channels):
            if k % 3 == 2:
                a, b, c0, d0 = design.low_shelf(r.uniform(0.02, 0.2), r.uniform(-6, 6))
            else:
                a, b, c0, d0 = design.band_pass(r.uniform(0.02, 0.45), r.uniform(0.002, 0.02))
            A.append(a); B.append(b); C0.append(c0); D0.append(d0)
        return np.array(A), np.array(B), np.array(C0), np.array(D0)

    A1, B1, C1, D1 = bank(1)
    A2, B2, C2, D2 = bank(2)
    cut = n // 3
    want = np.zeros((channels, n))
    for k in range(channels):
        so = np.zeros(6)
        want[k, :cut] = orc.iir(x[k, :cut].astype(np.float64), A1[k], B1[k], C1[k], D1[k], state=so)
        want[k, cut:] = orc.iir(x[k, cut:].astype(np.float64), A2[k], B2[k], C2[k], D2[k], state=so)  # retuned, history kept
    t = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev)
    xd = t(x)
    st = torch.zeros(channels, 4, dtype=torch.float64, device=dev)
    y = torch.cat([ops.iir_multi(xd[:, :cut].contiguous(), t(A1), t(B1), t(C1), t(D1), state=st),
                   ops.iir_multi(xd[:, cut:].contiguous(), t(A2), t(B2), t(C2), t(D2), state=st)], dim=1)
    got = y.cpu().numpy()
    for k in range(channels):
        assert rel_l2(got[k], want[k]) <= TIGHT, k
    # host pointers, default c0/d0 (= 1, 0): the band-pass channels only
    sel = [k for k in range(channels) if k % 3 != 2]
    sth = np.zeros((len(sel), 4))
    yh = ops.iir_multi(np.ascontiguousarray(x[sel][:, :cut]), np.ascontiguousarray(A1[sel]), np.ascontiguousarray(B1[sel]), state=sth)
    assert np.array_equal(yh, got[sel][:, :cut])
    # uniform bank == the shared-coefficient entry point, bit for bit
    a, b = orc.chebyshev(0.1, 4)
    rep = lambda v: t(np.tile(v, (channels, 1)))
    assert torch.equal(ops.iir_multi(xd, rep(a), rep(b)), ops.iir(xd, a, b, mode="batch"))


@pytest.mark.parametrize("fc", [0.01, 0.001])
def test_scan_many_tiles_slow_filters(ops, dev, fc):
    """Slow-decaying sections over enough tiles (8,193) that the tile carries are scanned by several workgroups with
    warm-up tiles (fc = 0.001) or summed locally from the last tile or two (fc = 0.01): against the sequential
    kernel on the same channel, carried state in and out."""
    import torch
    from dsp_amd import design
    a, b = design.chebyshev(fc, 4)
    n = (1 << 25) + 777
    g = torch.Generator(device=dev).manual_seed(int(fc * 1e4))
    x = torch.randn(1, n, device=dev, generator=g)
    st0 = torch.tensor([[0.3, -0.2, 0.1, 0.05, 0.4, 0.35, 0.3, 0.25]], dtype=torch.float64, device=dev)
    st_scan, st_seq = st0.clone(), st0.clone()
    y_scan = ops.iir(x, a, b, state=st_scan, mode="scan")
    y_seq = ops.iir(x, a, b, state=st_seq, mode="batch")
    err = torch.linalg.vector_norm((y_scan - y_seq).double()) / torch.linalg.vector_norm(y_seq.double())
    assert float(err) <= 2e-6, float(err)
    assert torch.allclose(st_scan, st_seq, rtol=1e-6, atol=1e-9)


def test_aliasing_rules(ops, dev, orc):
    """y == x is in-place in BATCH mode (and AUTO picks it); overlapping device buffers are refused elsewhere."""
    import torch
    from dsp_amd import capi
    a, b = orc.chebyshev(0.1, 4)
    x = torch.randn(3, 70000, device=dev)
    want = ops.iir(x, a, b, mode="batch")
    for mode in ("batch", "auto"):
        buf = x.clone()
        ops.iir(buf, a, b, mode=mode, out=buf)
        assert torch.equal(buf, want)
    with pytest.raises(capi.DspGpuError):
        buf = x.clone()
        ops.iir(buf, a, b, mode="scan", out=buf)
    big = torch.randn(3 * 70000 + 100, device=dev)
    with pytest.raises(capi.DspGpuError):
        ops.iir(big[:210000].view(3, 70000), a, b, mode="batch", out=big[100:210100].view(3, 70000))
    with pytest.raises(capi.DspGpuError):
        ops.convolve(big[:210000].view(3, 70000), np.ones(5), out=big[:210000].view(3, 70000))
    with pytest.raises(capi.DspGpuError):
        ops.moving_average(big[:210000].view(3, 70000), 5, out=big[4:210004].view(3, 70000))
    h = np.ascontiguousarray(x.cpu().numpy())
    assert np.array_equal(ops.iir(h, a, b, mode="scan", out=h), ops.iir(x, a, b, mode="scan").cpu().numpy())  # host buffers are staged


@pytest.mark.gpu
def test_scan_tables_cached_per_plan_across_streams(dev, orc):
    """The scan's matrix table lives on the device with its plan: repeated calls, two coefficient sets taking
    turns and a second stream (which has to wait for the upload made on the first) all read the right one."""
    import torch
    from dsp_amd import design, ops
    rng = np.random.default_rng(11)
    x = rng.standard_normal((2, 1 << 16)).astype(np.float32)
    xd = torch.from_numpy(x).to(dev)
    filters = [design.chebyshev(0.1, 4, highpass=False), design.chebyshev(0.03, 4, highpass=False),
               design.chebyshev(0.01, 4, highpass=False), design.chebyshev(0.2, 2, highpass=True)]
    want = [orc.iir(x.astype(np.float64), a, b) for a, b in filters]
    side = torch.cuda.Stream(device=dev)
    for rounds in range(3):
        for k, (a, b) in enumerate(filters):
            if (k + rounds) % 2:
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    got = ops.iir(xd, a, b, mode="scan")
                side.synchronize()
            else:
                got = ops.iir(xd, a, b, mode="scan")
            assert rel_l2(got.cpu().numpy(), want[k]) <= TOL, (rounds, k)


@pytest.mark.gpu
def test_high_order_sections_on_few_long_channels_are_sharded_in_time(ops, dev, orc):
    """Sections of more than eight poles (no scan kernel) on few long channels: AUTO shards the one-lane kernel in time
    (iir_batch_ext.hip, launch_iir_sharded) instead of walking a million samples on one lane -- same results as BATCH to fp32
    rounding, the oracle's bound, the carried state, chunked == unchunked."""
    import torch
    rng = np.random.default_rng(12)
    for poles, fc in ((10, 0.25), (12, 0.25), (10, 0.1)):
        a, b = orc.chebyshev(fc, poles)
        for channels, n in ((1, 150000), (3, 1 << 16)):
            x = rng.standard_normal((channels, n)).astype(np.float32)
            xd = torch.from_numpy(x).to(dev)
            want = orc.iir(x.astype(np.float64), a, b)
            got = ops.iir(xd, a, b, mode="auto").cpu().numpy()
            seq = ops.iir(xd, a, b, mode="batch").cpu().numpy()
            assert rel_l2(got, want) <= 1e-4, (poles, fc, channels, n)
            assert rel_l2(got, seq) <= 2e-7, (poles, fc, channels, n)
            st = torch.zeros(channels, 2 * poles, dtype=torch.float64, device=dev)
            h = (n // 2) // 128 * 128
            y0 = ops.iir(xd[:, :h].contiguous(), a, b, state=st, mode="auto")
            y1 = ops.iir(xd[:, h:].contiguous(), a, b, state=st, mode="auto")
            assert rel_l2(torch.cat([y0, y1], dim=1).cpu().numpy(), seq) <= 2e-7, (poles, fc, channels, n)
            st_seq = torch.zeros_like(st)
            ops.iir(xd, a, b, state=st_seq, mode="batch")
            assert float((st - st_seq).abs().max()) <= 1e-9 * max(float(st_seq.abs().max()), 1.0)
