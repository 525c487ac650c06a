"""GPU parity of the IIR recursion in the sample types and arithmetics beside fp32 / fast:

* DSP_GPU_IIR_BATCH_REF -- the reference's own arithmetic (coefficients.h:106-130: accumulator in the sample type,
  rounded after every `+=`; Coefficients<2> = BiQuad, coefficients.h:172-182: all double).  The bar is BIT EXACT
  against the reference-generated fixtures (`cheby_*_y32`, `bq_*_y32`, `*_y64`) and against the oracle on fresh
  seeds, for float and for double samples.  This is what closes the gap of the float instantiation of
  lowPassFilter<4>(0.01), which is 2.5e-3 away from its own double instantiation.
* dsp_gpu_iir_f64 -- double samples (SampleBuffer<double>, what every in-repo caller of the reference filters) in the
  fast arithmetic: double throughout with fused multiply-adds, so it differs from the reference's double instantiation
  only by the rounding of a different summation order: relative L2 <= 2e-12 (measured <= 1e-13 for the fixtures, 7e-13 for
  an 8-pole section on random data), except
  the ill-conditioned lowPassFilter<4>(0.01), whose feedback amplifies rounding differences to 1.3e-11 (bound 1e-10).

Everything goes through the C ABI (dsp_amd.ops -> libdsp_gpu.so).
"""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu
F64_FAST = 2e-12  # fast arithmetic on double samples against the reference's double instantiation


@pytest.fixture(scope="module")
def ops(dev):
    from dsp_amd import ops as _ops
    return _ops


CHEBY = ["lp2", "lp4", "lp6", "hp4", "lp4_lo", "lp8"]
BIQUADS = ["bp", "br", "ls", "hs", "pk"]


@pytest.mark.parametrize("tag", CHEBY)
def test_golden_chebyshev_reference_arithmetic_is_bit_exact(ops, dev, golden, tag):
    import torch
    x = golden["iir_noise_in"]
    a, b = golden[f"cheby_{tag}_a"], golden[f"cheby_{tag}_b"]
    got32 = ops.iir(torch.from_numpy(x).to(dev), a, b, mode="batch_ref").cpu().numpy()
    assert np.array_equal(got32, golden[f"cheby_{tag}_y32"]), rel_l2(got32, golden[f"cheby_{tag}_y32"])
    x64 = x.astype(np.float64)
    got64 = ops.iir_f64(torch.from_numpy(x64).to(dev), a, b, mode="batch_ref").cpu().numpy()
    assert np.array_equal(got64, golden[f"cheby_{tag}_y64"]), rel_l2(got64, golden[f"cheby_{tag}_y64"])
    # host-pointer route: same bits
    assert np.array_equal(ops.iir(x, a, b, mode="batch_ref"), got32)
    assert np.array_equal(ops.iir_f64(x64, a, b, mode="batch_ref"), got64)


@pytest.mark.parametrize("tag", BIQUADS)
def test_golden_biquads_reference_arithmetic_is_bit_exact(ops, dev, golden, tag):
    """recursive_filters.h:165-263 designs through Coefficients<2> (coefficients.h:172-182); shelves and peaking use c0/d0."""
    import torch
    x = golden["iir_noise_in"]
    a, b = golden[f"bq_{tag}_a"], golden[f"bq_{tag}_b"]
    c0, d0 = golden[f"bq_{tag}_cd"]
    got32 = ops.iir(torch.from_numpy(x).to(dev), a, b, c0, d0, mode="batch_ref").cpu().numpy()
    assert np.array_equal(got32, golden[f"bq_{tag}_y32"])
    got64 = ops.iir_f64(torch.from_numpy(x.astype(np.float64)).to(dev), a, b, c0, d0, mode="batch_ref").cpu().numpy()
    assert np.array_equal(got64, golden[f"bq_{tag}_y64"])


@pytest.mark.parametrize("tag", CHEBY + BIQUADS)
@pytest.mark.parametrize("mode", ["batch", "auto"])
def test_golden_double_samples_fast_arithmetic(ops, dev, golden, tag, mode):
    import torch
    x64 = golden["iir_noise_in"].astype(np.float64)
    if tag in BIQUADS:
        a, b = golden[f"bq_{tag}_a"], golden[f"bq_{tag}_b"]
        c0, d0 = golden[f"bq_{tag}_cd"]
        want = golden[f"bq_{tag}_y64"]
    else:
        a, b = golden[f"cheby_{tag}_a"], golden[f"cheby_{tag}_b"]
        c0, d0 = 1.0, 0.0
        want = golden[f"cheby_{tag}_y64"]
    got = ops.iir_f64(torch.from_numpy(x64).to(dev), a, b, c0, d0, mode=mode).cpu().numpy()
    assert got.dtype == np.float64
    # lowPassFilter<4>(0.01): four poles clustered at z = 1 amplify every rounding difference by the feedback noise gain
    # (~1e5): two double evaluations that differ only in summation order are 1.3e-11 apart there (the float
    # instantiation of the reference is 2.5e-3 from its double one for the same reason)
    bound = 1e-10 if tag == "lp4_lo" else F64_FAST
    assert rel_l2(got, want) <= bound, rel_l2(got, want)
    assert np.array_equal(got, ops.iir_f64(x64, a, b, c0, d0, mode=mode))  # host-pointer route


def test_carried_state_reference_arithmetic(ops, dev, golden):
    """The fixture was produced by the reference in two calls (1500 + 2596 samples) on one Coefficients<4> object."""
    import torch
    a, b = golden["cheby_lp4_a"], golden["cheby_lp4_b"]
    x = golden["iir_noise_in"]
    xd = torch.from_numpy(x).to(dev)
    st = torch.zeros(1, 8, dtype=torch.float64, device=dev)
    y = torch.cat([ops.iir(xd[:1500], a, b, state=st, mode="batch_ref"), ops.iir(xd[1500:], a, b, state=st, mode="batch_ref")])
    assert np.array_equal(y.cpu().numpy(), golden["cheby_lp4_y32_split"])
    # reference state layout: x[0..4], y[0..4] with x[0] = 0 scratch, y[1..4] = the float-rounded v[t-1..t-4]
    ref_state = golden["cheby_lp4_state_end"]
    got = st.cpu().numpy()[0]
    assert np.array_equal(got[:4], ref_state[1:5])
    assert np.array_equal(got[4:], ref_state[6:10])


@pytest.mark.parametrize("channels,n,poles", [(1, 1, 4), (1, 3, 4), (64, 64, 4), (65, 63, 2), (130, 257, 6), (3, 4099, 8),
                                              (1000, 129, 4), (5, 1, 2), (2, 70, 1), (2, 1000, 3), (2, 500, 5), (2, 500, 7),
                                              (3, 300, 9), (2, 200, 10), (2, 200, 12), (2, 150, 15), (2, 150, 20)])
@pytest.mark.parametrize("f64", [False, True])
def test_ragged_shapes_against_the_oracle(ops, dev, orc, channels, n, poles, f64):
    """Fresh seeds, every compiled order, initial state, ragged tiles, guard rows around the output.
    REF arithmetic: bit exact.  Fast arithmetic on double samples: inside F64_FAST."""
    import torch
    rng = np.random.default_rng(channels * 131 + n + poles)
    if poles % 2 == 0 and poles <= 8:
        a, b = orc.chebyshev(0.12, poles)
    else:  # odd / high orders: legal Coefficients<P>, kept stable by small feedback
        a = rng.standard_normal(poles + 1) * 0.3
        b = np.concatenate([[0.0], rng.uniform(-0.4, 0.4, poles) / poles])
    c0, d0 = 0.75, 0.25
    np_t = np.float64 if f64 else np.float32
    x = rng.standard_normal((channels, n)).astype(np_t)
    st0 = rng.standard_normal((channels, 2 * poles))
    st0[:, :poles] = st0[:, :poles].astype(np_t)  # x history holds samples
    if not f64 and poles != 2:
        st0[:, poles:] = st0[:, poles:].astype(np.float32)  # generic float section: y[] holds float-rounded values
    so = np.zeros((channels, 2 * (poles + 1)))  # oracle layout: x[0..P] (x[0] scratch), y[0..P] (y[0] unused)
    so[:, 1:poles + 1] = st0[:, :poles]
    so[:, poles + 2:] = st0[:, poles:]
    if poles == 2:  # Coefficients<2> is the BiQuad specialisation: state {x1, x2, y1, y2}
        sb = st0.copy()
        want = orc.biquad(x, a, b, c0, d0, state=sb)
        want_state = sb
    else:
        want = orc.iir(x, a, b, c0, d0, state=so)
        want_state = np.concatenate([so[:, 1:poles + 1], so[:, poles + 2:]], axis=1)
    run = ops.iir_f64 if f64 else ops.iir
    tdt = torch.float64 if f64 else torch.float32
    st = torch.from_numpy(st0.copy()).to(dev)
    guard = torch.full((channels + 2, n), 5.0, device=dev, dtype=tdt)
    run(torch.from_numpy(x).to(dev), a, b, c0, d0, state=st, mode="batch_ref", out=guard[1:-1])
    g = guard.cpu().numpy()
    assert (g[0] == 5).all() and (g[-1] == 5).all()
    assert np.array_equal(g[1:-1], want), rel_l2(g[1:-1], want)
    assert np.array_equal(st.cpu().numpy(), want_state)
    if f64:
        st = torch.from_numpy(st0.copy()).to(dev)
        got = run(torch.from_numpy(x).to(dev), a, b, c0, d0, state=st, mode="batch").cpu().numpy()
        assert rel_l2(got, want) <= F64_FAST
        assert np.allclose(st.cpu().numpy(), want_state, rtol=1e-11, atol=1e-13)


def test_unaligned_rows_and_in_place(ops, dev, orc):
    """Rows that are not 16-byte aligned (odd n, a view that starts one sample in) and y == x."""
    import torch
    rng = np.random.default_rng(5)
    a, b = orc.chebyshev(0.2, 4)
    for np_t, run in ((np.float32, ops.iir), (np.float64, ops.iir_f64)):
        x = rng.standard_normal((7, 1001)).astype(np_t)
        want = orc.iir(x, a, b)
        xd = torch.from_numpy(x).to(dev)
        assert np.array_equal(run(xd, a, b, mode="batch_ref").cpu().numpy(), want)
        flat = torch.zeros(7 * 1001 + 1, device=dev, dtype=xd.dtype)
        view = flat[1:].view(7, 1001)
        view.copy_(xd)
        run(view, a, b, mode="batch_ref", out=view)  # in place
        assert np.array_equal(view.cpu().numpy(), want)


@pytest.mark.parametrize("f64", [False, True])
def test_filter_bank_reference_arithmetic(ops, dev, orc, f64):
    """dsp_gpu_iir_multi_ex: one coefficient set per channel, REF arithmetic bit exact per channel; poles 4 and 2."""
    import torch
    rng = np.random.default_rng(77)
    np_t = np.float64 if f64 else np.float32
    for poles in (4, 2):
        channels, n = 70, 333
        x = rng.standard_normal((channels, n)).astype(np_t)
        A = np.zeros((channels, poles + 1))
        B = np.zeros((channels, poles + 1))
        for c in range(channels):
            A[c], B[c] = orc.chebyshev(0.05 + 0.4 * c / channels, poles)
        c0 = rng.uniform(0.5, 1.5, channels)
        d0 = rng.uniform(-0.5, 0.5, channels)
        want = np.stack([(orc.biquad if poles == 2 else orc.iir)(x[c], A[c], B[c], c0[c], d0[c]) for c in range(channels)])
        t = lambda v: torch.from_numpy(v).to(dev)
        got = ops.iir_multi(t(x), t(A), t(B), t(c0), t(d0), mode="batch_ref", f64=f64).cpu().numpy()
        assert np.array_equal(got, want)
        if f64:
            fast = ops.iir_multi(t(x), t(A), t(B), t(c0), t(d0), mode="batch", f64=True).cpu().numpy()
            assert rel_l2(fast, want) <= F64_FAST
        assert np.array_equal(ops.iir_multi(x, A, B, c0, d0, mode="batch_ref", f64=f64), want)  # host route


def test_double_samples_scan_is_sharded_in_time(ops, dev, orc):
    """SCAN on double samples (what AUTO takes for a few long channels): every channel cut into segments that start from
    states computed over dsp_gpu_iir_settle samples of warm-up (iir_batch_ext.hip) -- against the oracle's double recursion
    and against one lane per channel, with and without a carried state, single channels with a remainder, several channels
    with a dividing segment length; what cannot be sharded is refused and AUTO runs it one lane per channel."""
    import torch
    from dsp_amd import capi
    rng = np.random.default_rng(11)
    for fc, poles in ((0.1, 4), (0.25, 2), (0.05, 4)):
        a, b = orc.chebyshev(fc, poles)
        for channels, n in ((1, 100000), (1, 65536 + 777), (3, 1 << 16), (5, 30720)):
            x = rng.standard_normal((channels, n))
            xd = torch.from_numpy(x).to(dev)
            want = orc.iir(x, a, b)
            got = ops.iir_f64(xd, a, b, mode="scan").cpu().numpy()
            assert rel_l2(got, want) <= F64_FAST, (fc, poles, channels, n)
            assert rel_l2(got, ops.iir_f64(xd, a, b, mode="batch").cpu().numpy()) <= 1e-12
            assert rel_l2(ops.iir_f64(xd, a, b, mode="auto").cpu().numpy(), want) <= F64_FAST
            # two halves with the state carried = the whole
            st = torch.zeros(channels, 2 * poles, dtype=torch.float64, device=dev)
            h = (n // 2) // 64 * 64
            y0 = ops.iir_f64(xd[:, :h].contiguous(), a, b, state=st, mode="scan")
            st_b = st.clone()
            y1 = ops.iir_f64(xd[:, h:].contiguous(), a, b, state=st, mode="auto")
            assert rel_l2(torch.cat([y0, y1], dim=1).cpu().numpy(), want) <= F64_FAST, (fc, poles, channels, n)
            # the state handed back is the one a single lane hands back
            st_c = torch.zeros_like(st)
            ops.iir_f64(xd[:, :h].contiguous(), a, b, state=st_c, mode="batch")
            assert float((st_b - st_c).abs().max()) <= 1e-11 * max(float(st_c.abs().max()), 1.0)
    # more poles than AUTO asks the scan for with fp32 samples: doubles are sharded at every order
    a, b = orc.chebyshev(0.25, 12)
    x = rng.standard_normal((1, 200000))
    xd = torch.from_numpy(x).to(dev)
    got = ops.iir_f64(xd, a, b, mode="auto").cpu().numpy()
    assert rel_l2(got, ops.iir_f64(xd, a, b, mode="batch").cpu().numpy()) <= 1e-11
    assert rel_l2(got, orc.iir(x, a, b)) <= 1e-9  # (a 12-pole direct form: the one-lane run is no closer)
    # channel lengths without a 512-byte-row divisor (16-byte rows then), odd lengths (a few channels: one after the other, each
    # with its remainder), too short for two segments (refused; AUTO runs one lane per channel)
    a, b = orc.chebyshev(0.1, 4)
    for channels, n in ((4, 30000), (3, 100001), (2, 65535)):
        x = rng.standard_normal((channels, n))
        xd = torch.from_numpy(x).to(dev)
        st = torch.zeros(channels, 8, dtype=torch.float64, device=dev)
        got = ops.iir_f64(xd, a, b, state=st, mode="scan").cpu().numpy()
        assert rel_l2(got, orc.iir(x, a, b)) <= F64_FAST, (channels, n)
        st_seq = torch.zeros_like(st)
        ops.iir_f64(xd, a, b, state=st_seq, mode="batch")
        assert float((st - st_seq).abs().max()) <= 1e-11 * max(float(st_seq.abs().max()), 1.0), (channels, n)
    x = torch.zeros(4, 3000, dtype=torch.float64, device=dev)
    with pytest.raises(capi.DspGpuError) as e:
        ops.iir_f64(x, a, b, mode="scan")
    assert e.value.code == capi.EUNSUPPORTED
    ops.iir_f64(x, a, b, mode="auto")
    with pytest.raises(capi.DspGpuError) as e:  # a filter that does not settle within 8,192 samples
        a2, b2 = orc.chebyshev(0.001, 4)
        ops.iir_f64(torch.zeros(1, 1 << 20, dtype=torch.float64, device=dev), a2, b2, mode="scan")
    assert e.value.code == capi.EUNSUPPORTED


def test_full_size_double_batch_properties(ops, dev, orc):
    """A batch too large for the oracle in seconds: linearity and exact power-of-two scaling in both arithmetics,
    and REF == oracle on a few channels picked out of it."""
    import torch
    a, b = orc.chebyshev(0.1, 4)
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(8192, 4096, device=dev, dtype=torch.float64, generator=g)
    for mode in ("batch", "batch_ref"):
        y = ops.iir_f64(x, a, b, mode=mode)
        assert torch.equal(ops.iir_f64(x * 4.0, a, b, mode=mode), y * 4.0)
    pick = [0, 63, 64, 4095, 8191]
    want = orc.iir(x[pick].cpu().numpy(), a, b)
    assert np.array_equal(y[pick].cpu().numpy(), want)
