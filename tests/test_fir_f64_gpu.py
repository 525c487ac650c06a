"""GPU parity of the FIR filters on DOUBLE samples (dsp_gpu_convolve_f64, dsp_gpu_moving_average_f64) through the C ABI.

The reference's own callers filter SampleBuffer<double> with the kernel dsp::sinc returns in double
(eval/windowed_sinc_filter/main.cpp:105, eval/filter_compare/main.cpp:11-20, eval/moving_average_filter_demo/main.cpp:14).
The GPU computes exact double fmas and differs from the reference's loops only by the order of the sums:
tolerance 1e-13 relative L2 (measured 1e-16 ... 3e-15), written below; the `*_f64` fixtures are the reference's own output.
"""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu
TOL64 = 1e-13


@pytest.fixture(scope="module")
def ops(dev):
    from dsp_amd import ops as _ops
    return _ops


def test_golden_511_taps_double(ops, dev, golden):
    import torch
    x = golden["fir_noise_in"].astype(np.float64)
    taps = golden["sinc_0p1_511"]
    got = ops.convolve_f64(torch.from_numpy(x).to(dev), taps, "direct").cpu().numpy()
    assert got.dtype == np.float64
    assert np.all(got[:510] == 0.0)  # filter.h:133
    err = rel_l2(got, golden["fir_noise_511_f64"])
    assert err <= TOL64, err
    assert np.array_equal(got, ops.convolve_f64(x, taps, "auto"))  # host-pointer route, AUTO == DIRECT


def test_golden_five_tone_recipe_double(ops, dev, golden):
    """eval/windowed_sinc_filter/main.cpp:48-65,104-108: the 5-tone cosine through the 201-tap sinc, in double as there."""
    import torch
    x = golden["fir_tone_in"].astype(np.float64)
    got = ops.convolve_f64(torch.from_numpy(x).to(dev), golden["sinc_0p14_200"], "direct").cpu().numpy()
    err = rel_l2(got, golden["fir_tone_out_f64"])
    assert err <= TOL64, err


def test_filter_compare_impulse_double(ops, dev, orc):
    """eval/filter_compare/main.cpp:11-20: SincFilter{0.14, 200} applied to an impulse at 300 of 401 samples."""
    import torch
    taps = orc.sinc(0.14, 200)
    x = np.zeros(401)
    x[300] = 1.0
    got = ops.convolve_f64(torch.from_numpy(x).to(dev), taps).cpu().numpy()
    want = orc.fir(x, taps)
    assert np.array_equal(got, want)  # one non-zero product per output: exact


@pytest.mark.parametrize("channels,n,ntaps", [(1, 1, 1), (1, 5, 3), (3, 17, 17), (2, 4096, 511), (5, 4097, 64), (2, 9001, 1023),
                                              (1, 20000, 4096), (70, 300, 31), (1, 8191, 2), (4, 12288, 129)])
def test_ragged_shapes_against_the_oracle(ops, dev, orc, channels, n, ntaps):
    import torch
    rng = np.random.default_rng(channels * 7 + n + ntaps)
    x = rng.standard_normal((channels, n))
    taps = rng.standard_normal(ntaps) / np.sqrt(ntaps)
    want = orc.fir(x, taps)
    guard = torch.full((channels + 2, n), 5.0, device=dev, dtype=torch.float64)
    ops.convolve_f64(torch.from_numpy(x).to(dev), taps, "direct", out=guard[1:-1])
    g = guard.cpu().numpy()
    assert (g[0] == 5).all() and (g[-1] == 5).all()
    err = rel_l2(g[1:-1], want)
    assert err <= TOL64, err
    assert np.all(g[1:-1][:, :min(n, ntaps - 1)] == 0.0)


def test_unaligned_rows_double(ops, dev, orc):
    import torch
    rng = np.random.default_rng(2)
    x = rng.standard_normal((3, 5001))
    taps = orc.sinc(0.2, 101)
    flat = torch.zeros(3 * 5001 + 1, device=dev, dtype=torch.float64)
    view = flat[1:].view(3, 5001)
    view.copy_(torch.from_numpy(x))
    got = ops.convolve_f64(view, taps).cpu().numpy()
    assert rel_l2(got, orc.fir(x, taps)) <= TOL64


def test_refusals_double(ops, dev):
    import torch
    from dsp_amd import capi
    x = torch.zeros(2, 1000, dtype=torch.float64, device=dev)
    with pytest.raises(capi.DspGpuError) as e:
        ops.convolve_f64(x, np.ones(8), "overlap_save")
    assert e.value.code == capi.EUNSUPPORTED
    with pytest.raises(capi.DspGpuError) as e:
        ops.convolve_f64(x, np.ones(5000))
    assert e.value.code == capi.EINVAL
    with pytest.raises(capi.DspGpuError):
        ops.convolve_f64(x, np.ones(8), out=x)  # in place on device buffers


def test_full_size_double_properties(ops, dev, orc):
    """256 x 2^18 doubles, 511 taps: impulse -> taps, linearity, and the oracle on a slice of one channel."""
    import torch
    taps = orc.sinc(0.1, 511)
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn(256, 1 << 18, device=dev, dtype=torch.float64, generator=g)
    y = ops.convolve_f64(x, taps)
    assert torch.equal(ops.convolve_f64(x * 2.0, taps), y * 2.0)
    want = orc.fir(x[37, :20000].cpu().numpy(), taps)
    assert rel_l2(y[37, :20000].cpu().numpy(), want) <= TOL64
    imp = torch.zeros(1, 1 << 18, device=dev, dtype=torch.float64)
    imp[0, 4096] = 1.0
    got = ops.convolve_f64(imp, taps).cpu().numpy()[0]
    assert np.array_equal(got[4096:4096 + 511], taps) and np.count_nonzero(got) == np.count_nonzero(taps)


# ---- moving average ---------------------------------------------------------------------------------------------------

def test_moving_average_golden_double(ops, dev, golden):
    import torch
    x = golden["fir_noise_in"].astype(np.float64)
    got = ops.moving_average_f64(torch.from_numpy(x).to(dev), 511).cpu().numpy()
    assert got.dtype == np.float64
    # the reference's running sum drifts by ~1e-16 per step over the 8K samples; windows summed directly do not
    err = rel_l2(got, golden["mavg_noise_511_f64"])
    assert err <= 1e-12, err
    assert np.all(got[:255] == 0.0) and np.all(got[-255:] == 0.0)
    assert np.array_equal(got, ops.moving_average_f64(x, 511))  # host-pointer route


@pytest.mark.parametrize("channels,n,npoints", [(2, 4096 * 33 + 1000, 511), (1, 1 << 18, 64), (3, 4096 * 20, 2045)])
def test_moving_average_sequential_kernel_double(ops, dev, orc, channels, n, npoints, monkeypatch):
    """Long channels: moving_average_seq_kernel<double> (see tests/test_fir_gpu.py::test_mavg_sequential_kernel)."""
    import torch
    monkeypatch.setenv("DSP_GPU_MAVG_SEQ", "1")
    rng = np.random.default_rng(n + npoints)
    x = rng.standard_normal((channels, n))
    got = ops.moving_average_f64(torch.from_numpy(x).to(dev), npoints).cpu().numpy()
    want = orc.moving_average(x, npoints)
    assert rel_l2(got, want) <= 1e-12
    for lo in range(0, n - 4096, 4096):
        assert rel_l2(got[:, lo:lo + 4096], want[:, lo:lo + 4096]) <= 1e-11, lo


@pytest.mark.parametrize("channels,n,npoints", [(1, 16, 5), (3, 1000, 8), (2, 4097, 511), (5, 300, 1), (1, 9000, 2048), (70, 129, 33),
                                                (1, 511, 511), (2, 20001, 8191)])
def test_moving_average_ragged_double(ops, dev, orc, channels, n, npoints):
    import torch
    rng = np.random.default_rng(n + npoints)
    x = rng.standard_normal((channels, n))
    want = orc.moving_average(x, npoints)
    got = ops.moving_average_f64(torch.from_numpy(x).to(dev), npoints).cpu().numpy()
    err = rel_l2(got, want)
    assert err <= 1e-12, err
    assert np.array_equal(got == 0.0, want == 0.0)  # the zero edges (filter.h:88-98)


def test_moving_average_kat_double(ops, dev):
    """MovingAverageFilter{5}.apply(0..15) -> 0 0 2 3 ... 13 0 0."""
    import torch
    got = ops.moving_average_f64(torch.arange(16, dtype=torch.float64, device=dev), 5).cpu().numpy()
    assert np.array_equal(got, [0, 0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 0, 0])


@pytest.mark.parametrize("channels,n,ntaps", [(700, 64, 7), (333, 101, 31), (50, 1000, 511), (9, 10001, 129), (3, 65535, 33)])
def test_many_channels_as_one_signal_double(ops, dev, orc, channels, n, ntaps, monkeypatch):
    """dsp_gpu_convolve_f64 on many channels runs them as one signal (fir_f64_kernels.hip): exact zeros over the first
    ntaps - 1 outputs of every channel, nothing leaked across channel borders, odd totals, nothing written around the batch;
    against the oracle and the per-channel path (DSP_GPU_FIR_NO_FLATTEN=1)."""
    import torch
    rng = np.random.default_rng(channels + n + ntaps)
    h = rng.standard_normal(ntaps) / np.sqrt(ntaps)
    x = rng.standard_normal((channels, n)) * 3 + 1
    want = orc.fir(x, h)
    guard = torch.full((channels + 2, n), 9.0, device=dev, dtype=torch.float64)
    ops.convolve_f64(torch.from_numpy(x).to(dev), h, out=guard[1:-1])
    g = guard.cpu().numpy()
    assert (g[0] == 9).all() and (g[-1] == 9).all()
    got = g[1:-1]
    assert rel_l2(got, want) <= TOL64
    assert not got[:, : ntaps - 1].any()
    monkeypatch.setenv("DSP_GPU_FIR_NO_FLATTEN", "1")
    other = ops.convolve_f64(torch.from_numpy(x).to(dev), h).cpu().numpy()
    assert rel_l2(got, other) <= 2 * TOL64
