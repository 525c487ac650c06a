"""The DSP_GPU_MEM_HOST route (dsp_amd/csrc/host_path.hip): batches in host memory are cut into chunks of whole
units and pipelined through a page-locked ring (copy-in || H2D || kernels || D2H || copy-out); small calls take one
bounce each way.  Whatever the chunking, the result must equal the device-pointer call on the whole batch -- byte
for byte where the same kernel runs per unit (FFT, direct FIR, moving average, IIR)."""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(dev):
    from dsp_amd import ops as _ops
    return _ops


@pytest.mark.parametrize("batch,log2n", [(3, 10), (700, 12), (33, 16), (2, 21)])
def test_fft_host_equals_device(ops, dev, batch, log2n):
    import torch
    rng = np.random.default_rng(batch + log2n)
    x = (rng.standard_normal((batch, 1 << log2n)) + 1j * rng.standard_normal((batch, 1 << log2n))).astype(np.complex64)
    want = ops.fft0(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert np.array_equal(ops.fft0(x), want)                      # pageable: through the ring
    xp = torch.from_numpy(x).pin_memory()
    yp = torch.empty_like(xp).pin_memory()
    ops.fft0(xp.numpy(), out=yp.numpy())                          # page-locked caller buffers: direct DMA
    assert np.array_equal(yp.numpy(), want)
    y = x.copy()
    ops.fft0(y, out=y)                                            # host buffers may alias
    assert np.array_equal(y, want)


def test_host_buffers_that_overlap_partially(ops, dev):
    """Host buffers may overlap in any way: the pipelined route must not write chunk i-2 of `out` over input it has not
    staged yet.  y = x shifted by five rows (equal unit sizes, different arrays), and a real-input FFT whose float rows
    live inside the complex output array (in == out, in_unit != out_unit)."""
    import torch
    rng = np.random.default_rng(12)
    buf = np.zeros((705, 4096), dtype=np.complex64)
    x = buf[5:]
    x[...] = (rng.standard_normal((700, 4096)) + 1j * rng.standard_normal((700, 4096))).astype(np.complex64)
    want = ops.fft0(torch.from_numpy(x.copy()).to(dev)).cpu().numpy()
    y = buf[:700]
    ops.fft0(x, out=y)       # 23 MiB each way: cut into chunks unless the overlap is noticed
    assert np.array_equal(y, want)
    buf2 = np.zeros((705, 4096), dtype=np.complex64)
    x2 = buf2[:700]
    x2[...] = x.copy() if False else (rng.standard_normal((700, 4096)) + 1j * rng.standard_normal((700, 4096))).astype(np.complex64)
    want2 = ops.fft0(torch.from_numpy(x2.copy()).to(dev)).cpu().numpy()
    ops.fft0(x2, out=buf2[5:])  # the other direction: out ahead of in
    assert np.array_equal(buf2[5:], want2)

    out = np.zeros((900, 4096), dtype=np.complex64)
    xr = out.view(np.float32).reshape(-1)[:900 * 3000].reshape(900, 3000)
    xr[...] = rng.standard_normal((900, 3000)).astype(np.float32)
    want3 = ops.fft(torch.from_numpy(xr.copy()).to(dev), 3000).cpu().numpy()
    ops.fft(xr, 3000, out=out)
    assert np.array_equal(out, want3)


def test_real_fft_host_rows_with_padding(ops, dev):
    import torch
    rng = np.random.default_rng(5)
    x = rng.standard_normal((900, 3000)).astype(np.float32)       # 3000 -> 4096 points, ~10 MiB in, 29 MiB out
    want = ops.fft(torch.from_numpy(x).to(dev), 3000, shift=True).cpu().numpy()
    assert np.array_equal(ops.fft(x, 3000, shift=True), want)


@pytest.mark.parametrize("channels,n", [(1, 300), (64, 1 << 16), (5, 1 << 21), (1, 1 << 22)])
def test_filters_host_equal_device(ops, dev, orc, channels, n):
    import torch
    from dsp_amd import design
    rng = np.random.default_rng(channels * 7 + n)
    x = rng.standard_normal((channels, n)).astype(np.float32)
    xd = torch.from_numpy(x).to(dev)
    h = design.sinc(0.1, 101)
    assert np.array_equal(ops.convolve(x, h, "direct"), ops.convolve(xd, h, "direct").cpu().numpy())
    got = ops.convolve(x, h, "overlap_save")
    assert rel_l2(got, ops.convolve(xd, h, "overlap_save").cpu().numpy()) <= 2e-6  # chunks may pick the other OS kernel
    assert np.array_equal(ops.moving_average(x, 51), ops.moving_average(xd, 51).cpu().numpy())
    a, b = design.chebyshev(0.1, 4, highpass=False)
    for mode in ("batch", "scan", "auto"):
        st_h = rng.standard_normal((channels, 8))
        st_d = torch.from_numpy(st_h.copy()).to(dev)
        want = ops.iir(xd, a, b, state=st_d, mode=mode).cpu().numpy()
        st = st_h.copy()
        got = ops.iir(x, a, b, state=st, mode=mode)
        assert np.array_equal(got, want), mode
        assert np.array_equal(st, st_d.cpu().numpy()), mode


def test_small_call_latency_path_many_times(ops, dev, orc):
    """The single-bounce route re-uses one page-locked slot: back-to-back calls of changing sizes."""
    rng = np.random.default_rng(9)
    for n in (16, 1024, 64, 4096, 256, 2048, 8):
        x = (rng.standard_normal((3, n)) + 1j * rng.standard_normal((3, n))).astype(np.complex64)
        got = ops.fft0(x)
        assert rel_l2(got, orc.fft0(x.astype(np.complex128))) <= 1e-5 * max(np.log2(n), 1)
