"""dsp_gpu_fft_batch_f64: dsp::fft0 in the reference's own sample type, std::complex<double> (fft.h:31-32), through the
C ABI.  Checked against the golden fixtures (outputs of the reference itself, tests/golden/make_golden.py), the oracle
on seeded inputs at every length up to 2^20, and size-independent properties above that.  Tolerance: the reference
builds its twiddles by a recurrence in double (fft.h:43-44,52) whose error grows with N; the GPU takes each twiddle from
sincospi on an exact angle, so the bound is the reference's own accuracy: 1e-13 * log2 N relative L2."""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def tol(log2n):
    """Against the reference / oracle.  Its twiddle recurrence w *= wm (fft.h:43-44,52) drifts by about eps per step, so
    the reference itself is off by ~2e-17 * N at large N (measured 2.2e-12 at 2^18); below that 1e-13 * log2 N."""
    return max(1e-13 * max(log2n, 1), 2e-17 * (1 << log2n))


def tight(log2n):
    """Against numpy's pocketfft (accurate to ~eps * log N): what the GPU's own arithmetic achieves."""
    return 1e-14 * max(log2n, 1)


@pytest.fixture(scope="module")
def ops(dev):
    from dsp_amd import ops as _ops
    return _ops


def test_golden_vectors(ops, golden, dev):
    """Outputs of the reference's own fft0<double> (oracle/_ref, tests/golden/make_golden.py)."""
    import torch
    seen = 0
    for n in (2, 8, 64, 1024, 4096):
        x = golden[f"fft_{n}_in"].astype(np.complex128)
        k = int(np.log2(n))
        for inverse, key in ((False, f"fft_{n}_fwd"), (True, f"fft_{n}_inv")):
            want = golden[key]
            got = ops.fft0_f64(torch.from_numpy(x).to(dev), inverse=inverse).cpu().numpy()
            assert rel_l2(got, want) <= tol(k), (n, inverse)
            assert rel_l2(ops.fft0_f64(x, inverse=inverse), want) <= tol(k), (n, inverse)  # host pointers
            seen += 1
    assert seen == 10


@pytest.mark.parametrize("log2n", list(range(1, 21)))
def test_oracle_parity_every_length(ops, orc, dev, log2n):
    import torch
    rng = np.random.default_rng(640 + log2n)
    n = 1 << log2n
    batch = 5 if log2n <= 14 else 2
    x = rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))
    xd = torch.from_numpy(x).to(dev)
    for inverse in (False, True):
        got = ops.fft0_f64(xd, inverse=inverse).cpu().numpy()
        for b in (0, batch - 1):
            assert rel_l2(got[b], orc.fft0(x[b], inverse)) <= tol(log2n), (log2n, inverse, b)
        # FORWARD is e^{+i 2 pi k n / N} = N * ifft, INVERSE e^{-i} = fft (nothing normalises)
        indep = np.fft.fft(x, axis=-1) if inverse else np.fft.ifft(x, axis=-1) * n
        assert rel_l2(got, indep) <= tight(log2n), (log2n, inverse)
    buf = xd.clone()
    ops.fft0_f64(buf, out=buf)  # in place
    assert torch.equal(buf, ops.fft0_f64(xd))


@pytest.mark.parametrize("log2n,batch", [(12, 4096), (13, 517), (16, 700), (18, 67), (22, 3), (24, 1)])
def test_properties_at_size(ops, dev, log2n, batch):
    """Round trip = N x, Parseval, an exact-bin tone and linearity at sizes the oracle does not reach in seconds."""
    import torch
    g = torch.Generator(device=dev).manual_seed(log2n)
    n = 1 << log2n
    x = torch.view_as_complex(torch.randn(batch, n, 2, device=dev, generator=g, dtype=torch.float64))
    X = ops.fft0_f64(x)
    back = ops.fft0_f64(X, inverse=True)
    assert float((back / n - x).abs().max()) <= 1e-12 * float(x.abs().max())
    ex, eX = float((x.abs() ** 2).sum()), float((X.abs() ** 2).sum())
    assert abs(eX / n - ex) <= 1e-12 * ex
    k0 = 12345 % n
    t = torch.arange(n, device=dev, dtype=torch.float64)
    tone = torch.exp(-2j * np.pi * ((k0 * t) % n) / n)[None]  # FORWARD is e^{+i}: the tone e^{-i 2 pi k0 n / N} lands in bin k0
    T = ops.fft0_f64(tone.contiguous())
    assert abs(complex(T[0, k0]) - n) <= 1e-9 * n
    T[0, k0] = 0
    assert float(T.abs().max()) <= 1e-9 * n
    y = torch.view_as_complex(torch.randn(batch, n, 2, device=dev, generator=g, dtype=torch.float64))
    lin = ops.fft0_f64(2.5 * x - 0.75 * y) - (2.5 * X - 0.75 * ops.fft0_f64(y))
    assert float(lin.abs().max()) <= 1e-11 * float(X.abs().max())


@pytest.mark.parametrize("log2n", [4, 5, 6, 7, 8, 10, 12])
def test_many_tiles_and_a_ragged_last_one(ops, dev, log2n):
    """Several tiles per XCD (the workgroup -> tile map gives each XCD a contiguous eighth) and a last tile that is
    partly behind the batch, on the staged short-row route (rows of fewer than 128 points) and on the direct one; out of
    place and in place; against torch's double transform (FORWARD = N * ifft)."""
    import torch
    n = 1 << log2n
    per_tile = 4096 // n
    g = torch.Generator(device=dev).manual_seed(7000 + log2n)
    for batch in (8 * 3 * per_tile + 37 % per_tile + 1, 11 * per_tile - 1, per_tile + 1):
        x = torch.view_as_complex(torch.randn(batch, n, 2, device=dev, generator=g, dtype=torch.float64))
        guard = torch.full((batch + 2, n), 7.0 + 0j, device=dev, dtype=torch.complex128)
        out = guard[1:-1]
        for inverse in (False, True):
            ops.fft0_f64(x, inverse=inverse, out=out)
            want = torch.fft.fft(x, dim=-1) if inverse else torch.fft.ifft(x, dim=-1) * n
            err = torch.linalg.vector_norm(out - want, dim=-1) / torch.linalg.vector_norm(want, dim=-1)
            assert float(err.max()) <= tight(log2n) * 4, (log2n, batch, inverse, int(err.argmax()))
            assert bool((guard[0] == 7.0).all()) and bool((guard[-1] == 7.0).all())  # nothing written around the batch
        buf = x.clone()
        ops.fft0_f64(buf, out=buf)
        assert torch.equal(buf, ops.fft0_f64(x))


def test_fp32_route_agrees_with_double_route(ops, dev):
    """The benchmarked fp32 transform against the double one on the same float-exact samples."""
    import torch
    g = torch.Generator(device=dev).manual_seed(99)
    x = torch.view_as_complex(torch.randn(64, 4096, 2, device=dev, generator=g))
    d = ops.fft0_f64(x.to(torch.complex128))
    s = ops.fft0(x).to(torch.complex128)
    assert float(torch.linalg.vector_norm(s - d) / torch.linalg.vector_norm(d)) <= 1e-5 * 12


def test_limits(ops, dev):
    import torch
    from dsp_amd import capi
    x = torch.zeros(24, dtype=torch.complex128, device=dev)
    with pytest.raises(capi.DspGpuError):
        ops.fft0_f64(x)  # not a power of two
    L = capi.lib()
    assert L.dsp_gpu_fft_batch_f64(x.data_ptr(), x.data_ptr(), 25, 1, capi.FORWARD, capi.MEM_DEVICE, None) == capi.EINVAL
    assert L.dsp_gpu_fft_batch_f64(None, x.data_ptr(), 3, 1, capi.FORWARD, capi.MEM_DEVICE, None) == capi.EINVAL


@pytest.mark.parametrize("log2n,batch", [(10, 3000), (18, 5), (22, 1)])
def test_host_pointers_equal_device_pointers(ops, dev, log2n, batch):
    """numpy in / numpy out (the host route: chunked through the page-locked ring, or whole when one transform is larger
    than a ring slot) gives the device route's bits."""
    import torch
    rng = np.random.default_rng(64 + log2n)
    x = rng.standard_normal((batch, 1 << log2n)) + 1j * rng.standard_normal((batch, 1 << log2n))
    want = ops.fft0_f64(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert np.array_equal(ops.fft0_f64(x), want)
