"""GPU parity of the batched FFT (dsp_gpu_fft_batch / dsp_gpu_fft_real_batch) through the C ABI.

Tolerance (BASELINE.json north_star): relative L2 error <= 1e-5 * log2(N) against the reference's
double-precision result on the same float-exact inputs.
"""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def budget(log2n):
    return 1e-5 * max(log2n, 1)


@pytest.fixture(scope="module")
def ops(dev):
    from dsp_amd import ops as _ops
    return _ops


@pytest.mark.parametrize("n", [2, 8, 64, 1024, 4096])
def test_golden_vectors(ops, dev, golden, n):
    import torch
    x = golden[f"fft_{n}_in"]
    k = n.bit_length() - 1
    for inv, key in ((False, "fwd"), (True, "inv")):
        want = golden[f"fft_{n}_{key}"]
        got_d = ops.fft0(torch.from_numpy(x).to(dev), inv).cpu().numpy()
        got_h = ops.fft0(x, inv)  # host pointers, staged by the library
        assert rel_l2(got_d, want) <= budget(k)
        assert np.array_equal(got_d, got_h)


def test_kat_sign_and_scaling(ops, dev, golden):
    import torch
    x = golden["kat_fft8_in"].astype(np.complex64)
    fwd = ops.fft0(torch.from_numpy(x).to(dev))
    assert np.allclose(fwd.cpu().numpy(), golden["kat_fft8_fwd"], atol=1e-5)
    assert fwd.cpu().numpy()[1].imag < 0  # e^{+i} FORWARD: conjugate of numpy's forward (fft.h:22-29)
    back = ops.fft0(fwd, inverse=True).cpu().numpy()
    assert np.allclose(back, 8 * x, atol=1e-4)  # neither direction normalises


@pytest.mark.parametrize("log2n", list(range(1, 15)) + [16, 17, 18, 20])
def test_oracle_parity(ops, dev, orc, log2n):
    import torch
    rng = np.random.default_rng(1000 + log2n)
    n = 1 << log2n
    batch = 5 if log2n <= 12 else (2 if log2n <= 16 else 1)
    x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
    xd = torch.from_numpy(x).to(dev)
    for inv in (False, True):
        want = orc.fft0(x.astype(np.complex128), inv)
        got = ops.fft0(xd, inv).cpu().numpy()
        assert rel_l2(got, want) <= budget(log2n), (log2n, inv)
    assert np.array_equal(xd.cpu().numpy(), x)  # input untouched (fft.h:32: `in` is read only)


@pytest.mark.parametrize("log2n,batch", [(4, 1), (4, 255), (4, 257), (6, 63), (8, 17), (11, 3), (12, 1), (3, 1000), (1, 7)])
def test_ragged_batches(ops, dev, orc, log2n, batch):
    """Batches that do not fill the last workgroup (several small transforms share one)."""
    import torch
    rng = np.random.default_rng(batch * 13 + log2n)
    n = 1 << log2n
    x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
    guard = torch.full((batch + 2, n), 7 + 7j, dtype=torch.complex64, device=dev)
    out = guard[1:-1]
    ops.fft0(torch.from_numpy(x).to(dev), out=out)
    assert rel_l2(out.cpu().numpy(), orc.fft0(x.astype(np.complex128))) <= budget(log2n)
    g = guard.cpu().numpy()
    assert (g[0] == 7 + 7j).all() and (g[-1] == 7 + 7j).all()  # nothing written outside the batch


@pytest.mark.parametrize("log2n", [5, 12, 14, 18])
def test_in_place(ops, dev, orc, log2n):
    import torch
    rng = np.random.default_rng(log2n)
    n = 1 << log2n
    x = (rng.standard_normal((3, n)) + 1j * rng.standard_normal((3, n))).astype(np.complex64)
    xd = torch.from_numpy(x).to(dev)
    ops.fft0(xd, out=xd)
    assert rel_l2(xd.cpu().numpy(), orc.fft0(x.astype(np.complex128))) <= budget(log2n)


@pytest.mark.parametrize("tag,nf", [("impulse", 64), ("rect", 128), ("sinc", 128), ("pad", 300)])
def test_real_wrapper_golden(ops, dev, golden, tag, nf):
    """dsp::fft (fft.h:57-83) on the eval/fft_pairs recipes, incl. zero padding 300 -> 512."""
    import torch
    x = golden[f"pairs_{tag}_in"].astype(np.float32)
    want = golden[f"pairs_{tag}_out"]
    got = ops.fft(torch.from_numpy(x).to(dev), nf).cpu().numpy()
    assert got.shape == want.shape
    k = want.size.bit_length() - 1
    # the fixture input is double; compare against the oracle on the fp32-rounded input too
    assert rel_l2(got, want) <= budget(k) + 2e-7
    assert np.array_equal(got, ops.fft(x, nf))


@pytest.mark.parametrize("n_in,nf,batch", [(100, 128, 4), (300, 300, 3), (4096, 4096, 2), (5000, 4096, 2), (3, 2, 5),
                                           (20000, 20000, 2), (70000, 1 << 17, 1), (6000, 8192, 3), (9000, 16384, 2), (16384, 16384, 70),
                                           (32768, 32768, 9), (32767, 32768, 2), (65536, 65536, 3), (65535, 65536, 2), (1, 65536, 1),
                                           (100001, 1 << 17, 2), (1 << 18, 1 << 18, 2), (300000, 1 << 19, 1), (1 << 20, 1 << 20, 1)])
def test_real_wrapper_oracle(ops, dev, orc, n_in, nf, batch):
    import torch
    rng = np.random.default_rng(n_in + nf)
    x = rng.standard_normal((batch, n_in)).astype(np.float32)
    xd = torch.from_numpy(x).to(dev)
    for inv in (False, True):
        want = np.stack([orc.fft_real(x[b].astype(np.float64), nf, inv) for b in range(batch)])
        got = ops.fft(xd, nf, inverse=inv).cpu().numpy()
        k = want.shape[-1].bit_length() - 1
        assert rel_l2(got, want) <= budget(k)
        got_s = ops.fft(xd, nf, inverse=inv, shift=True).cpu().numpy()
        want_s = np.stack([orc.shift(w) for w in want])
        assert rel_l2(got_s, want_s) <= budget(k)


def test_chain_impulse_response_spectrum(ops, dev, golden):
    """eval/recursive_filters/main.cpp:104-121: filter response -> 1024-point fft (the fft half)."""
    import torch
    for tag in ("lp2", "pk"):
        resp = golden[f"chain_{tag}_response"].astype(np.float32)
        got = ops.fft(torch.from_numpy(resp).to(dev), 1024).cpu().numpy()
        assert rel_l2(got, golden[f"chain_{tag}_fft"]) <= budget(10)


def test_c2_full_size_properties(ops, dev):
    """BASELINE config 2 at full size (65,536 x 4096): size-independent properties."""
    import torch
    g = torch.Generator(device=dev).manual_seed(1234)
    B, N = 65536, 4096
    x = torch.view_as_complex(torch.randn(B, N, 2, device=dev, generator=g))
    X = ops.fft0(x)
    # Parseval: sum |X|^2 = N sum |x|^2, per transform
    ex = (x.real.double() ** 2 + x.imag.double() ** 2).sum(dim=1)
    eX = (X.real.double() ** 2 + X.imag.double() ** 2).sum(dim=1)
    assert float(((eX / (N * ex)) - 1).abs().max()) < 1e-5
    # round trip: INVERSE(FORWARD(x)) = N x
    back = ops.fft0(X, inverse=True)
    num = torch.linalg.vector_norm((back / N - x).reshape(-1).view(torch.float32).double())
    den = torch.linalg.vector_norm(x.reshape(-1).view(torch.float32).double())
    assert float(num / den) <= 2 * budget(12)
    # linearity on a slice: F(a x + b y) = a F(x) + b F(y)
    y = torch.roll(x[:256], 1, 0)
    lhs = ops.fft0(2.5 * x[:256] - 0.5j * y)
    rhs = 2.5 * X[:256] - 0.5j * ops.fft0(y)
    assert float(torch.linalg.vector_norm(lhs - rhs) / torch.linalg.vector_norm(rhs)) <= 2 * budget(12)
    # exact-bin tones: row r holds exp(-2 pi i k n / N) with k = r mod N  -> FORWARD peak N at bin k
    r = torch.arange(512, device=dev)
    n = torch.arange(N, device=dev)
    ph = (-(r[:, None] * n[None, :]) % N).double() * (2 * np.pi / N)
    tone = torch.complex(torch.cos(ph), torch.sin(ph)).to(torch.complex64)
    T = ops.fft0(tone)
    peak = T.abs().argmax(dim=1)
    # FORWARD is e^{+i}: x = e^{-i 2 pi k n/N} peaks at bin k
    assert torch.equal(peak, r)
    assert float((T.abs().max(dim=1).values - N).abs().max()) < 0.05


def test_large_single_transform_round_trip(ops, dev):
    import torch
    g = torch.Generator(device=dev).manual_seed(2030)
    n = 1 << 24
    x = torch.view_as_complex(torch.randn(1, n, 2, device=dev, generator=g))
    X = ops.fft0(x)
    back = ops.fft0(X, inverse=True) / n
    err = torch.linalg.vector_norm((back - x).view(torch.float32).double()) / torch.linalg.vector_norm(x.view(torch.float32).double())
    assert float(err) <= 2 * budget(24)
    # one exact-bin tone on top of the noise floor shows up where the e^{+i} convention puts it
    k = 1234567
    idx = torch.arange(n, device=dev, dtype=torch.int64)
    ph = ((idx * k) % n).double() * (2 * np.pi / n)
    tone = torch.complex(torch.cos(ph), -torch.sin(ph)).to(torch.complex64)[None]
    T = ops.fft0(tone)
    assert int(T.abs().argmax()) == k
    assert abs(float(T.abs().max()) / n - 1) < 1e-3


@pytest.mark.parametrize("log2n", [25, 26, 30])
def test_four_pass_transforms(ops, dev, log2n):
    """Above 2^24 points a transform takes four column passes (up to the reference's own ceiling of 2^30, its
    int indices, fft.h:36-37): against the six-step path (different kernels and factorisation), in place, and an
    exact-bin tone."""
    import torch
    from dsp_amd import dist_fft
    n = 1 << log2n
    g = torch.Generator(device=dev).manual_seed(log2n)
    x = torch.view_as_complex(torch.randn(n, 2, device=dev, generator=g))
    k = (n // 3) | 1
    idx = torch.arange(n, device=dev, dtype=torch.int64)
    ph = ((idx * k) % n).double() * (2 * np.pi / n)
    x += 3.0 * torch.complex(torch.cos(ph), -torch.sin(ph)).to(torch.complex64)
    del idx, ph
    X = ops.fft0(x[None])[0]
    assert int(X.abs().argmax()) == k
    assert abs(float(X[k].abs()) / n - 3.0) < 1e-2
    want = dist_fft.simulate([x], log2n)[0]
    den = float(torch.linalg.vector_norm(want.view(torch.float32).double()))
    assert float(torch.linalg.vector_norm((X - want).view(torch.float32).double())) / den <= budget(log2n)
    del want
    buf = x.clone()[None]
    ops.fft0(buf, out=buf)  # in place
    assert torch.equal(buf[0], X)
    del buf
    back = ops.fft0(X[None], inverse=True)[0] / n
    err = torch.linalg.vector_norm((back - x).view(torch.float32).double()) / torch.linalg.vector_norm(x.view(torch.float32).double())
    assert float(err) <= 2 * budget(log2n)


def test_errors(ops, dev):
    import torch
    from dsp_amd import capi
    with pytest.raises(capi.DspGpuError):
        ops.fft0(torch.zeros(3, 100, dtype=torch.complex64, device=dev))  # not a power of two
    with pytest.raises(capi.DspGpuError):
        ops.fft0(torch.zeros(3, 64, dtype=torch.complex128, device=dev))  # fp32 only
    with pytest.raises(capi.DspGpuError):
        ops.fft0(torch.zeros(1, 64, dtype=torch.complex64))  # CPU torch tensor: pass numpy for host memory


@pytest.mark.parametrize("log2n,batch", [(13, 1), (13, 1100), (14, 7), (14, 600)])
def test_single_pass_8k_16k(ops, dev, orc, log2n, batch):
    """2^13 / 2^14 points live in one workgroup's LDS (fft_big_kernel); the persistent workgroups walk several
    transforms each once the batch outgrows the chip (1100 > 8*64, 600 > 8*32)."""
    import torch
    n = 1 << log2n
    g = torch.Generator(device=dev).manual_seed(batch)
    x = torch.view_as_complex(torch.randn(batch, n, 2, device=dev, generator=g))
    X = ops.fft0(x)
    for r in sorted({0, batch // 2, batch - 1}):
        want = orc.fft0(x[r].cpu().numpy().astype(np.complex128))
        assert rel_l2(X[r].cpu().numpy(), want) <= budget(log2n)
    back = ops.fft0(X, inverse=True) / n
    err = torch.linalg.vector_norm((back - x).view(torch.float32).double()) / torch.linalg.vector_norm(x.view(torch.float32).double())
    assert float(err) <= 2 * budget(log2n)
    buf = x.clone()
    ops.fft0(buf, out=buf)
    assert torch.equal(buf, X)


@pytest.mark.parametrize("log2n", [13, 14, 15, 16, 18, 21, 25])
def test_real_half_length_route_equals_full_length(ops, dev, log2n, monkeypatch):
    """Rows of 2^15 samples and more ride a half-length complex transform (N/2 trick, spectrum unfolded in the last
    pass); the full-length route stays selectable: both must agree, with and without dsp::shift."""
    import torch
    g = torch.Generator(device=dev).manual_seed(log2n)
    n = 1 << log2n
    batch = 3 if log2n <= 18 else 1
    x = torch.randn(batch, n - 5, device=dev, generator=g)  # zero padded by 5 samples
    got = {}
    for half in ("1", "0"):
        monkeypatch.setenv("DSP_GPU_REAL_HALF_LARGE", half)
        monkeypatch.setenv("DSP_GPU_REAL_HALF_14", half)
        got[half] = (ops.fft(x, n), ops.fft(x, n, inverse=True, shift=True))
    if log2n != 15:  # (2^15 has no full-length single-pass form to compare with)
        for a, b in zip(got["1"], got["0"]):
            den = torch.linalg.vector_norm(b.view(torch.float32).double())
            assert float(torch.linalg.vector_norm((a - b).view(torch.float32).double()) / den) <= budget(log2n)
    # Hermitian symmetry of a real row's spectrum and the DC bin
    X = got["1"][0]
    assert float((X[:, 1:] - X[:, 1:].flip(1).conj()).abs().max()) <= 1e-3 * float(X.abs().max())
    assert float((X[:, 0].real - x.double().sum(1).float()).abs().max()) <= 2e-3 * float(x.abs().sum(1).max()) ** 0.5 + 1e-1


@pytest.mark.parametrize("batch", [1, 5, 300])
def test_single_pass_32768_points_equals_two_pass(ops, dev, orc, batch, monkeypatch):
    """2^15-point transforms take one trip through HBM (fft_plane_kernel: 512 threads x 64 points, the points cross
    between passes one fp32 plane at a time); DSP_GPU_FFT_PLANE=0 keeps the two-pass route.  Both against the oracle,
    against each other, forward / inverse round trip and in place."""
    import torch
    g = torch.Generator(device=dev).manual_seed(15 + batch)
    n = 1 << 15
    x = torch.view_as_complex(torch.randn(batch, n, 2, device=dev, generator=g))
    got = {}
    for plane in ("1", "0"):
        monkeypatch.setenv("DSP_GPU_FFT_PLANE", plane)
        got[plane] = (ops.fft0(x), ops.fft0(x, inverse=True))
    monkeypatch.setenv("DSP_GPU_FFT_PLANE", "1")
    for a, b in zip(got["1"], got["0"]):
        den = torch.linalg.vector_norm(b.view(torch.float32).double())
        assert float(torch.linalg.vector_norm((a - b).view(torch.float32).double()) / den) <= budget(15)
    for row in (0, batch - 1):
        want = orc.fft0(x[row].cpu().numpy().astype(np.complex128), False)
        assert rel_l2(got["1"][0][row].cpu().numpy(), want) <= budget(15)
    back = ops.fft0(got["1"][0], inverse=True)
    assert float((back / n - x).abs().max()) <= 1e-4 * float(x.abs().max())
    buf = x.clone()
    ops.fft0(buf, out=buf)
    assert torch.equal(buf, got["1"][0])


@pytest.mark.parametrize("n_in", [65536, 65531, 40000, 3])
@pytest.mark.parametrize("shift", [False, True])
def test_real_65536_single_pass_equals_other_routes(ops, dev, orc, n_in, shift, monkeypatch):
    """Rows of 2^16 real samples (eval/modulation/helper.h:43's size) ride the single-pass 2^15-point transform, spectrum
    unfolded in the same workgroup; zero padding comes from the buffer descriptor's bounds check.  Against the
    two-pass half-length route, the full-length route and the oracle."""
    import torch
    g = torch.Generator(device=dev).manual_seed(n_in + int(shift))
    x = torch.randn(9, n_in, device=dev, generator=g)
    nf = 1 << 16
    monkeypatch.setenv("DSP_GPU_FFT_PLANE", "1")
    got = ops.fft(x, nf, shift=shift)
    inv = ops.fft(x, nf, inverse=True, shift=shift)
    monkeypatch.setenv("DSP_GPU_FFT_PLANE", "0")
    two_pass = ops.fft(x, nf, shift=shift)
    monkeypatch.setenv("DSP_GPU_REAL_HALF_LARGE", "0")
    full = ops.fft(x, nf, shift=shift)
    for other in (two_pass, full):
        den = torch.linalg.vector_norm(other.view(torch.float32).double())
        assert float(torch.linalg.vector_norm((got - other).view(torch.float32).double()) / den) <= budget(16)
    padded = np.zeros(nf)
    padded[:n_in] = x[8].cpu().numpy()
    want = orc.fft0(padded.astype(np.complex128), False)
    if shift:
        want = np.roll(want, nf // 2)
    assert rel_l2(got[8].cpu().numpy(), want) <= budget(16)
    # the other direction of a real row is the conjugate spectrum
    den = torch.linalg.vector_norm(got.view(torch.float32).double())
    assert float(torch.linalg.vector_norm((inv - got.conj()).view(torch.float32).double()) / den) <= budget(16)


def test_random_shapes_against_numpy(ops, dev):
    """A seeded sweep over the dispatch table -- every kernel family (tiny, tile, single-workgroup 2^13 / 2^14 / 2^15,
    multi-pass; real rows full- and half-length), odd batches, zero padding, shift, both directions, in place --
    against numpy's FFT as an independent check (FORWARD = e^{+i} = N * ifft, nothing normalises)."""
    import torch
    rng = np.random.default_rng(20260)
    for case in range(60):
        log2n = int(rng.integers(1, 19))
        n = 1 << log2n
        batch = int(rng.integers(1, max(2, min(40, (1 << 21) // n) + 1)))
        inverse = bool(rng.integers(0, 2))
        if case % 2 == 0:  # complex
            x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
            want = np.fft.fft(x.astype(np.complex128), axis=-1) if inverse else np.fft.ifft(x.astype(np.complex128), axis=-1) * n
            xd = torch.from_numpy(x).to(dev)
            got = ops.fft0(xd, inverse=inverse)
            assert rel_l2(got.cpu().numpy(), want) <= budget(log2n), ("complex", log2n, batch, inverse)
            buf = xd.clone()
            ops.fft0(buf, inverse=inverse, out=buf)
            assert torch.equal(buf, got), ("in place", log2n, batch, inverse)
        else:  # real rows, zero padded to n, optionally shifted
            n_in = int(rng.integers(1, n + 1)) if rng.integers(0, 3) else n
            shift = bool(rng.integers(0, 2))
            x = rng.standard_normal((batch, n_in)).astype(np.float32)
            padded = np.zeros((batch, n))
            padded[:, :n_in] = x
            want = np.fft.fft(padded, axis=-1) if inverse else np.fft.ifft(padded, axis=-1) * n
            if shift:
                want = np.roll(want, n // 2, axis=-1)
            got = ops.fft(torch.from_numpy(x).to(dev), n, inverse=inverse, shift=shift).cpu().numpy()
            assert rel_l2(got, want) <= budget(log2n), ("real", log2n, batch, n_in, inverse, shift)
