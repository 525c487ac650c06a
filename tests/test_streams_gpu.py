"""Calls on different streams of one device do not share scratch memory (include/dsp_gpu.h, "Threads and streams").

The library's growing workspaces -- the ping-pong slab of FFTs above 2^15 points, the tile carries and the state
snapshot of the SCAN IIR, the history scratch of dsp_gpu_convolve_stream -- are keyed by (device, stream).  Three host
threads, each on a stream of its own, run workloads that all go through those workspaces at the same time; every result
must be bit-identical to the same call made alone on the default stream.
"""
import threading

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from dsp_amd import capi, design, ops  # noqa: E402

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def test_three_threads_three_streams_bit_identical_to_serial():
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(31)
    # two different batches of 2^20-point transforms (three passes through the scratch slab)
    xa = torch.view_as_complex(torch.randn(8, 1 << 20, 2, device=dev, generator=g))
    xb = torch.view_as_complex(torch.randn(8, 1 << 20, 2, device=dev, generator=g))
    # a stateful SCAN IIR over a few long channels (tile carries + state snapshot) ...
    a, b = design.chebyshev(0.1, 4, highpass=False)
    xi = torch.randn(3, 1 << 22, device=dev, generator=g)
    st0 = torch.randn(3, 8, device=dev, generator=g, dtype=torch.float64) * 0.1
    # ... and a streamed FIR with carried history (history scratch), both on the third stream
    taps = design.sinc(0.1, 511)
    xf = torch.randn(16, 1 << 16, device=dev, generator=g)

    def fir_stream():
        hist = torch.zeros(16, 510, device=dev)
        parts, pos = [], 0
        for lo in range(0, xf.shape[1], 1 << 14):
            parts.append(ops.convolve_stream(xf[:, lo:lo + (1 << 14)].contiguous(), taps, hist, pos, "overlap_save"))
            pos += 1 << 14
        return torch.cat(parts, dim=1), hist

    def iir_stateful():
        st = st0.clone()
        y = ops.iir(xi, a, b, state=st, mode="scan")
        return y, st

    # serial answers, default stream
    want_a, want_b = ops.fft0(xa), ops.fft0(xb)
    want_y, want_st = iir_stateful()
    want_f, want_h = fir_stream()
    torch.cuda.synchronize()

    rounds = 6
    errors, got = [], {}

    def worker(name, fn):
        try:
            s = torch.cuda.Stream(device=dev)
            res = None
            with torch.cuda.stream(s):
                for _ in range(rounds):
                    res = fn()
            s.synchronize()
            got[name] = res
        except BaseException as e:  # noqa: BLE001
            errors.append((name, e))

    threads = [threading.Thread(target=worker, args=("a", lambda: ops.fft0(xa))),
               threading.Thread(target=worker, args=("b", lambda: ops.fft0(xb))),
               threading.Thread(target=worker, args=("c", lambda: (iir_stateful(), fir_stream())))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    torch.cuda.synchronize()
    assert torch.equal(got["a"], want_a)
    assert torch.equal(got["b"], want_b)
    (y, st), (f, h) = got["c"]
    assert torch.equal(y, want_y) and torch.equal(st, want_st)
    assert torch.equal(f, want_f) and torch.equal(h, want_h)


def test_small_transforms_overlapping_on_three_streams():
    """Kernels that do NOT fill the chip really run side by side: 2 x 2^17 points per call (64 workgroups a pass, three
    passes through the scratch slab), 150 calls per stream, every output compared on the device.  With one workspace per
    device (DSP_GPU_SHARED_WORKSPACES=1, the library until round 2) this test fails: a stream's pass reads what another
    stream's pass has just written."""
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(33)
    xs = [torch.view_as_complex(torch.randn(2, 1 << 17, 2, device=dev, generator=g)) for _ in range(3)]
    wants = [ops.fft0(x) for x in xs]
    torch.cuda.synchronize()
    bad = [torch.zeros((), dtype=torch.bool, device=dev) for _ in range(3)]
    errors = []
    start = threading.Barrier(3)

    def worker(i):
        try:
            s = torch.cuda.Stream(device=dev)
            start.wait()
            with torch.cuda.stream(s):
                # hold the stream back while its 150 calls are queued, so that the three queues then drain side by side
                # (launched one by one from Python the kernels would finish before the next stream's arrive)
                torch.cuda._sleep(200_000_000)
                for _ in range(150):
                    bad[i] |= (ops.fft0(xs[i]) != wants[i]).any()
            s.synchronize()
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    torch.cuda.synchronize()
    assert not any(bool(b.item()) for b in bad)


def test_more_streams_than_workspace_sets():
    """A ninth stream takes over the least recently used set (after a device sync): still the right answers."""
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(32)
    x = torch.view_as_complex(torch.randn(2, 1 << 17, 2, device=dev, generator=g))
    want = ops.fft0(x)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(11)]
    outs = []
    for _ in range(2):
        for s in streams:
            with torch.cuda.stream(s):
                outs.append(ops.fft0(x))
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, want)


def test_host_calls_from_two_threads():
    """DSP_GPU_MEM_HOST calls from two host threads on one device (they serialise inside the library)."""
    _dev()
    rng = np.random.default_rng(5)
    xs = [(rng.standard_normal((64, 4096)) + 1j * rng.standard_normal((64, 4096))).astype(np.complex64) for _ in range(2)]
    want = [ops.fft0(x) for x in xs]
    got, errors = [None, None], []

    def worker(i):
        try:
            for _ in range(5):
                got[i] = ops.fft0(xs[i])
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for i in range(2):
        assert np.array_equal(got[i], want[i])


def test_overlap_save_draw_counters_are_per_stream(monkeypatch):
    """The one-shot overlap-save FIR's waves draw their block pairs from counters that a memset clears in front of every
    launch; those counters belong to the (device, stream) workspace set.  Two streams run the wave kernel (forced onto
    inputs this small) back to back, 40 calls each, on different inputs, every output compared bit for bit with the same call
    made alone.  (With shared counters one stream's memset could land in the middle of the other's launch; on this chip
    the two persistent launches mostly take turns, so the test guards the results rather than proving the hazard.)"""
    dev = _dev()
    monkeypatch.setenv("DSP_GPU_OS_WAVE_MIN_TILES", "1")
    g = torch.Generator(device=dev).manual_seed(77)
    taps = design.sinc(0.1, 511)
    xs = [torch.randn(48, 1 << 16, device=dev, generator=g) for _ in range(2)]
    want = [ops.convolve(x, taps, "overlap_save") for x in xs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    outs = [[torch.full_like(xs[i], 7.0) for _ in range(40)] for i in range(2)]
    for k in range(40):
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                ops.convolve(xs[i], taps, "overlap_save", out=outs[i][k])
    torch.cuda.synchronize()
    for i in range(2):
        for o in outs[i]:
            assert torch.equal(o, want[i])
