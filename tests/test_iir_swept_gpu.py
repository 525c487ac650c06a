"""GPU parity of the band-pass retuned at every sample (dsp_gpu_biquad_swept) through the C ABI.

The path: eval/signals/wind_generator.h:150-167 -- per sample bandPassFilter(fc[t], bw) (recursive_filters.h:165-186),
a and b copied into a BiQuad that keeps its history, one BiQuad step (coefficients.h:172-182).  The oracle restates that
loop (pinned bit for bit against the reference's own designer + BiQuad: test_oracle_vs_ref.py, and against the golden
vectors the reference build produced: test_oracle_golden.py).

Tolerance.  The kernel performs the reference's operations in the reference's order, in double, without fused
multiply-adds; the one thing that can differ from the CPU is the last bit of cos().  A one-ulp change of cos moves
K = (1 - 2Rc + R^2)/(2 - 2c) by ~1e-16/(2 pi fc)^2 relative, far below the fp32 the output is rounded to, so the outputs
agree except where a double lands within that distance of an fp32 rounding boundary.  The tests therefore require:
every sample within ONE fp32 ulp, at most 1 sample in 1,000 differing at all, relative L2 <= 1e-7.
"""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu
BW = float(np.float32(60) / np.float32(44100))  # wind_generator.h:159


@pytest.fixture(scope="module")
def ops(dev):
    from dsp_amd import ops as _ops
    return _ops


def sweeps(rng, channels, n):
    """Smooth centre-frequency sweeps, one per channel, as a wind speed would drive them (a recursion whose
    coefficients jump at random from sample to sample is not a stable filter, in the reference either)."""
    t = np.arange(n, dtype=np.float64)
    base = rng.uniform(0.01, 0.2, size=(channels, 1))
    depth = rng.uniform(0.0, 0.1, size=(channels, 1))
    rate = rng.uniform(0.001, 0.02, size=(channels, 1))
    phase = rng.uniform(0, 2 * np.pi, size=(channels, 1))
    return (base + depth * (0.5 + 0.5 * np.sin(rate * t + phase))).astype(np.float32)


def close_enough(got, want):
    got, want = np.asarray(got, dtype=np.float32), np.asarray(want, dtype=np.float32)
    assert got.shape == want.shape and np.isfinite(got).all()
    ulp = np.spacing(np.abs(want).astype(np.float32))
    diff = np.abs(got.astype(np.float64) - want.astype(np.float64))
    assert (diff <= ulp).all(), f"max error {float((diff / ulp).max()):.2f} ulp"
    assert np.count_nonzero(diff) <= max(1, got.size // 1000), f"{np.count_nonzero(diff)} of {got.size} samples differ"
    assert rel_l2(got, want.astype(np.float64)) <= 1e-7


@pytest.mark.parametrize("tag", ["wh0", "wh1"])
def test_golden_whistling(ops, dev, golden, tag):
    """The two Whistling voices of eval/signals/main.cpp:33-34; expected outputs produced by the reference build."""
    import torch
    x, fc = golden["iir_noise_in"], golden[f"swept_{tag}_fc"]
    bw = float(golden["swept_bw"][0])
    st = torch.zeros(1, 4, dtype=torch.float64, device=dev)
    got = ops.biquad_swept(torch.from_numpy(x).to(dev), torch.from_numpy(fc).to(dev), bw, state=st).cpu().numpy()
    close_enough(got, golden[f"swept_{tag}_y32"])
    assert np.allclose(st.cpu().numpy()[0], golden[f"swept_{tag}_state"], rtol=1e-9, atol=1e-12)
    host = ops.biquad_swept(x, fc, bw)  # host-pointer route
    assert np.array_equal(host, got)


def test_golden_bandreject(ops, dev, golden):
    import torch
    x, fc = golden["iir_noise_in"], golden["swept_wh0_fc"]
    got = ops.biquad_swept(torch.from_numpy(x).to(dev), torch.from_numpy(fc).to(dev), float(golden["swept_bw"][0]),
                           kind="bandreject").cpu().numpy()
    close_enough(got, golden["swept_wh0_reject_y32"])


@pytest.mark.parametrize("channels,n", [(1, 1), (1, 63), (3, 64), (64, 65), (70, 1001), (129, 4096), (200, 130)])
@pytest.mark.parametrize("kind", ["bandpass", "bandreject"])
def test_ragged_shapes(ops, dev, orc, channels, n, kind):
    """Partial tiles in time, partial groups of channels, rows that are not 16-byte aligned."""
    import torch
    rng = np.random.default_rng(channels * 7919 + n)
    x = rng.standard_normal((channels, n)).astype(np.float32)
    fc = sweeps(rng, channels, n)
    s_in = rng.standard_normal((channels, 4)) * 0.1
    s_orc = s_in.copy()
    want = orc.biquad_swept(x, fc, BW, kind, state=s_orc)
    st = torch.from_numpy(s_in.copy()).to(dev)
    got = ops.biquad_swept(torch.from_numpy(x).to(dev), torch.from_numpy(fc).to(dev), BW, kind=kind, state=st)
    close_enough(got.cpu().numpy(), want)
    assert np.allclose(st.cpu().numpy(), s_orc, rtol=1e-9, atol=1e-12)


def test_one_sweep_for_all_channels(ops, dev, orc):
    import torch
    rng = np.random.default_rng(5)
    x = rng.standard_normal((130, 3000)).astype(np.float32)
    t = np.arange(3000, dtype=np.float32)
    fc = ((np.float32(0.5) + np.float32(0.5) * np.sin(t / np.float32(150))) * np.float32(1000) + np.float32(1000)) / np.float32(44100)
    fc = fc.astype(np.float32)
    got = ops.biquad_swept(torch.from_numpy(x).to(dev), torch.from_numpy(fc).to(dev), BW).cpu().numpy()
    close_enough(got, orc.biquad_swept(x, fc, BW))
    each = ops.biquad_swept(torch.from_numpy(x).to(dev), torch.from_numpy(np.broadcast_to(fc, x.shape).copy()).to(dev), BW)
    assert np.array_equal(got, each.cpu().numpy())  # the shared sweep is the same arithmetic


@pytest.mark.parametrize("channels,n", [(1, 5), (70, 1001), (64, 128), (65, 257), (3, 4096)])
def test_one_sweep_ragged_shapes_and_state(ops, dev, orc, channels, n):
    """The shared-sweep kernel (coefficients designed once per wave and sample) on partial tiles, partial channel
    groups, unaligned rows, with state in and out: the oracle, and the per-channel kernel bit for bit."""
    import torch
    rng = np.random.default_rng(channels * 31 + n)
    x = rng.standard_normal((channels, n)).astype(np.float32)
    fc = sweeps(rng, 1, n)[0]
    s_in = rng.standard_normal((channels, 4)) * 0.1
    s_orc = s_in.copy()
    want = orc.biquad_swept(x, fc, BW, "bandreject", state=s_orc)
    st = torch.from_numpy(s_in.copy()).to(dev)
    xd = torch.from_numpy(x).to(dev)
    got = ops.biquad_swept(xd, torch.from_numpy(fc).to(dev), BW, kind="bandreject", state=st)
    close_enough(got.cpu().numpy(), want)
    assert np.allclose(st.cpu().numpy(), s_orc, rtol=1e-9, atol=1e-12)
    st2 = torch.from_numpy(s_in.copy()).to(dev)
    each = ops.biquad_swept(xd, torch.from_numpy(np.broadcast_to(fc, x.shape).copy()).to(dev), BW, kind="bandreject", state=st2)
    assert torch.equal(got, each) and torch.equal(st, st2)


def test_misaligned_base_pointers(ops, dev, orc):
    """x, fc and y that start 4 bytes off a 16-byte boundary (views into larger buffers) take the element-wise path."""
    import torch
    rng = np.random.default_rng(12)
    ch, n = 66, 512
    x = rng.standard_normal((ch, n)).astype(np.float32)
    fc = sweeps(rng, ch, n)
    bx = torch.zeros(ch * n + 1, device=dev)
    bf = torch.zeros(ch * n + 1, device=dev)
    by = torch.zeros(ch * n + 1, device=dev)
    bx[1:] = torch.from_numpy(x).to(dev).flatten()
    bf[1:] = torch.from_numpy(fc).to(dev).flatten()
    xv, fv, yv = bx[1:].view(ch, n), bf[1:].view(ch, n), by[1:].view(ch, n)
    assert xv.data_ptr() % 16 == 4
    ops.biquad_swept(xv, fv, BW, out=yv)
    close_enough(yv.cpu().numpy(), orc.biquad_swept(x, fc, BW))
    ops.biquad_swept(xv, fv[0], BW, out=yv)  # one sweep for all, same misalignment
    close_enough(yv.cpu().numpy(), orc.biquad_swept(x, fc[0], BW))


def test_blocks_with_carried_state_are_bit_exact(ops, dev):
    """Fed in audio-callback sized blocks (256 frames, eval/signals/main.cpp) with the BiQuad's members carried,
    the output is the unblocked one bit for bit."""
    import torch
    rng = np.random.default_rng(6)
    ch, n = 96, 256 * 9 + 100
    x = torch.from_numpy(rng.standard_normal((ch, n)).astype(np.float32)).to(dev)
    fc = torch.from_numpy(sweeps(rng, ch, n)).to(dev)
    whole_state = torch.zeros(ch, 4, dtype=torch.float64, device=dev)
    whole = ops.biquad_swept(x, fc, BW, state=whole_state)
    st = torch.zeros(ch, 4, dtype=torch.float64, device=dev)
    parts = []
    for t0 in range(0, n, 256):
        parts.append(ops.biquad_swept(x[:, t0:t0 + 256].contiguous(), fc[:, t0:t0 + 256].contiguous(), BW, state=st))
    assert torch.equal(torch.cat(parts, dim=1), whole)
    assert torch.equal(st, whole_state)


def test_constant_sweep_is_the_fixed_bandpass(ops, dev, orc):
    """With fc constant the retuned filter IS bandPassFilter(fc, bw) run by dsp_gpu_iir (linearity of the path in
    its input holds too: the design does not depend on x)."""
    import torch
    rng = np.random.default_rng(8)
    x = rng.standard_normal((66, 2048)).astype(np.float32)
    fc = np.full((66, 2048), np.float32(0.05), dtype=np.float32)
    a, b, c0, d0 = orc.biquad_design("bandpass", float(np.float32(0.05)), BW)
    xd, fd = torch.from_numpy(x).to(dev), torch.from_numpy(fc).to(dev)
    got = ops.biquad_swept(xd, fd, BW).cpu().numpy()
    fixed = ops.iir(xd, a, b, c0, d0, mode="batch").cpu().numpy()
    assert rel_l2(got, fixed.astype(np.float64)) <= 2e-6  # dsp_gpu_iir fuses its multiply-adds
    y2 = ops.biquad_swept(2 * xd, fd, BW).cpu().numpy()
    assert np.array_equal(y2, 2 * got)  # scaling by a power of two is exact in every operation


def test_in_place_and_aliasing(ops, dev):
    import ctypes as C
    import torch
    from dsp_amd import capi
    rng = np.random.default_rng(9)
    x = torch.from_numpy(rng.standard_normal((64, 512)).astype(np.float32)).to(dev)
    fc = torch.from_numpy(sweeps(rng, 64, 512)).to(dev)
    want = ops.biquad_swept(x, fc, BW)
    xin = x.clone()
    assert torch.equal(ops.biquad_swept(xin, fc, BW, out=xin), want)   # y == x
    fin = fc.clone()
    assert torch.equal(ops.biquad_swept(x, fin, BW, out=fin), want)    # y == fc, one sweep per channel
    L = capi.lib()
    vp = C.c_void_p
    p = x.data_ptr()
    # shifted overlap of x and y; y over a shared sweep
    assert L.dsp_gpu_biquad_swept(vp(p), vp(fc.data_ptr()), vp(p + 64), 64, 256, 256, 0, BW, None, 1, None) == capi.EINVAL
    assert L.dsp_gpu_biquad_swept(vp(p), vp(fc.data_ptr()), vp(fc.data_ptr()), 64, 512, 0, 0, BW, None, 1, None) == capi.EINVAL
    # a shared sweep that merely sits next to y (inside y's length of it, not inside y) is fine
    buf = torch.zeros(64 * 512 + 512, device=dev)
    buf[:512] = fc[0]
    ynear = buf[512:].view(64, 512)
    assert torch.equal(ops.biquad_swept(x, buf[:512], BW, out=ynear), ops.biquad_swept(x, fc[0].contiguous(), BW))


def test_full_size_properties(ops, dev, orc):
    """A chip-filling batch (65,536 channels x 1,024): a few channels against the oracle, and the whole batch through
    its size-independent properties -- blocked == unblocked bit for bit, exact power-of-two scaling."""
    import torch
    ch, n = 65536, 1024
    g = torch.Generator(device=dev).manual_seed(31)
    x = torch.randn(ch, n, generator=g, device=dev, dtype=torch.float32)
    t = torch.arange(n, device=dev, dtype=torch.float32)
    rate = torch.rand(ch, 1, generator=g, device=dev) * 0.02 + 0.001
    phase = torch.rand(ch, 1, generator=g, device=dev) * 6.28
    fc = (0.05 + 0.04 * torch.sin(rate * t + phase)).contiguous()
    st = torch.zeros(ch, 4, dtype=torch.float64, device=dev)
    y = ops.biquad_swept(x, fc, BW, state=st)
    pick = [0, 1, 63, 64, 4097, 32768, 65535]
    want = orc.biquad_swept(x[pick].cpu().numpy(), fc[pick].cpu().numpy(), BW)
    close_enough(y[pick].cpu().numpy(), want)
    st2 = torch.zeros(ch, 4, dtype=torch.float64, device=dev)
    halves = [ops.biquad_swept(x[:, :384].contiguous(), fc[:, :384].contiguous(), BW, state=st2),
              ops.biquad_swept(x[:, 384:].contiguous(), fc[:, 384:].contiguous(), BW, state=st2)]
    assert torch.equal(torch.cat(halves, dim=1), y) and torch.equal(st2, st)
    assert torch.equal(ops.biquad_swept(x * 0.5, fc, BW), y * 0.5)


def test_zero_frequency_divides_by_zero_like_the_reference(ops, dev, orc):
    """bandPassFilter(0, bw) divides by 2 - 2 cos(0) = 0 (recursive_filters.h:170): K = +inf, and the recursion turns infinite, then NaN,
    from that sample on -- in the reference, in the oracle and here, at the same samples; other channels are untouched."""
    import torch
    rng = np.random.default_rng(3)
    x = rng.standard_normal((3, 200)).astype(np.float32)
    fc = sweeps(rng, 3, 200)
    fc[1, 77] = 0.0
    want = orc.biquad_swept(x, fc, BW)
    got = ops.biquad_swept(torch.from_numpy(x).to(dev), torch.from_numpy(fc).to(dev), BW).cpu().numpy()
    assert np.isfinite(want[1, :77]).all() and not np.isfinite(want[1, 77:]).any()
    bad = ~np.isfinite(want)
    assert np.array_equal(~np.isfinite(got), bad) and np.array_equal(got[bad], want[bad], equal_nan=True)  # same inf / NaN
    close_enough(got[[0, 2]], want[[0, 2]])
    close_enough(got[1, :77], want[1, :77])


def test_errors(ops, dev):
    import torch
    from dsp_amd import capi
    x = torch.zeros(4, 64, device=dev)
    fc = torch.full((4, 64), 0.1, device=dev)
    with pytest.raises(capi.DspGpuError):
        ops.biquad_swept(x, fc[:, :32].contiguous(), BW)               # neither x's shape nor one sweep
    with pytest.raises(capi.DspGpuError):
        ops.biquad_swept(x, fc, BW, kind="lowshelf")
    with pytest.raises(capi.DspGpuError):
        ops.biquad_swept(x, fc, float("nan"))
    with pytest.raises(capi.DspGpuError):
        ops.biquad_swept(x, fc, BW, state=torch.zeros(4, 3, dtype=torch.float64, device=dev))
    with pytest.raises(capi.DspGpuError):
        ops.biquad_swept(x, fc.cpu().numpy(), BW)                      # fc must live where x lives


@pytest.mark.gpu
def test_few_long_voices_are_sharded_in_time(dev, monkeypatch):
    """One voice (the wind generator's shape) and a few voices with their own sweeps: segments started from states that a warm-up
    of 2 * 44 ln 2 / -ln R samples leaves (iir_swept_kernels.hip) against one lane per voice (DSP_GPU_SWEPT_NO_SHARDS=1): outputs
    to fp32 rounding, the state handed back, chunked == unchunked, a remainder, sweeps through the whole band."""
    import torch
    from dsp_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    for channels, n, bw in ((1, 300000 + 77, 0.01), (1, 1 << 18, 60.0 / 44100.0), (3, 1 << 17, 0.02), (2, 200000, 0.005), (2, 99999, 0.02)):
        x = torch.randn(channels, n, device=dev, generator=g)
        t = torch.arange(n, device=dev, dtype=torch.float32) / n
        fc = (0.02 + 0.45 * (0.5 + 0.5 * torch.sin(2 * np.pi * (3 + torch.arange(channels, device=dev)[:, None]) * t[None, :]))).contiguous()
        for kind in ("bandpass", "bandreject"):
            st = torch.zeros(channels, 4, dtype=torch.float64, device=dev)
            h = (n // 2) // 128 * 128
            y0 = ops.biquad_swept(x[:, :h].contiguous(), fc[:, :h].contiguous(), bw, kind=kind, state=st)
            y1 = ops.biquad_swept(x[:, h:].contiguous(), fc[:, h:].contiguous(), bw, kind=kind, state=st)
            whole = ops.biquad_swept(x, fc, bw, kind=kind)
            monkeypatch.setenv("DSP_GPU_SWEPT_NO_SHARDS", "1")
            st_seq = torch.zeros_like(st)
            seq = ops.biquad_swept(x, fc, bw, kind=kind, state=st_seq)
            monkeypatch.delenv("DSP_GPU_SWEPT_NO_SHARDS")
            den = float(torch.linalg.vector_norm(seq.double()))
            assert float(torch.linalg.vector_norm((whole - seq).double())) <= 2e-6 * den, (channels, n, bw, kind)
            assert float(torch.linalg.vector_norm((torch.cat([y0, y1], dim=1) - seq).double())) <= 2e-6 * den, (channels, n, bw, kind)
            assert float((st - st_seq).abs().max()) <= 1e-5 * max(float(st_seq.abs().max()), 1.0), (channels, n, bw, kind)
