"""Headless versions of the reference's two evaluation programs: the numeric part of each GUI demo,
run through the GPU entry points and checked against the oracle running the reference's own sequence of calls.

  * eval/recursive_filters/main.cpp:72-121  (applyFilter + createDate): 1024-sample impulse at N/2 -> one of the
    seven recursive filters -> dsp::fft -> |.|
  * eval/windowed_sinc_filter/main.cpp:49-124 (createSignal + create): windowed-sinc kernel -> dsp::fft -> |.|, the
    five-tone test signal (2, 10, 50, 250, 450 Hz at 1 kHz) -> SincFilter::apply -> dsp::fft of the part after
    the filter delay -> |.|
The filters are designed on the host (dsp_amd.design, bit-identical to the reference: tests/test_design_host.py).
"""
import numpy as np
import pytest

from conftest import rel_l2

N = 1024
SAMPLE_RATE = 1000


def recursive_designs(design, frequency=0.25, bandwidth=0.006, gain=1.0):
    """applyFilter's seven branches (recursive_filters/main.cpp:72-101) at the demo's default settings (:44-51)."""
    lp = design.low_pass(frequency, 2)
    hp = design.high_pass(frequency, 2)
    return {
        "band_pass": design.band_pass(frequency, bandwidth),
        "band_reject": design.band_reject(frequency, bandwidth),
        "low_pass": (lp[0], lp[1], 1.0, 0.0),
        "high_pass": (hp[0], hp[1], 1.0, 0.0),
        "low_shelf": design.low_shelf(frequency, gain),
        "high_shelf": design.high_shelf(frequency, gain),
        "peaking": design.peaking(frequency, gain, bandwidth),
    }


def five_tones(padding: int):
    """createSignal (windowed_sinc_filter/main.cpp:49-66)."""
    period = 1.0 / SAMPLE_RATE
    t = period * np.arange(SAMPLE_RATE + padding) - period * padding
    sig = np.zeros_like(t)
    for f, amp in ((2, 1.0), (10, 0.5), (50, 0.25), (250, 0.125), (450, 0.0625)):
        sig += np.cos(2 * np.pi * f * t) * amp
    return sig


def test_recursive_recipe_on_the_oracle(orc):
    """The recipe itself, CPU only: the band-pass response peaks at its centre bin, the band-reject one dips there."""
    from dsp_amd import design
    impulse = np.zeros(N)
    impulse[N // 2] = 1.0
    mags = {}
    for name, (a, b, c0, d0) in recursive_designs(design).items():
        ir = orc.iir(impulse, a, b, c0, d0)
        mags[name] = np.abs(orc.fft_real(ir, N))
    k = int(0.25 * N)
    assert np.argmax(mags["band_pass"][: N // 2]) == k
    assert np.argmin(mags["band_reject"][: N // 2]) == k
    assert mags["low_pass"][1] > 0.9 and mags["low_pass"][N // 2 - 1] < 0.1
    assert mags["high_pass"][1] < 0.1 and mags["high_pass"][N // 2 - 1] > 0.9


@pytest.mark.gpu
@pytest.mark.parametrize("frequency,bandwidth,gain", [(0.25, 0.006, 1.0), (0.1, 0.02, 6.0), (0.4, 0.001, -9.0)])
def test_recursive_filter_demo(orc, dev, frequency, bandwidth, gain):
    import torch
    from dsp_amd import design, ops
    impulse = np.zeros(N, dtype=np.float32)
    impulse[N // 2] = 1.0
    xd = torch.from_numpy(impulse).to(dev)
    designs = recursive_designs(design, frequency, bandwidth, gain)
    designs["peaking_q2"] = design.peaking(frequency, gain, 2.0)  # a usable Q next to the demo's
    for name, (a, b, c0, d0) in designs.items():
        ir = ops.iir(xd, a, b, c0, d0)                       # filter(input)
        mag = ops.fft(ir, N).abs().cpu().numpy()             # dsp::fft(...), abs(c)
        want_ir = orc.iir(impulse.astype(np.float64), a, b, c0, d0)
        if not np.isfinite(want_ir).all() or np.abs(want_ir).max() > 1e30:
            # the demo hands its `bandwidth` slider to peakingFilter as Q (main.cpp:96-97): at 0.006 the section is
            # unstable and the reference's own response overflows; the device must blow up the same way, not hide it
            got = ir.cpu().numpy()
            assert name == "peaking" and (not np.isfinite(got).all() or np.abs(got).max() > 1e30)
            continue
        want = np.abs(orc.fft_real(want_ir, N))
        assert rel_l2(ir.cpu().numpy(), want_ir) <= 1e-4, name   # north_star: IIR <= 1e-4
        assert rel_l2(mag, want) <= 1e-4 + 1e-5 * 10, name        # + the FFT's 1e-5 * log2(N)


@pytest.mark.gpu
@pytest.mark.parametrize("cutoff,length,window,inversion", [(0.1, 128, "blackman", "none"), (0.05, 300, "blackman", "inversion"),
                                                            (0.2, 511, "hamming", "none"), (0.1, 128, "identity", "reversal")])
def test_windowed_sinc_demo(orc, dev, cutoff, length, window, inversion):
    import torch
    from dsp_amd import design, ops
    kernel = design.sinc(cutoff, length, window=window, inversion=inversion)
    assert np.array_equal(kernel, orc.sinc(cutoff, length, window, inversion)[: kernel.size])
    # frequency response of the kernel: dsp::fft(kernel) zero-padded to the next power of two (:83-86)
    nk = 1 << int(np.ceil(np.log2(length)))
    kd = torch.from_numpy(kernel.astype(np.float32)).to(dev)
    mag = ops.fft(kd, nk).abs().cpu().numpy()
    want_mag = np.abs(orc.fft_real(kernel, nk))
    assert rel_l2(mag, want_mag) <= 1e-5 * np.log2(nk) + 1e-6
    # five tones through SincFilter::apply, spectrum of what follows the filter delay (:101-118)
    sig = five_tones(kernel.size)
    sd = torch.from_numpy(sig.astype(np.float32)).to(dev)
    for mode in ("direct", "overlap_save"):
        y = ops.convolve(sd, kernel, mode=mode)
        want_y = orc.fir(sig, kernel)
        assert rel_l2(y.cpu().numpy(), want_y) <= (1e-6 if mode == "direct" else 1e-5 * 12) + 2e-7  # + fp32 input rounding
        tail = y[kernel.size:].contiguous()
        nt = 1 << int(np.ceil(np.log2(tail.numel())))
        spec = ops.fft(tail, nt).abs().cpu().numpy()
        want_spec = np.abs(orc.fft_real(want_y[kernel.size:], nt))
        assert rel_l2(spec, want_spec) <= 1e-5 * np.log2(nt) + 1e-5 * 12
    if inversion == "none":  # low-pass: the 2 Hz tone passes, the 450 Hz one is gone
        k2, k450 = round(2 * nt / SAMPLE_RATE), round(450 * nt / SAMPLE_RATE)
        if cutoff * SAMPLE_RATE < 400:
            assert spec[k450 - 2:k450 + 3].max() < 1e-2 * spec[max(k2 - 2, 0):k2 + 3].max()
