"""BASELINE config 1 ("eval/fft_pairs"): 1024-point FFT -> IFFT round trip over frames of
resources/sine.wav, one channel.  The reference's eval/fft_pairs program is a GUI that does not do this
(SURVEY.md Q13), so this is the build's own headless harness: real -> complex through dsp::fft
(fft.h:57-83), FORWARD then INVERSE, divide by 1024 (nothing normalises, fft.h:40-54), compare with the
input.  CPU: the reference path (oracle).  GPU: the same frames through the C ABI.

The reference's own resources/sine.wav (2 ch, 16-bit PCM, 44.1 kHz) is used where /root/reference exists.  Elsewhere (the
GPU box: the reference cannot travel) the frames come from tests/golden/sine_wav_ch0_head.npz, the first 128 frames of that
file's channel 0 as the integers it holds (data, made by tests/golden/make_sine_fixture.py), written out and read back
through the same WAV reader.
"""
import os

import numpy as np
import pytest

from conftest import rel_l2
from dsp_amd import wav

REF_WAV = "/root/reference/resources/sine.wav"
FRAME = 1024


@pytest.fixture(scope="module")
def frames(tmp_path_factory):
    if os.path.exists(REF_WAV):
        x, rate = wav.read(REF_WAV)
        assert x.shape[1] == 2 and rate == 44100
    else:
        fix = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sine_wav_ch0_head.npz"))
        s = fix["pcm16"].astype(np.float64) / 32768.0
        p = tmp_path_factory.mktemp("wav") / "sine.wav"
        wav.write_pcm16(str(p), np.stack([s, s], axis=1), int(fix["rate"]))
        x, rate = wav.read(str(p))
        assert np.array_equal(np.round(x[:, 0].astype(np.float64) * 32768.0).astype(np.int16), fix["pcm16"])
    ch0 = x[:, 0]
    n = (ch0.size // FRAME) * FRAME
    return ch0[:n].reshape(-1, FRAME)


def test_wav_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    x = (rng.uniform(-0.9, 0.9, (1000, 2))).astype(np.float32)
    wav.write_pcm16(str(tmp_path / "a.wav"), x, 48000)
    y, rate = wav.read(str(tmp_path / "a.wav"))
    assert rate == 48000 and y.shape == x.shape
    assert np.abs(y - x).max() <= 1.0 / 32768 + 1e-7


def test_reference_cpu_path(frames, orc):
    sub = frames[:: max(1, len(frames) // 200)]  # ~200 frames keep the CPU suite short
    worst = 0.0
    for fr in sub:
        spec = orc.fft_real(fr.astype(np.float64), FRAME)
        back = orc.fft0(spec, inverse=True) / FRAME
        assert np.abs(back.imag).max() < 1e-12
        worst = max(worst, np.abs(back.real - fr).max())
    assert worst < 1e-12  # double round trip: ~1e-14 ("expect ~1e-16 .. 1e-14 in double")
    # fp32 instantiation of the same algorithm (the like-for-like CPU baseline)
    spec32 = orc.fft0_f32(sub.astype(np.complex64))
    back32 = orc.fft0_f32(spec32, inverse=True) / FRAME
    assert rel_l2(back32.real, sub) <= 1e-5 * 10


@pytest.mark.gpu
def test_gpu_round_trip(frames, orc, dev):
    import torch
    from dsp_amd import ops
    x = torch.from_numpy(np.ascontiguousarray(frames)).to(dev)
    spec = ops.fft(x, FRAME)                     # all frames in one batched call
    back = ops.fft0(spec, inverse=True) / FRAME
    assert float(back.imag.abs().max()) < 1e-5
    err = torch.linalg.vector_norm((back.real - x).double()) / torch.linalg.vector_norm(x.double())
    assert float(err) <= 1e-5 * 10
    # spectrum of a few frames against the reference path
    for i in (0, len(frames) // 2, len(frames) - 1):
        want = orc.fft_real(frames[i].astype(np.float64), FRAME)
        assert rel_l2(spec[i].cpu().numpy(), want) <= 1e-5 * 10
