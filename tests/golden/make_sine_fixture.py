#!/usr/bin/env python3
"""Fixture for BASELINE config 1 on machines without /root/reference (the GPU box): the first 128 frames of 1024 samples of
channel 0 of the reference's resources/sine.wav (2 ch, 16-bit PCM, 44.1 kHz), as the 16-bit integers the file holds.
Data only -- inputs of a test, read with dsp_amd/wav.py; run here, where the reference exists:

    python tests/golden/make_sine_fixture.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dsp_amd import wav  # noqa: E402

x, rate = wav.read("/root/reference/resources/sine.wav")
assert x.shape[1] == 2 and rate == 44100
head = x[: 128 * 1024, 0]
pcm = np.round(head.astype(np.float64) * 32768.0).astype(np.int16)
assert np.array_equal(pcm.astype(np.float32) / 32768.0, head.astype(np.float32)), "wav.read scales by 1/32768"
out = os.path.join(ROOT, "tests", "golden", "sine_wav_ch0_head.npz")
np.savez_compressed(out, pcm16=pcm, rate=np.int32(rate), frames_in_file=np.int64(x.shape[0]))
print(out, os.path.getsize(out), "bytes;", pcm.size, "samples of", x.shape[0])
