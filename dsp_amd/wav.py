"""Minimal RIFF/WAVE reader and writer (PCM 16-bit / 32-bit float) -- host plumbing.

The reference ingests audio through the choc submodule (dsp/src/audio_test.h:138-147), which is
not vendored; BASELINE config 1 only needs frames of resources/sine.wav (2 ch, 16-bit PCM,
44.1 kHz), so this is the whole of it.  numpy only.
"""
from __future__ import annotations

import struct

import numpy as np


def read(path: str):
    """-> (samples float32 [frames, channels] in [-1, 1), sample_rate)"""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        tag, size = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        body = data[pos + 8:pos + 8 + size]
        if tag == b"fmt ":
            fmt = struct.unpack_from("<HHIIHH", body, 0)
        elif tag == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    code, channels, rate, _, _, bits = fmt
    if code == 1 and bits == 16:
        x = np.frombuffer(pcm, dtype="<i2").astype(np.float32) / np.float32(32768.0)
    elif code == 3 and bits == 32:
        x = np.frombuffer(pcm, dtype="<f4").astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported format code {code}, {bits} bits")
    frames = x.size // channels
    return x[:frames * channels].reshape(frames, channels), rate


def write_pcm16(path: str, samples, rate: int) -> None:
    x = np.atleast_2d(np.asarray(samples, dtype=np.float32))
    if x.shape[0] < x.shape[1] and x.ndim == 2 and x.shape[0] <= 8:
        x = x.T  # accept [channels, frames]
    pcm = np.clip(np.round(x * 32768.0), -32768, 32767).astype("<i2").tobytes()
    channels = x.shape[1]
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, channels, rate, rate * channels * 2, channels * 2, 16))
        f.write(b"data" + struct.pack("<I", len(pcm)) + pcm)
