#!/usr/bin/env python3
"""Moving average and overlap-save FIR (511 points / taps) over channel lengths that are not powers of two, ~2^30 samples."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd import design, ops  # noqa: E402

dev = torch.device("cuda:0")
buf = torch.randn(1 << 30, device=dev)
out = torch.empty_like(buf)
taps = design.sinc(0.1, 511)
for ch, n in ((1024, 1 << 20), (1024, 1000000), (1365, 786432), (869, 1234568), (10000, 100000), (100000, 10000), (262144, 4096), (250000, 4000)):
    x, y = buf[: ch * n].view(ch, n), out[: ch * n].view(ch, n)
    for name, fn in (("mavg", lambda: ops.moving_average(x, 511, out=y)), ("fir_os", lambda: ops.convolve(x, taps, mode="overlap_save", out=y))):
        for _ in range(10):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{name:6s} {ch:7d} x {n:8d}: {ms:7.4f} ms {ch * n * 8 / ms / 8e9:5.3f} of 8 TB/s", flush=True)
