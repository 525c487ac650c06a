#!/usr/bin/env python3
"""511-tap FIR on many channels of 1,024 ... 65,536 samples (2^28 samples per call): overlap-save, direct and what AUTO takes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd import design, ops  # noqa: E402

dev = torch.device("cuda:0")
TOTAL = 1 << 28
buf = torch.randn(TOTAL, device=dev)
out = torch.empty_like(buf)
taps = design.sinc(0.1, 511)
lengths = [int(v) for v in sys.argv[1:]] or [1024, 2048, 3072, 4096, 6000, 8192, 10000, 16384, 32768, 65536, 262144]
for n in lengths:
    ch = TOTAL // n
    x, y = buf[: ch * n].view(ch, n), out[: ch * n].view(ch, n)
    line = f"{ch:7d} x {n:6d}:"
    for mode in ("overlap_save", "direct", "auto"):
        for _ in range(5):
            ops.convolve(x, taps, mode=mode, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.convolve(x, taps, mode=mode, out=y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += f"  {mode} {ms:7.3f} ms ({ch * n * 8 / ms / 8e9:5.3f})"
    print(line, flush=True)
