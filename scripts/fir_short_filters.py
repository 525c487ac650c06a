#!/usr/bin/env python3
"""Short filters (7 / 31 / 63 taps) over channel shapes, 2^28 samples per call: what AUTO takes and its rate (8 B/sample of 8 TB/s)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd import design, ops  # noqa: E402

dev = torch.device("cuda:0")
TOTAL = 1 << 28
buf = torch.randn(TOTAL, device=dev)
out = torch.empty_like(buf)
for n in (64, 100, 256, 1000, 1024, 4096, 10000, 65536, 1 << 20):
    ch = TOTAL // n
    x, y = buf[: ch * n].view(ch, n), out[: ch * n].view(ch, n)
    line = f"{ch:8d} x {n:8d}:"
    for ntaps in (7, 31, 63):
        h = design.sinc(0.1, ntaps)
        for mode in ("direct", "auto"):
            for _ in range(5):
                ops.convolve(x, h, mode=mode, out=y)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.convolve(x, h, mode=mode, out=y)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            line += f"  {ntaps}/{mode} {ms:6.3f} ({ch * n * 8 / ms / 8e9:5.3f})"
    print(line, flush=True)
