#!/usr/bin/env python3
"""Moving average over windows and channel shapes, 2^30 samples per call (fraction of 8 TB/s at 8 B/sample)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
buf = torch.randn(1 << 30, device=dev)
out = torch.empty_like(buf)
for ch, n in ((1024, 1 << 20), (1 << 20, 1024), (1 << 22, 256), (1 << 24, 64), (262144, 4096), (100000, 10000), (7, 150000000), (1, 1 << 30)):
    x, y = buf[: ch * n].view(ch, n), out[: ch * n].view(ch, n)
    line = f"{ch:8d} x {n:10d}:"
    for m in (3, 5, 31, 511, 4095, 8191):
        if m >= n:
            continue
        for _ in range(5):
            ops.moving_average(x, m, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.moving_average(x, m, out=y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += f"  {m}: {ms:6.3f} ({ch * n * 8 / ms / 8e9:5.3f})"
    print(line, flush=True)
