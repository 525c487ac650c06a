#!/usr/bin/env python3
"""dsp_gpu_iir (AUTO and the two forced modes) over (channels, n) shapes of about 2^28 samples, 4 poles."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd import design, ops  # noqa: E402

dev = torch.device("cuda:0")
a, b = design.low_pass(0.1, 4)
buf = torch.randn(1 << 28, device=dev)
out = torch.empty_like(buf)
for ch, n in ((1, 1 << 28), (1, 250000000), (3, 80000000), (16, 1 << 24), (100, 2500000), (1000, 250000), (4096, 65536), (10000, 25000),
              (20000, 12500), (30000, 8000)):
    x, y = buf[: ch * n].view(ch, n), out[: ch * n].view(ch, n)
    line = f"{ch:6d} x {n:9d}:"
    for mode in ("auto", "scan", "batch"):
        if mode == "batch" and ch < 1000:
            continue
        for _ in range(30 if mode != "batch" else 3):
            ops.iir(x, a, b, mode=mode, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.iir(x, a, b, mode=mode, out=y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += f"  {mode} {ms:7.3f} ms ({ch * n * 8 / ms / 8e9:5.3f})"
    print(line, flush=True)
