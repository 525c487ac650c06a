#!/usr/bin/env python3
"""Rate of the batch IIR (4 poles) over channel lengths, about 2^30 samples per call: is 4096 special?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd import design, ops  # noqa: E402

dev = torch.device("cuda:0")
a, b = design.low_pass(0.1, 4)
TOTAL = 1 << 30
buf = torch.randn(TOTAL, device=dev)
out = torch.empty_like(buf)
MODE = os.environ.get("IIR_MODE", "batch")
lengths = [int(v) for v in sys.argv[1:]] or [1024, 2048, 3840, 4000, 4096, 4224, 4352, 4608, 5120, 6144, 8192, 10000, 16384, 65536]
for n in lengths:
    ch = TOTAL // n
    x, y = buf[: ch * n].view(ch, n), out[: ch * n].view(ch, n)
    for _ in range(30):
        ops.iir(x, a, b, mode=MODE, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.iir(x, a, b, mode=MODE, out=y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{MODE:5s} {ch:7d} x {n:6d}: {ms:7.4f} ms {ch * n / ms / 1e6:7.1f} GSamples/s {ch * n * 8 / ms / 8e9:5.3f} of 8 TB/s", flush=True)
