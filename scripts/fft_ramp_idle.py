#!/usr/bin/env python3
"""As fft_ramp.py, then an idle gap (RAMP_IDLE seconds, default 2) and the same again: does the slow start come back after
idle (clock management) or only once per process / buffer (first touches)?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n = 4096
batch = (1 << 28) // n
x = torch.view_as_complex(torch.randn(batch, n, 2, device=dev))
out = torch.empty_like(x)
ops.fft0(x, out=out)  # one-time costs out of the way
torch.cuda.synchronize()
for phase in ("from process start", "after the idle gap"):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
    ev[0].record()
    for g in range(8):
        for _ in range(10):
            ops.fft0(x, out=out)
        ev[g + 1].record()
    torch.cuda.synchronize()
    print(phase + ", groups of 10:", " ".join(f"{ev[g].elapsed_time(ev[g + 1]) / 10:.3f}" for g in range(8)), flush=True)
    time.sleep(float(os.environ.get("RAMP_IDLE", "2")))
