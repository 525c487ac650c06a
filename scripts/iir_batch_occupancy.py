#!/usr/bin/env python3
"""IIR batch kernel (262,144 ch x 4,096) by pole count, for the A/B of workgroups per CU
(DSP_GPU_IIR_BATCH_LDS_PAD=<bytes of dynamic LDS> lowers it; unset = the library's choice)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd import design, ops  # noqa: E402

dev = torch.device("cuda:0")
x = torch.randn(262144, 4096, device=dev)
y = torch.empty_like(x)
for poles in (2, 4, 6, 8, 12):
    a, b = design.chebyshev(0.1, poles, highpass=False)
    for _ in range(3):
        ops.iir(x, a, b, mode="batch", out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.iir(x, a, b, mode="batch", out=y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"poles {poles:2d}: {ms:6.3f} ms  {x.numel() / ms / 1e6:6.1f} GSamples/s  {x.numel() * 8 / ms / 1e9 / 8:5.3f} of 8 TB/s", flush=True)
