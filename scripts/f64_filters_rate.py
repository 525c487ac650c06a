#!/usr/bin/env python3
"""Rates of the double-sample filters (dsp_gpu_moving_average_f64, dsp_gpu_iir_f64, dsp_gpu_convolve_f64) against their
lines: 16 B/sample of HBM traffic for the moving average and the IIR, 78.6 TFLOP/s of fp64 MFMA for the direct FIR."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd import design, ops  # noqa: E402

dev = torch.device("cuda:0")


def rate(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


x = torch.randn(1024, 1 << 19, device=dev, dtype=torch.float64)
y = torch.empty_like(x)
for m in (5, 511):
    s = rate(lambda: ops.moving_average_f64(x, m, out=y))
    print(f"moving_average_f64 {m:4d} points, 1024 x 2^19: {s * 1e3:7.3f} ms {x.numel() / s / 1e9:7.1f} GSamples/s {x.numel() * 16 / s / 8e12:5.2f} of 8 TB/s", flush=True)
ca, cb = design.low_pass(0.1, 4)
xb = torch.randn(262144, 2048, device=dev, dtype=torch.float64)
yb = torch.empty_like(xb)
s = rate(lambda: ops.iir_f64(xb, ca, cb, mode="batch", out=yb))
print(f"iir_f64 batch 4 poles, 262144 x 2048: {s * 1e3:7.3f} ms {xb.numel() / s / 1e9:7.1f} GSamples/s {xb.numel() * 16 / s / 8e12:5.2f} of 8 TB/s", flush=True)
del xb, yb
taps = design.sinc(0.1, 511)
xc = torch.randn(1024, 1 << 18, device=dev, dtype=torch.float64)
yc = torch.empty_like(xc)
s = rate(lambda: ops.convolve_f64(xc, taps, out=yc), iters=5)
mac = 511 * (xc.shape[1] - 510) * xc.shape[0]
print(f"convolve_f64 511 taps, 1024 x 2^18: {s * 1e3:7.3f} ms {mac / s / 1e12:6.2f} TMAC/s {2 * mac / s / 78.6e12:5.2f} of 78.6 TFLOP/s", flush=True)
