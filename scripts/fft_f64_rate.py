#!/usr/bin/env python3
"""Rate of dsp_gpu_fft_batch_f64 (complex double) over transform sizes, 2^26 points per call (1 GiB in, 1 GiB out)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
total = 1 << 26
for log2n in [int(a) for a in sys.argv[1:]] or (4, 8, 10, 12, 13, 14, 16, 18, 20, 24):
    n = 1 << log2n
    x = torch.view_as_complex(torch.randn(total // n, n, 2, device=dev, dtype=torch.float64))
    y = torch.empty_like(x)
    for _ in range(3):
        ops.fft0_f64(x, out=y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 10
    for _ in range(iters):
        ops.fft0_f64(x, out=y)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    print(f"N=2^{log2n:<2d} {ms:8.3f} ms {total / ms / 1e6:7.1f} GSamples/s  {total * 32 / ms / 1e9 / 8:5.2f} of 8 TB/s (32 B/point)", flush=True)
