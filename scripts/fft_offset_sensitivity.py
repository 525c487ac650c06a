#!/usr/bin/env python3
"""Does the 4096-point batched FFT's rate depend on where its buffers sit relative to one another?
65,536 x 4096 complex fp32; the output starts `shift` bytes into a larger allocation.  For the A/B of workgroups per CU
(DSP_GPU_ROW_LDS_PAD)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
batch, n = 65536, 4096
x = torch.view_as_complex(torch.randn(batch, n, 2, device=dev))
big = torch.empty(batch * n + (1 << 22), dtype=torch.complex64, device=dev)
print(f"x at {x.data_ptr():#x}, out base {big.data_ptr():#x}", flush=True)
for shift in (0, 256, 4096, 65536, 1 << 20, 3 << 20, 1 << 24):
    out = big[shift // 8: shift // 8 + batch * n].view(batch, n)
    for _ in range(10):
        ops.fft0(x, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.fft0(x, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"out shift {shift:>9d} B: {ms:6.4f} ms  {batch * n * 16 / ms / 1e9 / 8:5.3f} of 8 TB/s", flush=True)
