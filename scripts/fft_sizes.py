"""Dev tool: batched complex FFT throughput over transform sizes (2^28 points per call), GSamples/s and HBM fraction
(16 B/sample algorithmic).  Run on the GPU box:  python scripts/fft_sizes.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import ops

total = 1 << 28
ITERS = int(os.environ.get("FFT_ITERS", "10"))  # 50+ for sustained clocks: short runs read up to 5 % fast
x = torch.randn(total, 2, device="cuda").view(torch.complex64) if False else torch.view_as_complex(torch.randn(total, 2, device="cuda"))
out = torch.empty_like(x)
import os
for log2n in eval(os.environ.get("FFT_SIZES", "(4, 6, 8, 10, 11, 12, 13, 14, 16, 18, 20, 22, 24, 26, 28, 30)")):
    n = 1 << log2n
    if n > total:  # one transform larger than the sweep's batch: grow the buffers
        total = n
        del x, out
        x = torch.view_as_complex(torch.randn(total, 2, device="cuda"))
        out = torch.empty_like(x)
    xv, ov = x.view(total // n, n), out.view(total // n, n)
    for _ in range(ITERS // 5):
        ops.fft0(xv, out=ov)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        ops.fft0(xv, out=ov)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / ITERS
    print(f"N=2^{log2n:<2d} {ms:8.3f} ms  {total / ms / 1e6:7.1f} GSamples/s  {total * 16 / ms / 1e9 / 8:5.2f} of 8 TB/s", flush=True)
