"""Dev tool: batch (lane per channel) vs scan (block-parallel inside channels) over shapes of 2^28 samples in total:
where DSP_GPU_IIR_AUTO should switch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import ops, design
a, b = design.chebyshev(0.1, 4)
total = 1 << 28
x = torch.randn(total, device="cuda")
y = torch.empty_like(x)
for log2c in (0, 6, 10, 12, 13, 14, 15, 16, 17, 18):
    ch, n = 1 << log2c, total >> log2c
    row = [f"{ch:7d} ch x 2^{28 - log2c:<2d}"]
    for mode in ("batch", "scan"):
        if mode == "batch" and log2c < 6:
            row.append("batch      -   ")
            continue
        xv, yv = x.view(ch, n), y.view(ch, n)
        for _ in range(2): ops.iir(xv, a, b, mode=mode, out=yv)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): ops.iir(xv, a, b, mode=mode, out=yv)
        e1.record(); torch.cuda.synchronize()
        row.append(f"{mode} {e0.elapsed_time(e1) / 3:8.3f} ms")
    print("  ".join(row), flush=True)
