"""Dev tool: FIR / moving average over shapes of 2^28 samples in total, from one long channel to many short ones."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import ops, design
total = 1 << 28
x = torch.randn(total, device="cuda")
y = torch.empty_like(x)
h63, h511 = design.sinc(0.1, 63), design.sinc(0.1, 511)
for log2c in (0, 8, 12, 14, 16, 18):
    ch, n = 1 << log2c, total >> log2c
    xv, yv = x.view(ch, n), y.view(ch, n)
    row = [f"{ch:7d} ch x 2^{28 - log2c:<2d}"]
    for name, fn in (("direct63", lambda: ops.convolve(xv, h63, "direct", out=yv)),
                     ("direct511", lambda: ops.convolve(xv, h511, "direct", out=yv)),
                     ("os511", lambda: ops.convolve(xv, h511, "overlap_save", out=yv)),
                     ("mavg511", lambda: ops.moving_average(xv, 511, out=yv))):
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): fn()
        e1.record(); torch.cuda.synchronize()
        row.append(f"{name} {e0.elapsed_time(e1) / 3:7.3f} ms")
    print("  ".join(row), flush=True)
