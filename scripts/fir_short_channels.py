"""Dev tool: direct vs overlap-save FIR on many SHORT channels (n < 4096), 2^26 samples in total: where AUTO should switch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import ops, design
for n in (256, 512, 1024, 2048, 3000, 4096):
    ch = (1 << 26) // n
    x = torch.randn(ch, n, device="cuda")
    y = torch.empty_like(x)
    for ntaps in (31, 63, 127, 255, 511, 1023):
        if ntaps > n: continue
        h = design.sinc(0.1, ntaps)
        row = [f"{ch:7d} ch x {n:5d}  taps {ntaps:5d}"]
        for mode in ("direct", "overlap_save"):
            for _ in range(2): ops.convolve(x, h, mode, out=y)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ops.convolve(x, h, mode, out=y)
            e1.record(); torch.cuda.synchronize()
            row.append(f"{mode} {e0.elapsed_time(e1) / 5:7.3f} ms")
        print("  ".join(row), flush=True)
