"""Dev tool: direct (MFMA) vs overlap-save FIR time over tap counts, 1,024 ch x 2^20 samples: where AUTO should switch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dsp_amd import ops, design
x = torch.randn(1024, 1 << 20, device="cuda")
y = torch.empty_like(x)
for ntaps in (3, 7, 15, 31, 63, 95, 127, 191, 255, 511, 1023, 2047, 4095, 8191, 32767):
    h = design.sinc(0.1, ntaps)
    row = [f"taps {ntaps:5d}"]
    for mode in ("direct", "direct_valu", "overlap_save", "auto"):
        if mode == "direct_valu" and ntaps > 1024: continue
        if mode == "direct" and ntaps > 4096: continue
        for _ in range(2): ops.convolve(x, h, mode, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.convolve(x, h, mode, out=y)
        e1.record(); torch.cuda.synchronize()
        row.append(f"{mode} {e0.elapsed_time(e1) / 5:7.3f} ms")
    print("  ".join(row), flush=True)
