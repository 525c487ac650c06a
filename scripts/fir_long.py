"""Dev tool: overlap-save FIR time over long tap counts (16384-point blocks, partitioned above 8193), 1,024 ch x 2^20."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import ops, design
x = torch.randn(1024, 1 << 20, device="cuda")
y = torch.empty_like(x)
for ntaps in (511, 2049, 2051, 3001, 4095, 6001, 8191, 8193, 8195, 16383, 32767):
    h = design.sinc(0.1, ntaps)
    for _ in range(2): ops.convolve(x, h, "overlap_save", out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.convolve(x, h, "overlap_save", out=y)
    e1.record(); torch.cuda.synchronize()
    print(f"taps {ntaps:6d}  overlap_save {e0.elapsed_time(e1) / 5:7.3f} ms", flush=True)
