import os, sys, torch
sys.path.insert(0, os.getcwd())
from dsp_amd import design, ops
dev = torch.device("cuda:0")
TOTAL = 1 << 27
buf = torch.randn(TOTAL, device=dev, dtype=torch.float64); out = torch.empty_like(buf)
for n in (64, 256, 1000, 4096, 10000, 65536, 1 << 20):
    ch = TOTAL // n
    x, y = buf[:ch*n].view(ch, n), out[:ch*n].view(ch, n)
    line = f"{ch:8d} x {n:8d}:"
    for ntaps in (7, 31, 511):
        h = design.sinc(0.1, ntaps)
        for _ in range(3): ops.convolve_f64(x, h, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.convolve_f64(x, h, out=y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        line += f"  {ntaps}: {ms:7.3f} ms ({ch*n*16/ms/8e9:5.3f} of HBM, {2*ntaps*ch*n/ms/1e9/78.6:5.3f} of f64 MFMA)"
    print(line, flush=True)
