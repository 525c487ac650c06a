import os, sys, torch
sys.path.insert(0, os.getcwd())
from dsp_amd import ops
dev = torch.device("cuda:0")
buf = torch.randn(1 << 29, device=dev); out = torch.empty_like(buf)
for n in [int(a) for a in sys.argv[1:]] or (2048, 3000, 3500, 4096, 4500, 5000, 6000, 7000, 8192, 9000, 10000, 13000, 20000, 50000):
    ch = (1 << 29) // n
    x, y = buf[:ch*n].view(ch, n), out[:ch*n].view(ch, n)
    line = f"{ch:8d} x {n:6d}:"
    for m in (5, 511):
        for _ in range(5): ops.moving_average(x, m, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.moving_average(x, m, out=y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += f"  {m}: {ms:6.3f} ({ch*n*8/ms/8e9:5.3f})"
    print(line, flush=True)
