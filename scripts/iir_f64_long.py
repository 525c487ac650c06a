import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from dsp_amd import design, ops
dev = torch.device("cuda:0")
a, b = design.low_pass(0.1, 4)
for ch, n in ((1, 1 << 20), (1, 1 << 22), (16, 1 << 20), (256, 1 << 18), (1, 1 << 27), (1, 100000000), (8, 1 << 24)):
    x = torch.randn(ch, n, device=dev, dtype=torch.float64); y = torch.empty_like(x)
    for _ in range(3): ops.iir_f64(x, a, b, out=y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): ops.iir_f64(x, a, b, out=y)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / 5
    xf = x.float(); yf = torch.empty_like(xf)
    for _ in range(3): ops.iir(xf, a, b, out=yf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): ops.iir(xf, a, b, out=yf)
    torch.cuda.synchronize(); msf = (time.perf_counter() - t0) * 1e3 / 5
    print(f"{ch} x {n}: double {ms:9.3f} ms ({ch*n/ms/1e6:8.3f} GSamples/s, {ch*n*16/ms/8e9:5.3f} of 8 TB/s)   float {msf:7.3f} ms", flush=True)
