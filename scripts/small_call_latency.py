"""Dev tool: latency of small calls (the reference's demo sizes), host-pointer path (staged, synchronous) and
device-pointer path (asynchronous launch + one synchronise)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dsp_amd import ops, design
a, b = design.chebyshev(0.1, 4)
h = design.sinc(0.1, 511)
cases = [("fft real 1024 (host)", lambda x=np.random.randn(1024).astype(np.float32): ops.fft(x, 1024)),
         ("fir 511 taps, 4096 samples (host)", lambda x=np.random.randn(4096).astype(np.float32): ops.convolve(x, h)),
         ("iir 4-pole, 256 samples + state (host)", lambda x=np.random.randn(256).astype(np.float32), st=np.zeros((1, 8)): ops.iir(x, a, b, state=st)),
         ("moving average 51, 4096 samples (host)", lambda x=np.random.randn(4096).astype(np.float32): ops.moving_average(x, 51))]
xd = torch.randn(1024, device="cuda"); od = torch.empty(1024, dtype=torch.complex64, device="cuda")
def dev_fft():
    ops.fft(xd, 1024); torch.cuda.synchronize()
cases.append(("fft real 1024 (device + sync)", dev_fft))
for name, fn in cases:
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(200): fn()
    print(f"{name:42s} {(time.perf_counter() - t0) / 200 * 1e6:8.1f} us", flush=True)
