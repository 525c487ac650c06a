"""Dev tool: throughput of the host-pointer (mem = HOST) path: pageable numpy buffers in, out."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dsp_amd import ops, design
rng = np.random.default_rng(0)
x = (rng.standard_normal((16384, 4096)) + 1j * rng.standard_normal((16384, 4096))).astype(np.complex64)   # 512 MiB
for _ in range(2): y = ops.fft0(x)
t0 = time.perf_counter()
for _ in range(3): y = ops.fft0(x)
dt = (time.perf_counter() - t0) / 3
print(f"fft0 host path: {x.size / dt / 1e9:.2f} GSamples/s, {2 * x.nbytes / dt / 1e9:.1f} GB/s in+out, {dt * 1e3:.1f} ms", flush=True)
xr = rng.standard_normal((256, 1 << 20)).astype(np.float32)   # 1 GiB
h = design.sinc(0.1, 511)
for _ in range(2): y = ops.convolve(xr, h, "overlap_save")
t0 = time.perf_counter()
for _ in range(3): y = ops.convolve(xr, h, "overlap_save")
dt = (time.perf_counter() - t0) / 3
print(f"fir os host path: {xr.size / dt / 1e9:.2f} GSamples/s, {2 * xr.nbytes / dt / 1e9:.1f} GB/s in+out, {dt * 1e3:.1f} ms", flush=True)
# the same calls on page-locked host buffers (what a caller that pins its buffers gets)
import torch
xp = torch.view_as_complex(torch.randn(16384, 4096, 2).pin_memory()).numpy()
yp = torch.view_as_complex(torch.empty(16384, 4096, 2).pin_memory()).numpy()
for _ in range(2): ops.fft0(xp, out=yp)
t0 = time.perf_counter()
for _ in range(3): ops.fft0(xp, out=yp)
dt = (time.perf_counter() - t0) / 3
print(f"fft0 host path, pinned buffers: {xp.size / dt / 1e9:.2f} GSamples/s, {2 * xp.nbytes / dt / 1e9:.1f} GB/s in+out, {dt * 1e3:.1f} ms", flush=True)
yq = np.empty_like(x); yq[:] = 0
t0 = time.perf_counter()
for _ in range(3): ops.fft0(x, out=yq)
dt = (time.perf_counter() - t0) / 3
print(f"fft0 host path, pageable, output preallocated: {x.size / dt / 1e9:.2f} GSamples/s, {2 * x.nbytes / dt / 1e9:.1f} GB/s in+out, {dt * 1e3:.1f} ms", flush=True)
