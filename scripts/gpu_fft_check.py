"""First GPU contact: FFT parity against the oracle + a C2 timing.  (dev script, not a test)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import dsp_amd
from dsp_amd import ops
from oracle import load
orc = load("oracle")
rng = np.random.default_rng(1)
dev = torch.device("cuda:0")
worst = 0
for k in list(range(1, 13)) + [13, 14, 16, 17, 18, 20]:
    n = 1 << k
    batch = 3 if k <= 14 else 1
    x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
    for inv in (False, True):
        ref = orc.fft0(x.astype(np.complex128), inv)
        got = ops.fft0(torch.from_numpy(x).to(dev), inv).cpu().numpy()
        err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        got_h = ops.fft0(x, inv)
        errh = np.linalg.norm(got_h - ref) / np.linalg.norm(ref)
        print(f"log2n={k:2d} inv={int(inv)} rel_l2 dev={err:.3e} host={errh:.3e} budget={1e-5*k:.1e}", flush=True)
        worst = max(worst, err / (1e-5 * k))
print("worst err/budget", worst)
# C2 timing
x = torch.randn(65536, 4096, 2, device=dev).view(torch.float32)
x = torch.view_as_complex(x.reshape(65536, 4096, 2))
out = torch.empty_like(x)
for it in (3, 20, 20):
    ms = ops.fft_kernel_ms(x, out, False, it)
    print(f"C2 fwd: {ms:.4f} ms/launch  {65536*4096/ms/1e6:.1f} GSamples/s  {65536*4096*16/ms/1e9:.3f} TB/s", flush=True)
ms = ops.fft_kernel_ms(x, out, True, 20)
print(f"C2 inv: {ms:.4f} ms/launch  {65536*4096*16/ms/1e9:.3f} TB/s")
for k in (8, 10, 11):
    n = 1 << k
    xb = x.view(-1, n); ob = out.view(-1, n)
    ms = ops.fft_kernel_ms(xb, ob, False, 20)
    print(f"N=2^{k}: {ms:.4f} ms  {65536*4096*16/ms/1e9:.3f} TB/s")
xl = x.view(-1, 1 << 20)[:64].contiguous(); ol = torch.empty_like(xl)
ms = ops.fft_kernel_ms(xl, ol, False, 10)
print(f"N=2^20 x64: {ms:.4f} ms  {64*(1<<20)*16/ms/1e9:.3f} TB/s (algorithmic 16B/sample)")
