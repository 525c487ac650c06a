"""Dev tool: IIR scan over one 2^28-sample channel for filters of increasing memory (fused warm-up tiles vs
reduce -> carry -> apply, double vs double-double carries)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import ops, design
n = 1 << 28
x = torch.randn(1, n, device="cuda")
y = torch.empty_like(x)
for name, (a, b) in (("lp4 fc=0.1", design.chebyshev(0.1, 4)), ("lp4 fc=0.03", design.chebyshev(0.03, 4)),
                     ("lp4 fc=0.01", design.chebyshev(0.01, 4)), ("lp4 fc=0.001", design.chebyshev(0.001, 4)),
                     ("lp2 fc=0.01", design.chebyshev(0.01, 2)), ("hp6 fc=0.2", design.chebyshev(0.2, 6, highpass=True)),
                     ("lp8 fc=0.2", design.chebyshev(0.2, 8))):
    for _ in range(2): ops.iir(x, a, b, mode="scan", out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.iir(x, a, b, mode="scan", out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name:14s} {ms:7.3f} ms  {n / ms / 1e6:7.1f} GSamples/s", flush=True)
