import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from dsp_amd import design, ops
dev = torch.device("cuda:0")
for poles in (8, 10, 12):
    a, b = design.low_pass(0.25, poles)
    for ch, n in ((1, 1 << 20), (16, 1 << 20)):
        x = torch.randn(ch, n, device=dev); y = torch.empty_like(x)
        ops.iir(x, a, b, out=y); torch.cuda.synchronize()
        t0 = time.perf_counter(); ops.iir(x, a, b, out=y); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
        print(f"{poles} poles {ch} x {n}: float auto {ms:9.3f} ms", flush=True)
