"""Dev tool: AUTO (sharded in time for sections of more than eight poles) against BATCH (one lane per channel), 1 x 70,000 fp32 samples."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from dsp_amd import design, ops, capi
dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
x = rng.standard_normal((1, 70000)).astype(np.float32)
xd = torch.from_numpy(x).to(dev)
for poles in (10, 12, 14, 16, 18, 20):
    for fc in (0.1, 0.25, 0.4):
        a, b = design.low_pass(fc, poles)
        try: w = ops.iir_settle(a, b)
        except capi.DspGpuError: w = None
        ya = ops.iir(xd, a, b, mode="auto").double()
        yb = ops.iir(xd, a, b, mode="batch").double()
        r = lambda p, q: float(torch.linalg.vector_norm(p - q) / torch.linalg.vector_norm(q))
        print(f"poles {poles} fc {fc}: settle {w}  auto-vs-batch {r(ya, yb):.2e}", flush=True)
