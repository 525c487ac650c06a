import sys, numpy as np, torch
sys.path.insert(0, ".")
from dsp_amd import ops, design
from oracle import load
orc = load("oracle")
dev = torch.device("cuda:0")
rng = np.random.default_rng(9)
def rel(a, b): return float(np.linalg.norm(a - b) / np.linalg.norm(b))
for fc in (0.01, 0.004, 0.002, 0.001):
    a, b = design.chebyshev(fc, 4)
    for n in (256, 2048, 8192, 16384, 100000):
        x = rng.standard_normal((1, n)).astype(np.float32)
        want = orc.iir(x.astype(np.float64), a, b)
        xd = torch.from_numpy(x).to(dev)
        gs = ops.iir(xd, a, b, mode="scan").cpu().numpy()
        gb = ops.iir(xd, a, b, mode="batch").cpu().numpy()
        # where does the error start?
        d = np.abs(gs - want)[0]; thr = 1e-4 * np.abs(want).max()
        bad = np.nonzero(d > thr)[0]
        print(f"fc={fc} n={n}: scan {rel(gs, want):.2e} batch {rel(gb, want):.2e} first_bad={bad[0] if bad.size else None}", flush=True)
