import os, sys, torch
sys.path.insert(0, os.getcwd())
from dsp_amd import design, ops
dev = torch.device("cuda:0")
a, b = design.low_pass(0.1, 4)
which = sys.argv[1]
x = torch.randn(1 << 28, device=dev)
y = torch.empty_like(x)
shape = {"scan": (1, 1 << 28), "b4096": (65536, 4096), "b4352": (61680, 4352), "b8192": (32768, 8192), "b2048": (131072, 2048)}[which]
nel = shape[0] * shape[1]
xv, yv = x[:nel].view(shape), y[:nel].view(shape)
mode = "scan" if which == "scan" else "batch"
f = lambda: ops.iir(xv, a, b, mode=mode, out=yv)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
for _ in range(5): f()
torch.cuda.synchronize()
ev[0].record()
for i in range(60):
    f(); ev[i + 1].record()
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(60)]
first20 = sum(t[:20]) / 20
last20 = sorted(t[40:])[10]
print(f"{which:6s} first 20 after 5 warm-up: {first20:.4f} ms ({nel * 8 / first20 / 8e9:.3f}); median of launches 46-65: {last20:.4f} ms ({nel * 8 / last20 / 8e9:.3f})")
