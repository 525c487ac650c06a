import os, sys, torch
sys.path.insert(0, "/root/repo")
from dsp_amd import ops
dev = torch.device("cuda:0")
batch, n = 65536, int(os.environ.get("RAMP_N", "4096"))
batch = (1 << 28) // n
x = torch.view_as_complex(torch.randn(batch, n, 2, device=dev))
out = torch.empty_like(x)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(17)]
ev[0].record()
for g in range(16):
    for _ in range(20):
        ops.fft0(x, out=out)
    ev[g + 1].record()
torch.cuda.synchronize()
print(" ".join(f"{ev[g].elapsed_time(ev[g+1]) / 20:.3f}" for g in range(16)), flush=True)
