import torch, time, sys
sys.path.insert(0, '/root/repo')
from dsp_amd import ops, design
taps = design.sinc(0.1, 511)
dev = torch.device('cuda:0')
for ch, n in ((64, 1<<24), (1024, 1<<20), (1020, (1<<20) + 4096), (2048, 1<<19), (1024, 1<<20), (64, 1<<24)):
    x = torch.randn(ch, n, device=dev); y = torch.empty_like(x)
    for _ in range(3): ops.convolve(x, taps, 'overlap_save', out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.convolve(x, taps, 'overlap_save', out=y)
    e1.record(); torch.cuda.synchronize()
    print(ch, n, e0.elapsed_time(e1) / 10, 'ms')
    del x, y
