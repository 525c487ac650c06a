import sys, os
sys.path.insert(0, os.getcwd())
import torch
from dsp_amd import capi
total = 1 << 28
x = torch.randn(total, device="cuda")
out = torch.empty(total, dtype=torch.complex64, device="cuda")
for log2n in (16, 18):
    n = 1 << log2n
    for _ in range(5):
        capi.check(capi.lib().dsp_gpu_fft_real_batch(x.data_ptr(), n, n, out.data_ptr(), n, total // n, capi.FORWARD, 0, capi.MEM_DEVICE, torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
