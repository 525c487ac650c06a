import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import capi
total = 1 << 28
out = torch.empty(total, dtype=torch.complex64, device="cuda")
for log2n, n_in in ((12, 4096), (12, 4000), (12, 3001), (10, 1024), (10, 1000), (8, 256), (8, 200), (13, 8000), (15, 30000), (16, 65000), (18, 200000)):
    n = 1 << log2n
    batch = total // n
    x = torch.randn(batch * n_in, device="cuda")
    def run():
        capi.check(capi.lib().dsp_gpu_fft_real_batch(x.data_ptr(), n_in, n_in, out.data_ptr(), n, batch, capi.FORWARD, 0,
                                                     capi.MEM_DEVICE, torch.cuda.current_stream().cuda_stream))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    bytes_ = batch * (n_in * 4 + n * 8)
    print(f"N=2^{log2n:<2d} n_in={n_in:<7d} {ms:8.3f} ms {total / ms / 1e6:7.1f} Gbins/s  {bytes_ / ms / 1e9 / 8:5.2f} of 8 TB/s", flush=True)
