"""Dev tool: dsp::fft real-input wrapper (fft.h:57-83) throughput, 2^28 real samples per call, 12 B/sample algorithmic."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import ops, capi
total = 1 << 28
ITERS = int(os.environ.get("FFT_ITERS", "10"))  # 50+ for sustained clocks
x = torch.randn(total, device="cuda")
out = torch.empty(total, dtype=torch.complex64, device="cuda")
import ctypes as C
for log2n in eval(os.environ.get("REAL_SIZES", "(6, 8, 10, 12, 13, 14, 15, 16, 17, 18, 20, 22, 23, 24)")):
    n = 1 << log2n
    xv = x.view(total // n, n)
    def run():
        capi.check(capi.lib().dsp_gpu_fft_real_batch(xv.data_ptr(), n, n, out.data_ptr(), n, total // n, capi.FORWARD, 0,
                                                     capi.MEM_DEVICE, torch.cuda.current_stream().cuda_stream))
    for _ in range(max(3, ITERS // 5)): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / ITERS
    print(f"N=2^{log2n:<2d} {ms:8.3f} ms {total / ms / 1e6:7.1f} GSamples/s  {total * 12 / ms / 1e9 / 8:5.2f} of 8 TB/s", flush=True)
