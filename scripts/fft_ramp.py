#!/usr/bin/env python3
"""ms per launch of the batched complex FFT (2^28 points per launch, RAMP_N points per transform) over successive groups
of 20 launches from process start: how long the chip takes to reach its sustained rate (used for the A/B
of DSP_GPU_FFT_ROW_WGS_PER_CU, profiles/r02_fft_wgs_per_cu.txt)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd import ops  # noqa: E402

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
