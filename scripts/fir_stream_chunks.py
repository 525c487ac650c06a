#!/usr/bin/env python3
"""Streamed FIR (dsp_gpu_convolve_stream) in audio-callback-sized chunks: channels x chunk per call, history carried."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd import design, ops  # noqa: E402

dev = torch.device("cuda:0")
for ch, n in ((1 << 20, 256), (1 << 19, 512), (1 << 16, 4096), (1 << 12, 65536)):
    x = torch.randn(ch, n, device=dev)
    y = torch.empty_like(x)
    line = f"{ch:8d} x {n:6d}:"
    for ntaps in (31, 511):
        h = design.sinc(0.1, ntaps)
        for mode in ("direct", "overlap_save"):
            hist = torch.zeros(ch, ntaps - 1, device=dev)
            pos = 0
            for _ in range(3):
                ops.convolve_stream(x, h, hist, pos, mode=mode, out=y); pos += n
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.convolve_stream(x, h, hist, pos, mode=mode, out=y); pos += n
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            line += f"  {ntaps}/{mode[:6]} {ms:7.3f} ms ({ch * n * 8 / ms / 8e9:5.3f})"
    print(line, flush=True)
