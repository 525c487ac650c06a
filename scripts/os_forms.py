#!/usr/bin/env python3
"""Overlap-save FIR on C3's shape (1,024 x 2^20) at several tap counts, both forms of the wave kernel (dev tool, round 3):
DSP_GPU_OS_WAVE_FORM=agpr (round 2: one wave per SIMD, AGPR prefetch) against the default (two waves per SIMD, QUAD
transpose, sequential runs).  ms per call, HIP events over 20 calls."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1:
    import numpy as np
    import torch
    from dsp_amd import ops
    dev = torch.device("cuda:0")
    x = torch.randn(1024, 1 << 20, device=dev)
    y = torch.empty_like(x)
    for ntaps in (127, 255, 511, 767, 1025, 1535, 1793, 2049):
        h = np.random.default_rng(ntaps).standard_normal(ntaps) / ntaps ** 0.5
        for _ in range(5):
            ops.convolve(x, h, "overlap_save", out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.convolve(x, h, "overlap_save", out=y)
        e1.record()
        torch.cuda.synchronize()
        print(f"  {sys.argv[1]:8s} {ntaps:5d} taps  {e0.elapsed_time(e1) / 20:.3f} ms", flush=True)
else:
    for form in ("default", "agpr", "default", "agpr"):
        env = dict(os.environ)
        if form == "agpr":
            env["DSP_GPU_OS_WAVE_FORM"] = "agpr"
        subprocess.run([sys.executable, __file__, form], env=env, check=True)
