#!/usr/bin/env python3
"""One channel, every entry point, both sample types, four lengths: a coarse look for cliffs (ms per call and MSamples/s)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd import design, ops  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


h511, h31 = design.sinc(0.1, 511), design.sinc(0.1, 31)
shapes = [(1, 1 << 12), (1, 1 << 16), (1, 1 << 20), (1, 1 << 24)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for ch, n in shapes:
    x32 = torch.randn(ch, n, device=dev)
    x64 = x32.double()
    c32 = torch.view_as_complex(torch.randn(ch, 1 << (n.bit_length() - 1), 2, device=dev))
    c64 = c32.to(torch.complex128)
    fc = torch.full((ch, n), 0.1, device=dev)
    rows = [("fft0", lambda: ops.fft0(c32)), ("fft0_f64", lambda: ops.fft0_f64(c64)), ("fft real", lambda: ops.fft(x32, 1 << (n - 1).bit_length())),
            ("conv31", lambda: ops.convolve(x32, h31)), ("conv511", lambda: ops.convolve(x32, h511)),
            ("conv31_f64", lambda: ops.convolve_f64(x64, h31)), ("conv511_f64", lambda: ops.convolve_f64(x64, h511)),
            ("mavg", lambda: ops.moving_average(x32, 511)), ("mavg_f64", lambda: ops.moving_average_f64(x64, 511))]
    for poles in (2, 4, 8, 12):
        a, b = design.low_pass(0.25, poles)
        rows.append((f"iir{poles}", lambda a=a, b=b: ops.iir(x32, a, b)))
        rows.append((f"iir{poles}_f64", lambda a=a, b=b: ops.iir_f64(x64, a, b)))
    rows.append(("swept", lambda: ops.biquad_swept(x32, fc, 0.01)))
    line = f"{ch} x {n}:"
    for name, fn in rows:
        try:
            ms = t(fn)
            line += f" {name} {ms:.3f}"
        except Exception as e:  # noqa: BLE001
            line += f" {name} ERR({str(e)[:30]})"
    print(line, flush=True)
