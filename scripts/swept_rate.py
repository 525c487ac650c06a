#!/usr/bin/env python3
"""Rate of dsp_gpu_biquad_swept (the band-pass retuned every sample) on a chip-filling batch.

    python scripts/swept_rate.py [channels] [n]

Prints ms per launch (HIP events on the launch stream), GSamples/s and the traffic rate at 12 B/sample.
"""
import sys

import torch

sys.path.insert(0, ".")
from dsp_amd import ops  # noqa: E402


def main():
    ch = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(ch, n, generator=g, device=dev)
    y = torch.empty_like(x)
    t = torch.arange(n, device=dev, dtype=torch.float32)
    rate = torch.rand(ch, 1, generator=g, device=dev) * 0.02 + 0.001
    per_channel = (0.05 + 0.04 * torch.sin(rate * t)).contiguous()  # smooth sweeps: random jumps are not a stable filter
    wide = (0.23 + 0.22 * torch.sin(rate * t)).contiguous()  # all three quadrants of the cosine inside every wave
    for label, fc in (("per-channel sweep", per_channel), ("one sweep for all", per_channel[0].contiguous()),
                      ("sweeps 0.01..0.45", wide)):
        for kind in ("bandpass", "bandreject"):
            for _ in range(2):
                ops.biquad_swept(x, fc, 60.0 / 44100.0, kind=kind, out=y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 5
            e0.record()
            for _ in range(iters):
                ops.biquad_swept(x, fc, 60.0 / 44100.0, kind=kind, out=y)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            samples = ch * n
            traffic = samples * (8 + (4 if fc.dim() == 2 else 0))
            print(f"{label:18s} {kind:10s} {ch} x {n}: {ms:7.3f} ms  {samples / ms / 1e6:7.1f} GSamples/s  "
                  f"{traffic / ms / 1e9:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
