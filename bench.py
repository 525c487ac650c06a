#!/usr/bin/env python3
"""Benchmark of the MI355X hot path (BASELINE.json metric), one JSON line on stdout.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload fft|fir_direct|fir_os|iir_batch|iir_scan|mavg|...]

Default workload = BASELINE.json configs[1]: 65,536 x 4096-point complex fp32 FFT (FORWARD) of
synthetic N(0,1) noise, inputs resident in HBM.  A "step" is one pass of the hot path over that
batch through the C ABI (dsp_gpu_fft_batch).  N > 1: launched by torch.distributed.run, one rank
per GPU; channels are independent, so every rank owns its own batch of the same size (weak
scaling, no data-path collective) and `value` = samples all ranks processed / max-over-ranks time.

`secondary`    the default run also times the FIR / IIR / moving-average workloads BASELINE's metric names, each with
               the same W warm-up and K timed steps between the same barriers.  They are timed BEFORE the headline
               FFT (--headline-first: after it): the first memory-bound workload of a fresh process would be
               timed while the chip is still ramping, the MFMA-bound direct FIR that goes first
               does not care, and every figure of the line is then taken at sustained clocks.
`roofline`     algorithmic bytes (or flops) of one launch / average launch time measured with
               HIP events recorded on the stream the launches go to, over the timed region.
`cpu_baseline` the reference's own code (oracle/_ref, kind "reference") if that library
               travelled with the tree, else the oracle restatement (kind "port"), single
               thread (the reference is single-threaded), on a bounded sample of the workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: peak fp32 vector (= fp32 MFMA) rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="fft",
                    choices=["fft", "fir_direct", "fir_direct_valu", "fir_os", "iir_batch", "iir_scan", "mavg", "fft_dist",
                             "fft_32k", "fft_real_64k", "fft_f64", "biquad_swept", "biquad_swept_shared"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every GPU gets the full per-GPU batch; strong: the one-GPU batch is channel-sharded")
    ap.add_argument("--dist-exchange", default="torch", choices=["torch", "rccl"],
                    help="fft_dist only: all_to_all_single between the plan's phases, or dsp_gpu_fft_dist's own RCCL exchange")
    ap.add_argument("--dist-serial", action="store_true",
                    help="fft_dist with --dist-exchange rccl: the unpipelined schedule (A/B partner of the default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="default fft run only: skip the FIR / IIR / moving-average workloads reported under `secondary`")
    ap.add_argument("--headline-first", action="store_true",
                    help="default fft run only: time the headline FFT before the secondary workloads (cold chip) instead of after")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the CPU baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=int(os.environ.get("OMP_NUM_THREADS", "0") or 0),
                    help="also time the reference on this many host threads (0/1: skip); default $OMP_NUM_THREADS")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------
# workloads: each returns (step_fn, units_per_step, meta)
# ----------------------------------------------------------------------------------------------
# timed before the headline FFT in the default run, in this order: the MFMA-bound direct FIR first, then the moving
# average (not a BASELINE configuration) as the first memory-bound one, which is the one that still sees the ramp
SECONDARY = ("fir_direct", "mavg", "fir_os", "iir_batch", "iir_scan")
SHARD = (0, 1)  # (rank, world) when --scaling strong channel-shards the one-GPU batch
TIME_SHARD = False  # --scaling strong under torch.distributed (also at world 1 with DSP_BENCH_INIT_DIST): C4b cut in time


def _share(total):
    from dsp_amd import multigpu
    lo, hi = multigpu.shard_range(total, *SHARD)
    return hi - lo


def make_fft(torch, dev, ops, rank):
    batch, n = _share(65536), 4096
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.view_as_complex(torch.randn(batch, n, 2, device=dev, generator=g))
    out = torch.empty_like(x)

    def step():
        ops.fft0(x, False, out=out)

    meta = dict(metric="batched_fft_throughput", unit="GSamples/s", dtype="f32",
                config={"workload": "C2: 65,536 x 4096-point complex fp32 FFT (FORWARD), N(0,1) noise, per GPU",
                        "batch": batch, "n": n, "sharding": "channels, no collective"},
                bound="hbm", alg_per_unit=16.0, alg_unit="bytes/sample (8 B read + 8 B write)",
                kernel="fft_lds_kernel<12>")
    return step, batch * n, meta


def make_fft_other(torch, dev, ops, rank, which):
    """Not BASELINE configurations: the round-2 kernels that have no bench line of their own otherwise (profiles/)."""
    g = torch.Generator(device=dev).manual_seed(555 + rank)
    if which == "fft_32k":  # fft_plane_kernel: 2^15 points in one HBM pass
        batch, n = _share(8192), 1 << 15
        x = torch.view_as_complex(torch.randn(batch, n, 2, device=dev, generator=g))
        out = torch.empty_like(x)
        step = lambda: ops.fft0(x, False, out=out)  # noqa: E731
        per, kernel, wl, dtype = 16.0, "fft_plane_kernel", "8,192 x 32,768-point complex fp32 FFT", "f32"
    elif which == "fft_real_64k":  # the dsp::fft wrapper on rows of 2^16 samples (eval/modulation/helper.h:43's size)
        batch, n = _share(4096), 1 << 16
        x = torch.randn(batch, n, device=dev, generator=g)
        step = lambda: ops.fft(x, n)  # noqa: E731  (allocates its output like the reference's wrapper does)
        per, kernel, wl, dtype = 12.0, "fft_plane_kernel<REAL_HALF>", "4,096 rows of 65,536 real fp32 samples -> 65,536-bin spectra", "f32"
    else:  # the reference's own sample type
        batch, n = _share(16384), 4096
        x = torch.view_as_complex(torch.randn(batch, n, 2, device=dev, generator=g, dtype=torch.float64))
        out = torch.empty_like(x)
        step = lambda: ops.fft0_f64(x, False, out=out)  # noqa: E731
        per, kernel, wl, dtype = 32.0, "fft64_tile_kernel<12>", "16,384 x 4096-point complex DOUBLE FFT", "f64"
    meta = dict(metric=f"{which}_throughput", unit="GSamples/s", dtype=dtype, bound="hbm", alg_per_unit=per,
                alg_unit="bytes/sample", kernel=kernel, config={"workload": wl + " (FORWARD), per GPU", "batch": batch, "n": n})
    return step, batch * n, meta


def make_fir(torch, dev, ops, rank, mode):
    channels, n, ntaps = _share(1024), 1 << 20, 511
    taps = sinc_taps_host(0.1, ntaps)
    g = torch.Generator(device=dev).manual_seed(4321 + rank)
    x = torch.randn(channels, n, device=dev, generator=g)
    y = torch.empty_like(x)

    def step():
        ops.convolve(x, taps, mode, out=y)

    if mode in ("direct", "direct_valu"):
        meta = dict(metric="fir_direct_throughput", unit="GMAC/s", dtype="f32", bound="mfma",
                    alg_per_unit=2.0, alg_unit="flop/MAC",
                    kernel="fir_direct_kernel" if mode == "direct" else "fir_direct_valu_kernel (A/B partner, packed fp32 VALU)")
        units = channels * (n - (ntaps - 1)) * ntaps
    else:
        meta = dict(metric="fir_overlap_save_throughput", unit="GSamples/s", dtype="f32", bound="hbm",
                    alg_per_unit=8.0, alg_unit="bytes/sample (4 B read + 4 B write)", kernel="fir_overlap_save_kernel")
        units = channels * n
    meta["config"] = {"workload": f"C3: 511-tap windowed-sinc FIR ({mode}), 1,024 ch x 2^20 samples fp32 noise, per GPU",
                      "channels": channels, "n": n, "ntaps": ntaps, "sharding": "channels, no collective"}
    return step, units, meta


def make_mavg(torch, dev, ops, rank):
    channels, n, pts = _share(1024), 1 << 20, 511
    g = torch.Generator(device=dev).manual_seed(99 + rank)
    x = torch.randn(channels, n, device=dev, generator=g)
    y = torch.empty_like(x)

    def step():
        ops.moving_average(x, pts, out=y)

    meta = dict(metric="moving_average_throughput", unit="GSamples/s", dtype="f32", bound="hbm", alg_per_unit=8.0,
                alg_unit="bytes/sample", kernel="moving_average_kernel",
                config={"workload": "511-point centred moving average, 1,024 ch x 2^20 samples, per GPU",
                        "channels": channels, "n": n, "npoints": pts})
    return step, channels * n, meta


def make_iir(torch, dev, ops, rank, mode):
    a, b = cheby_lp4_host()
    if mode == "batch":
        channels, n = _share(262144), 4096
        wl = "C4a: 4-pole Chebyshev low-pass, 262,144 ch x 4,096 samples fp32 noise (batch-parallel), per GPU"
    else:
        channels, n = 1, 1 << 28
        wl = "C4b: 4-pole Chebyshev low-pass, one 2^28-sample channel (block-parallel scan), per GPU"
    g = torch.Generator(device=dev).manual_seed(777 + rank)
    time_sharded = mode == "scan" and TIME_SHARD
    if time_sharded:  # --scaling strong: the ONE channel is cut in time, one segment per GPU (multigpu.iir_time_sharded)
        from dsp_amd import multigpu
        lo, hi = multigpu.shard_range(n, *SHARD)
        n = hi - lo
        wl = f"C4b: 4-pole Chebyshev low-pass, one 2^28-sample channel cut in time over {SHARD[1]} GPU(s), {n} samples on this one"
    x = torch.randn(channels, n, device=dev, generator=g)
    y = torch.empty_like(x)

    def step():
        if time_sharded:
            multigpu.iir_time_sharded(x, a, b, rank=SHARD[0], world=SHARD[1], mode=mode)
        else:
            ops.iir(x, a, b, mode=mode, out=y)

    meta = dict(metric=f"iir_{mode}_throughput", unit="GSamples/s", dtype="f32", bound="hbm", alg_per_unit=8.0,
                alg_unit="bytes/sample (4 B read + 4 B write)", kernel=f"iir_{mode}_kernel",
                config={"workload": wl, "channels": channels, "n": n, "poles": 4})
    return step, channels * n, meta


def make_swept(torch, dev, ops, rank, shared=False):
    """the band-pass retuned every sample (eval/signals/wind_generator.h:150-167), a batch of voices;
    shared: all voices follow ONE sweep (designed once per sample, 8 B/sample)."""
    channels, n = _share(262144), 4096
    g = torch.Generator(device=dev).manual_seed(160 + rank)
    x = torch.randn(channels, n, device=dev, generator=g)
    t = torch.arange(n, device=dev, dtype=torch.float32)
    rate = torch.rand(channels, 1, device=dev, generator=g) * 0.02 + 0.001
    phase = torch.rand(channels, 1, device=dev, generator=g) * 6.28
    # wh1 of eval/signals/main.cpp:34: centre 1000 Hz + 1000 Hz x wind speed at 44.1 kHz; a slow sine per voice is the wind
    fc = ((1000.0 + 1000.0 * (0.5 + 0.5 * torch.sin(rate * t + phase))) / 44100.0).contiguous()
    y = torch.empty_like(x)
    bw = 60.0 / 44100.0
    if shared:
        fc = fc[0].contiguous()

    def step():
        ops.biquad_swept(x, fc, bw, out=y)

    meta = dict(metric="biquad_swept_throughput", unit="GSamples/s", dtype="f64", bound="hbm",
                alg_per_unit=8.0 if shared else 12.0,
                alg_unit="bytes/sample (x read, y written; the one sweep is read once per 64 voices)" if shared else
                         "bytes/sample (x and fc read, y written); the kernel is bound by its ~60 fp64 instructions per sample "
                         "before it is by HBM",
                kernel="biquad_swept_shared_kernel" if shared else "biquad_swept_kernel",
                config={"workload": "band-pass redesigned from fc[t] before every step (wind_generator.h:150-167), "
                                    "262,144 voices x 4,096 samples fp32, " +
                                    ("one sweep for all voices" if shared else "one sweep per voice") + ", per GPU",
                        "channels": channels, "n": n})
    return step, channels * n, meta


def make_fft_dist(torch, dev, world, rank, exchange, serial=False):
    """C5: ONE transform of world * 2^27 points (2^30 on 8 GPUs), six-step with all-to-all over xGMI."""
    from dsp_amd import dist_fft
    log2n = 27 + (world.bit_length() - 1)
    per_rank = (1 << log2n) // world
    g = torch.Generator(device=dev).manual_seed(2030 + rank)
    x = 0.01 * torch.view_as_complex(torch.randn(per_rank, 2, device=dev, generator=g))

    if exchange == "rccl":  # all of it inside the library: dsp_gpu_fft_dist with grouped ncclSend/ncclRecv
        comm = dist_fft.RcclCommunicator(rank, world)
        plan = dist_fft.NativePlan(log2n, rank, world, nccl_comm=comm.handle)
        y = torch.empty_like(x)

        def step():
            plan.run(x, False, out=y, serial=serial)
    else:

        def step():  # the library's plan handle runs the local phases; torch.distributed (RCCL) the exchanges
            dist_fft.fft_dist(x, log2n, False)

    meta = dict(metric="distributed_fft_throughput", unit="GSamples/s", dtype="f32", bound="hbm",
                alg_per_unit=48.0, alg_unit="bytes/sample (3 local passes x 16 B; the exchanges ride xGMI)",
                kernel="fft_column_kernel + fft_lds_kernel + swap_axes_kernel + transpose_blocks_kernel",
                config={"workload": f"C5: one 2^{log2n}-point complex fp32 FFT, six-step, {world} GPU(s), 2^27 points per GPU",
                        "log2n": log2n,
                        "exchange": (f"{plan.exchanges(serial=serial)} x grouped ncclSend/ncclRecv inside dsp_gpu_fft_dist"
                                     f" ({'serial' if serial or plan.exchanges() == 3 else 'pipelined'} schedule)")
                        if exchange == "rccl" else "3 x all_to_all_single (RCCL)"})
    return step, per_rank, meta


def sinc_taps_host(cf, length):
    """dsp::sinc (dsp.h:47-79), Blackman window: host-side design in double."""
    from dsp_amd import design
    return design.sinc(cf, length)


def cheby_lp4_host():
    """recursive::lowPassFilter<4>(0.1) coefficients (recursive_filters.h:153-157; KAT in SURVEY.md section 4)."""
    from dsp_amd import design
    return design.chebyshev(0.1, 4, highpass=False)


# ----------------------------------------------------------------------------------------------
def cpu_baseline(workload, seconds, threads=1):
    """Time the reference's CPU code on a bounded sample of the same workload (rank 0, N=1).
    threads > 1: the same per-channel function on `threads` host threads at once, each on its own slice (the
    "parallel for over channels around the unchanged function" of SURVEY.md 8d; ctypes drops the GIL in the call)."""
    if threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as pool:
            parts = list(pool.map(lambda k: _cpu_sample(workload, seconds, 1234 + k), range(threads)))
        dt = time.perf_counter() - t0
        out = dict(parts[0])
        out.update(value=sum(p["units"] for p in parts) / dt / 1e9, cores=threads, seconds=round(dt, 2))
        out.pop("units")
        return out
    out = _cpu_sample(workload, seconds, 1234)
    out.pop("units")
    return out


def _cpu_sample(workload, seconds, seed):
    from oracle import available, load
    kind = "reference" if available("ref") else "port"
    lib = load("ref" if kind == "reference" else "fast")
    rng = np.random.default_rng(seed)
    if workload == "fft_f32":  # the like-for-like figure, the same algorithm instantiated in float
        kind, lib = "port", load("fast")
        x = (rng.standard_normal((256, 4096)) + 1j * rng.standard_normal((256, 4096))).astype(np.complex64)
        call, per_call, unit = (lambda: lib.fft0_f32(x)), x.size, "GSamples/s"
        sample = "{calls} x 256 of 65,536 transforms, the restatement of fft0 instantiated in complex float (the reference has only double)"
    elif workload == "fft":
        x = (rng.standard_normal((256, 4096)) + 1j * rng.standard_normal((256, 4096)))
        call, per_call, unit = (lambda: lib.fft0(x)), x.size, "GSamples/s"
        sample = "{calls} x 256 of 65,536 transforms, fft0 in complex double as the reference computes (fft.h:31-55)"
    elif workload == "fft_dist":
        x = (rng.standard_normal(1 << 20) + 1j * rng.standard_normal(1 << 20))
        call, per_call, unit = (lambda: lib.fft0(x)), x.size, "GSamples/s"
        sample = "fft0 of 2^20 points in complex double (fft.h:31-55), {calls} times; 2^27+ does not fit a bounded CPU sample"
    elif workload in ("fir_direct", "fir_direct_valu", "fir_os"):
        h = sinc_taps_host(0.1, 511)
        x = rng.standard_normal(1 << 16).astype(np.float32)
        call = lambda: lib.fir(x, h)
        per_call = (x.size - 510) * (511 if workload != "fir_os" else 1)
        unit = "GMAC/s" if workload != "fir_os" else "GSamples/s"
        sample = "SincFilter::apply (filter.h:123-140), float samples, 511 taps, {calls} 2^16-sample slices of one channel"
    elif workload == "mavg":
        x = rng.standard_normal(1 << 22).astype(np.float32)
        call, per_call, unit = (lambda: lib.moving_average(x, 511)), x.size, "GSamples/s"
        sample = "MovingAverageFilter{{511}} (filter.h:81-101), float, {calls} 2^22-sample slices"
    elif workload in ("biquad_swept", "biquad_swept_shared"):
        x = rng.standard_normal(1 << 20).astype(np.float32)
        fc = ((1000.0 + 1000.0 * (0.5 + 0.5 * np.sin(0.01 * np.arange(x.size)))) / 44100.0).astype(np.float32)
        call, per_call, unit = (lambda: lib.biquad_swept(x, fc, 60.0 / 44100.0)), x.size, "GSamples/s"
        sample = ("bandPassFilter(fc[t], bw) + one BiQuad step per sample (wind_generator.h:156-164), float samples, "
                  "{calls} 2^20-sample slices of one voice")
    else:
        a, b = cheby_lp4_host()
        x = rng.standard_normal(1 << 22).astype(np.float32)
        call, per_call, unit = (lambda: lib.iir(x, a, b)), x.size, "GSamples/s"
        sample = "Coefficients<4>::operator()(SampleBuffer) (coefficients.h:88-130), float, {calls} 2^22-sample slices"
    calls = 0
    t0 = time.perf_counter()
    while True:  # at least one call, then until the budget is spent
        call()
        calls += 1
        if time.perf_counter() - t0 >= seconds:
            break
    dt = time.perf_counter() - t0
    units = calls * per_call
    return {"value": units / dt / 1e9, "unit": unit, "cores": 1, "kind": kind, "sample": sample.format(calls=calls),
            "seconds": round(dt, 2), "units": units}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with python -m torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # DSP_BENCH_INIT_DIST=1: also at world 1 (under torch.distributed.run --nproc-per-node 1), to rehearse the RCCL
    # set-up, barrier and reductions of the N > 1 run on a one-GPU box
    use_dist = world > 1 or bool(os.environ.get("DSP_BENCH_INIT_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from dsp_amd import multigpu, ops

    global SHARD
    global TIME_SHARD
    if args.scaling == "strong":
        SHARD = (rank, world)
        TIME_SHARD = use_dist

    def barrier():
        if use_dist:
            dist.barrier()

    def measure(workload, steps, warmup):
        """Warm up, then time exactly `steps` passes of one workload between barriers; returns the rank-0 record parts."""
        if workload == "fft_dist":
            step, units, meta = make_fft_dist(torch, dev, world, rank, args.dist_exchange, args.dist_serial)
        elif workload == "fft":
            step, units, meta = make_fft(torch, dev, ops, rank)
        elif workload in ("fft_32k", "fft_real_64k", "fft_f64"):
            step, units, meta = make_fft_other(torch, dev, ops, rank, workload)
        elif workload in ("fir_direct", "fir_direct_valu", "fir_os"):
            step, units, meta = make_fir(torch, dev, ops, rank, {"fir_direct": "direct", "fir_direct_valu": "direct_valu",
                                                                 "fir_os": "overlap_save"}[workload])
        elif workload == "mavg":
            step, units, meta = make_mavg(torch, dev, ops, rank)
        elif workload in ("biquad_swept", "biquad_swept_shared"):
            step, units, meta = make_swept(torch, dev, ops, rank, shared=workload.endswith("shared"))
        else:
            step, units, meta = make_iir(torch, dev, ops, rank, "batch" if workload == "iir_batch" else "scan")
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        # The launches go to torch's current stream (ops passes its handle through the C ABI), so
        # events recorded on that stream bracket exactly the kernels of the timed region.
        stream = torch.cuda.current_stream(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record(stream)
        for _ in range(steps):
            step()
        e1.record(stream)
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        kernel_ms = e0.elapsed_time(e1) / steps
        elapsed = multigpu.max_over_ranks(elapsed, dev)
        all_units = multigpu.sum_over_ranks(float(units), dev)  # ranks may own unequal shares under strong scaling
        del step
        torch.cuda.empty_cache()
        value = all_units * steps / elapsed / 1e9
        launch_s = kernel_ms * 1e-3
        if meta["bound"] == "hbm":
            achieved = units * meta["alg_per_unit"] / launch_s / 1e9
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS}
        else:
            achieved = units * meta["alg_per_unit"] / launch_s / 1e12
            roof = {"bound": "mfma", "achieved": achieved, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / FP32_PEAK_TFLOPS}
        roof.update({"traffic": measured_traffic(workload), "kernel": meta["kernel"],
                     "launch_ms": kernel_ms, "algorithmic": meta["alg_unit"]})
        return {"metric": meta["metric"], "value": value, "unit": meta["unit"], "ms_per_step": elapsed / steps * 1e3,
                "dtype": meta["dtype"], "config": meta["config"], "roofline": roof}

    # BASELINE.json's metric also names the FIR (GMAC/s) and IIR rates: the default run times those workloads too, at
    # their full C3 / C4 sizes, with the same barriers and step count, and reports them beside the headline value.
    # Order: the headline FFT is timed LAST (--headline-first restores the old order).  A memory-bound workload timed
    # first in a fresh process sees the chip still ramping (its first ~40 launches run up to 8 % slow);
    # the first one here is the MFMA-bound direct FIR, which does not care, and everything after it -- the headline
    # included -- is timed at the sustained clocks a serving process runs at.
    secondary = []
    main_rec = None
    if args.workload == "fft" and not args.no_secondary:
        if args.headline_first:
            main_rec = measure(args.workload, args.steps, args.warmup)
        for w in SECONDARY:
            secondary.append((w, measure(w, args.steps, args.warmup)))
    if main_rec is None:
        main_rec = measure(args.workload, args.steps, args.warmup)

    if rank == 0:
        line = {"metric": main_rec["metric"], "value": main_rec["value"], "unit": main_rec["unit"], "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_rec["ms_per_step"],
                "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": main_rec["dtype"],
                "data": "synthetic", "config": main_rec["config"], "roofline": main_rec["roofline"]}
        cpu = world == 1 and not args.no_cpu_baseline and args.workload not in ("fft_32k", "fft_real_64k", "fft_f64")
        if cpu:
            line["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_seconds)
            if args.cpu_threads > 1:  # reported beside it, never the target (the roofline fraction is)
                line["cpu_baseline_all_cores"] = cpu_baseline(args.workload, args.cpu_seconds / 2, args.cpu_threads)
            if args.workload == "fft":
                line["cpu_baseline_f32"] = cpu_baseline("fft_f32", args.cpu_seconds / 4)
        if secondary:
            line["secondary"] = []
            for w, rec in secondary:
                rec = dict(rec, workload=w, steps=args.steps, warmup=args.warmup)
                if cpu:
                    rec["cpu_baseline"] = cpu_baseline(w, min(args.cpu_seconds / 4, 3.0))
                line["secondary"].append(rec)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


def measured_traffic(workload):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json), or None."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(p):
        return None
    try:
        return json.load(open(p)).get(workload, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


if __name__ == "__main__":
    main()
