#!/usr/bin/env python3
"""Benchmark of the MI355X hot path (BASELINE.json metric), one JSON line on stdout.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload fft|fir_direct|fir_os|iir_batch|iir_scan|mavg|...]

Default workload = BASELINE.json configs[1]: 65,536 x 4096-point complex fp32 FFT (FORWARD) of
synthetic N(0,1) noise, inputs resident in HBM.  A "step" is one pass of the hot path over that
batch through the C ABI (dsp_gpu_fft_batch).  N > 1: launched by torch.distributed.run, one rank
per GPU; channels are independent, so every rank owns its own batch of the same size (weak
scaling, no data-path collective) and `value` = samples all ranks processed / max-over-ranks time.

`secondary`    the default run also times the FIR / IIR / moving-average workloads BASELINE's metric names, each with
               the same W warm-up and K timed steps between the same barriers.  They are timed between the two
               measurements of the headline FFT (`cold` first, `value` last): the first memory-bound workload of a
               fresh process is timed while the chip is still ramping, the MFMA-bound direct FIR that
               follows does not care, and every other figure of the line is then taken at sustained clocks.
`roofline`     algorithmic bytes (or flops) of one launch / average launch time measured with
               HIP events recorded on the stream the launches go to, over the timed region.
`step_ms`      mean of the timed region plus median and min of per-step event times (a second pass of K steps with an
               event after every step, straight after the timed region; the timed region itself carries no extra events).
`parity`       rel-L2 of a slice of the SAME device buffers the timed steps wrote, against the CPU oracle (the checker,
               never the thing measured; outside the timed region; rank 0 at N=1 only, skipped by --no-cpu-baseline).
`cold`         default run only: the headline FFT measured FIRST in the fresh process (chip still ramping, DESIGN.md
               4.1) beside `value`, which is measured last at sustained clocks.
`dist`         default run at N > 1: the distributed transform (C5) checks itself on the ranks of this run -- parity at 2^24
               points, local / exchange / serial / pipelined times -- after everything else has been measured, in child
               processes of its own; a failure there shows as `dist.error`, the rest of the line is unaffected.
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
    ap.add_argument("--no-dist-report", action="store_true",
                    help="default fft run at N > 1: skip the `dist` section (the distributed transform's self-check)")
    ap.add_argument("--dist-budget", type=float, default=90.0,
                    help="seconds the `dist` section of the default N > 1 run may take before the line goes out without it")
    ap.add_argument("--dist-child", action="store_true", help=argparse.SUPPRESS)  # the `dist` section's own process
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="default fft run only: skip the FIR / IIR / moving-average workloads reported under `secondary`")
    ap.add_argument("--headline-first", action="store_true",
                    help="default fft run only: report the headline FFT as measured FIRST (cold chip) as `value` and the "
                         "measurement after the secondary workloads as `sustained` (default: the other way round, `cold`)")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="budget of the CPU baseline sample")
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

    def parity():  # first 8 transforms of the timed buffers against fft0 in double (fft.h:31-55)
        k = min(8, batch)
        xs, ys = x[:k].cpu().numpy(), out[:k].cpu().numpy()

        def check(orc):
            want = orc.fft0(xs.astype(np.complex128), False)
            return _rel_l2(ys, want), 1e-5 * 12, f"{k} of {batch} transforms vs the oracle's fft0 (complex double)"
        return check

    meta["parity"] = parity
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
                    alg_per_unit=8.0, alg_unit="bytes/sample (4 B read + 4 B write)",
                    kernel="fir_os_wave2_kernel<2> (+ fir_overlap_save_kernel<12> on the 2 x 1,024 edge block pairs, ~36 us, and a 1 KB memset of the draw counters)")
        units = channels * n
    meta["config"] = {"workload": f"C3: 511-tap windowed-sinc FIR ({mode}), 1,024 ch x 2^20 samples fp32 noise, per GPU",
                      "channels": channels, "n": n, "ntaps": ntaps, "sharding": "channels, no collective"}

    def parity():  # the first 2^16 samples of the last channel (a causal filter: a prefix is self-contained)
        m = 1 << 16
        xs, ys = x[-1, :m].cpu().numpy(), y[-1, :m].cpu().numpy()

        def check(orc):
            want = orc.fir(xs.astype(np.float64), taps)
            tol = 1e-6 if mode.startswith("direct") else 1e-5 * 12
            return _rel_l2(ys, want), tol, "2^16-sample prefix of one channel vs SincFilter::apply in double"
        return check

    meta["parity"] = parity
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

    def parity():  # a centred window: outputs up to m - pts/2 - 1 of a prefix of m samples are the full signal's
        m = 1 << 16
        xs, ys = x[-1, :m].cpu().numpy(), y[-1, : m - pts].cpu().numpy()

        def check(orc):
            want = orc.moving_average(xs.astype(np.float64), pts)[: m - pts]
            return _rel_l2(ys, want), 1e-6, "2^16-sample prefix of one channel vs MovingAverageFilter in double"
        return check

    meta["parity"] = parity
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

    def parity():  # zero initial state: a prefix is self-contained
        m = min(n, 1 << 18)
        xs, ys = x[-1:, :m].cpu().numpy(), y[-1:, :m].cpu().numpy()

        def check(orc):
            want = orc.iir(xs.astype(np.float64), a, b)
            return _rel_l2(ys, want), 1e-4, f"{m}-sample prefix of one channel vs Coefficients<4> in double"
        return check

    if not time_sharded:
        meta["parity"] = parity
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
    import torch.distributed as dist

    from dsp_amd import dist_fft, ops
    log2n = 27 + (world.bit_length() - 1)
    per_rank = (1 << log2n) // world
    g = torch.Generator(device=dev).manual_seed(2030 + rank)
    x = 0.01 * torch.view_as_complex(torch.randn(per_rank, 2, device=dev, generator=g))
    stream = torch.cuda.current_stream(dev)

    def timed(fn, reps=3):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    if exchange == "rccl":  # all of it inside the library: dsp_gpu_fft_dist with grouped ncclSend/ncclRecv
        comm = dist_fft.RcclCommunicator(rank, world)
        plan = dist_fft.NativePlan(log2n, rank, world, nccl_comm=comm.handle)
        y = torch.empty_like(x)

        def step():
            plan.run(x, False, out=y, serial=serial)

        def transform(t, lg):  # the same route at another size (parity check)
            c = dist_fft.NativePlan(lg, rank, world, nccl_comm=comm.handle)
            try:
                return c.run(t, False, serial=serial)
            finally:
                torch.cuda.synchronize()
                c.close()
    else:

        def step():  # the library's plan handle runs the local phases; torch.distributed (RCCL) the exchanges
            dist_fft.fft_dist(x, log2n, False)

        def transform(t, lg):
            return dist_fft.fft_dist(t, lg, False)

    def report():
        """Run BEFORE the timed region, on every rank: (1) parity of the distributed route at 2^24 points against the
        single-GPU transform of the gathered input; (2) event-timed anatomy: local passes alone, each exchange alone,
        the serial and the pipelined schedule, and the share of the exchange time the pipeline hides."""
        lg = 24
        n_loc = (1 << lg) // world
        gg = torch.Generator(device=dev).manual_seed(77 + rank)
        xl = torch.view_as_complex(torch.randn(n_loc, 2, device=dev, generator=gg))
        got = transform(xl, lg)
        if world > 1:
            mine = torch.view_as_real(xl).contiguous()
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            full = torch.view_as_complex(torch.cat(parts))
        else:
            full = xl
        want = ops.fft0(full.reshape(1, -1), False).reshape(-1)[rank * n_loc:(rank + 1) * n_loc]
        err = float((torch.linalg.vector_norm(got - want) / torch.linalg.vector_norm(want)).item())
        rep = {"parity_2p24": {"rel_l2_vs_single_gpu": err, "tol": 2e-5 * lg, "ok": bool(err <= 2e-5 * lg),
                               "what": f"2^24-point transform over {world} rank(s) against dsp_gpu_fft_batch of the gathered input"}}
        # anatomy at the benchmark's size
        phases = dist_fft.native_phases(log2n, rank, world, False)

        def local_only():
            t = x
            for ph in phases:
                t = ph(t)

        rep["local_passes_ms"] = timed(local_only)
        send = torch.view_as_real(x).contiguous()
        recv = torch.empty_like(send)
        if dist.is_initialized():
            rep["exchange_ms"] = [timed(lambda: dist.all_to_all_single(recv, send)) for _ in range(3)]
            rep["exchange_what"] = (f"one all_to_all_single of {send.numel() * 4 // max(world, 1)} B per peer, "
                                    "event-timed alone, three times")
        if exchange == "rccl":
            rep["serial_ms"] = timed(lambda: plan.run(x, False, out=y, serial=True))
            rep["pipelined_ms"] = timed(lambda: plan.run(x, False, out=y, serial=False))
            ex = rep["serial_ms"] - rep["local_passes_ms"]
            rep["exchange_in_serial_ms"] = ex
            rep["overlap_fraction"] = (rep["serial_ms"] - rep["pipelined_ms"]) / ex if ex > 0 else None
            rep["overlap_what"] = ("(serial - pipelined) / (serial - local passes alone): the share of the exchange time the "
                                   "pipelined schedule hides; scripts/dist_overlap.py computes the same from a kernel trace")
            rep["gsamples_per_s"] = {"serial": (1 << log2n) / rep["serial_ms"] / 1e6,
                                     "pipelined": (1 << log2n) / rep["pipelined_ms"] / 1e6,
                                     "what": f"one 2^{log2n}-point transform over {world} rank(s), this rank's event times"}
            # the other exchange route (torch.distributed between the plan's phases) against the same answer
            if dist.is_initialized():
                got_t = dist_fft.fft_dist(xl, lg, False)
                err_t = float((torch.linalg.vector_norm(got_t - want) / torch.linalg.vector_norm(want)).item())
                rep["parity_2p24_torch_exchange"] = {"rel_l2_vs_single_gpu": err_t, "tol": 2e-5 * lg,
                                                     "ok": bool(err_t <= 2e-5 * lg)}
        else:
            rep["overlap_fraction"] = 0.0
            rep["overlap_what"] = "phases and all_to_all_single alternate on one stream: nothing overlaps by construction"
        del send, recv
        return rep

    meta = dict(metric="distributed_fft_throughput", unit="GSamples/s", dtype="f32", bound="hbm",
                alg_per_unit=48.0, alg_unit="bytes/sample (3 local passes x 16 B; the exchanges ride xGMI)",
                kernel="fft_column_kernel + fft_lds_kernel + swap_axes_kernel + transpose_blocks_kernel",
                config={"workload": f"C5: one 2^{log2n}-point complex fp32 FFT, six-step, {world} GPU(s), 2^27 points per GPU",
                        "log2n": log2n,
                        "exchange": (f"{plan.exchanges(serial=serial)} x grouped ncclSend/ncclRecv inside dsp_gpu_fft_dist"
                                     f" ({'serial' if serial or plan.exchanges() == 3 else 'pipelined'} schedule)")
                        if exchange == "rccl" else "3 x all_to_all_single (RCCL)"},
                dist_report=report)
    return step, per_rank, meta


def _rel_l2(got, want):
    got, want = np.asarray(got), np.asarray(want)
    return float(np.linalg.norm(got.astype(want.dtype) - want) / max(float(np.linalg.norm(want)), 1e-300))


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
    if args.dist_child:
        return dist_child_main(args)
    # stdout carries THE line and nothing else: RCCL prints a version banner to file descriptor 1 when a communicator is
    # made, so everything but the line goes to stderr from here on
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)
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

    def measure(workload, steps, warmup, want_parity=True):
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
        extra = {}
        if "dist_report" in meta:  # self-verification of the multi-GPU route, before anything is timed
            extra["dist"] = meta["dist_report"]()
            if not extra["dist"]["parity_2p24"]["ok"]:
                raise SystemExit(f"fft_dist parity failed on rank {rank}: {extra['dist']['parity_2p24']}")
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
        # per-step times: a second pass of the same K steps straight after, one event after every step (kept out of the
        # timed region so that it stays exactly the K launches the contract names)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        marks[0].record(stream)
        for i in range(steps):
            step()
            marks[i + 1].record(stream)
        torch.cuda.synchronize()
        per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
        step_ms = {"mean": kernel_ms, "median": per_step[len(per_step) // 2] if len(per_step) % 2 else
                   0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2]), "min": per_step[0],
                   "max": per_step[-1],
                   "what": "mean: HIP events around the timed region / K; median, min, max: K more steps, an event after each"}
        elapsed = multigpu.max_over_ranks(elapsed, dev)
        all_units = multigpu.sum_over_ranks(float(units), dev)  # ranks may own unequal shares under strong scaling
        if use_dist:  # what every rank processed per step (strong scaling: the shares; weak: N equal batches)
            mine = torch.tensor([float(units)], dtype=torch.float64, device=dev)
            shares = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(shares, mine)
            extra["per_rank_units"] = [int(t.item()) for t in shares]
        if want_parity and world == 1 and not args.no_cpu_baseline and "parity" in meta:
            # a slice of the buffers the timed steps just wrote comes to the host now; the CPU oracle checks it after ALL
            # workloads have been timed (resolve_parity), so that no workload starts behind a stretch of CPU-only work
            extra["_parity_check"] = meta["parity"]()
        meta.pop("parity", None)  # the closures hold the device buffers
        meta.pop("dist_report", None)
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
        traffic = measured_traffic(workload)
        roof.update({"traffic": traffic,
                     "traffic_source": "profiles/traffic.json (rocprofv3 PMC passes committed with the tree, not this run)"
                     if traffic is not None else None,
                     "kernel": meta["kernel"], "launch_ms": kernel_ms, "algorithmic": meta["alg_unit"]})
        rec = {"metric": meta["metric"], "value": value, "unit": meta["unit"], "ms_per_step": elapsed / steps * 1e3,
               "dtype": meta["dtype"], "config": meta["config"], "roofline": roof, "step_ms": step_ms}
        rec.update(extra)
        return rec

    # N > 1 (or the world-1 rehearsal): prove that RCCL really spans the ranks the launcher promised before timing anything
    rccl_ranks = None
    if use_dist:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
        if rccl_ranks != args.gpus or dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but RCCL sums {rccl_ranks} ranks (world size {dist.get_world_size()})")
        if rank == 0:
            print(f"[bench] rccl_ranks == {rccl_ranks} == --gpus {args.gpus}: ok", file=sys.stderr, flush=True)

    # BASELINE.json's metric also names the FIR (GMAC/s) and IIR rates: the default run times those workloads too, at
    # their full C3 / C4 sizes, with the same barriers and step count, and reports them beside the headline value.
    # Order: the headline FFT is timed FIRST (reported as `cold`: a memory-bound workload timed first in a fresh process
    # sees the chip still ramping -- its first ~40 launches run up to 8 % slow) AND LAST (`value`: the
    # sustained clocks a serving process runs at).  The first secondary workload is the MFMA-bound direct FIR, which
    # does not care about the ramp.  --headline-first reports the cold measurement as `value` instead.
    secondary = []
    main_rec = cold_rec = None
    if args.workload == "fft" and not args.no_secondary:
        cold_rec = measure(args.workload, args.steps, args.warmup, want_parity=False)
        for w in SECONDARY:
            secondary.append((w, measure(w, args.steps, args.warmup)))
    main_rec = measure(args.workload, args.steps, args.warmup)

    def resolve_parity(rec):
        check = rec.pop("_parity_check", None)
        if check is not None:
            from oracle import load  # the checker, never the thing measured
            err, tol, what = check(load("oracle"))
            rec["parity"] = {"rel_l2": err, "tol": tol, "ok": bool(err <= tol), "what": what}

    for rec in [main_rec, cold_rec] + [r for _, r in secondary]:
        if rec is not None:
            resolve_parity(rec)
    if cold_rec is not None and args.headline_first:
        main_rec, cold_rec = dict(cold_rec, parity=main_rec.get("parity")), main_rec

    def emit(dist_rec=None):
        """Rank 0 prints THE line."""
        if rank == 0:
            line = {"metric": main_rec["metric"], "value": main_rec["value"], "unit": main_rec["unit"], "n_gpus": world,
                    "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_rec["ms_per_step"],
                    "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": main_rec["dtype"],
                    "data": "synthetic", "config": main_rec["config"], "roofline": main_rec["roofline"],
                    "step_ms": main_rec["step_ms"]}
            for k in ("parity", "dist", "per_rank_units"):
                if main_rec.get(k) is not None:
                    line[k] = main_rec[k]
            if dist_rec is not None:
                line["dist"] = dist_rec
            if rccl_ranks is not None:
                line["rccl_ranks"] = rccl_ranks
            if cold_rec is not None:
                line["cold" if not args.headline_first else "sustained"] = {
                    "value": cold_rec["value"], "ms_per_step": cold_rec["ms_per_step"], "launch_ms": cold_rec["roofline"]["launch_ms"],
                    "frac": cold_rec["roofline"]["frac"], "step_ms": cold_rec["step_ms"],
                    "what": ("the same K steps measured first in the fresh process, before the secondary workloads"
                             if not args.headline_first else "the same K steps measured last, after the secondary workloads")}
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
                        rec["cpu_baseline"] = cpu_baseline(w, min(args.cpu_seconds / 4, 1.5))
                        if args.cpu_threads > 1:
                            rec["cpu_baseline_all_cores"] = cpu_baseline(w, min(args.cpu_seconds / 6, 1.0), args.cpu_threads)
                    line["secondary"].append(rec)
            sys.stdout.flush()
            with os.fdopen(line_fd, "w") as out:
                out.write(json.dumps(line) + "\n")

    # Default run under torch.distributed: the line also carries `dist`, the distributed transform (C5, the one path
    # with a real exchange step) checking itself on the ranks of this very run.  It runs AFTER everything above has been
    # measured and in child processes of its own (one per rank, a process group of their own), so that a failure, a
    # hang or a crash in it costs this section and nothing else: the line then goes out with `dist.error`.
    dist_rec = None
    if use_dist and args.workload == "fft" and not args.no_dist_report and not args.no_secondary:
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        dist_rec = run_dist_children(dist, rank, world, local_rank, args.dist_budget)
    emit(dist_rec)
    if use_dist:
        dist.destroy_process_group()


def run_dist_children(dist, rank, world, local_rank, budget_s):
    """One child process per rank runs `bench.py --dist-child`; rank 0's child prints the report as JSON."""
    import socket
    import subprocess
    import tempfile
    box = [None]
    if rank == 0:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            box[0] = sk.getsockname()[1]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(box[0]), "RANK": str(rank), "WORLD_SIZE": str(world),
                "LOCAL_RANK": str(local_rank)})
    with tempfile.TemporaryFile() as err:
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--dist-child", "--gpus", str(world),
                                  "--dist-budget", str(budget_s)], env=env, stdout=subprocess.PIPE, stderr=err, cwd=ROOT)
        try:
            out, _ = child.communicate(timeout=budget_s + 15.0)
        except subprocess.TimeoutExpired:
            child.kill()
            out, _ = child.communicate()
            return {"error": f"rank {rank}'s child did not finish within {budget_s + 15.0:.0f} s"}
        if rank != 0:
            return None
        try:
            return json.loads(out.decode().strip().splitlines()[-1])
        except Exception:  # no report: say why, with the end of the child's stderr
            err.seek(0)
            tail = err.read().decode(errors="replace")[-600:]
            return {"error": f"rank 0's child ended with status {child.returncode} and no report", "stderr_tail": tail}


def dist_child_main(args):
    """`--dist-child`: the `dist` section of the default N > 1 line (run_dist_children), in a process group of its own."""
    import threading

    import torch
    import torch.distributed as dist
    world = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rep = {}

    def out_of_time():  # a rank that waits for a peer that died: say so and leave before the parent has to kill us
        if rank == 0:
            print(json.dumps(dict(rep, error=f"the dist section did not finish within {args.dist_budget:.0f} s")), flush=True)
        os._exit(3)

    timer = threading.Timer(args.dist_budget, out_of_time)
    timer.daemon = True
    timer.start()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist.init_process_group("nccl", device_id=dev)
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    rep["ranks"] = int(ones.item())
    rehearse = os.environ.get("DSP_BENCH_DIST_CHILD_TEST")  # scripts/bench_rehearsal.sh: the parent's line must survive both
    if rehearse == "crash":
        os.abort()
    if rehearse == "hang":
        time.sleep(1e6)
    try:
        _, _, meta = make_fft_dist(torch, dev, world, rank, "rccl")
        rep.update(meta["dist_report"]())
        rep["config"] = meta["config"]
        torch.cuda.synchronize()
    except Exception as e:  # reported, not raised: the parent's line says what went wrong
        rep["error"] = f"rank {rank}: {type(e).__name__}: {e}"[:600]
    ok = "error" not in rep and rep.get("parity_2p24", {}).get("ok", False)
    oks = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(oks)  # every rank, whatever happened to it (a peer stuck elsewhere: the timer above)
    rep["ranks_with_parity_ok"] = int(oks.item())
    timer.cancel()
    if rank == 0:
        print(json.dumps(rep), flush=True)
    sys.stdout.flush()
    os._exit(0)  # no teardown of communicators that may be half-dead


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
