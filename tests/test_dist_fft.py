"""The multi-GPU six-step FFT (dsp_amd/dist_fft.py): index logic on CPU (virtual ranks and a real
world_size-2 gloo group, with a numpy stand-in for the local steps), the HIP local steps on the GPU."""
import os

import numpy as np
import pytest

from conftest import rel_l2


class NumpyLocalOps:
    """Stand-in for GpuLocalOps on CPU torch tensors (TEST ONLY).  Reference convention: FORWARD uses
    e^{+i...} and nothing normalises, i.e. FORWARD = n * ifft, INVERSE = fft."""

    @staticmethod
    def _np(t):
        return t.numpy().astype(np.complex128)

    @staticmethod
    def _t(a):
        import torch
        return torch.from_numpy(np.ascontiguousarray(a).astype(np.complex64).reshape(-1))

    def _dft(self, a, axis, inverse):
        return np.fft.fft(a, axis=axis) if inverse else np.fft.ifft(a, axis=axis) * a.shape[axis]

    def swap_axes(self, t, a, b, c, tw_log2n=0, row0=0, inverse=False):
        x = self._np(t).reshape(a, b, c)
        if tw_log2n:
            n = 1 << tw_log2n
            row = (row0 + np.arange(b))[None, :, None]
            col = (np.arange(a)[:, None, None] * c + np.arange(c)[None, None, :])
            x = x * np.exp((-2j if inverse else 2j) * np.pi * ((row * col) % n) / n)
        return self._t(x.transpose(1, 0, 2))

    def fft_columns(self, t, n, cols, inverse=False):
        return self._t(self._dft(self._np(t).reshape(n, cols), 0, inverse))

    def fft_rows(self, t, rows, n, inverse=False):
        return self._t(self._dft(self._np(t).reshape(rows, n), 1, inverse))

    def transpose_blocks(self, t, blocks, rows, cols):
        return self._t(self._np(t).reshape(blocks, rows, cols).transpose(2, 0, 1))


_SPECTRA = {}


def _oracle_spectrum(orc, x, log2n, inverse):
    """orc.fft0 of the callback tests' signal, kept across their parameter combinations (2^24 points take seconds)."""
    key = (log2n, bool(inverse))
    if key not in _SPECTRA:
        _SPECTRA[key] = orc.fft0(x.astype(np.complex128), inverse)
    return _SPECTRA[key]


def _signal(log2n, seed=2030):
    rng = np.random.default_rng(seed)
    n = 1 << log2n
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)


@pytest.mark.parametrize("log2n,world", [(6, 1), (8, 2), (9, 2), (10, 4), (11, 4), (12, 8)])
@pytest.mark.parametrize("inverse", [False, True])
def test_virtual_ranks_cpu(orc, log2n, world, inverse):
    import torch
    from dsp_amd import dist_fft
    x = _signal(log2n)
    parts = [torch.from_numpy(p.copy()) for p in np.split(x, world)]
    out = dist_fft.simulate(parts, log2n, inverse, ops=NumpyLocalOps())
    got = np.concatenate([o.numpy() for o in out])
    assert rel_l2(got, orc.fft0(x.astype(np.complex128), inverse)) <= 1e-5 * log2n


def test_plan_rejects_bad_shapes():
    from dsp_amd import dist_fft
    with pytest.raises(ValueError):
        dist_fft.make_plan(10, 3)
    with pytest.raises(ValueError):
        dist_fft.make_plan(4, 8)
    p = dist_fft.make_plan(30, 8)
    assert (p.n1, p.n2, p.rows, p.cols) == (1 << 15, 1 << 15, 4096, 4096)
    assert p.rows * p.cols * 8 == 128 << 20  # one 128 MiB block per peer pair


def _gloo_worker(rank, world, rdzv, log2n, outdir):
    import torch
    import torch.distributed as dist
    from dsp_amd import dist_fft, multigpu
    dist.init_process_group("gloo", init_method=f"file://{rdzv}", rank=rank, world_size=world)
    x = _signal(log2n)
    lo, hi = multigpu.shard_range(x.size, rank, world)
    for inverse in (False, True):
        mine = torch.from_numpy(x[lo:hi].copy())
        out = dist_fft.fft_dist(mine, log2n, inverse, ops=NumpyLocalOps())
        np.save(os.path.join(outdir, f"out_{int(inverse)}_{rank}.npy"), out.numpy())
    # the bench's timing reduction: max over ranks, sum over ranks
    assert multigpu.max_over_ranks(1.0 + rank) == float(world)
    assert multigpu.sum_over_ranks(1.0) == float(world)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_gloo_process_group(tmp_path, orc, world):
    """The real torch.distributed path (all_to_all_single) with one process per rank, gloo on CPU."""
    import torch.multiprocessing as mp
    log2n = 10
    rdzv = str(tmp_path / "rdzv")
    mp.spawn(_gloo_worker, args=(world, rdzv, log2n, str(tmp_path)), nprocs=world, join=True)
    x = _signal(log2n)
    for inverse in (False, True):
        got = np.concatenate([np.load(tmp_path / f"out_{int(inverse)}_{r}.npy") for r in range(world)])
        assert rel_l2(got, orc.fft0(x.astype(np.complex128), inverse)) <= 1e-4


def test_shard_range_covers_everything():
    from dsp_amd import multigpu
    for total, world in ((65536, 8), (1024, 4), (10, 3), (262144, 2)):
        spans = [multigpu.shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    with pytest.raises(ValueError):
        multigpu.shard_range(8, 2, 2)


def test_native_plan_arguments():
    """dsp_gpu_fft_dist_plan_create: argument checks and the slice size (no GPU work)."""
    import ctypes as C
    from dsp_amd import capi
    L = capi.lib()
    null_fn = capi.EXCHANGE_FN(0)
    h = C.c_void_p()
    for log2n, rank, world in ((7, 0, 1), (33, 0, 1), (20, 0, 3), (20, 2, 2), (20, -1, 2), (9, 0, 16), (8, 0, 16)):
        assert L.dsp_gpu_fft_dist_plan_create(C.byref(h), log2n, rank, world, null_fn, None) == capi.EINVAL, (log2n, rank, world)
        assert not h.value
    assert L.dsp_gpu_fft_dist_plan_create(None, 20, 0, 1, null_fn, None) == capi.EINVAL
    assert L.dsp_gpu_fft_dist_plan_create_rccl(C.byref(h), 20, 0, 1, None) == capi.EINVAL
    for log2n, world in ((30, 8), (20, 4), (8, 1), (9, 8)):
        assert L.dsp_gpu_fft_dist_plan_create(C.byref(h), log2n, world - 1, world, null_fn, None) == capi.OK
        assert L.dsp_gpu_fft_dist_local_points(h) == (1 << log2n) // world
        assert dist_fft_plan_matches(log2n, world)
        assert L.dsp_gpu_fft_dist_plan_destroy(h) == capi.OK
    assert L.dsp_gpu_fft_dist_plan_destroy(None) == capi.OK
    assert L.dsp_gpu_fft_dist_local_points(None) == 0


def dist_fft_plan_matches(log2n, world):
    from dsp_amd import dist_fft
    p = dist_fft.make_plan(log2n, world)       # the host-side statement accepts the same shapes
    return p.rows * p.n2 == (1 << log2n) // world


# ---- GPU: the HIP local steps ---------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("log2n,world", [(16, 1), (16, 2), (20, 2), (20, 4), (20, 8), (18, 8), (21, 8)])
def test_virtual_ranks_gpu(orc, dev, log2n, world):
    import torch
    from dsp_amd import dist_fft
    x = _signal(log2n, seed=log2n)
    for inverse in (False, True):
        parts = [torch.from_numpy(p.copy()).to(dev) for p in np.split(x, world)]
        out = dist_fft.simulate(parts, log2n, inverse)
        got = torch.cat(out).cpu().numpy()
        assert rel_l2(got, orc.fft0(x.astype(np.complex128), inverse)) <= 1e-5 * log2n


@pytest.mark.gpu
def test_local_steps_against_numpy(dev):
    import torch
    from dsp_amd import dist_fft
    g, ref = dist_fft.GpuLocalOps(), NumpyLocalOps()
    rng = np.random.default_rng(5)
    mk = lambda n: (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    for (a, b, c) in ((4, 8, 256), (1, 16, 64), (8, 512, 128)):
        x = mk(a * b * c)
        for tw, row0, inv in ((0, 0, False), (24, 4096 * 3, False), (30, 4096 * 7, True)):
            got = g.swap_axes(torch.from_numpy(x).to(dev), a, b, c, tw, row0, inv).cpu().numpy()
            want = ref.swap_axes(torch.from_numpy(x), a, b, c, tw, row0, inv).numpy()
            assert rel_l2(got, want) <= 1e-6
    for n, cols in ((16, 256), (256, 16), (1024, 256), (32768, 256), (65536, 256)):
        x = mk(n * cols)
        for inv in (False, True):
            got = g.fft_columns(torch.from_numpy(x).to(dev), n, cols, inv).cpu().numpy()
            want = ref.fft_columns(torch.from_numpy(x), n, cols, inv).numpy()
            assert rel_l2(got, want) <= 1e-5 * np.log2(n)
    for (blocks, rows, cols) in ((1, 32, 32), (3, 50, 70), (8, 128, 512)):
        x = mk(blocks * rows * cols)
        got = g.transpose_blocks(torch.from_numpy(x).to(dev), blocks, rows, cols).cpu().numpy()
        assert np.array_equal(got, ref.transpose_blocks(torch.from_numpy(x), blocks, rows, cols).numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("log2n2,rows,world", [(12, 16, 8), (12, 5, 1), (13, 8, 4), (14, 3, 2), (15, 4, 16), (16, 2, 8)])
def test_fused_rows_exchange_equals_three_steps(dev, log2n2, rows, world):
    """dsp_gpu_fft_rows_exchange (two HBM passes) against unpack*twiddle -> row FFT -> pack done separately."""
    import torch
    from dsp_amd import dist_fft
    g = dist_fft.GpuLocalOps()
    n2 = 1 << log2n2
    C = n2 // world
    rng = np.random.default_rng(log2n2 * 100 + world)
    x = (rng.standard_normal(rows * n2) + 1j * rng.standard_normal(rows * n2)).astype(np.complex64)
    xd = torch.from_numpy(x).to(dev)
    tw_log2n, row0 = log2n2 + 14, 12345 * rows
    for inv in (False, True):
        fused = g.fft_rows_exchange(xd, rows, n2, world, tw_log2n, row0, inv)
        assert fused is not None
        t = g.swap_axes(xd, world, rows, C, tw_log2n=tw_log2n, row0=row0, inverse=inv)
        t = g.fft_rows(t, rows, n2, inv)
        want = g.swap_axes(t, rows, world, C)
        err = torch.linalg.vector_norm((fused - want).view(torch.float32).double()) / torch.linalg.vector_norm(want.view(torch.float32).double())
        assert float(err) <= 2e-6, (inv, float(err))
    # shapes the fused passes do not cover are refused, not mangled
    assert g.fft_rows_exchange(xd[: 4 * 256], 4, 256, 1, 20, 0) is None


@pytest.mark.gpu
def test_c5_shape_on_one_gpu(dev):
    """2^24 points over 8 virtual ranks with the real kernels: against the single-GPU transform, plus
    exact-bin tones (the C5 recipe: tones + small noise)."""
    import torch
    from dsp_amd import dist_fft, ops
    log2n, G = 24, 8
    n = 1 << log2n
    g = torch.Generator(device=dev).manual_seed(2030)
    x = 0.01 * torch.view_as_complex(torch.randn(n, 2, device=dev, generator=g))
    idx = torch.arange(n, device=dev, dtype=torch.int64)
    bins = [3, 4097, 65536 + 11, n // 2 + 5, n - 1]
    for k in bins:
        ph = ((idx * k) % n).double() * (2 * np.pi / n)
        x = x + torch.complex(torch.cos(ph), -torch.sin(ph)).to(torch.complex64)
    out = torch.cat(dist_fft.simulate(list(x.reshape(G, -1)), log2n))
    single = ops.fft0(x[None])[0]
    err = torch.linalg.vector_norm((out - single).view(torch.float32).double()) / torch.linalg.vector_norm(single.view(torch.float32).double())
    assert float(err) <= 1e-5 * log2n
    top = torch.topk(out.abs(), len(bins)).indices.cpu().tolist()
    assert sorted(top) == sorted(bins)


@pytest.mark.gpu
def test_nccl_world_of_one(dev, tmp_path, orc):
    """The torch.distributed code path itself on the GPU (RCCL group of one rank)."""
    import torch
    import torch.distributed as dist
    from dsp_amd import dist_fft
    dist.init_process_group("nccl", init_method=f"file://{tmp_path}/rdzv", rank=0, world_size=1, device_id=dev)
    try:
        x = _signal(16)
        out = dist_fft.fft_dist(torch.from_numpy(x).to(dev), 16).cpu().numpy()
        assert rel_l2(out, orc.fft0(x.astype(np.complex128))) <= 1.6e-4
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("log2n", [16, 18, 24])
def test_native_plan_whole_transform_world_of_one(dev, log2n):
    """dsp_gpu_fft_dist end to end with one rank (no exchange): against the single-GPU transform."""
    import torch
    from dsp_amd import dist_fft, ops
    g = torch.Generator(device=dev).manual_seed(log2n)
    x = torch.view_as_complex(torch.randn(1 << log2n, 2, device=dev, generator=g))
    keep = x.clone()
    plan = dist_fft.NativePlan(log2n, 0, 1)
    try:
        for inverse in (False, True):
            got = plan.run(x, inverse)
            want = ops.fft0(x[None], inverse=inverse)[0]
            err = torch.linalg.vector_norm((got - want).view(torch.float32).double()) / torch.linalg.vector_norm(want.view(torch.float32).double())
            assert float(err) <= 1e-5 * log2n
            assert torch.equal(x, keep)          # the input is not modified
        with pytest.raises(Exception):
            plan.run(x, out=x)                   # in == out is refused
    finally:
        plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("log2n", [16, 19, 24])
def test_native_plan_transposed_out_world_of_one(dev, log2n):
    """DSP_GPU_DIST_TRANSPOSED_OUT on one rank: out[r][k2] = X[r + N1 k2]."""
    import torch
    from dsp_amd import dist_fft, ops
    g = torch.Generator(device=dev).manual_seed(log2n)
    x = torch.view_as_complex(torch.randn(1 << log2n, 2, device=dev, generator=g))
    n1 = 1 << ((log2n + 1) // 2)
    plan = dist_fft.NativePlan(log2n, 0, 1)
    try:
        for inverse in (False, True):
            got = plan.run(x, inverse, transposed=True).reshape(n1, -1)
            want = ops.fft0(x[None], inverse=inverse)[0].reshape(-1, n1).T  # X[k1 + N1 k2] -> [k1][k2]
            err = torch.linalg.vector_norm((got - want).contiguous().view(torch.float32).double()) / torch.linalg.vector_norm(want.contiguous().view(torch.float32).double())
            assert float(err) <= 1e-5 * log2n
    finally:
        plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("serial", [False, True])
@pytest.mark.parametrize("transposed", [False, True])
@pytest.mark.parametrize("log2n,world", [(16, 2), (20, 4), (18, 8), (22, 8), (24, 1), (24, 2), (24, 8)])
def test_native_plan_with_exchange_callback(orc, dev, log2n, world, transposed, serial):
    """dsp_gpu_fft_dist_ex with a caller-supplied exchange: `world` host threads, one plan each, on one GPU.  The
    callback trades blocks with device copies on the stream it is handed (the caller's in the serial schedule, the
    plan's own in the pipelined one) and honours the contract of include/dsp_gpu.h with events: a receiver copies only
    once the sender's stream has produced the block, and a sender's stream moves on only once every receiver has
    copied.  A baton keeps one thread at a time inside the library (calls that use the library's scratch must not
    overlap on a device).  Pipelined and serial schedules must agree bit for bit."""
    import ctypes as C
    import threading
    import torch
    from dsp_amd import dist_fft
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    hip.hipEventDestroy.argtypes = [C.c_void_p]
    D2D, NO_TIMING = 3, 2
    x = _signal(log2n, seed=7 + log2n)
    stream = torch.cuda.current_stream(dev).cuda_stream
    baton, barrier = threading.Lock(), threading.Barrier(world)
    max_calls = 12
    sends = [[None] * world for _ in range(max_calls)]
    calls = [0] * world
    errors = []

    def new_event():
        e = C.c_void_p()
        assert hip.hipEventCreateWithFlags(C.byref(e), NO_TIMING) == 0
        return e

    ready = [[new_event() for _ in range(world)] for _ in range(max_calls)]
    done = [[new_event() for _ in range(world)] for _ in range(max_calls)]

    def rendezvous():
        baton.release()
        try:
            barrier.wait(timeout=60)
        finally:
            baton.acquire()

    def make_exchange(rank):
        def exchange(send, recv, nbytes, w, s):
            k = calls[rank]
            calls[rank] += 1
            sends[k][rank] = send
            rc = 0 if hip.hipEventRecord(ready[k][rank], s) == 0 else -2  # my blocks are ready at this point of my stream
            rendezvous()
            for src in range(w):  # block `rank` of every sender, in sender order
                if hip.hipStreamWaitEvent(s, ready[k][src], 0) != 0:
                    rc = -2
                if hip.hipMemcpyAsync(recv + src * nbytes, sends[k][src] + rank * nbytes, nbytes, D2D, s) != 0:
                    rc = -2
            if hip.hipEventRecord(done[k][rank], s) != 0:
                rc = -2
            rendezvous()
            for dst in range(w):  # my send buffer is free once every receiver has copied its block out of it
                if hip.hipStreamWaitEvent(s, done[k][dst], 0) != 0:
                    rc = -2
            return rc
        return exchange

    try:
        for inverse in (False, True):
            parts = [torch.from_numpy(p.copy()).to(dev) for p in np.split(x, world)]
            outs = [torch.full_like(p, float("nan")) for p in parts]
            for c in range(world):
                calls[c] = 0
            plans = [dist_fft.NativePlan(log2n, r, world, exchange=make_exchange(r)) for r in range(world)]
            flags = plans[0].flags(transposed, serial)
            expected = plans[0].exchanges(transposed, serial)
            assert expected == (2 if transposed else 3) or (not serial and expected in (4, 6, 8, 12))
            if log2n >= 24 and not serial:
                assert expected > 3, "a transform of this size is expected to take the pipelined schedule"

            def worker(r):
                torch.cuda.set_device(dev)
                with baton:
                    try:
                        plans[r].capi.check(plans[r].capi.lib().dsp_gpu_fft_dist_ex(plans[r]._handle, parts[r].data_ptr(),
                                                                                   outs[r].data_ptr(), int(inverse), flags, stream))
                    except Exception as e:  # noqa: BLE001
                        errors.append(e)
                        barrier.abort()

            threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
            for t in threads:
                t.start()
            for t in threads:
                t.join(timeout=120)
            torch.cuda.synchronize()
            assert not errors, errors
            assert calls == [expected] * world
            got = torch.cat(outs).cpu().numpy()
            want = _oracle_spectrum(orc, x, log2n, inverse)
            if transposed:  # the ranks' row blocks stacked: [k1][k2] = X[k1 + N1 k2]
                n1 = 1 << ((log2n + 1) // 2)
                want = want.reshape(-1, n1).T.reshape(-1)
            assert rel_l2(got, want) <= 1e-5 * log2n
            if not serial and expected > 3:  # the serial schedule runs the same kernels on the same values
                for c in range(world):
                    calls[c] = 0
                outs2 = [torch.full_like(p, float("nan")) for p in parts]
                flags = plans[0].flags(transposed, True)

                def worker2(r):
                    torch.cuda.set_device(dev)
                    with baton:
                        try:
                            plans[r].capi.check(plans[r].capi.lib().dsp_gpu_fft_dist_ex(plans[r]._handle, parts[r].data_ptr(),
                                                                                       outs2[r].data_ptr(), int(inverse), flags, stream))
                        except Exception as e:  # noqa: BLE001
                            errors.append(e)
                            barrier.abort()

                threads = [threading.Thread(target=worker2, args=(r,)) for r in range(world)]
                for t in threads:
                    t.start()
                for t in threads:
                    t.join(timeout=120)
                torch.cuda.synchronize()
                assert not errors, errors
                assert np.array_equal(torch.cat(outs2).cpu().numpy().view(np.uint32), got.view(np.uint32))
            for p in plans:
                p.close()
    finally:
        torch.cuda.synchronize()
        for row in ready + done:
            for e in row:
                hip.hipEventDestroy(e)


@pytest.mark.gpu
@pytest.mark.parametrize("log2n", [16, 24])
def test_native_plan_rccl_world_of_one(dev, orc, log2n):
    """The built-in exchange (grouped ncclSend/ncclRecv) on a communicator of one rank made with RCCL's own API.
    2^24 points take the pipelined schedule (exchanges on the plan's stream): same bits as the serial one."""
    import torch
    from dsp_amd import dist_fft

    comm = dist_fft.RcclCommunicator(0, 1)
    try:
        x = _signal(log2n, seed=7 + log2n)
        plan = dist_fft.NativePlan(log2n, 0, 1, nccl_comm=comm.handle)
        assert plan.exchanges() == (12 if log2n == 24 else 3)
        xd = torch.from_numpy(x).to(dev)
        out = plan.run(xd).cpu().numpy()
        assert rel_l2(out, _oracle_spectrum(orc, x, log2n, False)) <= 1e-5 * log2n
        assert np.array_equal(plan.run(xd, serial=True).cpu().numpy().view(np.uint32), out.view(np.uint32))
        t = plan.run(xd, transposed=True)
        assert torch.equal(t, plan.run(xd, transposed=True, serial=True))
        plan.close()
    finally:
        torch.cuda.synchronize()
        comm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("log2n", list(range(8, 16)))
def test_small_plans_every_world(orc, dev, log2n):
    """Every shape dsp_gpu_fft_dist_plan_create accepts must also run: small transforms on many ranks leave fewer
    columns per rank than a column-pass tile (2^10 on one rank: 32 against 128; 2^14 on 8 ranks: 16 against 32) and
    take dsp_gpu_fft_columns' transpose route.  Plans the shape rules refuse must fail at creation."""
    import torch
    from dsp_amd import capi, dist_fft
    x = _signal(log2n, seed=100 + log2n)
    for world in (1, 2, 4, 8):
        l1 = (log2n + 1) // 2
        n1, n2 = 1 << l1, 1 << (log2n - l1)
        if n1 % world or n2 % world or n2 // world < 2:
            with pytest.raises(capi.DspGpuError):
                dist_fft.NativePlan(log2n, 0, world)
            continue
        for inverse in (False, True):
            parts = [torch.from_numpy(p.copy()).to(dev) for p in np.split(x, world)]
            got = torch.cat(dist_fft.simulate(parts, log2n, inverse)).cpu().numpy()
            assert rel_l2(got, orc.fft0(x.astype(np.complex128), inverse)) <= 1e-5 * log2n, (log2n, world, inverse)


@pytest.mark.gpu
def test_fft_columns_any_width(dev):
    import torch
    from dsp_amd import dist_fft
    g, ref = dist_fft.GpuLocalOps(), NumpyLocalOps()
    rng = np.random.default_rng(77)
    for n, cols in ((16, 2), (64, 24), (256, 1), (512, 6), (1024, 32), (4096, 10), (16384, 16), (65536, 2)):
        x = (rng.standard_normal(n * cols) + 1j * rng.standard_normal(n * cols)).astype(np.complex64)
        for inv in (False, True):
            got = g.fft_columns(torch.from_numpy(x).to(dev), n, cols, inv).cpu().numpy()
            want = ref.fft_columns(torch.from_numpy(x), n, cols, inv).numpy()
            assert rel_l2(got, want) <= 1e-5 * np.log2(n), (n, cols, inv)


@pytest.mark.gpu
def test_exchange_callback_exception_propagates(dev):
    """A Python exchange that raises must fail the transform (not return success on unfilled buffers)."""
    import torch
    from dsp_amd import dist_fft

    class Boom(RuntimeError):
        pass

    def exchange(send, recv, nbytes, world, stream):
        raise Boom("exchange failed")

    plan = dist_fft.NativePlan(16, 0, 1, exchange=exchange)
    x = torch.zeros(1 << 16, dtype=torch.complex64, device=dev)
    with pytest.raises(Boom):
        plan.run(x)
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fail_at", [0, 4, 7, 11])
def test_pipelined_exchange_failure_leaves_the_plan_usable(orc, dev, fail_at):
    """An exchange that fails in the middle of the pipelined schedule (world 1, twelve exchange calls) must fail the
    transform with its code, leave nothing that hangs the streams, and the same plan must then transform correctly."""
    import ctypes as C
    import torch
    from dsp_amd import capi, dist_fft
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    log2n = 24
    x = _signal(log2n, seed=7 + log2n)
    calls = {"n": 0, "fail": fail_at}

    def exchange(send, recv, nbytes, world, stream):
        k = calls["n"]
        calls["n"] += 1
        if k == calls["fail"]:
            return capi.EINVAL
        return 0 if hip.hipMemcpyAsync(recv, send, nbytes * world, 3, stream) == 0 else capi.EHIP

    plan = dist_fft.NativePlan(log2n, 0, 1, exchange=exchange)
    try:
        assert plan.exchanges() == 12
        xd = torch.from_numpy(x).to(dev)
        with pytest.raises(capi.DspGpuError):
            plan.run(xd)
        torch.cuda.synchronize()
        assert calls["n"] == fail_at + 1  # nothing was exchanged after the failure
        calls["n"], calls["fail"] = 0, -1
        out = plan.run(xd).cpu().numpy()
        assert calls["n"] == 12
        assert rel_l2(out, _oracle_spectrum(orc, x, log2n, False)) <= 1e-5 * log2n
    finally:
        torch.cuda.synchronize()
        plan.close()
