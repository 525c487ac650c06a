"""One long IIR channel split over GPUs in time (SURVEY.md 8e, C4b across GPUs): dsp_gpu_iir_settle and
dsp_amd.multigpu.iir_time_sharded.  CPU: the settle length against the oracle's own decay, and the sharding logic with
the oracle as the local filter over a gloo process group.  GPU: virtual ranks on one device against the single-GPU run."""
import os

import numpy as np
import pytest

from conftest import rel_l2


def _oracle_iir_factory(orc):
    """ops.iir's signature on CPU tensors, computed by the oracle in its DOUBLE instantiation (state carried in and out).
    (The float instantiation rounds the fed-back output at every step, coefficients.h:108-118: two runs that differ by
    2^-44 then keep independent rounding noise of ~1e-7, which says nothing about the sharding.)"""
    import torch

    def iir(x, a, b, c0=1.0, d0=0.0, state=None, mode="auto"):
        xs = x.numpy().astype(np.float64)
        out = np.empty_like(xs)
        P = len(a) - 1
        for c in range(xs.shape[0]):
            so = None
            if state is not None:  # library layout {x[t-1..t-P], v[t-1..t-P]} <-> oracle layout x[0..P], y[0..P] (slot 0 unused)
                so = np.zeros(2 * (P + 1))
                so[1:P + 1] = state[c, :P].numpy()
                so[P + 2:] = state[c, P:].numpy()
            out[c] = orc.iir(xs[c], np.asarray(a), np.asarray(b), c0=c0, d0=d0, state=so)
            if state is not None:
                state[c, :P] = torch.from_numpy(so[1:P + 1].copy())
                state[c, P:] = torch.from_numpy(so[P + 2:].copy())
        return torch.from_numpy(out)
    return iir


def test_settle_length_matches_the_decay(golden):
    from dsp_amd import capi, ops
    for tag, lo, hi in (("cheby_lp4", 64, 1024), ("cheby_lp2", 64, 1024), ("cheby_hp4", 64, 1024), ("cheby_lp4_lo", 1024, 65536)):
        a, b = golden[f"{tag}_a"], golden[f"{tag}_b"]
        w = ops.iir_settle(a, b)
        assert w % 64 == 0 and lo <= w <= hi, (tag, w)
        # the zero-input response from a unit state is below 2^-40 of it after w samples (and was not 64 samples before
        # half that far)
        P = len(a) - 1
        for unit in range(P):
            v = np.zeros(P)
            v[unit] = 1.0
            peak = []
            for t in range(w + 8):
                nv = float(np.dot(b[1:], v))  # feedback ADDED (coefficients.h:116-118)
                v = np.concatenate(([nv], v[:-1]))
                peak.append(abs(nv))
            assert max(peak[w:]) <= 2.0 ** -40, (tag, unit)
    # an undamped resonator never forgets
    with pytest.raises(capi.DspGpuError):
        ops.iir_settle(np.array([1.0, 0.0, 0.0]), np.array([0.0, 2.0 * np.cos(0.3), -1.0]))


def test_settle_takes_every_order_the_library_filters():
    """Orders above the scan's eight poles too (Coefficients<2..20>, recursive_filters.h:30-31): round 3 found the noise-gain
    recursion behind dsp_gpu_iir_settle keeping its history in an array of nine (a stack overrun from ten poles on, a crash at
    sixteen).  Every order answers with a whole number of 64-sample blocks or a clean refusal, and the answer is a decay."""
    from dsp_amd import capi, design, ops
    answered = 0
    for poles in range(2, 21, 2):
        for fc in (0.1, 0.25, 0.4):
            a, b = design.low_pass(fc, poles)
            try:
                w = ops.iir_settle(a, b)
            except capi.DspGpuError as e:
                assert e.code == capi.EUNSUPPORTED, (poles, fc)
                continue
            answered += 1
            assert w % 64 == 0 and 64 <= w <= 64 * 3840, (poles, fc, w)
            v = np.zeros(poles)
            v[0] = 1.0
            for t in range(w + 4):  # zero-input run from a unit state, in extended precision where numpy has it
                nv = float(np.dot(np.asarray(b[1:], dtype=np.longdouble), v.astype(np.longdouble)))
                v = np.concatenate(([nv], v[:-1]))
            assert np.abs(v).max() <= 2.0 ** -30, (poles, fc, w)
    assert answered >= 20


def _gloo_worker(rank, world, rdzv, outdir):
    import torch
    import torch.distributed as dist
    import oracle
    from dsp_amd import multigpu, ops
    dist.init_process_group("gloo", init_method=f"file://{rdzv}", rank=rank, world_size=world)
    orc = oracle.load("fast")
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "dsp_golden.npz"))
    a, b = g["cheby_lp4_a"], g["cheby_lp4_b"]
    x = np.random.default_rng(5).standard_normal((2, 1 << 14))
    lo, hi = multigpu.shard_range(x.shape[1], rank, world)
    y, final = multigpu.iir_time_sharded(torch.from_numpy(x[:, lo:hi].copy()), a, b, rank=rank, world=world,
                                         iir=_oracle_iir_factory(orc), settle=ops.iir_settle(a, b))
    np.save(os.path.join(outdir, f"y_{rank}.npy"), y.numpy())
    np.save(os.path.join(outdir, f"final_{rank}.npy"), final.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_two_ranks_with_the_oracle_as_local_filter(tmp_path, orc, golden):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_gloo_worker, args=(world, str(tmp_path / "rdzv"), str(tmp_path)), nprocs=world, join=True)
    a, b = golden["cheby_lp4_a"], golden["cheby_lp4_b"]
    x = np.random.default_rng(5).standard_normal((2, 1 << 14))
    got = np.concatenate([np.load(tmp_path / f"y_{r}.npy") for r in range(world)], axis=1)
    for c in range(2):
        st = np.zeros(10)
        want = orc.iir(x[c], a, b, state=st)
        assert rel_l2(got[c], want) <= 1e-12
        lib_state = np.concatenate((st[1:5], st[6:10]))
        for r in range(world):
            assert np.allclose(np.load(tmp_path / f"final_{r}.npy")[c], lib_state, rtol=1e-9, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,world,log2n", [("cheby_lp4", 4, 22), ("cheby_lp4_lo", 2, 22), ("cheby_lp8", 8, 20), ("bq_pk", 3, 21)])
def test_virtual_ranks_match_the_single_gpu_run(dev, golden, tag, world, log2n):
    """`world` segments of one signal on one GPU, the all-gather emulated by a list: equal to the unsharded run to
    fp32 output rounding, end state included; an initial state is honoured by rank 0."""
    import torch
    from dsp_amd import multigpu, ops
    a, b = golden[f"{tag}_a"], golden[f"{tag}_b"]
    c0, d0 = (golden[f"{tag}_cd"] if f"{tag}_cd" in golden else (1.0, 0.0))
    g = torch.Generator(device=dev).manual_seed(log2n + world)
    n = (1 << log2n) + 1000  # ragged: the last rank's segment is longer
    x = torch.randn(2, n, device=dev, generator=g)
    P = len(a) - 1
    start = torch.randn(2, 2 * P, device=dev, generator=g, dtype=torch.float64) * 0.1
    st_ref = start.clone()
    want = ops.iir(x, a, b, c0, d0, state=st_ref)
    bounds = [multigpu.shard_range(n, r, world) for r in range(world)]
    # pass 1 of every rank before anybody's pass 2: the gather needs all end states
    zero_states = []
    for r, (lo, hi) in enumerate(bounds):
        st = torch.zeros(2, 2 * P, dtype=torch.float64, device=dev)
        if r == 0:
            st.copy_(start)
        ops.iir(x[:, lo:hi].contiguous(), a, b, c0, d0, state=st)
        zero_states.append(st)
    outs = []
    for r, (lo, hi) in enumerate(bounds):
        y, final = multigpu.iir_time_sharded(x[:, lo:hi].contiguous(), a, b, c0, d0, rank=r, world=world,
                                             gather_states=lambda t: zero_states, state=start if r == 0 else None)
        outs.append(y)
        assert torch.allclose(final, st_ref, rtol=1e-7, atol=1e-10)
    got = torch.cat(outs, dim=1)
    err = torch.linalg.vector_norm((got - want).double()) / torch.linalg.vector_norm(want.double())
    assert float(err) <= 2e-6, float(err)
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
