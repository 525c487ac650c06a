"""One FFT spread over several GPUs: the six-step decomposition with all-to-all exchanges (config C5).

dsp::fft0 (reference fft.h:31-55) of N = 2^log2n points where every rank holds a contiguous N/G
slice of the input and receives the same slice of the natural-order output.  With N = N1*N2,
n = n1*N2 + n2, k = k1 + N1*k2 and W_M = exp(+-2 pi i / M) (+ for the reference's FORWARD):

    X[k1 + N1 k2] = sum_n2 W_N2^{n2 k2} * W_N^{n2 k1} * ( sum_n1 x[n1 N2 + n2] W_N1^{n1 k1} )

    pack -> [all-to-all] -> FFT over n1 -> [all-to-all] -> unpack * twiddle -> FFT over n2
         -> pack -> [all-to-all] -> transpose to natural order
(the middle phase runs as two fused HBM passes, dsp_gpu_fft_rows_exchange, where the shape allows)

The exchange is the only collective on the whole hot path; it is point-to-point by nature (every
pair of GPUs trades one block), which is what the xGMI mesh wants: `all_to_all_single` over RCCL,
one block of N/G^2 points per peer (128 MiB for 2^30 points on 8 GPUs), no ring.  The local steps
are the HIP kernels behind dsp_gpu_fft_columns / dsp_gpu_swap_axes / dsp_gpu_fft_batch /
dsp_gpu_transpose_blocks.  One process per GPU drives this through torch.distributed; `simulate`
runs the same phases for G virtual ranks inside one process (how the index logic is tested on a
single GPU, and on CPU with a stand-in for the local steps).
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class Plan:
    log2n: int
    world: int
    n1: int
    n2: int

    @property
    def rows(self) -> int:  # R: rows of the [N1][N2] view per rank
        return self.n1 // self.world

    @property
    def cols(self) -> int:  # C: columns per rank after the first exchange
        return self.n2 // self.world


def make_plan(log2n: int, world: int) -> Plan:
    if world < 1 or world & (world - 1):
        raise ValueError("world size must be a power of two")
    l1 = (log2n + 1) // 2
    n1, n2 = 1 << l1, 1 << (log2n - l1)
    if n1 % world or n2 % world or n2 // world < 2:
        raise ValueError(f"2^{log2n} points cannot be spread over {world} ranks")
    return Plan(log2n, world, n1, n2)


class GpuLocalOps:
    """Local steps on torch CUDA complex64 tensors through the C ABI (current torch stream)."""

    def __init__(self):
        import torch

        from . import capi
        self.torch, self.capi = torch, capi

    def _stream(self, t):
        return self.torch.cuda.current_stream(t.device).cuda_stream

    def swap_axes(self, t, a, b, c, tw_log2n=0, row0=0, inverse=False):
        out = self.torch.empty_like(t)
        self.capi.check(self.capi.lib().dsp_gpu_swap_axes(t.data_ptr(), out.data_ptr(), a, b, c, tw_log2n, row0,
                                                          int(inverse), self._stream(t)))
        return out

    def fft_columns(self, t, n, cols, inverse=False):
        out = self.torch.empty_like(t)
        self.capi.check(self.capi.lib().dsp_gpu_fft_columns(t.data_ptr(), out.data_ptr(), n.bit_length() - 1, cols,
                                                            int(inverse), self._stream(t)))
        return out

    def fft_rows(self, t, rows, n, inverse=False):
        out = self.torch.empty_like(t)
        self.capi.check(self.capi.lib().dsp_gpu_fft_batch(t.data_ptr(), out.data_ptr(), n.bit_length() - 1, rows,
                                                          int(inverse), self.capi.MEM_DEVICE, self._stream(t)))
        return out

    def fft_rows_exchange(self, t, rows, n2, world, tw_log2n, row0, inverse=False):
        """unpack * twiddle -> row FFT -> pack in two fused passes; None when the shape is not covered."""
        out = self.torch.empty_like(t)
        rc = self.capi.lib().dsp_gpu_fft_rows_exchange(t.data_ptr(), out.data_ptr(), n2.bit_length() - 1, rows, world,
                                                       tw_log2n, row0, int(inverse), self._stream(t))
        if rc == self.capi.EUNSUPPORTED:
            return None
        self.capi.check(rc)
        return out

    def transpose_blocks(self, t, blocks, rows, cols):
        out = self.torch.empty_like(t)
        self.capi.check(self.capi.lib().dsp_gpu_transpose_blocks(t.data_ptr(), out.data_ptr(), blocks, rows, cols,
                                                                 self._stream(t)))
        return out


class NativePlan:
    """One rank's dsp_gpu_fft_dist_plan (include/dsp_gpu.h): the phase sequencing lives in the library
    (dsp_amd/csrc/dist_plan.hip); this object only owns the handle.  `exchange` (optional) is a Python callable
    (send_ptr, recv_ptr, block_bytes, world, stream) -> int used by `run`; `nccl_comm` selects the built-in RCCL
    exchange instead."""

    def __init__(self, log2n: int, rank: int, world: int, exchange=None, nccl_comm=None):
        import ctypes as C

        import torch

        from . import capi
        self.torch, self.capi = torch, capi
        self.log2n, self.rank, self.world = log2n, rank, world
        self._handle = C.c_void_p()
        if nccl_comm is not None:
            capi.check(capi.lib().dsp_gpu_fft_dist_plan_create_rccl(C.byref(self._handle), log2n, rank, world, nccl_comm))
        else:
            # keep the ctypes thunk alive as long as the plan: the library calls it from dsp_gpu_fft_dist
            self._error = None

            def thunk(user, send, recv, nbytes, w, stream):
                # an exception must not escape into C (ctypes would report it as "unraisable" and hand the library a 0 =
                # success, which then transforms buffers that were never filled): keep it, fail the call, re-raise in run()
                try:
                    return int(exchange(send, recv, nbytes, w, stream))
                except BaseException as e:  # noqa: BLE001
                    self._error = e
                    return capi.EHIP

            self._thunk = capi.EXCHANGE_FN(thunk if exchange is not None else 0)
            capi.check(capi.lib().dsp_gpu_fft_dist_plan_create(C.byref(self._handle), log2n, rank, world, self._thunk, None))
        self.local_points = capi.lib().dsp_gpu_fft_dist_local_points(self._handle)

    def _stream(self, t):
        return self.torch.cuda.current_stream(t.device).cuda_stream

    def phase(self, index: int, t, inverse: bool = False):
        out = self.torch.empty_like(t)
        self.capi.check(self.capi.lib().dsp_gpu_fft_dist_phase(self._handle, index, t.data_ptr(), out.data_ptr(),
                                                               int(inverse), self._stream(t)))
        return out

    def flags(self, transposed: bool = False, serial: bool = False) -> int:
        return (self.capi.DIST_TRANSPOSED_OUT if transposed else 0) | (self.capi.DIST_SERIAL if serial else 0)

    def exchanges(self, transposed: bool = False, serial: bool = False) -> int:
        """How many times one transform calls the plan's exchange (chunks of the pipelined schedule x 3 or 2)."""
        return int(self.capi.lib().dsp_gpu_fft_dist_exchanges(self._handle, self.flags(transposed, serial)))

    def run(self, t, inverse: bool = False, out=None, transposed: bool = False, serial: bool = False):
        """The whole transform through dsp_gpu_fft_dist_ex (exchange callback or RCCL communicator of the plan).
        transposed: keep the spectrum as out[r][k2] = X[(rank*N1/G + r) + N1*k2] -- one exchange and one pass fewer.
        serial: the unpipelined schedule (every exchange on the caller's stream, nothing overlaps)."""
        if t.numel() != self.local_points:
            raise ValueError("x_local must hold N / world points")
        out = self.torch.empty_like(t) if out is None else out
        rc = self.capi.lib().dsp_gpu_fft_dist_ex(self._handle, t.data_ptr(), out.data_ptr(), int(inverse),
                                                 self.flags(transposed, serial), self._stream(t))
        err, self._error = getattr(self, "_error", None), None
        if err is not None:
            raise err
        self.capi.check(rc)
        return out

    def close(self):
        if self._handle:
            self.capi.lib().dsp_gpu_fft_dist_plan_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RcclCommunicator:
    """An ncclComm_t of our own for the library's built-in exchange (dsp_gpu_fft_dist_plan_create_rccl), made with
    RCCL's C API through ctypes: rank 0 draws the unique id, torch.distributed (any backend) carries its 128 bytes
    to the other ranks.  The GPU of this rank must be current."""

    def __init__(self, rank: int, world: int, group=None):
        import ctypes as C
        import os

        import torch
        import torch.distributed as dist

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]

        try:
            self._rccl = C.CDLL("librccl.so.1")  # the copy torch has loaded already, else the system's
        except OSError:
            self._rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
        self._rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        self._rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        uid = UniqueId()
        if rank == 0 and self._rccl.ncclGetUniqueId(C.byref(uid)) != 0:
            raise RuntimeError("ncclGetUniqueId failed")
        if world > 1:
            box = [None]
            box[0] = C.string_at(C.addressof(uid), 128) if rank == 0 else None
            dist.broadcast_object_list(box, src=0, group=group)
            C.memmove(C.addressof(uid), box[0], 128)
        self.handle = C.c_void_p()
        rc = self._rccl.ncclCommInitRank(C.byref(self.handle), world, uid, rank)
        if rc != 0:
            raise RuntimeError(f"ncclCommInitRank failed ({rc})")

    def close(self):
        if self.handle:
            self._rccl.ncclCommDestroy(self.handle)
            self.handle = None


_plans: dict = {}


def native_phases(log2n: int, rank: int, world: int, inverse: bool):
    """The same four local phases as `phases`, run by the library's plan handle (the product path on a GPU).
    Plans are kept per (log2n, rank, world): a plan may own a device work buffer."""
    plan = _plans.get((log2n, rank, world))
    if plan is None:
        plan = _plans[(log2n, rank, world)] = NativePlan(log2n, rank, world)
    return [lambda t, i=i: plan.phase(i, t, inverse) for i in range(4)]


def phases(plan: Plan, rank: int, ops, inverse: bool):
    """The four local phases of one rank; an all-to-all goes between consecutive phases.  The host-side statement
    of what dsp_amd/csrc/dist_plan.hip sequences: used with stand-in `ops` to test the index logic on CPU (gloo),
    and with GpuLocalOps to cross-check the library's plan against its own building blocks.

    Every phase maps a flat complex tensor of N/G points to a flat tensor of N/G points laid out as
    G equal blocks (block g goes to / came from rank g)."""
    G, R, C = plan.world, plan.rows, plan.cols

    def pack_columns(x):          # [R][G][C] -> [G][R][C]: block g = my rows, rank g's columns
        return ops.swap_axes(x, R, G, C)

    def transform_n1(recv):       # [N1][C]: FFT over n1; result rows k1 already grouped by destination rank
        return ops.fft_columns(recv, plan.n1, C, inverse)

    def transform_n2(recv):       # [G][R][C] -> [R][N2] with W_N^{k1 n2}; FFT over n2; pack columns k2 by rank
        fused = getattr(ops, "fft_rows_exchange", None)
        if fused is not None:     # the three steps in two HBM passes where the kernels cover the shape
            out = fused(recv, R, plan.n2, G, plan.log2n, rank * R, inverse)
            if out is not None:
                return out
        rows = ops.swap_axes(recv, G, R, C, tw_log2n=plan.log2n, row0=rank * R, inverse=inverse)
        spec = ops.fft_rows(rows, R, plan.n2, inverse)
        return ops.swap_axes(spec, R, G, C)

    def natural_order(recv):      # [G][R][C] (k1 = g R + r, my k2) -> [C][N1]: X[k1 + N1 k2]
        return ops.transpose_blocks(recv, G, R, C)

    return [pack_columns, transform_n1, transform_n2, natural_order]


def fft_dist(x_local, log2n: int, inverse: bool = False, group=None, ops=None):
    """Distributed fft0: x_local = this rank's contiguous N/G slice (torch complex64); returns this rank's
    slice of the natural-order spectrum.  Collective: every rank of `group` must call it."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    plan = make_plan(log2n, world)
    if x_local.numel() != (1 << log2n) // world:
        raise ValueError("x_local must hold N / world points")
    t = x_local.reshape(-1).contiguous()
    if ops is None and world == 1:  # one rank: the library's whole-transform call (no exchange, no pack)
        native_phases(log2n, 0, 1, inverse)
        return _plans[(log2n, 0, 1)].run(t, inverse)
    steps = phases(plan, rank, ops, inverse) if ops is not None else native_phases(log2n, rank, world, inverse)
    for i, step in enumerate(steps):
        t = step(t)
        if i + 1 < len(steps) and world > 1:
            send = torch.view_as_real(t.contiguous())
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send, group=group)
            t = torch.view_as_complex(recv)
    return t


def simulate(parts, log2n: int, inverse: bool = False, ops=None):
    """Run the same phases for len(parts) virtual ranks in this process; the exchanges are block copies.
    parts[r] = rank r's input slice.  Returns the list of output slices."""
    import torch

    G = len(parts)
    plan = make_plan(log2n, G)
    steps = [phases(plan, r, ops, inverse) if ops is not None else native_phases(log2n, r, G, inverse) for r in range(G)]
    cur = [p.reshape(-1).contiguous() for p in parts]
    for i in range(4):
        cur = [steps[r][i](cur[r]) for r in range(G)]
        if i < 3:
            blocks = [c.reshape(G, -1) for c in cur]
            cur = [torch.cat([blocks[src][dst] for src in range(G)]) for dst in range(G)]
    return cur
