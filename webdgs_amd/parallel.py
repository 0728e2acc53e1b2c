"""View-sharded data parallelism over RCCL/xGMI; the reference has none (batch 1, one GPUDevice).

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL on ROCm; ``gloo`` in CPU tests and 1-GPU rehearsals).  Every
rank holds a full replica of the point cloud.  A global step processes ``world_size * views_per_rank`` views: each rank renders and
back-propagates its own views and sums the per-view fp16 gradients into one fp32 block locally (after K17, so per-view values are
exactly the reference's).  Then the bandwidth-optimal exchange:

    reduce-scatter (sum) of the block + visibility counts   -> rank r holds the sums for ITS slice of the Gaussians
    Adam + re-pack on the owned slice only                   -> 1/world_size of the optimizer pass per rank
    all-gather of the re-packed 32-byte rows                 -> every replica's point cloud is current again

Batch semantics (new; reduce to the reference at batch 1): ``g = sum_views g_view``; Adam runs where
``sum_views (tile_counts > 0) > 0``.  Optimizer state of a slice lives on its owner and is gathered (``Exchange.broadcast``) only
before a densify rebuild or an export.  Per step and rank: (W-1)/W x (60 + 32) bytes per Gaussian (c3, W = 8: 80 MB) instead of
the all-reduce's 2 (W-1)/W x 60 (105 MB).

``Exchange`` is the seam: ``TorchExchange`` drives ``torch.distributed`` (stream-ordered on nccl, host-fenced on gloo),
``CapiExchange`` drives the library's own communicator (``wdgs_comm_*``, RCCL on the kernels' stream) for hosts without torch.
"""
from __future__ import annotations

import contextlib
import os
from typing import Optional, Sequence

import torch
import torch.distributed as dist

GRAD_FLOATS = 14  # pos3, opacity, rot4, log-sigma3, rgb3 (GaussianGradient component order)


def init_from_env(backend: Optional[str] = None) -> tuple[int, int, int]:
    """Initialises the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # WDGS_DIST_BACKEND=gloo lets several ranks share one GPU (functional rehearsal of the N > 1 path on a 1-GPU box)
            backend = os.environ.get("WDGS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


_side_streams: dict = {}


@contextlib.contextmanager
def collective_stream(device=None):
    """Runs the enclosed ``torch.distributed`` calls of an ``nccl`` (RCCL) group on a side stream of torch's, ordered after the current
    stream on entry and the current stream after it on exit.  Why: c10d issues a blocking collective on the CURRENT stream, leaves an event
    there and has its watchdog thread query that event until the collective has finished; HIP refuses the query of an event whose stream is
    recording a graph (``hipErrorCapturedEvent``: "operation not permitted on an event last recorded in a capturing stream"), and the
    watchdog answers with ``abort()``.  The current stream is the one the Trainer records its command buffers on, often right after an
    exchange -- so a collective must not leave its event there.  Without an initialised nccl group (gloo, single process) this is a no-op."""
    if not (dist.is_initialized() and torch.cuda.is_available() and dist.get_backend() == "nccl"):
        yield
        return
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    side = _side_streams.get(dev)
    if side is None:
        side = _side_streams[dev] = torch.cuda.Stream(dev)
    current = torch.cuda.current_stream(dev)
    side.wait_stream(current)
    try:
        with torch.cuda.stream(side):
            yield
    finally:
        current.wait_stream(side)


def shard_views(view_ids: Sequence[int], rank: int, world: int) -> list[int]:
    """The views of one global batch that ``rank`` processes: a strided split (views rank, rank+world, ...).

    Every view is assigned to exactly one rank; ranks differ by at most one view.
    """
    return [v for i, v in enumerate(view_ids) if i % world == rank]


def allreduce_gradients(grad_f32: torch.Tensor, visible_i32: torch.Tensor, group=None) -> None:
    """Sums the ``[N*14]`` fp32 gradient block and the ``[N]`` int32 visibility counts over all ranks, in place."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        with collective_stream(grad_f32.device if grad_f32.is_cuda else None):
            dist.all_reduce(grad_f32, op=dist.ReduceOp.SUM, group=group)
            dist.all_reduce(visible_i32, op=dist.ReduceOp.SUM, group=group)


def allreduce_counts(counts: torch.Tensor, group=None) -> torch.Tensor:
    """Sums integer per-Gaussian counters (densify metric counts) over all ranks, in place."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        with collective_stream(counts.device if counts.is_cuda else None):
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return counts


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        with collective_stream():
            dist.barrier()


def shutdown() -> None:
    """Tears the default process group down (after every Trainer and HipDevice of this process has been destroyed)."""
    if dist.is_initialized():
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        dist.destroy_process_group()


def slice_points(num_points: int, world: int) -> int:
    """Gaussians per rank: ceil(N / world) rounded up to 64 (rank r owns ``[r*slice, min((r+1)*slice, N))``)."""
    per = (max(int(num_points), 1) + world - 1) // world
    return (per + 63) // 64 * 64


def owned_range(num_points: int, world: int, rank: int) -> tuple[int, int]:
    """(first, count) of the slice ``rank`` owns."""
    sl = slice_points(num_points, world)
    first = min(rank * sl, num_points)
    return first, max(0, min((rank + 1) * sl, num_points) - first)


class _RawCuda:
    """Zero-copy view of library-owned device memory for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr: int, count: int, typestr: str):
        self.__cuda_array_interface__ = dict(shape=(int(count),), typestr=typestr, data=(int(ptr), False), version=2)


def _tensor_at(device, ptr: int, count: int, dtype: torch.dtype) -> torch.Tensor:
    typestr = {torch.float32: "<f4", torch.int32: "<i4", torch.uint8: "|u1"}[dtype]
    return torch.as_tensor(_RawCuda(ptr, count, typestr), device=device.torch_device)


class Exchange:
    """What the Trainer needs from a communicator.  Pointers are device addresses; counts are elements."""

    world_size, rank = 1, 0
    name = "none"

    def exchange_gradients(self, grad_ptr: int, visible_ptr: int, flag_ptr: int, slice_pts: int) -> None:
        """Reduce-scatter (sum) grad f32[W*slice*14] and visible u32[W*slice] in place (rank r's slice ends up summed) and sum the
        one-word flag over all ranks."""

    def allgather_rows(self, rows_ptr: int, slice_pts: int) -> None:
        """In-place all-gather of u32[W*slice*8]: every rank contributes its slice."""

    def broadcast(self, ptr: int, nbytes: int, root: int) -> None:
        """In-place broadcast of ``nbytes`` at ``ptr`` from ``root``."""

    def allreduce_counts(self, ptr: int, count: int) -> None:
        """In-place u32 sum."""

    def barrier(self) -> None:
        pass

    def destroy(self) -> None:
        pass


class TorchExchange(Exchange):
    """``torch.distributed`` as the transport.  On ``nccl`` (= RCCL) every collective is enqueued behind the work already on the
    device's stream -- the Trainer's HIP stream IS torch's current stream (``HipDevice``), c10d orders its own stream after the
    current one on entry and the current one after its own on exit -- so there is no host synchronisation anywhere in a step.
    ``gloo`` (CPU tests; several ranks on one GPU) stages device tensors through the host on streams of its own: there both sides of
    every collective are fenced on the host (``fenced``), and reduce-scatter is emulated by an all-reduce (gloo has none)."""

    def __init__(self, device, group=None, force: bool = False):
        self.device, self.group = device, group
        self.force = bool(force)  # issue the collectives even in a world of one (identities): exercises the backend on a 1-GPU box
        self._views: dict = {}
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(group) if dist.is_initialized() else "none"
        # WDGS_DP_FENCE=1 / 0 overrides (diagnostics): fences on nccl, or none on gloo
        fence_env = os.environ.get("WDGS_DP_FENCE", "")
        self.fenced = (self.backend != "nccl") if fence_env == "" else (fence_env == "1")
        self.name = f"torch.distributed/{self.backend}" + ("+host-fences" if self.fenced else "")

    def _t(self, ptr: int, count: int, dtype: torch.dtype) -> torch.Tensor:
        key = (int(ptr), int(count), dtype)
        t = self._views.get(key)
        if t is None:
            if len(self._views) > 64:
                self._views.clear()
            t = self._views[key] = _tensor_at(self.device, ptr, count, dtype)
        return t

    def _idle(self) -> bool:
        return self.world_size <= 1 and not self.force

    # The collectives run in place -- rank r's output is slice r of the input, NCCL's in-place layout -- so a step moves no byte it
    # does not have to.  Should a c10d build refuse aliased arguments, the exchange falls back, once and loudly, to a staging tensor
    # (the same values; one extra device copy of the slice).
    _in_place = True

    # what an argument check that refuses aliased input / output tensors says (c10d words such refusals with one of these); nothing else --
    # an RCCL failure, a timeout, a lost connection, anything unforeseen -- may switch one rank to another call sequence than its peers'
    _ALIASING_WORDS = ("alias", "overlap", "in-place", "inplace", "in place", "same tensor", "same storage", "same memory", "share memory", "shares memory")

    def _out_of_place(self, error: Exception) -> None:
        """The in-place collectives have never been refused by the c10d builds this ran on (no such message in any run record).  Should a
        build refuse aliased arguments it says so in an argument check, before anything is enqueued: ONLY an error that says so switches the
        exchange to staging.  Every other error is re-raised -- retrying a collective on one rank would leave the ranks issuing different
        collectives."""
        import sys
        text = str(error).lower()
        if not any(w in text for w in self._ALIASING_WORDS):
            raise error
        print(f"[webdgs_amd.parallel] in-place collective refused ({error}); staging through a scratch tensor from now on", file=sys.stderr, flush=True)
        self._in_place = False

    def _reduce_scatter(self, t: torch.Tensor, count: int) -> None:
        mine = t[self.rank * count:(self.rank + 1) * count]
        if self._in_place:
            try:
                dist.reduce_scatter_tensor(mine, t, op=dist.ReduceOp.SUM, group=self.group)
                return
            except RuntimeError as e:
                self._out_of_place(e)
        out = torch.empty_like(mine)
        dist.reduce_scatter_tensor(out, t, op=dist.ReduceOp.SUM, group=self.group)
        mine.copy_(out)

    @contextlib.contextmanager
    def _section(self):
        """The collectives of one exchange: host-fenced on gloo, on the side stream on nccl (``collective_stream``)."""
        if self.fenced:
            self.device.torch_stream.synchronize()
        with collective_stream(self.device.torch_device):
            yield
        if self.fenced:
            torch.cuda.synchronize(self.device.torch_device)

    def exchange_gradients(self, grad_ptr, visible_ptr, flag_ptr, slice_pts):
        if self._idle():
            return
        w, r = self.world_size, self.rank
        g = self._t(grad_ptr, w * slice_pts * GRAD_FLOATS, torch.float32)
        v = self._t(visible_ptr, w * slice_pts, torch.int32)
        f = self._t(flag_ptr, 1, torch.int32)
        with self._section():
            if self.backend == "nccl":
                self._reduce_scatter(g, slice_pts * GRAD_FLOATS)
                self._reduce_scatter(v, slice_pts)
            else:
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
                dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.group)

    def allgather_rows(self, rows_ptr, slice_pts):
        if self._idle():
            return
        w, r = self.world_size, self.rank
        t = self._t(rows_ptr, w * slice_pts * 8, torch.int32)
        mine = t[r * slice_pts * 8:(r + 1) * slice_pts * 8]
        with self._section():
            if self.backend == "nccl":
                if self._in_place:
                    try:
                        dist.all_gather_into_tensor(t, mine, group=self.group)
                    except RuntimeError as e:
                        self._out_of_place(e)
                if not self._in_place:
                    dist.all_gather_into_tensor(t, mine.clone(), group=self.group)
            else:
                dist.all_gather([t[i * slice_pts * 8:(i + 1) * slice_pts * 8] for i in range(w)], mine.clone(), group=self.group)

    def broadcast(self, ptr, nbytes, root):
        if self._idle() or nbytes == 0:
            return
        with self._section():
            dist.broadcast(self._t(ptr, nbytes, torch.uint8), src=root, group=self.group)

    def allreduce_counts(self, ptr, count):
        if self._idle() or count == 0:
            return
        with self._section():
            dist.all_reduce(self._t(ptr, count, torch.int32), op=dist.ReduceOp.SUM, group=self.group)

    def barrier(self):
        if self.world_size > 1:
            with collective_stream(self.device.torch_device):
                dist.barrier(group=self.group)


class CapiExchange(Exchange):
    """The library's own communicator (``wdgs_comm_*``): RCCL calls queued on the device's stream by libwebdgs_hip.so itself --
    the path a TypeScript / C++ host uses (no torch in the loop).  The unique id travels over the torch process group here."""

    def __init__(self, device, group=None):
        self.device = device
        self.comm = Communicator.fromProcessGroup(device, group)
        self.world_size, self.rank = self.comm.world_size, self.comm.rank
        self.name = "wdgs_comm (RCCL on the device stream)"
        self.force = True  # the communicator exists: its collectives run (identities in a world of one)
        self._group = group

    def exchange_gradients(self, grad_ptr, visible_ptr, flag_ptr, slice_pts):
        from . import _lib
        _lib.check(self.device.lib.wdgs_comm_exchange_gradients(self.comm.handle, grad_ptr, visible_ptr, flag_ptr, int(slice_pts)))

    def allgather_rows(self, rows_ptr, slice_pts):
        from . import _lib
        _lib.check(self.device.lib.wdgs_comm_allgather_rows(self.comm.handle, rows_ptr, int(slice_pts)))

    def broadcast(self, ptr, nbytes, root):
        from . import _lib
        _lib.check(self.device.lib.wdgs_comm_broadcast(self.comm.handle, ptr, int(nbytes), int(root)))

    def allreduce_counts(self, ptr, count):
        from . import _lib
        _lib.check(self.device.lib.wdgs_comm_allreduce_counts(self.comm.handle, ptr, int(count)))

    def barrier(self):
        if dist.is_initialized() and dist.get_world_size(self._group) > 1:
            with collective_stream(self.device.torch_device):
                dist.barrier(group=self._group)

    def destroy(self):
        self.comm.destroy()


def default_exchange(device, world_size: int) -> Exchange:
    """``WDGS_COMM=capi`` selects the C-ABI communicator; otherwise torch.distributed when a process group exists."""
    if os.environ.get("WDGS_COMM", "") == "capi":
        return CapiExchange(device)
    if world_size > 1 or dist.is_initialized():
        return TorchExchange(device)
    return Exchange()


class Communicator:
    """The C-ABI communicator (``wdgs_comm_*``, include/webdgs.h): RCCL driven by the library itself, for hosts that have no
    ``torch.distributed`` (the N-API addon, a C++ trainer).  ``unique_id`` is the 128-byte id of rank 0
    (``Communicator.uniqueId()``), shipped to the other ranks over any host channel; with a torch process group it is
    broadcast through it (``Communicator.fromProcessGroup``).  Reductions are queued on the device's stream."""

    def __init__(self, device, unique_id: bytes, world_size: int, rank: int):
        import ctypes as C

        from . import _lib
        if len(unique_id) != 128:
            raise ValueError("unique_id must be 128 bytes")
        self.device, self.world_size, self.rank = device, int(world_size), int(rank)
        ident = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        h = C.c_void_p()
        _lib.check(device.lib.wdgs_comm_create(device.handle, ident, self.world_size, self.rank, C.byref(h)))
        self.handle = h

    @staticmethod
    def uniqueId() -> bytes:
        import ctypes as C

        from . import _lib
        ident = (C.c_uint8 * 128)()
        _lib.check(_lib.load().wdgs_comm_get_unique_id(ident))
        return bytes(ident)

    @classmethod
    def fromProcessGroup(cls, device, group=None) -> "Communicator":
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        box = [cls.uniqueId() if rank == 0 else None]
        if world > 1:
            with collective_stream(device.torch_device):
                dist.broadcast_object_list(box, src=0, group=group)
        return cls(device, box[0], world, rank)

    def allreduceGradients(self, gradF32, visibleCounts, numPoints: int) -> None:
        from . import _lib
        _lib.check(self.device.lib.wdgs_comm_allreduce_gradients(self.handle, gradF32.ptr, visibleCounts.ptr, int(numPoints)))

    def allreduceCounts(self, counts, count: int) -> None:
        from . import _lib
        _lib.check(self.device.lib.wdgs_comm_allreduce_counts(self.handle, counts.ptr, int(count)))

    def destroy(self) -> None:
        if self.handle:
            self.device.lib.wdgs_comm_destroy(self.handle)
            self.handle = None
