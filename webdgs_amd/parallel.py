"""View-sharded data parallelism over RCCL/xGMI; the reference has none (batch 1, one GPUDevice).

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL on ROCm; ``gloo`` in CPU tests).  Every rank holds
a full replica of the point cloud and optimizer state.  A global step processes ``world_size * views_per_rank`` views:
each rank renders and back-propagates its own views, sums the per-view fp16 gradients into one fp32 block locally
(after K17, so per-view values are exactly the reference's), then ONE all-reduce (sum) of that block plus the u32
visibility counts precedes a single Adam step that every rank applies identically -- replicas stay bit-identical
because all ranks consume the same reduced buffer.  Batch semantics (new; reduce to the reference at batch 1):
``g = sum_views g_view``; Adam runs where ``sum_views (tile_counts > 0) > 0``.

Payload: 14 f32 + 1 u32 = 60 B per Gaussian (c3: 60 MB): one fp32 all-reduce of the gradient block and one int32
all-reduce of the visibility counts (4 B per Gaussian) -- integer so the mask is exact for any number of views.
"""
from __future__ import annotations

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


def shard_views(view_ids: Sequence[int], rank: int, world: int) -> list[int]:
    """The views of one global batch that ``rank`` processes: a strided split (views rank, rank+world, ...).

    Every view is assigned to exactly one rank; ranks differ by at most one view.
    """
    return [v for i, v in enumerate(view_ids) if i % world == rank]


def allreduce_gradients(grad_f32: torch.Tensor, visible_i32: torch.Tensor, group=None) -> None:
    """Sums the ``[N*14]`` fp32 gradient block and the ``[N]`` int32 visibility counts over all ranks, in place."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(grad_f32, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(visible_i32, op=dist.ReduceOp.SUM, group=group)


def allreduce_counts(counts: torch.Tensor, group=None) -> torch.Tensor:
    """Sums integer per-Gaussian counters (densify metric counts) over all ranks, in place."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return counts


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def shutdown() -> None:
    """Tears the default process group down (after every Trainer and HipDevice of this process has been destroyed)."""
    if dist.is_initialized():
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        dist.destroy_process_group()


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
