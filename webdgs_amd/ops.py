"""Host-side mirror of the reference's operator classes over the C ABI (``include/webdgs.h``).

Class and method names follow ``/root/reference/src/renderers/*.ts``, ``src/sort/sort_dynamic.ts``,
``src/prefix/prefix.ts`` and ``src/utils/allocate-pointcloud.ts`` so that code shaped like
``src/trainer.ts`` drives them unchanged; ``GPUDevice / GPUBuffer / GPUTextureView / GPUCommandEncoder`` become
``HipDevice / HipBuffer / HipEncoder``.  PyTorch appears only as the allocator for caller-owned device memory
(point clouds, cameras, images, optimizer state) and as the owner of the HIP stream; every kernel is in
``libwebdgs_hip.so``.  There is no CPU path here.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import re
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import CapacityError, StateError, WdgsError, check  # noqa: F401  (re-exported)


# ----------------------------------------------------------------------------- device / buffers
MAX_LANES = 4  # WDGS_MAX_LANES (include/webdgs.h)
MAX_BATCH_VIEWS = 16  # WDGS_MAX_BATCH_VIEWS


class HipBuffer:
    """A span of device memory (``GPUBuffer``): either a view into library-owned memory or a torch-backed allocation."""

    def __init__(self, device: "HipDevice", ptr: int, size: int, owner: Optional[torch.Tensor] = None, label: str = ""):
        self.device, self.ptr, self.size, self._owner, self.label = device, int(ptr or 0), int(size), owner, label
        self.destroyed = False
        # called before the buffer's CONTENT is handed to the host (read / tensor): a producer that keeps part of it elsewhere brings it up
        # to date first (the optimizer's deferred SH writes: Optimizer.setDeferredSH)
        self.before_read = None

    def read(self, dtype=np.uint32, count: Optional[int] = None, offset: int = 0) -> np.ndarray:
        """mapAsync + getMappedRange: synchronous copy to host."""
        dt = np.dtype(dtype)
        if self.before_read is not None:
            self.before_read()
        n = (self.size - offset) // dt.itemsize if count is None else int(count)
        out = np.empty(n, dt)
        if n:
            check(self.device.lib.wdgs_copy_to_host(self.device.handle, out.ctypes.data, self.ptr + offset, n * dt.itemsize))
        return out

    def write(self, data: np.ndarray, offset: int = 0) -> None:
        """queue.writeBuffer: stream-ordered."""
        a = np.ascontiguousarray(data)
        if a.nbytes + offset > self.size:
            raise _lib.WdgsError(_lib.WDGS_E_INVALID, f"write of {a.nbytes} bytes at {offset} exceeds buffer size {self.size}")
        check(self.device.lib.wdgs_copy_to_device(self.device.handle, self.ptr + offset, a.ctypes.data, a.nbytes))
        self.device._keepalive.append(a)

    def clear(self) -> None:
        check(self.device.lib.wdgs_memset(self.device.handle, self.ptr, 0, self.size))

    def tensor(self) -> torch.Tensor:
        if self._owner is None:
            raise _lib.StateError(_lib.WDGS_E_STATE, "buffer is library-owned; no torch tensor behind it")
        if self.before_read is not None:
            self.before_read()
        return self._owner

    def destroy(self) -> None:
        self.destroyed = True
        self._owner = None


class HipCommandBuffer:
    """``GPUCommandBuffer`` backed by an instantiated HIP graph; unlike WebGPU's it may be submitted repeatedly."""

    def __init__(self, device: "HipDevice", handle):
        self.device, self.handle = device, handle

    def destroy(self) -> None:
        if self.handle:
            self.device.lib.wdgs_command_buffer_destroy(self.handle)
            self.handle = None


class HipEncoder:
    """``GPUCommandEncoder``.  Eager (default): ops enqueue on the device stream as they are encoded and ``finish()`` is a
    marker.  ``record=True``: encodes are captured into a HIP graph; ``finish()`` returns a replayable ``HipCommandBuffer``."""

    def __init__(self, device: "HipDevice", label: str = "", record: bool = False):
        self.device, self.label, self.record = device, label, record
        self._open = False
        if record:
            check(device.lib.wdgs_encoder_begin(device.handle))
            self._open = True

    def __enter__(self) -> "HipEncoder":
        return self

    def __exit__(self, exc_type, exc, tb) -> bool:
        """``with device.createCommandEncoder(record=True) as enc:`` -- an exception inside the block (a failed encode, a
        Python error) drops the recording instead of leaving the device's stream in capture mode."""
        if exc_type is not None:
            self.abort()
        return False

    def abort(self) -> None:
        """Discards an unfinished recording (``wdgs_encoder_abort``); a no-op for eager or finished encoders."""
        if self._open:
            self._open = False
            check(self.device.lib.wdgs_encoder_abort(self.device.handle))

    def clearBuffer(self, buf: HipBuffer) -> None:
        buf.clear()

    def copyBufferToBuffer(self, src: HipBuffer, src_off: int, dst: HipBuffer, dst_off: int, size: int) -> None:
        """``encoder.copyBufferToBuffer`` (trainer.ts:445): device to device, stream-ordered, recordable."""
        if src_off + size > src.size or dst_off + size > dst.size:
            raise _lib.WdgsError(_lib.WDGS_E_INVALID, f"copyBufferToBuffer: {size} bytes at {src_off} -> {dst_off} exceed the buffers ({src.size}, {dst.size})")
        if src.before_read is not None:
            src.before_read()
        check(self.device.lib.wdgs_copy_buffer_to_buffer(self.device.handle, dst.ptr + dst_off, src.ptr + src_off, int(size)))

    def finish(self):
        if not self.record:
            return self
        if not self._open:
            raise _lib.StateError(_lib.WDGS_E_STATE, "encoder already finished or aborted")
        self._open = False  # wdgs_encoder_finish leaves capture mode whether or not it succeeds
        h = C.c_void_p()
        check(self.device.lib.wdgs_encoder_finish(self.device.handle, C.byref(h)))
        return HipCommandBuffer(self.device, h)


class _Queue:
    def __init__(self, device: "HipDevice"):
        self.device = device

    def submit(self, cmds) -> None:
        """Eager encoders have already put their work on the stream; recorded command buffers are launched here."""
        for c in cmds or ():
            if isinstance(c, HipCommandBuffer):
                check(self.device.lib.wdgs_queue_submit(self.device.handle, c.handle))

    def onSubmittedWorkDone(self, callback=None) -> None:
        """Without a callback: block until the stream drains (the awaited Promise of trainer.ts:639-645).  With one: return
        at once; ``callback()`` runs on a runtime thread when the work submitted so far is done (it must not touch the GPU)."""
        if callback is None:
            self.device.synchronize()
            return
        cb = _lib.DoneCallback(lambda _user: callback())
        self.device._keepalive.append(cb)  # the trampoline must outlive its call; released at the next synchronize()
        check(self.device.lib.wdgs_queue_on_done(self.device.handle, cb, None))

    def mark(self) -> int:
        """A ticket for "everything submitted to the current lane so far" -- the Promise of ``onSubmittedWorkDone()``, kept for later."""
        t = C.c_uint64(0)
        check(self.device.lib.wdgs_queue_mark(self.device.handle, C.byref(t)))
        return int(t.value)

    def wait(self, ticket: int) -> None:
        """Awaits a ticket: blocks until that work is done, then raises deferred device-side errors like ``synchronize()``."""
        check(self.device.lib.wdgs_queue_wait(self.device.handle, C.c_uint64(int(ticket))))

    def writeBuffer(self, buf: HipBuffer, offset: int, data: np.ndarray) -> None:
        buf.write(data, offset)


class CapacityReports:
    """Capacity reports that reached the wrong owner.  The library reports a truncated tile-entry list at the next host wait on the DEVICE, to whoever
    waits (csrc/api.hip: deferred_checks consumes every pass's word and names the passes); a Trainer and a Viewer that share a device each handle the
    reports that name their own passes -- and leave the others HERE, where the passes' owner looks at its own next wait (dropped instead, a
    report consumed by the wrong owner was lost to the right one)."""

    _NAMED = re.compile(r"\(forward pass (0x[0-9a-fA-F]+)\)")

    def __init__(self, keep: int = 16):
        self.pending: list = []
        self.keep = keep

    @classmethod
    def passes_named(cls, error) -> set:
        return {int(h, 16) for h in cls._NAMED.findall(str(error))}

    def post(self, error) -> None:
        self.pending.append(error)
        del self.pending[:-self.keep]

    def take(self, own_handles) -> Optional[Exception]:
        """The oldest pending report that names one of ``own_handles`` (removed), or None."""
        own = {int(h) for h in own_handles}
        for i, e in enumerate(self.pending):
            if self.passes_named(e) & own:
                return self.pending.pop(i)
        return None


class HipDevice:
    """``GPUDevice`` + ``GPUQueue`` on one MI355X: a HIP ordinal and torch's current stream on it."""

    def __init__(self, ordinal: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("webdgs_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path")
        self.lib = _lib.load()
        self.ordinal = ordinal
        self.torch_device = torch.device("cuda", ordinal)
        torch.cuda.set_device(ordinal)
        # One explicit stream shared by torch (allocations' fills, H2D copies, RCCL hand-off) and the library's kernels.
        # torch's default stream has a NULL handle, which wdgs_device_create reads as "create your own stream" -- that
        # would leave torch.zeros() fills unordered against our kernels -- so make a real stream current instead.
        self.torch_stream = torch.cuda.Stream(self.torch_device)
        torch.cuda.set_stream(self.torch_stream)
        stream = self.torch_stream.cuda_stream
        assert stream, "torch returned a NULL stream handle"
        h = C.c_void_p()
        check(self.lib.wdgs_device_create(ordinal, C.c_void_p(stream), C.byref(h)))
        self.handle = h
        self.queue = _Queue(self)
        self.capacityReports = CapacityReports()
        self._keepalive: list = []

    def createCommandEncoder(self, label: str = "", record: bool = False) -> HipEncoder:
        return HipEncoder(self, label, record)

    def createBuffer(self, size: int, label: str = "") -> HipBuffer:
        """Zero-filled like a WebGPU buffer."""
        t = torch.zeros(max(1, (int(size) + 3) // 4), dtype=torch.int32, device=self.torch_device)
        return HipBuffer(self, t.data_ptr(), int(size), t, label)

    def bufferFrom(self, array: np.ndarray, label: str = "") -> HipBuffer:
        a = np.ascontiguousarray(array)
        if a.nbytes == 0:  # an empty point cloud still needs a valid (never dereferenced) device address
            t = torch.zeros(4, dtype=torch.uint8, device=self.torch_device)
            return HipBuffer(self, t.data_ptr(), 0, t, label)
        t = torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(self.torch_device)
        return HipBuffer(self, t.data_ptr(), a.nbytes, t, label)

    def view(self, ptr: int, size: int, label: str = "") -> HipBuffer:
        return HipBuffer(self, ptr, size, None, label)

    def synchronize(self) -> None:
        check(self.lib.wdgs_device_synchronize(self.handle))
        self._keepalive.clear()

    def memoryInfo(self) -> dict:
        """``device.limits`` as far as memory goes (trainer.ts:147): bytes free / total on the device, and what the library's allocation cache holds."""
        f, t, c = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        check(self.lib.wdgs_device_memory_info(self.handle, C.byref(f), C.byref(t), C.byref(c)))
        return dict(free=int(f.value), total=int(t.value), cached=int(c.value))

    def selectLane(self, lane: int) -> None:
        """Directs every later encode / submit to lane 0 (this device's stream) or an internal one, 1..MAX_LANES-1 (``include/webdgs.h``: lanes)."""
        check(self.lib.wdgs_device_select_lane(self.handle, int(lane)))

    def laneMark(self, lane: int, mark: int) -> None:
        """Remembers the current end of ``lane`` in mark ``mark`` (``wdgs_device_lane_mark``) for ``laneWaitMark`` calls made later."""
        check(self.lib.wdgs_device_lane_mark(self.handle, int(lane), int(mark)))

    def laneWaitMark(self, lane: int, mark: int) -> None:
        check(self.lib.wdgs_device_lane_wait_mark(self.handle, int(lane), int(mark)))

    def laneOrder(self, waiter: int, signal: int) -> None:
        """What lane ``waiter`` gets from now on runs after what lane ``signal`` has been given so far (device-side, no host wait)."""
        check(self.lib.wdgs_device_lane_order(self.handle, int(waiter), int(signal)))

    def setProfiling(self, enabled: bool) -> None:
        check(self.lib.wdgs_device_set_profiling(self.handle, 1 if enabled else 0))

    def kernelTimes(self, reset: bool = False) -> dict:
        n = C.c_uint32(0)
        arr = (_lib.KernelTime * 128)()
        check(self.lib.wdgs_device_get_kernel_times(self.handle, arr, 128, C.byref(n)))
        out = {arr[i].name.decode(): (arr[i].launches, arr[i].total_ms) for i in range(min(n.value, 128))}
        if reset:
            check(self.lib.wdgs_device_reset_kernel_times(self.handle))
        return out

    def destroy(self) -> None:
        """Drains the stream and frees the library-side device.  Ops and command buffers should be destroyed first; one that is
        destroyed later only releases its memory (include/webdgs.h "teardown order").  Idempotent."""
        if self.handle:
            self.lib.wdgs_encoder_abort(self.handle)
            self.lib.wdgs_device_destroy(self.handle)
            self.handle = None
        self._keepalive.clear()


@dataclasses.dataclass
class PointCloud:
    """``PointCloud`` of ``src/utils/load-pointcloud.ts:16-23``."""

    type: str
    num_points: int
    sh_deg: int
    gaussian_3d_buffer: HipBuffer
    sh_buffer: HipBuffer
    # set while an Optimizer trains this cloud with deferred SH writes (``Optimizer.setDeferredSH``): the compact SH-DC array every forward pass
    # built on the cloud reads in place of the rows' first six bytes -- the trainer's passes and a Viewer's alike, without the host knowing
    dc_words: Optional[HipBuffer] = None


def createPointCloud(device: HipDevice, gaussians: np.ndarray, sh: np.ndarray, sh_deg: int) -> PointCloud:
    n = int(gaussians.shape[0])
    return PointCloud("full", n, int(sh_deg), device.bufferFrom(gaussians.astype(np.uint32, copy=False), "gaussian_3d_buffer"),
                      device.bufferFrom(sh.astype(np.uint32, copy=False), "sh_buffer"))


def allocatePointCloudLike(device: HipDevice, template: PointCloud, options: dict) -> PointCloud:
    """``allocatePointCloudLike`` (``src/utils/allocate-pointcloud.ts:8-44``)."""
    n = max(1, int(options["numPoints"]))
    tp = max(1, int(template.num_points))
    bpg = max(1, template.gaussian_3d_buffer.size // tp)
    bps = max(1, template.sh_buffer.size // tp)
    return PointCloud(template.type or "normal", n, template.sh_deg or 0, device.createBuffer(n * bpg, "resized gaussian_3d_buffer"),
                      device.createBuffer(n * bps, "resized sh_buffer"))


# ----------------------------------------------------------------------------- scanner / sorter
class PrefixScanner:
    """``PrefixScanner`` (``src/prefix/prefix.ts:26-43``)."""

    def __init__(self, max_elements: int, device: HipDevice):
        self.device, self.max_elements = device, int(max_elements)
        h = C.c_void_p()
        check(device.lib.wdgs_prefix_scanner_create(device.handle, self.max_elements, C.byref(h)))
        self.handle = h
        self.input_buffer = device.view(device.lib.wdgs_prefix_scanner_input(h), 4 * self.max_elements, "prefix-input")
        self.output_buffer = device.view(device.lib.wdgs_prefix_scanner_output(h), 4 * self.max_elements, "prefix-output")

    def set_count(self, count: int) -> dict:
        check(self.device.lib.wdgs_prefix_scanner_set_count(self.handle, int(count)))
        return {"num_workgroups": (int(count) + 4095) // 4096}

    def scan(self, encoder: Optional[HipEncoder] = None) -> None:
        check(self.device.lib.wdgs_prefix_scanner_scan(self.handle))

    def destroy(self) -> None:
        if self.handle:
            self.device.lib.wdgs_prefix_scanner_destroy(self.handle)
            self.handle = None


def get_prefix_scanner(max_elements: int, device: HipDevice) -> PrefixScanner:
    return PrefixScanner(max_elements, device)


class DynamicSortStuff:
    """``DynamicSortStuff`` (``src/sort/sort_dynamic.ts:9-24``): count comes from ``stats_buffer[0]`` on the device."""

    def __init__(self, max_capacity: int, device: HipDevice, stats_buffer: HipBuffer):
        self.device = device
        h = C.c_void_p()
        check(device.lib.wdgs_sorter_create(device.handle, int(max_capacity), C.c_void_p(stats_buffer.ptr), C.byref(h)))
        self.handle = h
        cap = device.lib.wdgs_sorter_capacity(h)
        self.capacity = cap
        self.ping_pong = [dict(sort_depths_buffer=device.view(device.lib.wdgs_sorter_keys(h, i), 4 * cap),
                               sort_indices_buffer=device.view(device.lib.wdgs_sorter_values(h, i), 4 * cap)) for i in range(2)]
        self.final_out_index = 0

    def sort(self, encoder: Optional[HipEncoder] = None, key_bits: int = 32) -> None:
        check(self.device.lib.wdgs_sorter_sort(self.handle, int(key_bits)))
        self.final_out_index = self.device.lib.wdgs_sorter_final_out_index(self.handle)

    def destroy(self) -> None:
        if self.handle:
            self.device.lib.wdgs_sorter_destroy(self.handle)
            self.handle = None


def get_dynamic_sorter(max_capacity: int, device: HipDevice, stats_buffer: HipBuffer) -> DynamicSortStuff:
    return DynamicSortStuff(max_capacity, device, stats_buffer)


# ----------------------------------------------------------------------------- TiledForwardPass
class TiledForwardPass:
    """``TiledForwardPass`` (``src/renderers/tiled-forward-pass.ts:62-534``)."""

    def __init__(self, device: HipDevice, pointCloud: PointCloud, cameraBuffer: HipBuffer, config: dict):
        self.device, self.pointCloud, self.cameraBuffer = device, pointCloud, cameraBuffer
        self.destroyed = False
        cfg = _lib.TiledForwardConfig(
            num_points=pointCloud.num_points, sh_deg=pointCloud.sh_deg, viewport_width=int(config["viewportWidth"]),
            viewport_height=int(config["viewportHeight"]), gaussian_scale=float(config.get("gaussianScale", 1.0)),
            point_size_px=float(config.get("pointSizePx", 3.0)), max_splat_radius_px=float(config.get("maxSplatRadiusPx", 128.0)),
            render_mode=1 if config.get("renderMode", "gaussian") == "gaussian" else 0,
            max_tile_entries=int(config.get("maxTileEntries", 0)), compat_caps=1 if config.get("compatCaps", False) else 0)
        h = C.c_void_p()
        check(device.lib.wdgs_tiled_forward_create(device.handle, C.byref(cfg), C.byref(h)))
        self.handle = h
        self._dc_source: Optional[HipBuffer] = None

    def syncDcSource(self) -> None:
        """Follows the cloud's ``dc_words`` (set by ``Optimizer.setDeferredSH``): a pass built on a cloud that is being trained renders the
        trained colours even if its host never heard of deferred SH writes (the reference's Viewer shares the Trainer's PointCloud,
        main.ts:389, 524)."""
        want = getattr(self.pointCloud, "dc_words", None)
        if want is not getattr(self, "_dc_source", None):
            self.setDcSource(want)

    def encode(self, encoder: Optional[HipEncoder] = None, options: Optional[dict] = None) -> None:
        skip = 1 if (options or {}).get("skipSort") else 0
        self.syncDcSource()
        check(self.device.lib.wdgs_tiled_forward_encode(self.handle, self.pointCloud.gaussian_3d_buffer.ptr, self.pointCloud.sh_buffer.ptr,
                                                        self.cameraBuffer.ptr, skip))

    def encodeProjected(self, encoder: Optional["HipEncoder"] = None) -> None:
        """The rest of ``encode`` (scan, emit, sort) for a pass whose K1 ran through ``projectViews`` (view-batched step; no reference
        counterpart)."""
        check(self.device.lib.wdgs_tiled_forward_encode_projected(self.handle))

    def isProjected(self) -> bool:
        """True while the pass holds a projection (``projectViews``) that no ``encode*`` call made through this host has consumed or
        overwritten since (the rest of the pass can run once per projection)."""
        return bool(self.device.lib.wdgs_tiled_forward_is_projected(self.handle))

    def setCameraBuffer(self, buffer: HipBuffer) -> None:
        self.cameraBuffer = buffer

    def setDcSource(self, dcWords: Optional[HipBuffer]) -> None:
        """K1 takes the SH-DC halves from the optimizer's compact array (``Optimizer.setDeferredSH``) instead of the cloud's rows; ``None``
        restores the rows.  No reference counterpart (include/webdgs.h)."""
        self._dc_source = dcWords  # (kept alive)
        check(self.device.lib.wdgs_tiled_forward_set_dc_source(self.handle, dcWords.ptr if dcWords is not None else None))

    def setPointCloud(self, pointCloud: PointCloud) -> bool:
        """Adopts a point cloud of another size (``wdgs_tiled_forward_resize``: buffers reused, or re-allocated with headroom) instead of
        the destroy + construct of ``applyPointCloudSwap`` (trainer.ts:201-237).  False -- nothing changed -- if the SH degree differs."""
        if pointCloud.sh_deg != self.pointCloud.sh_deg:
            return False
        check(self.device.lib.wdgs_tiled_forward_resize(self.handle, pointCloud.num_points))
        self.pointCloud = pointCloud
        return True

    def setGaussianScale(self, value: float) -> None:
        check(self.device.lib.wdgs_tiled_forward_set_gaussian_scale(self.handle, float(value)))

    def setPointSize(self, value: float) -> None:
        check(self.device.lib.wdgs_tiled_forward_set_point_size(self.handle, float(value)))

    def setRenderMode(self, mode: str) -> None:
        check(self.device.lib.wdgs_tiled_forward_set_render_mode(self.handle, 1 if mode == "gaussian" else 0))

    def setViewport(self, width: int, height: int) -> None:
        check(self.device.lib.wdgs_tiled_forward_set_viewport(self.handle, int(width), int(height)))

    def _res(self) -> _lib.TiledForwardResources:
        r = _lib.TiledForwardResources()
        check(self.device.lib.wdgs_tiled_forward_get_resources(self.handle, C.byref(r)))
        return r

    def getResources(self) -> dict:
        r, d, n = self._res(), self.device, max(1, self.pointCloud.num_points)
        return dict(splatBuffer=d.view(r.splat_buffer, 24 * n), depthsBuffer=d.view(r.depths_buffer, 4 * n),
                    tileKeysBuffer=d.view(r.tile_keys_buffer, 4 * r.max_tile_entries), tileIndicesBuffer=d.view(r.tile_indices_buffer, 4 * r.max_tile_entries),
                    tileOffsetsBuffer=d.view(r.tile_offsets_buffer, 4 * n), tileCountsBuffer=d.view(r.tile_counts_buffer, 4 * n),
                    statsBuffer=d.view(r.stats_buffer, 16), numTilesX=r.num_tiles_x, numTilesY=r.num_tiles_y, totalTiles=r.total_tiles,
                    maxTileEntries=r.max_tile_entries, settings=np.array(list(r.settings), np.float32))

    def getSortedIndicesBuffer(self) -> HipBuffer:
        return self.getResources()["tileIndicesBuffer"]

    def getSortedKeysBuffer(self) -> HipBuffer:
        return self.getResources()["tileKeysBuffer"]

    def getTileOffsetsBuffer(self) -> HipBuffer:
        return self.getResources()["tileOffsetsBuffer"]

    def getStatsBuffer(self) -> HipBuffer:
        return self.getResources()["statsBuffer"]

    def check(self) -> np.ndarray:
        """Synchronises; raises ``CapacityError`` if the last encode overflowed ``maxTileEntries``. Returns the 4 stats words."""
        st = (C.c_uint32 * 4)()
        check(self.device.lib.wdgs_tiled_forward_check(self.handle, st))
        return np.array(list(st), np.uint32)

    def setLongLists(self, threshold: int, maxItems: int = 0, maxRows: int = 0) -> None:
        """Long tile lists (``include/webdgs.h``: wdgs_tiled_forward_set_long_lists): tiles with more than ``threshold`` entries get per-pixel lists
        (0: off); ``maxItems`` / ``maxRows`` size the work (0: keep).  Synchronises; drop command buffers recorded against the pass."""
        check(self.device.lib.wdgs_tiled_forward_set_long_lists(self.handle, int(threshold), int(maxItems), int(maxRows)))

    def longListStats(self) -> dict:
        """The last frame's long-list work: what it wanted and what the pass has room for (synchronises)."""
        st = (C.c_uint32 * 12)()
        check(self.device.lib.wdgs_tiled_forward_long_list_stats(self.handle, st))
        keys = ("blocksWanted", "itemsWanted", "forwardQueue", "backwardQueue", "rowsUsed", "rowsWanted", "stalled", "_", "maxItems", "maxBlocks", "maxRows", "threshold")
        return {k: int(v) for k, v in zip(keys, st) if k != "_"}

    def destroy(self) -> None:
        if self.destroyed:
            return
        self.destroyed = True
        self.device.lib.wdgs_tiled_forward_destroy(self.handle)
        self.handle = None


# ----------------------------------------------------------------------------- TiledRasterizer
class TiledRasterizer:
    """``TiledRasterizer`` (``src/renderers/tiled-rasterizer.ts:34-368``) without the swap-chain blit."""

    def __init__(self, config: dict):
        self.device: HipDevice = config["device"]
        self.forwardPass: TiledForwardPass = config["forwardPass"]
        self.destroyed = False
        self.width = self.height = 0
        h = C.c_void_p()
        check(self.device.lib.wdgs_tiled_rasterizer_create(self.device.handle, self.forwardPass.handle, 1 if config.get("compatCaps") else 0, C.byref(h)))
        self.handle = h

    def encode(self, encoder: Optional[HipEncoder], width: int, height: int) -> None:
        check(self.device.lib.wdgs_tiled_rasterizer_encode(self.handle, int(width), int(height)))
        self.width, self.height = int(width), int(height)

    def _get(self, fn, nbytes: int) -> HipBuffer:
        p = C.c_void_p()
        check(fn(self.handle, C.byref(p)))
        return self.device.view(p.value, nbytes)

    def getOutputTextureView(self) -> HipBuffer:
        return self._get(self.device.lib.wdgs_tiled_rasterizer_get_output, 4 * self.width * self.height)

    def getAlphaTextureView(self) -> HipBuffer:
        return self._get(self.device.lib.wdgs_tiled_rasterizer_get_alpha, 4 * self.width * self.height)

    def getNContribTextureView(self) -> HipBuffer:
        return self._get(self.device.lib.wdgs_tiled_rasterizer_get_n_contrib, 4 * self.width * self.height)

    def getTileOffsetsBuffer(self) -> HipBuffer:
        tiles = ((self.width + 15) // 16) * ((self.height + 15) // 16)
        return self._get(self.device.lib.wdgs_tiled_rasterizer_get_tile_offsets, 4 * (tiles + 1))

    def blitToTexture(self, encoder: Optional[HipEncoder], target: HipBuffer, width: Optional[int] = None, height: Optional[int] = None) -> None:
        """``blitToTexture(encoder, targetView)`` (tiled-rasterizer.ts:333-357): ``target`` is an rgba8 image buffer of
        ``width x height`` (default: the rasterizer's own size); raises before the first ``encode`` like the reference."""
        w, h = int(width or self.width), int(height or self.height)
        if self.handle is not None and w * h * 4 > target.size:
            raise _lib.WdgsError(_lib.WDGS_E_INVALID, f"blitToTexture: target of {target.size} bytes is smaller than {w}x{h} rgba8")
        check(self.device.lib.wdgs_tiled_rasterizer_blit(self.handle, target.ptr, max(w, 0), max(h, 0)))

    def destroy(self) -> None:
        if self.destroyed:
            return
        self.destroyed = True
        self.device.lib.wdgs_tiled_rasterizer_destroy(self.handle)
        self.handle = None


# ----------------------------------------------------------------------------- TiledBackwardPass
def _training_config(tc: Optional[dict]) -> _lib.TrainingConfig:
    tc = tc or {}
    return _lib.TrainingConfig(float(tc.get("lambda_l1", 0.8)), float(tc.get("lambda_l2", 0.0)), float(tc.get("lambda_dssim", 0.2)),
                               float(tc.get("c1", 0.01 * 0.01)), float(tc.get("c2", 0.03 * 0.03)))


class TiledBackwardPass:
    """``TiledBackwardPass`` (``src/renderers/tiled-backward-pass.ts:71-861``)."""

    def __init__(self, device: HipDevice, pointCloud: PointCloud, config: dict):
        self.device, self.pointCloud = device, pointCloud
        self.destroyed = False
        self.viewportWidth, self.viewportHeight = int(config["viewportWidth"]), int(config["viewportHeight"])
        self.trainingConfig = dict(config.get("trainingConfig") or {})
        cfg = _lib.TiledBackwardConfig(pointCloud.num_points, pointCloud.sh_deg, self.viewportWidth, self.viewportHeight, _training_config(self.trainingConfig),
                                       float(config.get("gaussianScale", 1.0)), float(config.get("pointSizePx", 3.0)),
                                       float(config.get("maxSplatRadiusPx", 128.0)))
        h = C.c_void_p()
        check(device.lib.wdgs_tiled_backward_create(device.handle, C.byref(cfg), C.byref(h)))
        self.handle = h

    @staticmethod
    def _resources(res: dict) -> _lib.TiledBackwardResources:
        def p(k):
            b = res.get(k)
            return b.ptr if b is not None else None
        return _lib.TiledBackwardResources(p("splatBuffer"), p("tileOffsetsBuffer"), p("tileIndicesBuffer"), p("cameraBuffer"), p("alphaTexture"),
                                           p("nContribTexture"))

    def setPointCloud(self, pointCloud: PointCloud) -> bool:
        """See ``TiledForwardPass.setPointCloud`` (``wdgs_tiled_backward_resize``)."""
        if pointCloud.sh_deg != self.pointCloud.sh_deg:
            return False
        check(self.device.lib.wdgs_tiled_backward_resize(self.handle, pointCloud.num_points))
        self.pointCloud = pointCloud
        return True

    def encode(self, encoder: Optional[HipEncoder], predictedTexture: HipBuffer, targetTexture: HipBuffer, forwardResources: dict, options=None) -> None:
        r = self._resources(forwardResources)
        check(self.device.lib.wdgs_tiled_backward_encode(self.handle, predictedTexture.ptr, targetTexture.ptr, C.byref(r), self.pointCloud.gaussian_3d_buffer.ptr))

    def encodeRaster(self, encoder: Optional[HipEncoder], predictedTexture: HipBuffer, targetTexture: HipBuffer, forwardResources: dict) -> None:
        """First half of ``encode`` (K15 loss gradient, clear, K16 backward raster): ``wdgs_tiled_backward_encode_raster``."""
        r = self._resources(forwardResources)
        check(self.device.lib.wdgs_tiled_backward_encode_raster(self.handle, predictedTexture.ptr, targetTexture.ptr, C.byref(r)))

    def encodeGeometry(self, encoder: Optional[HipEncoder], cameraBuffer: HipBuffer, accumulate: Optional[dict] = None) -> None:
        """Second half of ``encode`` (K17).  ``accumulate`` (a batched step): ``dict(sums, visible, tileCounts, guard, stats, first)`` --
        K17 then also adds this view's gradient to the step's fp32 block and folds the forward pass's overflow word (``stats`` + 8
        bytes) into the guard word, instead of ``storeGradients`` / ``accumulateGradients`` + ``guardAccumulate`` afterwards."""
        into = None
        if accumulate is not None:
            a = accumulate
            into = C.byref(_lib.ViewAccumulate(a["sums"].ptr, a["visible"].ptr, a["tileCounts"].ptr, a["guard"].ptr, a["stats"].ptr + 8, 1 if a["first"] else 0))
        check(self.device.lib.wdgs_tiled_backward_encode_geometry(self.handle, cameraBuffer.ptr, self.pointCloud.gaussian_3d_buffer.ptr, into))

    def computeLossOnly(self, encoder, predictedTexture: HipBuffer, targetTexture: HipBuffer) -> None:
        check(self.device.lib.wdgs_tiled_backward_compute_loss_only(self.handle, predictedTexture.ptr, targetTexture.ptr))

    def computeMetricMap(self, encoder, predictedTexture: HipBuffer, targetTexture: HipBuffer, options: Optional[dict] = None) -> None:
        thr = float((options or {}).get("threshold", 0.5))
        check(self.device.lib.wdgs_tiled_backward_compute_metric_map(self.handle, predictedTexture.ptr, targetTexture.ptr, thr))

    def computeMetricCounts(self, encoder, resources: dict, options: Optional[dict] = None) -> None:
        r = self._resources(resources)
        clear = 1 if (options or {}).get("clear", True) else 0
        n_inst = resources["tileIndicesBuffer"].size // 4
        check(self.device.lib.wdgs_tiled_backward_compute_metric_counts(self.handle, C.byref(r), n_inst, clear))

    def normalizeMetricCounts(self, encoder, options: dict) -> None:
        check(self.device.lib.wdgs_tiled_backward_normalize_metric_counts(self.handle, max(1, int(options["divisor"]))))

    def setViewport(self, width: int, height: int) -> None:
        check(self.device.lib.wdgs_tiled_backward_set_viewport(self.handle, int(width), int(height)))
        self.viewportWidth, self.viewportHeight = int(width), int(height)

    def setTrainingConfig(self, next_cfg: dict) -> None:
        self.trainingConfig.update({k: v for k, v in next_cfg.items() if v is not None})
        tc = _training_config(self.trainingConfig)
        check(self.device.lib.wdgs_tiled_backward_set_training_config(self.handle, C.byref(tc)))

    def setGradientOutput(self, enabled: bool) -> None:
        """Whether ``Optimizer.stepWithGeometry`` also writes K17's packed gradient to ``getGradientsBuffer()`` (default: yes, as the
        reference's K17 does).  A host that never reads it saves 32 bytes per Gaussian and step; ``encode`` / ``encodeGeometry`` always write it."""
        check(self.device.lib.wdgs_tiled_backward_set_gradient_output(self.handle, 1 if enabled else 0))

    def getGradientsBuffer(self) -> HipBuffer:
        return self.device.view(self.device.lib.wdgs_tiled_backward_gradients(self.handle), 32 * max(1, self.pointCloud.num_points))

    def setMetricCountsTarget(self, counts: Optional[HipBuffer]) -> None:
        """``computeMetricCounts`` of this pass adds into ``counts`` (another pass's ``getMetricCountsBuffer()``) instead of its own array;
        ``None`` restores its own.  Lets several passes take the metric views of one densify event on different lanes (integer atomics:
        any order gives the same bits).  No reference counterpart."""
        self._metric_target = counts  # (kept alive)
        check(self.device.lib.wdgs_tiled_backward_set_metric_counts_target(self.handle, counts.ptr if counts is not None else None))

    def getMetricCountsBuffer(self) -> HipBuffer:
        return self.device.view(self.device.lib.wdgs_tiled_backward_metric_counts(self.handle), 4 * max(1, self.pointCloud.num_points))

    def getLossTextureView(self) -> HipBuffer:
        return self.device.view(self.device.lib.wdgs_tiled_backward_loss_image(self.handle), 16 * self.viewportWidth * self.viewportHeight)

    def getMetricMapTextureView(self) -> HipBuffer:
        return self.device.view(self.device.lib.wdgs_tiled_backward_metric_map(self.handle), 4 * self.viewportWidth * self.viewportHeight)

    def getMetricMapTexture(self) -> HipBuffer:
        """``getMetricMapTexture`` (tiled-backward-pass.ts): texture and view are the same r32uint image buffer here."""
        return self.getMetricMapTextureView()

    def getAccumulatorsBuffer(self) -> HipBuffer:
        """INTERNAL (parity tests): i32[N*12] fixed-point accumulators of the last encode."""
        return self.device.view(self.device.lib.wdgs_tiled_backward_accumulators(self.handle), 48 * max(1, self.pointCloud.num_points))

    def getMetricMinMaxBuffer(self) -> HipBuffer:
        return self.device.view(self.device.lib.wdgs_tiled_backward_metric_minmax(self.handle), 8)

    def destroy(self) -> None:
        if self.destroyed:
            return
        self.destroyed = True
        self.device.lib.wdgs_tiled_backward_destroy(self.handle)
        self.handle = None


def downsampleRGBA8(device: HipDevice, src: HipBuffer, src_w: int, src_h: int, dst: HipBuffer, dst_w: int, dst_h: int) -> None:
    """The GT down-sample render pass of ``trainer.ts:303-328`` (``blit.wgsl`` ``fs_main`` with a linear sampler)."""
    check(device.lib.wdgs_downsample_rgba8(device.handle, src.ptr, src_w, src_h, dst.ptr, dst_w, dst_h))


def imageSSE(device: HipDevice, a: HipBuffer, b: HipBuffer, num_pixels: int) -> int:
    """Exact sum of squared rgb8 differences of two rgba8 images (synchronises)."""
    out = device.createBuffer(8, "sse")
    check(device.lib.wdgs_image_sse_rgb8(device.handle, a.ptr, b.ptr, int(num_pixels), out.ptr))
    return int(out.read(np.uint64, count=1)[0])


def imagePSNR(device: HipDevice, a: HipBuffer, b: HipBuffer, num_pixels: int) -> float:
    """PSNR in dB over the rgb channels of two rgba8 images; +inf when identical."""
    sse = imageSSE(device, a, b, num_pixels)
    return float("inf") if sse == 0 else 10.0 * float(np.log10(255.0 * 255.0 * 3.0 * num_pixels / sse))


# ----------------------------------------------------------------------------- Optimizer
DEFAULT_ADAM_HYPERPARAMETERS = dict(lr_pos=0.00016, lr_color=0.0025, lr_opacity=0.05, lr_scale=0.005, lr_rot=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8)
_STATE_FIELDS = ("optPosBuffer", "optRotBuffer", "optScaleBuffer", "optOpacityBuffer", "paramSH", "stateSH")


def allocateOptimizerStateBuffers(device: HipDevice, numPoints: int) -> dict:
    """``allocateOptimizerStateBuffers`` (``src/renderers/optimizer.ts:27-38``): zero-filled."""
    sizes = (C.c_size_t * 6)()
    check(device.lib.wdgs_optimizer_state_sizes(int(numPoints), C.byref(sizes)))
    return {k: device.createBuffer(sizes[i], k) for i, k in enumerate(_STATE_FIELDS)}


def _state_struct(buffers: dict) -> _lib.OptimizerState:
    return _lib.OptimizerState(*[buffers[k].ptr for k in _STATE_FIELDS])


class Optimizer:
    """``Optimizer`` (``src/renderers/optimizer.ts:40-363``)."""

    def __init__(self, device: HipDevice, pointCloud: PointCloud, params: Optional[dict] = None, initialState: Optional[dict] = None):
        self.device, self.numPoints = device, pointCloud.num_points
        self.params = {**DEFAULT_ADAM_HYPERPARAMETERS, **(params or {})}
        self.destroyed = False
        hp = _lib.AdamHyperparameters(*[float(self.params[k]) for k in DEFAULT_ADAM_HYPERPARAMETERS])
        if initialState and initialState.get("buffers"):
            self.buffers = initialState["buffers"]  # adopted (optimizer.ts:81-88)
            it = int(initialState.get("iteration") or 0)
            fresh = False
        else:
            self.buffers = allocateOptimizerStateBuffers(device, self.numPoints)
            it, fresh = 0, True
        st = _state_struct(self.buffers)
        h = C.c_void_p()
        check(device.lib.wdgs_optimizer_create(device.handle, self.numPoints, C.byref(hp), None, None, C.byref(st), 0, it, C.byref(h)))
        self.handle = h
        self._deferred_cloud: Optional[PointCloud] = None
        if fresh:  # initBuffers (optimizer.ts:145-253): K20 unpack into the zero-filled state
            check(device.lib.wdgs_optimizer_init_from_point_cloud(h, pointCloud.gaussian_3d_buffer.ptr, pointCloud.sh_buffer.ptr))

    def getIteration(self) -> int:
        return int(self.device.lib.wdgs_optimizer_get_iteration(self.handle))

    def getHyperparameters(self) -> dict:
        return dict(self.params)

    def setHyperparameters(self, next_params: dict) -> None:
        self.params.update(next_params)
        hp = _lib.AdamHyperparameters(*[float(self.params[k]) for k in DEFAULT_ADAM_HYPERPARAMETERS])
        check(self.device.lib.wdgs_optimizer_set_hyperparameters(self.handle, C.byref(hp)))

    def getStateBuffers(self) -> dict:
        """``getStateBuffers`` (``optimizer.ts:267-278``).  Goes through ``wdgs_optimizer_get_state`` so the compact SH-DC copy the
        kernels train is written back into ``paramSH`` / ``stateSH`` (stream-ordered) before anyone reads them."""
        st = _lib.OptimizerState()
        check(self.device.lib.wdgs_optimizer_get_state(self.handle, C.byref(st)))
        # position, log-scale and SH-DC {param, m, v} are trained in a compact copy: a handle the host keeps across steps is brought up to date
        # whenever its content is read (the reference's GPUBuffers are live)
        for b in self.buffers.values():
            if b.before_read is None:
                b.before_read = self._flush_state
        return self.buffers

    def _flush_state(self) -> None:
        if not self.destroyed and self.handle:
            st = _lib.OptimizerState()
            check(self.device.lib.wdgs_optimizer_get_state(self.handle, C.byref(st)))

    # ---- deferred SH writes (include/webdgs.h: wdgs_optimizer_set_deferred_sh; no reference counterpart)
    def setDeferredSH(self, pointCloud: PointCloud, enabled: bool = True) -> Optional[HipBuffer]:
        """On: the steps write the trained SH-DC halves to a compact array instead of the cloud's 96-byte rows; returns that array (give it
        to every forward pass that renders the cloud: ``TiledForwardPass.setDcSource``).  The cloud's SH buffer is brought up to date
        by ``flushSH`` -- automatically before any host read of it (``HipBuffer.before_read``), explicitly before device-side readers
        that have no dc source (a viewer's own forward pass, ``DensifyPrunePass.encodeScatter``).  Off: flushes and restores the
        reference's write pattern."""
        check(self.device.lib.wdgs_optimizer_set_deferred_sh(self.handle, pointCloud.sh_buffer.ptr, 1 if enabled else 0))
        self._deferred_cloud = pointCloud if enabled else None
        pointCloud.sh_buffer.before_read = (lambda: self.flushSH(pointCloud)) if enabled else None
        ptr = self.device.lib.wdgs_optimizer_dc_words(self.handle)
        words = self.device.view(ptr, 8 * max(1, self.numPoints), "sh-dc words") if (enabled and ptr) else None
        pointCloud.dc_words = words
        return words

    def flushSH(self, pointCloud: PointCloud) -> None:
        """Writes the deferred SH-DC halves into the cloud's rows (a no-op when nothing was trained since the last flush)."""
        if not self.destroyed and self.handle:
            check(self.device.lib.wdgs_optimizer_flush_sh(self.handle, pointCloud.sh_buffer.ptr))

    def applyRepackedRows(self, rows: HipBuffer, skipFirst: int, skipCount: int, guard: Optional[HipBuffer], pointCloud: PointCloud) -> None:
        """``applyRepackedRows`` for this optimizer's replica: with deferred SH writes the gathered halves go to the compact array."""
        check(self.device.lib.wdgs_optimizer_apply_repacked_rows(self.handle, rows.ptr, int(skipFirst), int(skipCount), guard.ptr if guard is not None else None,
                                                                 pointCloud.gaussian_3d_buffer.ptr, pointCloud.sh_buffer.ptr))

    def advanceIteration(self, count: int = 1) -> None:
        """Bumps the host-side step counter when a recorded command buffer containing ``step`` is re-submitted."""
        check(self.device.lib.wdgs_optimizer_advance_iteration(self.handle, int(count)))

    def step(self, encoder, coefficients: PointCloud, gradientsBuffer: HipBuffer, tileCountsBuffer: HipBuffer) -> None:
        check(self.device.lib.wdgs_optimizer_step(self.handle, coefficients.gaussian_3d_buffer.ptr, coefficients.sh_buffer.ptr, gradientsBuffer.ptr, tileCountsBuffer.ptr))

    def stepWithGeometry(self, encoder, coefficients: PointCloud, backwardPass: "TiledBackwardPass", cameraBuffer: HipBuffer, tileCountsBuffer: HipBuffer) -> None:
        """``step`` fused with K17 (``wdgs_optimizer_step_with_geometry``): call after ``backwardPass.encodeRaster`` for the view.  One
        pass over the Gaussians: geometry backward (the packed gradient still lands in the pass's gradient buffer), Adam, re-pack."""
        check(self.device.lib.wdgs_optimizer_step_with_geometry(self.handle, backwardPass.handle, cameraBuffer.ptr, coefficients.gaussian_3d_buffer.ptr,
                                                                coefficients.sh_buffer.ptr, tileCountsBuffer.ptr))

    def stepF32(self, encoder, coefficients: PointCloud, gradF32: HipBuffer, visibleCounts: HipBuffer) -> None:
        """Data-parallel step on fp32 gradients summed over views."""
        check(self.device.lib.wdgs_optimizer_step_f32(self.handle, coefficients.gaussian_3d_buffer.ptr, coefficients.sh_buffer.ptr, gradF32.ptr, visibleCounts.ptr))

    def stepF32Range(self, encoder, coefficients: PointCloud, gradF32: HipBuffer, visibleCounts: HipBuffer, first: int, count: int,
                     rowsOut: Optional[HipBuffer] = None) -> None:
        """Adam + re-pack on Gaussians ``[first, first + count)`` -- the slice this rank owns after the reduce-scatter; ``rowsOut``
        also receives the re-packed 32-byte rows for the all-gather."""
        check(self.device.lib.wdgs_optimizer_step_f32_range(self.handle, coefficients.gaussian_3d_buffer.ptr, coefficients.sh_buffer.ptr, gradF32.ptr,
                                                            visibleCounts.ptr, int(first), int(count), rowsOut.ptr if rowsOut is not None else None))

    def setGuard(self, flagBuffer: Optional[HipBuffer], offset: int = 0) -> None:
        """While the u32 at ``flagBuffer + offset`` is non-zero at execution time, ``step*`` leave every buffer untouched."""
        check(self.device.lib.wdgs_optimizer_set_guard(self.handle, (flagBuffer.ptr + offset) if flagBuffer is not None else None))

    def stateChanged(self) -> None:
        """The state arrays were rewritten from outside (slices gathered from other ranks): refresh internal copies."""
        check(self.device.lib.wdgs_optimizer_state_changed(self.handle))

    def destroy(self) -> None:
        if self.destroyed:
            return
        pc = self._deferred_cloud
        if pc is not None and self.device.handle and not pc.sh_buffer.destroyed:  # the cloud outlives its optimizer: leave its SH rows current
            try:
                self.setDeferredSH(pc, False)
            except _lib.WdgsError:
                pc.sh_buffer.before_read = None
        self.destroyed = True
        self.device.lib.wdgs_optimizer_destroy(self.handle)
        self.handle = None


def accumulateGradients(device: HipDevice, numPoints: int, gradientsBuffer: HipBuffer, tileCountsBuffer: HipBuffer, accF32: HipBuffer, visibleCounts: HipBuffer) -> None:
    check(device.lib.wdgs_accumulate_gradients(device.handle, int(numPoints), gradientsBuffer.ptr, tileCountsBuffer.ptr, accF32.ptr, visibleCounts.ptr))


def storeGradients(device: HipDevice, numPoints: int, gradientsBuffer: HipBuffer, tileCountsBuffer: HipBuffer, accF32: HipBuffer, visibleCounts: HipBuffer) -> None:
    """Overwrite form of ``accumulateGradients`` for the first view of a batch (no clearing pass needed before it)."""
    check(device.lib.wdgs_store_gradients(device.handle, int(numPoints), gradientsBuffer.ptr, tileCountsBuffer.ptr, accF32.ptr, visibleCounts.ptr))


def guardAccumulate(device: HipDevice, flag: HipBuffer, src: HipBuffer, srcOffset: int = 0, overwrite: bool = False) -> None:
    """``flag = (overwrite ? 0 : flag) | (src != 0)``: folds per-view overflow words into the guard word of a batched step."""
    check(device.lib.wdgs_guard_accumulate(device.handle, flag.ptr, src.ptr + srcOffset, 1 if overwrite else 0))


def _ptr_array(values) -> "C.Array":
    vals = [v.value if isinstance(v, C.c_void_p) else int(v) for v in values]
    return (C.c_void_p * len(vals))(*vals)


def projectViews(forwardPasses: list, cameraBuffers: list, pointCloud: PointCloud) -> None:
    """K1 of ALL the views of a batched step in one launch (``wdgs_tiled_forward_project_views``): Gaussian and SH row are read once and
    projected under every camera into that view's own forward pass.  Follow with ``forwardPasses[v].encodeProjected(encoder)``."""
    dev = forwardPasses[0].device
    for f in forwardPasses:
        f.syncDcSource()
    check(dev.lib.wdgs_tiled_forward_project_views(_ptr_array([f.handle for f in forwardPasses]), _ptr_array([c.ptr for c in cameraBuffers]), len(forwardPasses),
                                                   pointCloud.gaussian_3d_buffer.ptr, pointCloud.sh_buffer.ptr))


def geometryViews(backwardPasses: list, cameraBuffers: list, forwardPasses: list, sums: HipBuffer, visible: HipBuffer, guard: HipBuffer, pointCloud: PointCloud,
                  writeGradients: bool = False, continues: bool = False) -> None:
    """K17 of ALL the views of a batched step in one launch (``wdgs_tiled_backward_encode_geometry_views``): what ``encodeGeometry(camera,
    accumulate=dict(first=(v == 0), ...))`` per view produces -- the step's fp32 gradient block, visibility counts and guard word -- bit for
    bit, with the Gaussians read once and the block written once.  ``continues``: these views follow earlier ones of the same step that
    were handed over in a previous call (groups, in view order)."""
    dev = backwardPasses[0].device
    counts = [f.getResources()["tileCountsBuffer"].ptr for f in forwardPasses]
    stats = [f.getStatsBuffer().ptr + 8 for f in forwardPasses]   # the overflow word of each view's stats block
    check(dev.lib.wdgs_tiled_backward_encode_geometry_views(_ptr_array([b.handle for b in backwardPasses]), _ptr_array([c.ptr for c in cameraBuffers]), _ptr_array(counts),
                                                            _ptr_array(stats), len(backwardPasses), pointCloud.gaussian_3d_buffer.ptr, sums.ptr, visible.ptr, guard.ptr,
                                                            1 if writeGradients else 0, 1 if continues else 0))


def applyRepackedRows(device: HipDevice, numPoints: int, rows: HipBuffer, skipFirst: int, skipCount: int, guard: Optional[HipBuffer],
                      pointCloud: PointCloud) -> None:
    """Writes the rows the other ranks published (all-gather) into this replica's point cloud, skipping the own slice."""
    check(device.lib.wdgs_apply_repacked_rows(device.handle, int(numPoints), rows.ptr, int(skipFirst), int(skipCount), guard.ptr if guard is not None else None,
                                              pointCloud.gaussian_3d_buffer.ptr, pointCloud.sh_buffer.ptr))


# ----------------------------------------------------------------------------- DensifyPrunePass
class DensifyPrunePass:
    """``DensifyPrunePass`` (``src/renderers/densify-prune.ts:75-687``), strategy ``gpu_rebuild``."""

    def __init__(self, device: HipDevice, config: Optional[dict] = None):
        self.device = device
        self.config = dict(strategy="cpu_rebuild", numViews=1, cloneThreshold=0, splitThreshold=0, pruneThreshold=0, maxNewPointsPerStep=0,
                           maxBufferBytes=128 * 1024 * 1024)
        self.config.update(config or {})
        self.numPoints = 0
        c = self._cfg()
        h = C.c_void_p()
        check(device.lib.wdgs_densify_prune_create(device.handle, C.byref(c), C.byref(h)))
        self.handle = h

    def _cfg(self) -> _lib.DensifyConfig:
        c = self.config
        return _lib.DensifyConfig(max(1, int(c.get("numViews") or 1)), max(0, int(c.get("cloneThreshold") or 0)),
                                  float(c["splitThreshold"] if c.get("splitThreshold") is not None else 1e9), float(c.get("pruneThreshold") or 0.0),
                                  max(0, int(c.get("maxNewPointsPerStep") or 0)), int(c.get("maxBufferBytes") or 0))

    def setConfig(self, next_cfg: dict) -> None:
        self.config.update(next_cfg)
        c = self._cfg()
        check(self.device.lib.wdgs_densify_prune_set_config(self.handle, C.byref(c)))

    def getConfig(self) -> dict:
        return dict(self.config)

    def ensureSize(self, numPoints: int) -> None:
        check(self.device.lib.wdgs_densify_prune_ensure_size(self.handle, int(numPoints)))
        self.numPoints = int(numPoints)

    def encodePrepare(self, encoder, inputs: dict) -> dict:
        pc: PointCloud = inputs["pointCloud"]
        mc = inputs.get("metricCountsBuffer")
        out = _lib.DensifyPrepared()
        check(self.device.lib.wdgs_densify_prune_encode_prepare(self.handle, pc.num_points, pc.gaussian_3d_buffer.ptr, mc.ptr if mc is not None else None, C.byref(out)))
        self.numPoints = pc.num_points
        n, d = max(1, pc.num_points), self.device
        return dict(actionBuffer=d.view(out.action_buffer, 4 * n), outCountBuffer=d.view(out.out_count_buffer, 4 * n),
                    outOffsetBuffer=d.view(out.out_offset_buffer, 4 * n), outTotalBuffer=d.view(out.out_total_buffer, 4), maxOutPoints=int(out.max_out_points))

    # ---- the stages encodePrepare is made of (densify-prune.ts:327-456), for hosts that sequence them themselves
    def _buffers(self) -> dict:
        out = _lib.DensifyPrepared()
        check(self.device.lib.wdgs_densify_prune_get_buffers(self.handle, C.byref(out)))
        n, d = max(1, self.numPoints), self.device
        return dict(actionBuffer=d.view(out.action_buffer, 4 * n), outCountBuffer=d.view(out.out_count_buffer, 4 * n),
                    outOffsetBuffer=d.view(out.out_offset_buffer, 4 * n), outTotalBuffer=d.view(out.out_total_buffer, 4), maxOutPoints=int(out.max_out_points))

    def computeMaxOutPoints(self, pointCloud: PointCloud) -> int:
        m = C.c_uint32(0)
        check(self.device.lib.wdgs_densify_prune_compute_max_out_points(self.handle, pointCloud.num_points, C.byref(m)))
        return int(m.value)

    def encodeDecision(self, encoder, inputs: dict) -> dict:
        pc: PointCloud = inputs["pointCloud"]
        mc = inputs.get("metricCountsBuffer")
        check(self.device.lib.wdgs_densify_prune_encode_decision(self.handle, pc.num_points, pc.gaussian_3d_buffer.ptr, mc.ptr if mc is not None else None))
        self.numPoints = pc.num_points
        b = self._buffers()
        return dict(actionBuffer=b["actionBuffer"], outCountBuffer=b["outCountBuffer"])

    def encodePrefixSum(self, encoder) -> HipBuffer:
        check(self.device.lib.wdgs_densify_prune_encode_prefix_sum(self.handle, self.numPoints))
        return self._buffers()["outOffsetBuffer"]

    def encodeCapToMax(self, encoder, outOffsetBuffer: Optional[HipBuffer], maxOutPoints: int) -> None:
        check(self.device.lib.wdgs_densify_prune_encode_cap_to_max(self.handle, self.numPoints, max(0, int(maxOutPoints))))

    def encodeTotalOut(self, encoder, outOffsetBuffer: Optional[HipBuffer] = None) -> HipBuffer:
        check(self.device.lib.wdgs_densify_prune_encode_total_out(self.handle, self.numPoints))
        return self._buffers()["outTotalBuffer"]

    def getOutTotalBuffer(self) -> HipBuffer:
        return self._buffers()["outTotalBuffer"]

    def getActionBuffer(self) -> HipBuffer:
        return self._buffers()["actionBuffer"]

    def getOutCountBuffer(self) -> HipBuffer:
        return self._buffers()["outCountBuffer"]

    def readTotal(self) -> int:
        t = C.c_uint32(0)
        check(self.device.lib.wdgs_densify_prune_read_total(self.handle, C.byref(t)))
        return int(t.value)

    def encodeScatter(self, encoder, inputs: dict, outputs: dict) -> None:
        pc: PointCloud = inputs["pointCloud"]
        out_pc: PointCloud = outputs["outPointCloud"]
        out_n = int(inputs["outNumPoints"])
        if out_pc.num_points != out_n:
            raise _lib.WdgsError(_lib.WDGS_E_INVALID, f"encodeScatter: outPointCloud.num_points ({out_pc.num_points}) != outNumPoints ({out_n})")
        in_st, out_st = inputs.get("optimizerState"), outputs.get("outOptimizerState")
        a = C.byref(_state_struct(in_st)) if in_st else None
        b = C.byref(_state_struct(out_st)) if out_st else None
        check(self.device.lib.wdgs_densify_prune_encode_scatter(self.handle, pc.num_points, pc.gaussian_3d_buffer.ptr, pc.sh_buffer.ptr, a, out_n,
                                                                1 if inputs.get("resetNewOptimizerState", True) else 0, out_pc.gaussian_3d_buffer.ptr,
                                                                out_pc.sh_buffer.ptr, b))

    def applyActions(self, *_a, **_k):
        raise NotImplementedError("DensifyPrunePass.applyActions is unimplemented in the reference (densify-prune.ts:680-686)")

    def destroy(self) -> None:
        if self.handle:
            self.device.lib.wdgs_densify_prune_destroy(self.handle)
            self.handle = None
