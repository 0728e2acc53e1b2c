"""``Trainer`` -- counterpart of ``/root/reference/src/trainer.ts`` driving the HIP operator classes.

Public surface and sequencing follow the reference (method names as in ``trainer.ts:177-566``):
``step()`` = random view -> camera upload -> forward -> rasterize -> backward -> Adam -> submit+sync -> it/s EMA ->
scheduled ``runDensifyPruneMultiView()`` (``trainer.ts:568-660``, ``373-497``).  Divergences, all host-side and documented in
DESIGN.md: the rasterizer's grid always follows the current viewport (fixes SURVEY Q19); metric views each render with
their own camera (fixes Q12: uploads are stream-ordered here); the point-cloud swap is applied inside the step that
produced it instead of on the next animation frame.  New: ``world_size > 1`` runs view-sharded data parallelism
(``webdgs_amd.parallel``).
"""
from __future__ import annotations

import math
import os
import random
import time
from typing import Optional

import numpy as np
import torch

from . import loaders, ops, parallel


class Trainer:
    def __init__(self, device: ops.HipDevice, trainingConfig: Optional[dict] = None, seed: int = 0, world_size: int = 1, rank: int = 0,
                 views_per_rank: int = 1, maxTileEntries: int = 0, use_command_buffers: bool = True, exchange: Optional[parallel.Exchange] = None,
                 overlap_views: Optional[bool] = None, pipeline_depth: int = 1, batch_views: Optional[bool] = None):
        self.device = device
        self.trainingConfig = dict(trainingConfig or dict(lambda_l1=0.8, lambda_l2=0.0, lambda_dssim=0.2))  # trainer.ts:100-104
        self.optimizerHyperparameters = dict(ops.DEFAULT_ADAM_HYPERPARAMETERS)
        self.world_size, self.rank, self.views_per_rank = int(world_size), int(rank), max(1, int(views_per_rank))
        self.maxTileEntries = int(maxTileEntries)
        self._grown_tile_entries = 0  # (auto sizing only: what an overflow has made of the lists, step())
        self._foreign_overflow_warned = False
        # the transport of the data-parallel exchange (parallel.Exchange); world_size == 1 needs none
        self.exchange = exchange if exchange is not None else (parallel.default_exchange(device, self.world_size) if self.world_size > 1 else parallel.Exchange())
        # sliced step (reduce-scatter / owned-slice Adam / all-gather): any real exchange, also a forced one in a world of one
        self._sliced = self.world_size > 1 or bool(getattr(self.exchange, "force", False))
        if self.world_size > 1 and (self.exchange.world_size, self.exchange.rank) != (self.world_size, self.rank):
            raise ValueError(f"exchange is rank {self.exchange.rank} of {self.exchange.world_size}, trainer is rank {self.rank} of {self.world_size}")
        # Recorded command buffers (HIP graphs), one per view set: the reference re-records its encoder every step; here the
        # recording is kept and re-submitted, because every size the kernels need is read on the device.
        self.use_command_buffers = bool(use_command_buffers)
        self._cmd_cache: dict = {}
        self._eager_steps = 0
        # A batched step deals its views to several op sets in turn; once their command buffers are recorded, set s replays on the
        # device's lane s, so the bandwidth-bound stages of one view (project, sort, loss, geometry backward) run beside the
        # VALU-bound rasterization kernels of another.  Results do not depend on it: the fp32 block is still filled in view order.
        # overlap_views: None = default (WDGS_LANES, else DEFAULT_LANES), False / 1 = one lane, True or n = that many lanes.
        if overlap_views is None:
            overlap_views = int(os.environ.get("WDGS_LANES", self.DEFAULT_LANES))
        elif isinstance(overlap_views, bool):
            overlap_views = self.DEFAULT_LANES if overlap_views else 1
        self._lanes = max(1, min(int(overlap_views), self.views_per_rank, ops.MAX_LANES))
        # View-batched kernels (round 3): a batched step projects every Gaussian for ALL its views in one launch (K1) and turns all the
        # views' accumulators into the step's fp32 gradient block in one launch (K17) -- Gaussian and SH row read once per step instead
        # of once per view, the fp32 block written once, and no cross-lane ordering of per-view K17s.  Every view then needs buffers of
        # its own: one op set per view of the batch (the lanes -- in-order streams -- stay at `_lanes`).  False (WDGS_BATCH_VIEWS=0): the
        # per-view kernels of round 2, one op set per lane.  The results are identical.
        self.batch_views = self.views_per_rank > 1 and (os.environ.get("WDGS_BATCH_VIEWS", "1") != "0" if batch_views is None else bool(batch_views))
        self._op_sets = min(self.views_per_rank, ops.MAX_BATCH_VIEWS) if self.batch_views else self._lanes
        # views per batched launch: 0 = all the step's views in one K1 and one K17 (measured best at c3, 8 views: 1875 views/s; groups of
        # 3 on their own lane, overlapping the other groups' rasterization, 1798: the batched kernels then re-read the cloud per group and
        # take issue slots from the VALU-bound rasterization kernels they run beside -- profiles/r03i_*)
        self.view_group = int(os.environ.get("WDGS_VIEW_GROUP", "0"))
        self._camera_buffers: list = []
        # every rank draws the same view sequence (same seed), then takes its shard
        self._rng = random.Random(seed)

        self.forwardPass = self.rasterizer = self.backwardPass = self.optimizer = None
        # pipeline_depth 1: every step awaits its own completion, as trainer.ts:639-645 does.  2: step() returns once the PREVIOUS
        # step has finished, so the host prepares and submits step k+1 while step k runs (the device never idles across the step
        # boundary); a capacity error then surfaces one step late -- the device-side guard has already kept that step's Adam from
        # running on incomplete gradients.  Densify steps, stop() and every read-back drain the pipeline first.
        self.pipeline_depth = max(1, min(int(pipeline_depth), 4))
        self.reuse_passes = True  # applyPointCloudSwap resizes the passes instead of rebuilding them (False: the reference's teardown)
        self.fuse_geometry_adam = True  # the single-view step runs K17, Adam and the re-pack as one kernel (False: the reference's three)
        self.keep_gradients = os.environ.get("WDGS_KEEP_GRADIENTS", "0") == "1"  # the fused step also fills backwardPass.getGradientsBuffer()
        # Adam writes the trained SH-DC halves to a compact array that K1 reads, instead of 6 bytes into every 96-byte SH row each step; the
        # rows are flushed at hand-over points (Optimizer.setDeferredSH).  False: the reference's write pattern.  Results are identical.
        self.deferred_sh = os.environ.get("WDGS_DEFERRED_SH", "1") != "0"
        self._dc_words: Optional[ops.HipBuffer] = None
        self._gradient_output_applied: Optional[bool] = None
        self._tickets: list = []
        self._more_op_sets: list = []  # [forwardPass, rasterizer, backwardPass] of lanes 1.. (set 0 is the three above)
        self.metricsForwardPass = self.metricsRasterizer = self.metricsPass = None
        self.metricsViewportWidth = self.metricsViewportHeight = 0
        self.metricsTarget: Optional[ops.HipBuffer] = None
        # The metric views of a densify event are independent until normalizeMetricCounts (counts are integer atomics: any order gives the
        # same bits): they are dealt to `metric_lanes` op sets, each on a device lane of its own, all adding into set 0's counts
        # (TiledBackwardPass.setMetricCountsTarget).  The reference walks them one after the other through one pass (trainer.ts:391-430).
        self.metric_lanes = max(1, min(int(os.environ.get("WDGS_METRIC_LANES", self.DEFAULT_LANES)), ops.MAX_LANES))
        self._more_metric_sets: list = []  # [forwardPass, rasterizer, metricsPass, target, cameraBuffer] of metric lanes 1..
        self.pointCloud: Optional[ops.PointCloud] = None
        self.cameraBuffer = device.createBuffer(272, "camera uniform")
        self.metricsCameraBuffer = device.createBuffer(272, "metrics camera uniform")

        self.isTraining = False
        self.iteration = 0
        self.maxIterations = 10_000
        self.stepItersPerSec = 0.0
        self.stepMs = 0.0
        self.lastDensifyPruneIteration: Optional[int] = None
        self.lastViewportWidth = self.lastViewportHeight = 1
        self.pendingPointCloudSwap: Optional[dict] = None
        self.trainCameras: list = []   # dicts: camera (float32[68]), width, height
        self.images: list = []         # dicts: texture (HipBuffer rgba8), width, height

        self.densifyPruneConfig = dict(  # trainer.ts:147-164
            schedule=dict(enabled=True, warmupIterations=500, interval=100, stopIterations=15_000),
            metricViews=10, metricDownscale=2, metricThreshold=0.5, maxBufferBytes=128 * 1024 * 1024, maxNewPointsPerStep=5000,
            pruneOpacity=0.01, cloneThresholdCount=500, splitScaleThreshold=1.0)
        self.densifyPrune = ops.DensifyPrunePass(device, self._densify_op_config())
        # batched / data-parallel step (parallel.py): fp32 gradient block, visibility counts, re-packed rows, guard word
        self._dp_grad: Optional[ops.HipBuffer] = None
        self._dp_visible: Optional[ops.HipBuffer] = None
        self._dp_rows: Optional[ops.HipBuffer] = None
        self._dp_flag: Optional[ops.HipBuffer] = None
        self._agree_word: Optional[ops.HipBuffer] = None
        # long tile lists (csrc/longlist.h): None = the library's defaults; dict(threshold=, maxItems=, maxRows=) for the passes this trainer builds
        self.longLists: Optional[dict] = None
        self._state_sliced = False  # optimizer state of non-owned slices is stale until syncOptimizerState()
        self.exchange_timing = False  # bracket the collectives with events on the device stream (bench.py)
        self._exchange_events: list = []

    DEFAULT_LANES = 3
    _OP_NAMES = ("forwardPass", "rasterizer", "backwardPass", "metricsForwardPass", "metricsRasterizer", "metricsPass", "optimizer")

    def _destroy_more_metric_sets(self) -> None:
        for more in self._more_metric_sets:
            for op in more[:3]:
                op.destroy()
        self._more_metric_sets = []

    def _destroy_more_op_sets(self) -> None:
        for ops_of_lane in self._more_op_sets:
            for op in ops_of_lane:
                op.destroy()
        self._more_op_sets = []

    # ------------------------------------------------------------------ configuration
    def _densify_op_config(self) -> dict:
        c = self.densifyPruneConfig
        return dict(strategy="gpu_rebuild", numViews=c["metricViews"], maxBufferBytes=c["maxBufferBytes"], maxNewPointsPerStep=c["maxNewPointsPerStep"],
                    pruneThreshold=c["pruneOpacity"], cloneThreshold=c["cloneThresholdCount"], splitThreshold=c["splitScaleThreshold"])

    def setPointCloud(self, pointCloud: ops.PointCloud) -> None:
        self.applyPointCloudSwap(dict(pointCloud=pointCloud))

    def requestPointCloudSwap(self, pointCloud: ops.PointCloud, optimizerInitialState: Optional[dict] = None) -> None:
        self.pendingPointCloudSwap = dict(pointCloud=pointCloud, optimizerInitialState=optimizerInitialState)

    def consumePointCloudSwapRequest(self) -> Optional[dict]:
        req, self.pendingPointCloudSwap = self.pendingPointCloudSwap, None
        return req

    def requestResizeTo(self, numPoints: int) -> None:
        if self.pointCloud is None:
            return
        self.requestPointCloudSwap(ops.allocatePointCloudLike(self.device, self.pointCloud, dict(numPoints=numPoints)))

    def applyPointCloudSwap(self, request: dict) -> None:
        """``trainer.ts:201-237``: adopt the new cloud (+ optimizer state) and bring the op graph to its size.  The reference destroys
        every pass and constructs new ones; here the passes are kept and resized (``setPointCloud``: buffers reused, or re-allocated
        with headroom when the cloud outgrew them) -- only the optimizer, which adopts the rebuilt state arrays, is new.  Passes that
        cannot follow (another SH degree) are rebuilt as the reference does."""
        self.drain()
        self._synchronize()
        oldParams = self.optimizer.getHyperparameters() if self.optimizer else None
        if self.optimizer is not None:
            self.optimizer.destroy()
            self.optimizer = None
        old = self.pointCloud
        self.pointCloud = request["pointCloud"]
        self._invalidate_command_buffers()
        passes = [self.forwardPass, self.backwardPass, self.metricsForwardPass, self.metricsPass] + [op for more in self._more_op_sets + self._more_metric_sets for op in (more[0], more[2])]
        kept = self.reuse_passes and old is not None and all(p.setPointCloud(self.pointCloud) for p in passes if p is not None)
        if not kept:
            for name in self._OP_NAMES:
                op = getattr(self, name)
                if op is not None:
                    op.destroy()
                setattr(self, name, None)
            self._destroy_more_op_sets()
            self._destroy_more_metric_sets()
        self.optimizer = ops.Optimizer(self.device, self.pointCloud, oldParams or self.optimizerHyperparameters, request.get("optimizerInitialState"))
        self.optimizerHyperparameters = dict(self.optimizer.getHyperparameters())
        self._dc_words = self.optimizer.setDeferredSH(self.pointCloud, True) if self.deferred_sh else None
        for fw in self._forward_passes():
            fw.setDcSource(self._dc_words)
        if old is not None and old is not self.pointCloud:
            old.gaussian_3d_buffer.destroy()
            old.sh_buffer.destroy()
        self._dp_grad = self._dp_visible = self._dp_rows = self._dp_flag = None
        self._state_sliced = False
        self.ensurePipelines(self.lastViewportWidth, self.lastViewportHeight)

    def _forward_passes(self) -> list:
        return [p for p in [self.forwardPass, self.metricsForwardPass] + [more[0] for more in self._more_op_sets + self._more_metric_sets] if p is not None]

    def flushPointCloud(self) -> None:
        """Brings the point cloud's SH rows up to date with what has been trained (``Optimizer.flushSH``): call before a device-side reader
        that does not go through this trainer's forward passes -- a viewer rendering the same cloud, a custom kernel.  Host reads of
        ``pointCloud.sh_buffer`` do it by themselves."""
        if self.optimizer is not None and self.pointCloud is not None:
            self.optimizer.flushSH(self.pointCloud)

    def _invalidate_command_buffers(self) -> None:
        """Recorded kernels bake pointers, viewport, hyper-parameters and loss weights: any change drops the recordings."""
        deferred = None
        if self._tickets and self.device.handle:  # steps in flight replay these recordings: wait before destroying them
            self._tickets = []
            try:
                self._synchronize()
            except ops.CapacityError as e:  # (reported once the recordings are gone: the caller must still hear of it)
                deferred = e
        for c in self._cmd_cache.values():
            c.destroy()
        self._cmd_cache = {}
        self._eager_steps = 0
        if deferred is not None:
            raise deferred

    def setDataset(self, cameras: list, images: list) -> None:
        """``cameras[i]`` pairs with ``images[i]`` (trainer.ts:575-577).  Accepted shapes: the reference's own --
        ``CameraData`` dicts from ``loaders`` and ``images.LoadedImage`` objects (the camera block is then built for the image
        size as ``Camera.set_preset`` + ``update_buffer`` do, trainer.ts:583-586) -- or ready dicts carrying ``camera`` (68
        floats) and ``width/height/texture``."""
        cameras, images = list(cameras), list(images)
        images = [im if isinstance(im, dict) else dict(name=im.name, width=im.width, height=im.height,
                                                       texture=im.texture if im.texture is not None else self.device.bufferFrom(im.bitmap))
                  for im in images]
        cameras = [c if "camera" in c else dict(c, camera=loaders.cameraUniforms(c, im["width"], im["height"])) for c, im in zip(cameras, images)]
        self.trainCameras, self.images = cameras, images
        # one resident 272-byte camera block per training view (the reference rewrites a single uniform buffer every step)
        self._camera_buffers = [self.device.bufferFrom(np.asarray(c["camera"], np.float32)) for c in self.trainCameras]
        self._invalidate_command_buffers()

    def getTrainingConfig(self) -> dict:
        return dict(self.trainingConfig)

    def setTrainingConfig(self, next_cfg: dict) -> None:
        self.trainingConfig.update({k: v for k, v in next_cfg.items() if v is not None})
        self._invalidate_command_buffers()
        for p in [self.backwardPass, self.metricsPass] + [more[2] for more in self._more_op_sets + self._more_metric_sets]:
            if p is not None:
                p.setTrainingConfig(next_cfg)

    def getOptimizerHyperparameters(self) -> dict:
        return dict(self.optimizer.getHyperparameters() if self.optimizer else self.optimizerHyperparameters)

    def setOptimizerHyperparameters(self, next_params: dict) -> None:
        self.optimizerHyperparameters.update(next_params)
        self._invalidate_command_buffers()
        if self.optimizer:
            self.optimizer.setHyperparameters(next_params)

    def setDensifyPruneConfig(self, next_cfg: dict) -> None:
        sched = {**self.densifyPruneConfig["schedule"], **(next_cfg.get("schedule") or {})}
        self.densifyPruneConfig = {**self.densifyPruneConfig, **next_cfg, "schedule": sched}
        self.densifyPrune.setConfig(self._densify_op_config())

    def start(self) -> None:
        if self.pointCloud is None or not self.trainCameras:
            print("Cannot start training: Missing point cloud or dataset.")
            return
        self.isTraining = True
        self.iteration = 0
        self.stepItersPerSec = 0.0
        self.stepMs = 0.0
        self.lastDensifyPruneIteration = None
        # the lanes this trainer will use come into being now (the library creates a lane's stream and event at its first use, ~5 ms each:
        # otherwise the first densify event, which is the first to touch the metric lanes, pays for them -- profiles/r08t_event_timing.txt)
        for k in range(1, max(self._lanes, self.metric_lanes)):
            self.device.laneOrder(k, 0)
            self.device.laneOrder(0, k)

    def stop(self) -> None:
        self.isTraining = False
        if self.device.handle and self._tickets:
            self.drain()

    def getIsTraining(self) -> bool:
        return self.isTraining

    def setMaxIterations(self, n: int) -> None:
        self.maxIterations = max(1, int(n))

    def getMaxIterations(self) -> int:
        return self.maxIterations

    def getIteration(self) -> int:
        return self.iteration

    def getPointCount(self) -> int:
        return self.pointCloud.num_points if self.pointCloud else 0

    def getLastStepMs(self) -> float:
        return self.stepMs

    def getItersPerSec(self) -> float:
        return self.stepItersPerSec

    def getLastDensifyPruneIteration(self) -> Optional[int]:
        return self.lastDensifyPruneIteration

    def getNextDensifyPruneIteration(self) -> Optional[int]:
        s = self.densifyPruneConfig["schedule"]
        if not s["enabled"]:
            return None
        warmup, interval, stop, i = s["warmupIterations"], max(1, s["interval"]), s["stopIterations"], self.iteration
        if i >= stop:
            return None
        if i < warmup:
            return min(warmup, stop)
        nxt = warmup + math.ceil((i + 1 - warmup) / interval) * interval
        return nxt if nxt <= stop else None

    # ------------------------------------------------------------------ pipelines
    def _new_forward_pass(self, cameraBuffer, w: int, h: int) -> ops.TiledForwardPass:
        fw = ops.TiledForwardPass(self.device, self.pointCloud, cameraBuffer, dict(viewportWidth=w, viewportHeight=h, renderMode="gaussian", maxTileEntries=self._tile_entries()))
        fw.setDcSource(self._dc_words)
        if self.longLists is not None:   # (None: the library's defaults -- threshold 2048, room for 1024 chunk slots and 8192 rows)
            fw.setLongLists(int(self.longLists.get("threshold", 2048)), int(self.longLists.get("maxItems", 0)), int(self.longLists.get("maxRows", 0)))
        return fw

    def _grow_long_lists(self) -> None:
        """Long tile lists (csrc/longlist.h) work in scratch of a fixed size; a frame whose long tiles find no room is composited the ordinary way --
        correct, but as slow as its longest list.  The work's header says what the last frame wanted: looked at where the host waits anyway (a densify
        event), and every pass is given room for 1.5 x that -- up to ``longLists.maxItemsCap`` / ``maxRowsCap`` (8 192 chunk slots: 33 000 entries of
        long tiles; 65 536 rows).  A frame that wants more than the caps is FULL of long tiles (a dense cloud at a small viewport): the path is not for
        it (longlist.h: ll_frame_on) and the scratch is left alone.  (Command buffers recorded against the old scratch are dropped.)"""
        import warnings
        cap_items, cap_rows = int((self.longLists or {}).get("maxItemsCap", 8192)), int((self.longLists or {}).get("maxRowsCap", 65536))
        have_items = have_rows = need_items = need_rows = 0
        for fw in self._forward_passes():
            st = fw.longListStats()
            if not st["threshold"]:
                continue
            if st["stalled"]:
                warnings.warn(f"a long-list task gave up waiting (code {st['stalled']:#x}): the frame's long tiles are not to be trusted", RuntimeWarning, stacklevel=3)
            have_items, have_rows = max(have_items, st["maxItems"]), max(have_rows, st["maxRows"])
            if st["itemsWanted"] <= cap_items:
                need_items, need_rows = max(need_items, st["itemsWanted"]), max(need_rows, st["rowsWanted"])
        items = max(have_items, min(int(need_items * 1.5), cap_items)) if need_items > have_items else have_items
        rows = max(have_rows, min(int(need_rows * 1.5), cap_rows)) if need_rows > have_rows else have_rows
        if (items, rows) == (have_items, have_rows):
            return
        self.longLists = dict(self.longLists or {}, maxItems=items, maxRows=rows)
        warnings.warn(f"long-list scratch enlarged to {self.longLists['maxItems']} chunk slots and {self.longLists['maxRows']} rows", RuntimeWarning, stacklevel=3)
        self._invalidate_command_buffers()
        for fw in self._forward_passes():
            fw.setLongLists(int(self.longLists.get("threshold", 2048)), self.longLists["maxItems"], self.longLists["maxRows"])

    def ensurePipelines(self, width: int, height: int) -> None:
        if (max(1, int(width)), max(1, int(height))) != (self.lastViewportWidth, self.lastViewportHeight):
            self._invalidate_command_buffers()
        self.lastViewportWidth, self.lastViewportHeight = max(1, int(width)), max(1, int(height))
        w, h = self.lastViewportWidth, self.lastViewportHeight
        if self.forwardPass is None:
            self.forwardPass = self._new_forward_pass(self.cameraBuffer, w, h)
        else:
            self.forwardPass.setViewport(w, h)
        if self.rasterizer is None:
            self.rasterizer = ops.TiledRasterizer(dict(device=self.device, forwardPass=self.forwardPass, format="rgba8unorm"))
        if self.backwardPass is None:
            self.backwardPass = ops.TiledBackwardPass(self.device, self.pointCloud, dict(viewportWidth=w, viewportHeight=h, trainingConfig=self.trainingConfig))
            self._gradient_output_applied = None  # (a fresh pass writes the packed gradient, the C ABI's default: _apply_gradient_output decides)
        else:
            self.backwardPass.setViewport(w, h)
        for more in self._more_op_sets:
            more[0].setViewport(w, h)
            more[2].setViewport(w, h)
        while len(self._more_op_sets) < self._op_sets - 1:
            fw = self._new_forward_pass(self.cameraBuffer, w, h)
            self._more_op_sets.append([fw, ops.TiledRasterizer(dict(device=self.device, forwardPass=fw, format="rgba8unorm")),
                                       ops.TiledBackwardPass(self.device, self.pointCloud, dict(viewportWidth=w, viewportHeight=h, trainingConfig=self.trainingConfig))])
        if self.optimizer is not None and self.world_size * self.views_per_rank == 1:
            # a step whose tile-entry list overflowed is skipped on the device (and reported by the next synchronize)
            self.optimizer.setGuard(self.forwardPass.getStatsBuffer(), 8)

    def ensureMetricsPipelines(self, baseWidth: int, baseHeight: int) -> tuple[int, int]:
        down = max(1, int(self.densifyPruneConfig["metricDownscale"]))
        w, h = max(1, baseWidth // down), max(1, baseHeight // down)
        if self.metricsForwardPass and self.metricsViewportWidth == w and self.metricsViewportHeight == h:
            return w, h
        for name in ("metricsForwardPass", "metricsRasterizer", "metricsPass"):
            op = getattr(self, name)
            if op is not None:
                op.destroy()
            setattr(self, name, None)
        self._destroy_more_metric_sets()
        self.metricsViewportWidth, self.metricsViewportHeight = w, h
        self.metricsForwardPass = self._new_forward_pass(self.metricsCameraBuffer, w, h)
        self.metricsRasterizer = ops.TiledRasterizer(dict(device=self.device, forwardPass=self.metricsForwardPass, format="rgba8unorm"))
        self.metricsPass = ops.TiledBackwardPass(self.device, self.pointCloud, dict(viewportWidth=w, viewportHeight=h, trainingConfig=self.trainingConfig))
        self.metricsTarget = self.device.createBuffer(4 * w * h, "metrics-gt-downsampled")
        return w, h

    def _metric_set(self, k: int) -> tuple:
        """(forwardPass, rasterizer, metricsPass, downsampled-GT buffer, camera buffer) of metric lane ``k``; sets 1.. are built on first use."""
        if k == 0:
            return self.metricsForwardPass, self.metricsRasterizer, self.metricsPass, self.metricsTarget, self.metricsCameraBuffer
        w, h = self.metricsViewportWidth, self.metricsViewportHeight
        while len(self._more_metric_sets) < k:
            cam = self.device.createBuffer(272, "metrics camera uniform")
            fw = self._new_forward_pass(cam, w, h)
            self._more_metric_sets.append([fw, ops.TiledRasterizer(dict(device=self.device, forwardPass=fw, format="rgba8unorm")),
                                           ops.TiledBackwardPass(self.device, self.pointCloud, dict(viewportWidth=w, viewportHeight=h, trainingConfig=self.trainingConfig)),
                                           self.device.createBuffer(4 * w * h, "metrics-gt-downsampled"), cam])
        return tuple(self._more_metric_sets[k - 1])

    @staticmethod
    def metrics_camera(camera: np.ndarray, width: int, height: int) -> np.ndarray:
        """The metrics camera of ``trainer.ts:399-401``: ``set_preset`` keeps the view's pose and derives
        ``fovY = 2 atan(height / (2 fy))`` from the view's own height and fy (``camera.ts:196-205``), ``on_update_canvas`` turns
        it into ``focal = 0.5 canvasHeight / tan(fovY / 2)`` for the metrics canvas (``camera.ts:138-147``) and ``update_buffer``
        rebuilds projection and inverses.  The atan/tan round trip is kept: it is what the reference evaluates."""
        from . import synth
        cam = np.asarray(camera, np.float32)
        fov_y = 2.0 * math.atan(float(cam[65]) / (2.0 * float(cam[67])))
        focal = 0.5 * height / math.tan(fov_y * 0.5)
        fov_x = 2.0 * math.atan(width / (2.0 * focal))
        znear, zfar = 0.01, 100.0
        top, right = math.tan(fov_y / 2.0) * znear, math.tan(fov_x / 2.0) * znear
        out = np.zeros(68, np.float32)
        out[0:16] = cam[0:16]
        out[32], out[37] = 2.0 * znear / (2.0 * right), -2.0 * znear / (2.0 * top)   # camera.ts:29-56, column-major
        out[42], out[43], out[46] = zfar / (zfar - znear), 1.0, -(zfar * znear) / (zfar - znear)
        out[16:32] = synth.mat4_inverse(out[0:16])
        out[48:64] = synth.mat4_inverse(out[32:48])
        out[64:68] = (width, height, focal, focal)
        return out

    # ------------------------------------------------------------------ one training step
    def _ops_of(self, op_set: int) -> tuple:
        return tuple(self._more_op_sets[op_set - 1]) if op_set > 0 else (self.forwardPass, self.rasterizer, self.backwardPass)

    def _encode_view(self, encoder, index: int, op_set: int = 0, geometry: bool = True, projected: bool = False) -> None:
        forwardPass, rasterizer, backwardPass = self._ops_of(op_set)
        image = self.images[index]
        cam = self._camera_buffers[index]  # camera.set_preset + update_buffer (trainer.ts:583-586): the view's resident block
        forwardPass.setCameraBuffer(cam)
        if projected:   # K1 ran for all the views of the step at once (ops.projectViews): scan, emit, sort remain
            forwardPass.encodeProjected(encoder)
        else:
            forwardPass.encode(encoder)
        rasterizer.encode(encoder, image["width"], image["height"])
        res = dict(splatBuffer=forwardPass.getResources()["splatBuffer"], tileOffsetsBuffer=rasterizer.getTileOffsetsBuffer(),
                   tileIndicesBuffer=forwardPass.getSortedIndicesBuffer(), cameraBuffer=cam,
                   alphaTexture=rasterizer.getAlphaTextureView(), nContribTexture=rasterizer.getNContribTextureView())
        if geometry:
            backwardPass.encode(encoder, rasterizer.getOutputTextureView(), image["texture"], res)
        else:  # a batched step: K17 follows separately, in view order, and adds to the step's fp32 block (_step_batched)
            backwardPass.encodeRaster(encoder, rasterizer.getOutputTextureView(), image["texture"], res)

    def warmupCommandBuffers(self) -> int:
        """Records every view's command buffers up front (the first pass over a dataset does this anyway; calling it before a timed
        region keeps recording out of it).  The number of steps taken is a function of the dataset size ONLY -- never of which
        buffers this rank happens to hold already: under data parallelism every step is a collective, so all ranks must take the same
        number of them (a rank-local "skip what is recorded" here once left one rank in the exchange and its peer at the barrier).
        Two passes: the first step of a fresh trainer runs eagerly (first-use allocations), any later one records."""
        if not self.use_command_buffers or not self.isTraining or self.pointCloud is None:
            return 0
        n_views, taken = self.world_size * self.views_per_rank, 0
        for v in range(len(self.trainCameras)):
            for _ in range(2 if v == 0 else 1):
                self.step([v] * n_views)
                taken += 1
        return taken

    def step(self, view_ids: Optional[list] = None) -> None:
        """One training iteration (trainer.ts:568-660).  ``view_ids``: the global batch's views (default: drawn at random, as the
        reference picks ``Math.random()`` per step); with ``world_size > 1`` each rank takes its shard.

        Tile-entry capacity: with ``maxTileEntries`` left at 0 the forward passes size their entry lists from the cloud (30 entries per
        Gaussian, at least 2^20) -- a cloud that training has thinned out and whose survivors have grown can outrun that (c3 does, after
        ~3 000 iterations of the default schedule).  The reference truncates such a list silently; the library skips the step on the device
        and reports it.  The Trainer then doubles the lists (``_grow_tile_entry_capacity``), warns, and training goes on -- the step or two
        that were skipped are lost iterations.  A capacity the caller pinned is never touched: the error is the caller's."""
        try:
            self._step(view_ids)
        except ops.CapacityError as e:
            if not self._grow_tile_entry_capacity(e):
                raise

    def _tile_entries(self) -> int:
        """``maxTileEntries`` for a new forward pass: the caller's, or what ``_grow_tile_entry_capacity`` has arrived at (0 = the library's own sizing)."""
        return self.maxTileEntries or self._grown_tile_entries

    def _grow_tile_entry_capacity(self, error) -> bool:
        import re
        import warnings
        if self.maxTileEntries != 0 or not self.device.handle:
            return False
        mine = self._own_overflow(error) or []   # (a report about other owners' passes only never gets here: _wait / _synchronize)
        now = max([int(fw.getResources()["maxTileEntries"]) for fw in self._forward_passes()] + [self._grown_tile_entries, 1 << 20])
        new = min(max(2 * now, int(max(mine) * 1.5) if mine else 0), 0xFFFFF000)
        if new <= now:
            return False
        warnings.warn(f"tile-entry lists grown from {now} to {new} entries after an overflow ({error}); the step that overflowed was skipped", RuntimeWarning, stacklevel=3)
        self._grown_tile_entries = new
        self._tickets = []
        try:
            self.device.synchronize()
        except ops.CapacityError:
            pass  # (a step still in flight overflowed as well)
        self._invalidate_command_buffers()
        # forward passes own the lists: every pass set is rebuilt around lists of the new size (as a cloud the passes cannot follow rebuilds them)
        for name in self._OP_NAMES:
            if name != "optimizer" and getattr(self, name) is not None:
                getattr(self, name).destroy()
                setattr(self, name, None)
        self._destroy_more_op_sets()
        self._destroy_more_metric_sets()
        self._gradient_output_applied = None
        self.ensurePipelines(self.lastViewportWidth, self.lastViewportHeight)
        for fw in self._forward_passes():
            fw.setDcSource(self._dc_words)
        return True

    def _step(self, view_ids: Optional[list] = None) -> None:
        if not self.isTraining or self.pointCloud is None:
            return
        stepStart = time.perf_counter()
        n_views = self.world_size * self.views_per_rank
        if view_ids is None:
            view_ids = [self._rng.randrange(len(self.trainCameras)) for _ in range(n_views)]
        mine = parallel.shard_views(view_ids, self.rank, self.world_size)
        image0 = self.images[mine[0]]
        self.ensurePipelines(image0["width"], image0["height"])

        s = self.densifyPruneConfig["schedule"]
        nextIteration = self.iteration + 1
        warmup, interval, stop = s["warmupIterations"], max(1, s["interval"]), s["stopIterations"]
        shouldDensify = s["enabled"] and warmup <= nextIteration <= stop and (nextIteration == warmup or (nextIteration - warmup) % interval == 0)

        try:
            if n_views == 1:
                self._step_single_view(mine[0])
            else:
                self._step_batched(mine)
        except BaseException as orig:
            # a failed encode (capacity, first-use allocation inside a recording, a Python error) must not leave the stream in
            # capture mode or half-recorded command buffers behind: drop the recording and fall back to a clean eager state
            if self.device.handle:
                self.device.lib.wdgs_encoder_abort(self.device.handle)
            try:
                self._invalidate_command_buffers()  # (waits for the steps still in flight before it destroys what they replay)
            except ops.CapacityError as deferred:
                # a step still in flight reported its own (deferred) overflow while the recordings were dropped: this step's error
                # stays the one raised, the deferred one rides along as its cause
                raise orig from deferred
            finally:
                self._tickets = []
            raise
        try:
            self._finish_step(n_views)
        except BaseException:
            self._tickets = []  # (the error has been consumed; whatever is still in flight is awaited by the next synchronize)
            raise

        self.iteration += 1
        self.stepMs = (time.perf_counter() - stepStart) * 1000.0
        inst = 1000.0 / self.stepMs if self.stepMs > 0 else 0.0
        self.stepItersPerSec = inst if self.stepItersPerSec == 0 else self.stepItersPerSec * 0.9 + inst * 0.1
        if shouldDensify:
            self.drain()
            self._grow_long_lists()
            self.runDensifyPruneMultiView()
            req = self.consumePointCloudSwapRequest()
            if req is not None:
                self.applyPointCloudSwap(req)
        if self.iteration >= self.maxIterations:
            self.stop()
            self.drain()

    def _run(self, key: tuple, encode) -> bool:
        """Submits the command buffer recorded under ``key``; the first time, ``encode(encoder)`` is encoded -- eagerly while the
        pipelines still allocate on first use, into a recorded command buffer (HIP graph) afterwards.  True if it was replayed."""
        cmd = self._cmd_cache.get(key)
        if cmd is not None:
            self.device.queue.submit([cmd])
            return True
        record = self.use_command_buffers and self._eager_steps >= 1
        with self.device.createCommandEncoder("trainer-" + str(key[0]), record=record) as encoder:
            encode(encoder)
            cmd = encoder.finish()
        if record:
            self._cmd_cache[key] = cmd
        self.device.queue.submit([cmd])
        return False

    def _apply_gradient_output(self) -> None:
        """The fused K17 + Adam step keeps the gradient in registers; nothing in the trainer reads the packed copy (trainer.ts hands it to
        ``optimizer.step()`` only), so the step does not write it unless asked to.  ``keep_gradients`` and ``fuse_geometry_adam`` are read
        HERE, at every step, so a host that flips them after ``setPointCloud`` is obeyed (a recorded fused step baked the old choice: the
        recordings are dropped on a change) -- ``backwardPass.getGradientsBuffer()`` never silently holds stale data."""
        want = bool(self.keep_gradients or not self.fuse_geometry_adam)
        if self._gradient_output_applied is want:
            return
        if self._gradient_output_applied is not None:
            self._invalidate_command_buffers()
        self.backwardPass.setGradientOutput(want)
        self._gradient_output_applied = want

    def _step_single_view(self, view: int) -> None:
        """The reference's step (trainer.ts:603-645): one view, Adam straight from the packed fp16 gradients."""
        self._apply_gradient_output()
        tileCounts = self.forwardPass.getResources()["tileCountsBuffer"]

        def encode(encoder):
            if self.fuse_geometry_adam:  # K1..K16, then K17 + Adam + re-pack in one pass over the Gaussians
                self._encode_view(encoder, view, geometry=False)
                self.optimizer.stepWithGeometry(encoder, self.pointCloud, self.backwardPass, self._camera_buffers[view], tileCounts)
            else:
                self._encode_view(encoder, view)
                self.optimizer.step(encoder, self.pointCloud, self.backwardPass.getGradientsBuffer(), tileCounts)
        if self._run(("step", view), encode):
            self.optimizer.advanceIteration(1)
        elif not self.use_command_buffers or self._eager_steps < 1:
            self._eager_steps += 1

    def _step_batched(self, mine: list) -> None:
        """[views -> fp32 block] -> exchange -> [Adam on the owned slice] -> all-gather -> [apply the other ranks' rows].  One
        recorded command buffer per (view, op set) -- K1..K16 -- plus one each for Adam and apply, whatever the batch's composition;
        K17 is launched eagerly per view, ordered across the lanes, and adds the view's gradient to the fp32 block itself."""
        n, w = self.pointCloud.num_points, self.world_size
        sl = parallel.slice_points(n, w)
        if self._dp_grad is None:  # (allocated before any recording is opened; sized world*slice so the collectives run in place)
            self._dp_grad = self.device.createBuffer(4 * parallel.GRAD_FLOATS * w * sl, "dp-grad-f32")
            self._dp_visible = self.device.createBuffer(4 * w * sl, "dp-visible")
            self._dp_flag = self.device.createBuffer(16, "dp-guard")
            self._dp_rows = self.device.createBuffer(32 * w * sl, "dp-repacked-rows") if self._sliced else None
            self.optimizer.setGuard(self._dp_flag, 0)
        first, count = parallel.owned_range(n, w, self.rank)
        eager_before = self._eager_steps
        dev = self.device
        if self.batch_views and len(mine) <= self._op_sets:
            self._views_batched(mine)
        else:
            self._views_one_by_one(mine)
        self._timed(lambda: self.exchange.exchange_gradients(self._dp_grad.ptr, self._dp_visible.ptr, self._dp_flag.ptr, sl))
        if self._run(("adam",), lambda encoder: self.optimizer.stepF32Range(encoder, self.pointCloud, self._dp_grad, self._dp_visible, first, count, self._dp_rows)):
            self.optimizer.advanceIteration(1)
        if self._sliced:
            self._timed(lambda: self.exchange.allgather_rows(self._dp_rows.ptr, sl))
            self._run(("apply",), lambda encoder: self.optimizer.applyRepackedRows(self._dp_rows, first, count, self._dp_flag, self.pointCloud))
            self._state_sliced = w > 1
        if not self.use_command_buffers or eager_before < 1:
            self._eager_steps += 1

    def _views_batched(self, mine: list) -> None:
        """The step's views in groups (`view_group`; default: one group = all of them): [K1 of the group, one launch] -> per view, dealt to the lanes: scan, emit, sort,
        composite, loss, backward raster (one recorded command buffer per (view, place in the batch)) -> [K17 of the group, one launch,
        into the step's fp32 block, groups in view order].  The batched launches run on a lane of their own, so a group's projection and
        the previous group's K17 -- bandwidth- and latency-bound -- execute beside the other groups' rasterization kernels; each view
        waits only for its group's K1, each K17 only for its group's views."""
        dev, L = self.device, self._lanes
        sets = [self._ops_of(k) for k in range(len(mine))]
        cams = [self._camera_buffers[v] for v in mine]
        lanes = L > 1 and self.use_command_buffers and all(("viewp", v, k) in self._cmd_cache for k, v in enumerate(mine))
        U = L if L < ops.MAX_LANES else 0          # the lane of the batched launches
        G = self.view_group or len(mine)
        groups = [list(range(g, min(g + G, len(mine)))) for g in range(0, len(mine), G)]
        try:
            if lanes:
                for s in range(1, ops.MAX_LANES if U else L):
                    dev.laneOrder(s, 0)  # every lane starts behind whatever lane 0 holds (the previous step's Adam, a densify rebuild)
            for gi, grp in enumerate(groups):   # every group's projection first, back to back on the batched launches' lane
                if lanes:
                    dev.selectLane(U)
                ops.projectViews([sets[k][0] for k in grp], [cams[k] for k in grp], self.pointCloud)
                if lanes:
                    dev.laneMark(U, gi)
            for gi, grp in enumerate(groups):
                for k in grp:
                    if lanes:
                        dev.laneWaitMark(k % L, gi)  # view k starts behind its OWN group's projection
                        dev.selectLane(k % L)
                    self._run(("viewp", mine[k], k), lambda encoder, v=mine[k], k=k: self._encode_view(encoder, v, k, geometry=False, projected=True))
                    if lanes:
                        dev.laneOrder(U, k % L)      # the group's K17 follows its views (and the earlier groups' K17: lane order)
                if lanes:
                    dev.selectLane(U)
                ops.geometryViews([sets[k][2] for k in grp], [cams[k] for k in grp], [sets[k][0] for k in grp], self._dp_grad, self._dp_visible, self._dp_flag,
                                  self.pointCloud, continues=gi > 0)
        finally:
            if lanes:
                dev.lib.wdgs_encoder_abort(dev.handle)  # (a no-op unless an encode above failed mid-recording)
                dev.selectLane(0)
                for s in range(1, ops.MAX_LANES if U else L):
                    dev.laneOrder(0, s)  # join: the exchange and the optimizer step follow every lane

    def _views_one_by_one(self, mine: list) -> None:
        """Round 2's form: per view K1..K16 recorded, K17 eager per view, ordered across the lanes (one op set per lane)."""
        dev = self.device
        # lanes carry replays only: a step that still encodes eagerly (first-use allocations) or records keeps to lane 0
        L = min(self._lanes, self._op_sets)
        lanes = L > 1 and self.use_command_buffers and all(("view", v, k % L) in self._cmd_cache for k, v in enumerate(mine))
        try:
            for s in range(1, L if lanes else 0):
                dev.laneOrder(s, 0)  # every lane starts behind whatever lane 0 holds (the previous step's Adam, a densify rebuild)
            for k, v in enumerate(mine):
                s = k % L
                forwardPass, _, backwardPass = self._ops_of(s)
                if lanes:
                    dev.selectLane(s)
                self._run(("view", v, s), lambda encoder, v=v, s=s: self._encode_view(encoder, v, s, geometry=False))
                if lanes and k > 0:
                    dev.laneOrder(s, (k - 1) % L)  # the fp32 block is filled in view order: this view's sums follow the previous view's
                # K17, outside the recording (its accumulate target depends on the view's place in the batch): the first view
                # overwrites the fp32 block (no clearing pass), the others add to it; the guard word collects the views' overflow words
                backwardPass.encodeGeometry(None, self._camera_buffers[v], dict(sums=self._dp_grad, visible=self._dp_visible, first=(k == 0),
                                                                                tileCounts=forwardPass.getResources()["tileCountsBuffer"],
                                                                                guard=self._dp_flag, stats=forwardPass.getStatsBuffer()))
        finally:
            if lanes:
                dev.lib.wdgs_encoder_abort(dev.handle)  # (a no-op unless an encode above failed mid-recording)
                dev.selectLane(0)
                for s in range(1, L):
                    dev.laneOrder(0, s)  # join: the exchange and the optimizer step follow every lane

    def _timed(self, collective) -> None:
        if not self.exchange_timing:
            collective()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(self.device.torch_stream)
        collective()
        e1.record(self.device.torch_stream)
        self._exchange_events.append((e0, e1))

    def exchangeMilliseconds(self) -> float:
        """Device time spent between the events bracketing the collectives since ``exchange_timing`` was switched on (synchronises)."""
        self.device.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self._exchange_events)
        self._exchange_events = []
        return float(ms)

    def _metric_overflow(self):
        """The capacity report of this rank's metric forward passes (their sticky words are consumed), or None.  Synchronises."""
        found = None
        for fw in [self.metricsForwardPass] + [more[0] for more in self._more_metric_sets]:
            if fw is None:
                continue
            try:
                fw.check()
            except ops.CapacityError as e:
                found = found or e
        return found

    def _agree(self, flag: bool) -> bool:
        """True on every rank if ``flag`` is true on any: one u32 summed over the ranks on the device, read back (a host wait -- only used inside a
        densify event, which reads the rebuilt cloud's size back anyway)."""
        if self.world_size <= 1:
            return bool(flag)
        if self._agree_word is None:
            self._agree_word = self.device.createBuffer(4, "agreement word")
        self._agree_word.write(np.array([1 if flag else 0], np.uint32))
        self.exchange.allreduce_counts(self._agree_word.ptr, 1)
        return int(self._agree_word.read(np.uint32, 1)[0]) != 0

    def _own_overflow(self, error) -> Optional[list]:
        """The entries needed by THIS trainer's passes among those a capacity report names (csrc/api.hip: deferred_checks names every pass that
        overflowed), ``[]`` if it names only other owners' passes, ``None`` if it names none (a step skipped on every rank)."""
        import re
        named = re.findall(r"(\d+) entries needed, max_tile_entries = \d+ \(forward pass (0x[0-9a-fA-F]+)\)", str(error))
        if not named:
            return None
        own = {int(fw.handle.value or 0) for fw in self._forward_passes()}
        return [int(n) for n, h in named if int(h, 16) in own]

    def _not_ours(self, error) -> bool:
        """True (after saying so once) for a capacity report about passes this trainer does not own -- a Viewer rendering the same cloud on this
        device: the report is device-wide, whoever waits first gets it, and it is the pass's owner who has to enlarge its lists."""
        if self._own_overflow(error) != []:
            return False
        self.device.capacityReports.post(error)   # (for the passes' owner: it looks there at its own next wait)
        if not self._foreign_overflow_warned:
            import warnings
            warnings.warn(f"a forward pass that is not this trainer's overflowed its tile-entry lists ({error}); its owner has to enlarge them", RuntimeWarning, stacklevel=4)
            self._foreign_overflow_warned = True
        return True

    def _wait(self, ticket) -> None:
        try:
            self.device.queue.wait(ticket)
        except ops.CapacityError as e:
            if not self._not_ours(e):
                raise
        self._reports_left_for_us()

    def _synchronize(self) -> None:
        try:
            self.device.synchronize()
        except ops.CapacityError as e:
            if not self._not_ours(e):
                raise
        self._reports_left_for_us()

    def _reports_left_for_us(self) -> None:
        """A report about THIS trainer's passes that another owner's wait consumed (a Viewer reading its frame): raised here, as if this wait had got it."""
        if self.device.capacityReports.pending:
            e = self.device.capacityReports.take(int(fw.handle.value or 0) for fw in self._forward_passes())
            if e is not None:
                raise e

    def _finish_step(self, n_views: int) -> None:
        """``await onSubmittedWorkDone()`` (trainer.ts:639-645) + the deferred capacity check, for the step ``pipeline_depth - 1``
        submissions ago.  In a batched step the guard word was summed over all ranks by the exchange and the optimizer kernels leave
        a host-visible note when they skip themselves, so every rank raises at the same step."""
        self._tickets.append(self.device.queue.mark())
        while len(self._tickets) >= self.pipeline_depth:
            self._wait(self._tickets.pop(0))

    def drain(self) -> None:
        """Awaits every step still in flight (a no-op at ``pipeline_depth`` 1)."""
        tickets, self._tickets = self._tickets, []
        for t in tickets:
            self._wait(t)

    def syncOptimizerState(self) -> None:
        """Brings every rank's optimizer state up to date: after sliced steps a rank holds current (param, m, v) only for the
        Gaussians it owns; each owner broadcasts its slice of the six state arrays.  A no-op on one rank.  Called before a
        densify rebuild; call it before reading ``optimizer.getStateBuffers()`` in a multi-rank run."""
        if not self._state_sliced or self.world_size <= 1:
            return
        n, w = self.pointCloud.num_points, self.world_size
        bufs = self.optimizer.getStateBuffers()  # flushes the compact SH-DC copy into paramSH / stateSH first
        rows = dict(optPosBuffer=48, optRotBuffer=48, optScaleBuffer=48, optOpacityBuffer=12, paramSH=192, stateSH=384)
        for root in range(w):
            first, count = parallel.owned_range(n, w, root)
            for k, rb in rows.items():
                self.exchange.broadcast(bufs[k].ptr + first * rb, count * rb, root)
        self.optimizer.stateChanged()
        self._state_sliced = False

    def destroy(self) -> None:
        """Deterministic teardown: command buffers, then every op, then the buffers this trainer allocated.  The device itself
        belongs to the caller (``HipDevice.destroy()`` comes after this, and before ``torch.distributed.destroy_process_group``)."""
        if self.device.handle:
            self.device.lib.wdgs_encoder_abort(self.device.handle)
            self._tickets = []
            try:
                self.device.synchronize()
            except ops.CapacityError:
                pass  # (a deferred report about a step of a trainer that is going away)
        self._invalidate_command_buffers()
        for name in self._OP_NAMES + ("densifyPrune",):
            op = getattr(self, name, None)
            if op is not None:
                op.destroy()
            setattr(self, name, None)
        self._destroy_more_op_sets()
        self._destroy_more_metric_sets()
        self._dp_grad = self._dp_visible = self._dp_rows = self._dp_flag = self.metricsTarget = None
        self._camera_buffers = []
        self.pointCloud = None
        self.isTraining = False

    # ------------------------------------------------------------------ densify / prune
    def runDensifyPruneMultiView(self) -> None:
        if self.pointCloud is None or self.optimizer is None or not self.trainCameras or not self.images:
            return
        baseW, baseH = self.lastViewportWidth, self.lastViewportHeight
        mW, mH = self.ensureMetricsPipelines(baseW, baseH)
        c = self.densifyPruneConfig
        viewsTarget = max(1, int(c["metricViews"]))
        encoder = self.device.createCommandEncoder("densify-prune multiview metrics")
        counts = self.metricsPass.getMetricCountsBuffer()
        encoder.clearBuffer(counts)
        dev, L = self.device, self.metric_lanes
        usedViews, attempts, taken = 0, 0, 0
        try:
            while attempts < viewsTarget * 4 and usedViews < viewsTarget:
                attempts += 1
                idx = self._rng.randrange(len(self.trainCameras))
                camData, image = self.trainCameras[idx], self.images[idx]
                if image["width"] != baseW or image["height"] != baseH:
                    continue
                # every rank walks the same view list; the work is sharded round-robin and the counts are all-reduced below
                take = (usedViews % self.world_size) == self.rank
                usedViews += 1
                if not take:
                    continue
                k = taken % L   # this rank's views in turn on its metric lanes; every lane's pass adds into set 0's counts
                taken += 1
                fw, rast, mpass, target, cam = self._metric_set(k)
                if k > 0:
                    if taken <= L:   # the lane's first view of this event: behind the clear (and whatever else lane 0 holds)
                        mpass.setMetricCountsTarget(counts)
                        dev.laneOrder(k, 0)
                    dev.selectLane(k)
                cam.write(self.metrics_camera(camData["camera"], mW, mH))
                fw.encode(encoder)
                rast.encode(encoder, mW, mH)
                ops.downsampleRGBA8(dev, image["texture"], baseW, baseH, target, mW, mH)
                mpass.computeMetricMap(encoder, rast.getOutputTextureView(), target, dict(threshold=c["metricThreshold"]))
                mpass.computeMetricCounts(encoder, dict(splatBuffer=fw.getResources()["splatBuffer"], tileOffsetsBuffer=rast.getTileOffsetsBuffer(),
                                                       tileIndicesBuffer=fw.getSortedIndicesBuffer(), nContribTexture=rast.getNContribTextureView()), dict(clear=False))
                if k > 0:
                    dev.selectLane(0)
        finally:
            dev.selectLane(0)
            for k in range(1, min(L, taken)):
                dev.laneOrder(0, k)   # join: normalize / prepare / the exchange follow every lane
        if usedViews == 0:
            return
        if self.world_size > 1:
            # A metric pass whose tile-entry list overflowed has counted a truncated view: the event is void -- and it has to be void on EVERY rank
            # (the views are sharded, so one rank alone may overflow; were it to bail out while its peers rebuild the cloud, the replicas
            # would part and the next exchange hang).  Each rank looks at its own metric passes, the flags are summed over the ranks, and all skip the
            # event together, before any count has been exchanged; a rank that overflowed raises on the way out, which makes step() enlarge its lists.
            overflow = self._metric_overflow()
            if self._agree(overflow is not None):
                if overflow is not None:
                    raise overflow
                return
            # u32 sum on the device, in place in the pass's own buffer (SURVEY 8(e) "Determinism")
            self.exchange.allreduce_counts(self.metricsPass.getMetricCountsBuffer().ptr, self.pointCloud.num_points)
        self.metricsPass.normalizeMetricCounts(encoder, dict(divisor=usedViews))
        self.densifyPrune.ensureSize(self.pointCloud.num_points)
        prepared = self.densifyPrune.encodePrepare(encoder, dict(pointCloud=self.pointCloud, metricCountsBuffer=self.metricsPass.getMetricCountsBuffer()))
        outTotal = self.densifyPrune.readTotal()  # the one 4-byte read-back (trainer.ts:440-458)
        inN = self.pointCloud.num_points
        outN = min(outTotal, prepared["maxOutPoints"])
        if outN == 0 or outN == inN:
            return
        self.syncOptimizerState()  # every rank rebuilds the whole cloud, so every rank needs the whole state
        self.optimizer.flushSH(self.pointCloud)  # the rebuild copies the cloud's SH rows: bring the deferred DC halves in first
        outPointCloud = ops.allocatePointCloudLike(self.device, self.pointCloud, dict(numPoints=outN))
        outState = ops.allocateOptimizerStateBuffers(self.device, outN)
        self.densifyPrune.encodeScatter(encoder, dict(pointCloud=self.pointCloud, optimizerState=self.optimizer.getStateBuffers(),
                                                      outOffsetBuffer=prepared["outOffsetBuffer"], outNumPoints=outN, resetNewOptimizerState=True),
                                        dict(outPointCloud=outPointCloud, outOptimizerState=outState))
        self._synchronize()
        self.requestPointCloudSwap(outPointCloud, dict(iteration=self.optimizer.getIteration(), buffers=outState))
        self.lastDensifyPruneIteration = self.iteration
