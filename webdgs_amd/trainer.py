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
import random
import time
from typing import Optional

import numpy as np
import torch

from . import loaders, ops, parallel


class Trainer:
    def __init__(self, device: ops.HipDevice, trainingConfig: Optional[dict] = None, seed: int = 0, world_size: int = 1, rank: int = 0,
                 views_per_rank: int = 1, maxTileEntries: int = 0, use_command_buffers: bool = True):
        self.device = device
        self.trainingConfig = dict(trainingConfig or dict(lambda_l1=0.8, lambda_l2=0.0, lambda_dssim=0.2))  # trainer.ts:100-104
        self.optimizerHyperparameters = dict(ops.DEFAULT_ADAM_HYPERPARAMETERS)
        self.world_size, self.rank, self.views_per_rank = int(world_size), int(rank), max(1, int(views_per_rank))
        self.maxTileEntries = int(maxTileEntries)
        # Recorded command buffers (HIP graphs), one per view set: the reference re-records its encoder every step; here the
        # recording is kept and re-submitted, because every size the kernels need is read on the device.
        self.use_command_buffers = bool(use_command_buffers)
        self._cmd_cache: dict = {}
        self._eager_steps = 0
        self._camera_buffers: list = []
        # every rank draws the same view sequence (same seed), then takes its shard
        self._rng = random.Random(seed)

        self.forwardPass = self.rasterizer = self.backwardPass = self.optimizer = None
        self.metricsForwardPass = self.metricsRasterizer = self.metricsPass = None
        self.metricsViewportWidth = self.metricsViewportHeight = 0
        self.metricsTarget: Optional[ops.HipBuffer] = None
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
        self._dp_grad: Optional[ops.HipBuffer] = None
        self._dp_visible: Optional[ops.HipBuffer] = None

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
        """``trainer.ts:201-237``: tear down the op graph, adopt the new cloud (+ optimizer state), rebuild."""
        self.device.synchronize()
        oldParams = self.optimizer.getHyperparameters() if self.optimizer else None
        for name in ("forwardPass", "rasterizer", "backwardPass", "metricsForwardPass", "metricsRasterizer", "metricsPass", "optimizer"):
            op = getattr(self, name)
            if op is not None:
                op.destroy()
            setattr(self, name, None)
        old = self.pointCloud
        self.pointCloud = request["pointCloud"]
        self.optimizer = ops.Optimizer(self.device, self.pointCloud, oldParams or self.optimizerHyperparameters, request.get("optimizerInitialState"))
        self.optimizerHyperparameters = dict(self.optimizer.getHyperparameters())
        if old is not None and old is not self.pointCloud:
            old.gaussian_3d_buffer.destroy()
            old.sh_buffer.destroy()
        self._dp_grad = self._dp_visible = None
        self._invalidate_command_buffers()
        self.ensurePipelines(self.lastViewportWidth, self.lastViewportHeight)

    def _invalidate_command_buffers(self) -> None:
        """Recorded kernels bake pointers, viewport, hyper-parameters and loss weights: any change drops the recordings."""
        for cmds in self._cmd_cache.values():
            for c in cmds:
                c.destroy()
        self._cmd_cache = {}
        self._eager_steps = 0

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
        for p in (self.backwardPass, self.metricsPass):
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

    def stop(self) -> None:
        self.isTraining = False

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
    def ensurePipelines(self, width: int, height: int) -> None:
        if (max(1, int(width)), max(1, int(height))) != (self.lastViewportWidth, self.lastViewportHeight):
            self._invalidate_command_buffers()
        self.lastViewportWidth, self.lastViewportHeight = max(1, int(width)), max(1, int(height))
        w, h = self.lastViewportWidth, self.lastViewportHeight
        if self.forwardPass is None:
            self.forwardPass = ops.TiledForwardPass(self.device, self.pointCloud, self.cameraBuffer,
                                                    dict(viewportWidth=w, viewportHeight=h, renderMode="gaussian", maxTileEntries=self.maxTileEntries))
        else:
            self.forwardPass.setViewport(w, h)
        if self.rasterizer is None:
            self.rasterizer = ops.TiledRasterizer(dict(device=self.device, forwardPass=self.forwardPass, format="rgba8unorm"))
        if self.backwardPass is None:
            self.backwardPass = ops.TiledBackwardPass(self.device, self.pointCloud, dict(viewportWidth=w, viewportHeight=h, trainingConfig=self.trainingConfig))
        else:
            self.backwardPass.setViewport(w, h)

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
        self.metricsViewportWidth, self.metricsViewportHeight = w, h
        self.metricsForwardPass = ops.TiledForwardPass(self.device, self.pointCloud, self.metricsCameraBuffer,
                                                       dict(viewportWidth=w, viewportHeight=h, renderMode="gaussian", maxTileEntries=self.maxTileEntries))
        self.metricsRasterizer = ops.TiledRasterizer(dict(device=self.device, forwardPass=self.metricsForwardPass, format="rgba8unorm"))
        self.metricsPass = ops.TiledBackwardPass(self.device, self.pointCloud, dict(viewportWidth=w, viewportHeight=h, trainingConfig=self.trainingConfig))
        self.metricsTarget = self.device.createBuffer(4 * w * h, "metrics-gt-downsampled")
        return w, h

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
    def _encode_view(self, encoder, index: int) -> None:
        image = self.images[index]
        cam = self._camera_buffers[index]  # camera.set_preset + update_buffer (trainer.ts:583-586): the view's resident block
        self.forwardPass.setCameraBuffer(cam)
        self.forwardPass.encode(encoder)
        self.rasterizer.encode(encoder, image["width"], image["height"])
        res = dict(splatBuffer=self.forwardPass.getResources()["splatBuffer"], tileOffsetsBuffer=self.rasterizer.getTileOffsetsBuffer(),
                   tileIndicesBuffer=self.forwardPass.getSortedIndicesBuffer(), cameraBuffer=cam,
                   alphaTexture=self.rasterizer.getAlphaTextureView(), nContribTexture=self.rasterizer.getNContribTextureView())
        self.backwardPass.encode(encoder, self.rasterizer.getOutputTextureView(), image["texture"], res)

    def warmupCommandBuffers(self) -> int:
        """Runs training steps on every view in turn until each view's command buffer is recorded (the first pass over a
        dataset does this anyway; calling it up front keeps recording out of a timed region).  Returns the steps taken."""
        if not self.use_command_buffers or not self.isTraining or self.pointCloud is None:
            return 0
        n_views, taken = self.world_size * self.views_per_rank, 0
        for v in range(len(self.trainCameras)):
            ids = [v] * n_views
            while tuple(parallel.shard_views(ids, self.rank, self.world_size)) not in self._cmd_cache and taken < 4 * len(self.trainCameras) + 4:
                self.step(ids)
                taken += 1
        return taken

    def step(self, view_ids: Optional[list] = None) -> None:
        """One training iteration (trainer.ts:568-660).  ``view_ids``: the global batch's views (default: drawn at random, as the
        reference picks ``Math.random()`` per step); with ``world_size > 1`` each rank takes its shard."""
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

        key = tuple(mine)
        cmds = self._cmd_cache.get(key)
        if cmds is None:
            # the first steps run eagerly (they allocate textures); afterwards each view set is recorded once and replayed
            record = self.use_command_buffers and self._eager_steps >= 1
            try:
                cmds = self._encode_and_submit_step(mine, n_views, record)
            except BaseException:
                # a failed encode (capacity, first-use allocation inside a recording, a Python error) must not leave the stream in
                # capture mode or half-recorded command buffers behind: drop the recording and fall back to a clean eager state
                self.device.lib.wdgs_encoder_abort(self.device.handle)
                self._invalidate_command_buffers()
                raise
            if record:
                self._cmd_cache[key] = cmds
            else:
                self._eager_steps += 1
        else:
            self.device.queue.submit([cmds[0]])
            if len(cmds) > 1:
                self._allreduce()
                self.device.queue.submit([cmds[1]])
            self.optimizer.advanceIteration(1)
        self.device.queue.onSubmittedWorkDone()

        self.iteration += 1
        self.stepMs = (time.perf_counter() - stepStart) * 1000.0
        inst = 1000.0 / self.stepMs if self.stepMs > 0 else 0.0
        self.stepItersPerSec = inst if self.stepItersPerSec == 0 else self.stepItersPerSec * 0.9 + inst * 0.1
        if shouldDensify:
            self.runDensifyPruneMultiView()
            req = self.consumePointCloudSwapRequest()
            if req is not None:
                self.applyPointCloudSwap(req)
        if self.iteration >= self.maxIterations:
            self.stop()

    def _encode_and_submit_step(self, mine: list, n_views: int, record: bool) -> list:
        """Encodes (eagerly, or into command buffers when ``record``) and submits one global step; returns the command buffers."""
        tileCounts = self.forwardPass.getResources()["tileCountsBuffer"]
        if n_views == 1:
            with self.device.createCommandEncoder("trainer-step", record=record) as encoder:
                self._encode_view(encoder, mine[0])
                self.optimizer.step(encoder, self.pointCloud, self.backwardPass.getGradientsBuffer(), tileCounts)
                cmds = [encoder.finish()]
            self.device.queue.submit(cmds)
            return cmds
        n = self.pointCloud.num_points
        if self._dp_grad is None:  # (allocated before any recording is opened)
            self._dp_grad = self.device.createBuffer(4 * parallel.GRAD_FLOATS * n, "dp-grad-f32")
            self._dp_visible = self.device.createBuffer(4 * n, "dp-visible")
        with self.device.createCommandEncoder("trainer-step", record=record) as encoder:
            for k, v in enumerate(mine):  # the first view overwrites the fp32 block (no clearing pass), the others add to it
                self._encode_view(encoder, v)
                (ops.storeGradients if k == 0 else ops.accumulateGradients)(self.device, n, self.backwardPass.getGradientsBuffer(), tileCounts,
                                                                            self._dp_grad, self._dp_visible)
            first = encoder.finish()
        self.device.queue.submit([first])
        self._allreduce()
        with self.device.createCommandEncoder("trainer-step-adam", record=record) as encoder2:
            self.optimizer.stepF32(encoder2, self.pointCloud, self._dp_grad, self._dp_visible)
            second = encoder2.finish()
        self.device.queue.submit([second])
        return [first, second]

    def destroy(self) -> None:
        """Deterministic teardown: command buffers, then every op, then the buffers this trainer allocated.  The device itself
        belongs to the caller (``HipDevice.destroy()`` comes after this, and before ``torch.distributed.destroy_process_group``)."""
        if self.device.handle:
            self.device.lib.wdgs_encoder_abort(self.device.handle)
            self.device.synchronize()
        self._invalidate_command_buffers()
        for name in ("forwardPass", "rasterizer", "backwardPass", "metricsForwardPass", "metricsRasterizer", "metricsPass", "optimizer", "densifyPrune"):
            op = getattr(self, name, None)
            if op is not None:
                op.destroy()
            setattr(self, name, None)
        self._dp_grad = self._dp_visible = self.metricsTarget = None
        self._camera_buffers = []
        self.pointCloud = None
        self.isTraining = False

    def _allreduce(self) -> None:
        if self.world_size <= 1:
            return
        n = self.pointCloud.num_points
        g = self._dp_grad.tensor().view(torch.float32)[: parallel.GRAD_FLOATS * n]
        vis = self._dp_visible.tensor()[:n]
        # Host-side fences on both sides of the exchange instead of cross-stream event waits: the collective runs on the
        # process group's own stream, and a 2-rank rehearsal on one GPU (gloo, ranks in lock-step) showed the Adam launch
        # that follows overtaking the copy-back of the reduced block when ordered by stream-wait-event alone.  Two fences
        # cost ~20 us of a >1.3 ms step and make the order independent of the backend's stream handling.
        self.device.torch_stream.synchronize()
        parallel.allreduce_gradients(g, vis)
        torch.cuda.synchronize(self.device.torch_device)

    # ------------------------------------------------------------------ densify / prune
    def runDensifyPruneMultiView(self) -> None:
        if self.pointCloud is None or self.optimizer is None or not self.trainCameras or not self.images:
            return
        baseW, baseH = self.lastViewportWidth, self.lastViewportHeight
        mW, mH = self.ensureMetricsPipelines(baseW, baseH)
        c = self.densifyPruneConfig
        viewsTarget = max(1, int(c["metricViews"]))
        encoder = self.device.createCommandEncoder("densify-prune multiview metrics")
        encoder.clearBuffer(self.metricsPass.getMetricCountsBuffer())
        usedViews, attempts = 0, 0
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
            self.metricsCameraBuffer.write(self.metrics_camera(camData["camera"], mW, mH))
            self.metricsForwardPass.encode(encoder)
            self.metricsRasterizer.encode(encoder, mW, mH)
            ops.downsampleRGBA8(self.device, image["texture"], baseW, baseH, self.metricsTarget, mW, mH)
            self.metricsPass.computeMetricMap(encoder, self.metricsRasterizer.getOutputTextureView(), self.metricsTarget, dict(threshold=c["metricThreshold"]))
            self.metricsPass.computeMetricCounts(encoder, dict(splatBuffer=self.metricsForwardPass.getResources()["splatBuffer"],
                                                               tileOffsetsBuffer=self.metricsRasterizer.getTileOffsetsBuffer(),
                                                               tileIndicesBuffer=self.metricsForwardPass.getSortedIndicesBuffer(),
                                                               nContribTexture=self.metricsRasterizer.getNContribTextureView()), dict(clear=False))
        if usedViews == 0:
            return
        if self.world_size > 1:
            n = self.pointCloud.num_points
            mc = self.metricsPass.getMetricCountsBuffer()
            t = torch.empty(n, dtype=torch.int32, device=self.device.torch_device)
            t.copy_(torch.from_numpy(mc.read(np.int32, count=n)))  # library-owned buffer -> torch tensor for the collective
            self.device.torch_stream.synchronize()  # fences as in _allreduce
            parallel.allreduce_counts(t)
            torch.cuda.synchronize(self.device.torch_device)
            mc.write(t.cpu().numpy())
        self.metricsPass.normalizeMetricCounts(encoder, dict(divisor=usedViews))
        self.densifyPrune.ensureSize(self.pointCloud.num_points)
        prepared = self.densifyPrune.encodePrepare(encoder, dict(pointCloud=self.pointCloud, metricCountsBuffer=self.metricsPass.getMetricCountsBuffer()))
        outTotal = self.densifyPrune.readTotal()  # the one 4-byte read-back (trainer.ts:440-458)
        inN = self.pointCloud.num_points
        outN = min(outTotal, prepared["maxOutPoints"])
        if outN == 0 or outN == inN:
            return
        outPointCloud = ops.allocatePointCloudLike(self.device, self.pointCloud, dict(numPoints=outN))
        outState = ops.allocateOptimizerStateBuffers(self.device, outN)
        self.densifyPrune.encodeScatter(encoder, dict(pointCloud=self.pointCloud, optimizerState=self.optimizer.getStateBuffers(),
                                                      outOffsetBuffer=prepared["outOffsetBuffer"], outNumPoints=outN, resetNewOptimizerState=True),
                                        dict(outPointCloud=outPointCloud, outOptimizerState=outState))
        self.device.synchronize()
        self.requestPointCloudSwap(outPointCloud, dict(iteration=self.optimizer.getIteration(), buffers=outState))
        self.lastDensifyPruneIteration = self.iteration
