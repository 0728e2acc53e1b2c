"""Preview path: forward pass + rasterizer + blit, mirroring ``src/viewer.ts:8-115`` without the browser.

The reference's ``Viewer`` owns a ``Camera`` bound to a canvas, builds a ``TiledForwardPass`` in ``'pointcloud'`` render mode
and a ``TiledRasterizer`` on ``setPointCloud`` (viewer.ts:46-66), and per frame encodes forward -> rasterize ->
``blitToTexture(swapChainView)`` (viewer.ts:72-87).  Here the canvas is a ``width x height`` rgba8 device buffer (the
"swap-chain image"); ``readFrame`` / ``savePNG`` take the place of presentation.  Camera interaction (``CameraControl``) is
UI and stays out; the camera is set from a ``CameraData`` dict or a ready 68-float block.
"""
from __future__ import annotations

import struct
import zlib
from typing import Optional

import numpy as np

from . import loaders
from ._lib import CapacityError
from .ops import HipBuffer, HipDevice, HipEncoder, PointCloud, TiledForwardPass, TiledRasterizer


def encodePNG(rgba: np.ndarray) -> bytes:
    """Minimal PNG writer (8-bit RGBA, filter 0, one IDAT) for ``[H, W, 4]`` uint8 frames."""
    a = np.ascontiguousarray(rgba, np.uint8)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError("encodePNG expects an [H, W, 4] uint8 array")
    h, w = a.shape[:2]
    raw = np.concatenate([np.zeros((h, 1), np.uint8), a.reshape(h, w * 4)], axis=1).tobytes()

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")


class Camera:
    """``Camera`` (``src/camera/camera.ts:100-205``) without the browser: ``canvas`` is any object with ``width`` and ``height``.  Owns the
    272-byte uniform buffer a forward pass reads and rewrites it, stream-ordered, on every ``update_buffer()``."""

    def __init__(self, canvas, device: HipDevice):
        self.canvas, self.device = canvas, device
        self.uniform_buffer: HipBuffer = device.createBuffer(272, "camera uniform")
        self.uniforms = np.zeros(68, np.float32)
        self.reset()

    def reset(self) -> None:
        """camera.ts:113-119: position (0, 0, 5), identity rotation, fovY 45 degrees."""
        self._preset = dict(position=(0.0, 0.0, 5.0))
        self.on_update_canvas()

    def on_update_canvas(self) -> None:
        self.update_buffer()

    def update_buffer(self) -> None:
        self.uniforms = loaders.cameraUniforms(self._preset, int(self.canvas.width), int(self.canvas.height))
        self.uniform_buffer.write(self.uniforms)

    def set_preset(self, preset: dict) -> None:
        """camera.ts:196-205: a CameraData dict (pose kept where the preset has none; fovY from height and fy when both are there)."""
        merged = dict(self._preset)
        for k in ("position", "rotation"):
            if preset.get(k) is not None:
                merged[k] = preset[k]
        if preset.get("fx") and preset.get("fy") and preset.get("height"):
            merged.update(fx=preset["fx"], fy=preset["fy"], height=preset["height"])
        self._preset = merged
        self.on_update_canvas()


class Viewer:
    """``Viewer`` (``src/viewer.ts``): ``setPointCloud``, ``render(encoder)``, pass-through setters, resize handling."""

    def __init__(self, device: HipDevice, width: int, height: int, format: str = "rgba8unorm"):
        self.device = device
        self.presentationFormat = format
        self.width, self.height = int(width), int(height)
        self.forwardPass: Optional[TiledForwardPass] = None
        self.rasterizer: Optional[TiledRasterizer] = None
        self.pointCloud: Optional[PointCloud] = None
        self._camera_data: Optional[dict] = None
        self.cameraBuffer: HipBuffer = device.createBuffer(272, "camera uniforms")
        self.frameBuffer: HipBuffer = device.createBuffer(4 * self.width * self.height, "swap-chain image")
        self.setCamera(dict(position=(0.0, 0.0, 5.0)))  # Camera defaults (camera.ts:113-136)

    # ---- camera (camera.ts:138-205 via loaders.cameraUniforms)
    def setCamera(self, camera) -> None:
        """``camera``: a CameraData dict (``loaders.loadCameraJson`` / ``mergeColmap`` entries) or 68 floats."""
        if isinstance(camera, dict):
            self._camera_data = camera
            block = loaders.cameraUniforms(camera, self.width, self.height)
        else:
            self._camera_data = None
            block = np.asarray(camera, np.float32).reshape(68)
        self.cameraBuffer.write(block)

    def setPointCloud(self, pointCloud: PointCloud) -> None:
        self._settings = dict(renderMode="pointcloud")  # (viewer.ts:46-66: a new cloud starts in point-cloud mode, default scale and point size)
        self._tile_entries = 0
        self.pointCloud = pointCloud
        self._build_passes()

    def _build_passes(self) -> None:
        if self.forwardPass is not None:
            self.forwardPass.destroy()
        if self.rasterizer is not None:
            self.rasterizer.destroy()
        self.forwardPass = TiledForwardPass(self.device, self.pointCloud, self.cameraBuffer,
                                            dict(viewportWidth=self.width, viewportHeight=self.height, renderMode="pointcloud", maxTileEntries=self._tile_entries))
        self.rasterizer = TiledRasterizer(dict(device=self.device, forwardPass=self.forwardPass, format=self.presentationFormat))
        for k, apply in (("renderMode", self.forwardPass.setRenderMode), ("gaussianScale", self.forwardPass.setGaussianScale), ("pointSize", self.forwardPass.setPointSize)):
            if k in self._settings:
                apply(self._settings[k])

    def update(self, dt: float) -> None:
        """Camera-control integration step of the reference (viewer.ts:68-70); no interactive control here."""

    def render(self, commandEncoder: Optional[HipEncoder] = None) -> None:
        if self.forwardPass is None or self.rasterizer is None or self.pointCloud is None:
            return
        self.forwardPass.encode(commandEncoder)
        self.rasterizer.encode(commandEncoder, self.width, self.height)
        self.rasterizer.blitToTexture(commandEncoder, self.frameBuffer, self.width, self.height)

    # ---- pass-through setters / getters (viewer.ts:89-104)
    def setRenderMode(self, mode: str) -> None:
        if self.forwardPass is not None:
            self._settings["renderMode"] = mode
            self.forwardPass.setRenderMode(mode)

    def setGaussianScale(self, value: float) -> None:
        if self.forwardPass is not None:
            self._settings["gaussianScale"] = value
            self.forwardPass.setGaussianScale(value)

    def setPointSize(self, value: float) -> None:
        if self.forwardPass is not None:
            self._settings["pointSize"] = value
            self.forwardPass.setPointSize(value)

    def getForwardPass(self) -> Optional[TiledForwardPass]:
        return self.forwardPass

    def resize(self, width: int, height: int) -> None:
        """``handleResize`` (viewer.ts:106-113): new canvas size -> camera block and forward-pass viewport follow."""
        self.width, self.height = int(width), int(height)
        self.frameBuffer.destroy()
        self.frameBuffer = self.device.createBuffer(4 * self.width * self.height, "swap-chain image")
        if self._camera_data is not None:
            self.cameraBuffer.write(loaders.cameraUniforms(self._camera_data, self.width, self.height))
        if self.forwardPass is not None:
            self.forwardPass.setViewport(self.width, self.height)

    # ---- presentation
    def readFrame(self) -> np.ndarray:
        """The presented image as ``[H, W, 4]`` uint8 (synchronises).  If the frame's tile-entry list outran what the library sized for the cloud
        (the reference would show the truncated picture; the library reports it), the viewer's passes are rebuilt around larger lists and the
        frame is rendered again.  (A report about this viewer's pass may have been consumed by another owner's wait -- a Trainer on the same device:
        it was left in ``device.capacityReports``, and is answered here.)"""
        import re
        for _ in range(4):  # this viewer's own pass: its word is consumed by its own check
            if self.forwardPass is None:
                break
            try:
                self.forwardPass.check()
                left = self.device.capacityReports.take([int(self.forwardPass.handle.value or 0)])   # (consumed by another owner's wait, left for us)
                if left is not None:
                    raise left
                break
            except CapacityError as e:
                own = int(self.forwardPass.handle.value or 0)   # (a report names up to four passes: this viewer's line)
                named = [(int(n), int(c)) for n, c, h in re.findall(r"(\d+) entries needed, max_tile_entries = (\d+) \(forward pass (0x[0-9a-fA-F]+)\)", str(e)) if int(h, 16) == own]
                if not named:
                    raise
                self._tile_entries = min(max(2 * named[0][1], int(named[0][0] * 1.5)), 0xFFFFF000)
                self._build_passes()
                self.render(None)
        return self.frameBuffer.read(np.uint8, 4 * self.width * self.height).reshape(self.height, self.width, 4)

    def savePNG(self, path: str) -> None:
        with open(path, "wb") as f:
            f.write(encodePNG(self.readFrame()))

    def destroy(self) -> None:
        if self.forwardPass is not None:
            self.forwardPass.destroy()
        if self.rasterizer is not None:
            self.rasterizer.destroy()
        self.forwardPass = self.rasterizer = None
