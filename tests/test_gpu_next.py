"""GPU: the "next" rows of SURVEY 8(f) wired into the hot path -- a PLY file and COLMAP/JSON cameras drive the renderer, the
exact SSE/PSNR kernel, and PSNR against the ground truth rises while training."""
import json

import numpy as np
import pytest

from webdgs_amd import loaders, ops, synth
from webdgs_amd.trainer import Trainer

import harness

pytestmark = pytest.mark.gpu


def test_image_sse_is_exact(hip_device):
    rng = np.random.default_rng(0)
    for npix in (1, 63, 4097, 640 * 480):
        a = rng.integers(0, 256, (npix, 4), dtype=np.uint8)
        b = rng.integers(0, 256, (npix, 4), dtype=np.uint8)
        ref = int(((a[:, :3].astype(np.int64) - b[:, :3].astype(np.int64)) ** 2).sum())
        ba, bb = hip_device.bufferFrom(a), hip_device.bufferFrom(b)
        assert ops.imageSSE(hip_device, ba, bb, npix) == ref
        assert ops.imagePSNR(hip_device, ba, ba, npix) == float("inf")
        assert abs(ops.imagePSNR(hip_device, ba, bb, npix) - 10 * np.log10(255.0 ** 2 * 3 * npix / ref)) < 1e-9


def test_ply_and_json_cameras_drive_the_renderer_like_the_synthetic_arrays(hip_device, orc):
    cfg = harness.small_config("c3", num_points=4000, width=160, height=96)
    g, sh = synth.make_gaussians(cfg)
    blk = synth.circle_cameras(cfg, 4)[1]
    pc_host = loaders.loadPointCloud(loaders.exportPly(g, sh, cfg.sh_deg))
    view = blk[0:16].reshape(4, 4).T.astype(np.float64)
    cam = loaders.loadCameraJson(json.dumps(dict(id=0, img_name="v", width=cfg.width, height=cfg.height, fx=1.0, fy=cfg.fy,
                                                 position=list(-view[:3, :3].T @ view[:3, 3]), rotation=view[:3, :3].tolist())).encode())[0]
    cam_block = loaders.cameraUniforms(cam)
    pipe = harness.HipPipeline(hip_device, cfg, pc_host.gaussians, pc_host.sh, cam_block)
    try:
        pipe.forward()
        got = pipe.collect_forward()
        ref = orc.forward(g, sh, cam_block, synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0))
        harness.assert_bits_equal(got["rgba8"], ref["rgba8"], "image from PLY + JSON camera")
        harness.assert_bits_equal(got["sorted_values"], ref["sorted_values"][:ref["total_entries"]], "sort order from PLY + JSON camera")
    finally:
        pipe.destroy()


def test_training_raises_psnr(hip_device):
    cfg = harness.small_config("c2", num_points=6000, width=160, height=112, s0=0.01)
    g, sh, _ = harness.scene(cfg)
    dev = hip_device
    tg, tsh = synth.make_target_scene(g, sh)
    cams = synth.circle_cameras(cfg, 3)
    cameras, images = [], []
    for c in cams:
        p = harness.HipPipeline(dev, cfg, tg, tsh, c)
        p.forward()
        images.append(dict(texture=dev.bufferFrom(p.rast.getOutputTextureView().read(np.uint8)), width=cfg.width, height=cfg.height))
        cameras.append(dict(camera=c, width=cfg.width, height=cfg.height))
        p.destroy()
    t = Trainer(dev, seed=11)
    t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg)); t.setDataset(cameras, images); t.start()

    def psnr():
        vals = []
        for i in range(3):
            t.forwardPass.setCameraBuffer(t._camera_buffers[i]); t.forwardPass.encode(None); t.rasterizer.encode(None, cfg.width, cfg.height)
            vals.append(ops.imagePSNR(dev, t.rasterizer.getOutputTextureView(), images[i]["texture"], cfg.width * cfg.height))
        return float(np.mean(vals))

    t.step()
    before = psnr()
    for _ in range(60):
        t.step()
    after = psnr()
    assert after > before + 0.5, (before, after)
