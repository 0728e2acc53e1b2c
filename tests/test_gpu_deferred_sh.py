"""GPU: deferred SH writes (include/webdgs.h wdgs_optimizer_set_deferred_sh).  With deferral on -- the Trainer's
default -- Adam writes the trained SH-DC halves to a compact array that K1 reads, and the cloud's 96-byte rows are brought up to date
only at hand-over points.  Every hand-over path must see CURRENT rows after N steps, and the trajectory must be bit-identical to the
reference's write pattern (deferral off)."""
import numpy as np
import pytest

from webdgs_amd import loaders, ops, synth
from webdgs_amd.trainer import Trainer
from webdgs_amd.viewer import Viewer

import harness
from harness import assert_bits_equal

pytestmark = pytest.mark.gpu


def _dataset(dev, cfg, n_views=3):
    g, sh = synth.make_gaussians(cfg)
    tg, tsh = synth.make_target_scene(g, sh)
    cams = synth.circle_cameras(cfg, n_views)
    tp = harness.HipPipeline(dev, cfg, tg, tsh, cams[0])
    cameras, images = [], []
    for i in range(n_views):
        tp.camera.write(cams[i])
        tp.forward()
        images.append(dict(texture=dev.bufferFrom(tp.rast.getOutputTextureView().read(np.uint8)), width=cfg.width, height=cfg.height))
        cameras.append(dict(camera=cams[i], width=cfg.width, height=cfg.height))
    tp.destroy()
    return g, sh, cameras, images


def _trainer(dev, cfg, g, sh, cameras, images, deferred, **kw):
    t = Trainer(dev, seed=5, **kw)
    t.deferred_sh = deferred
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset(cameras, images)
    t.setDensifyPruneConfig(dict(schedule=dict(enabled=True, warmupIterations=6, interval=5, stopIterations=100), metricViews=2, cloneThresholdCount=2,
                                 maxNewPointsPerStep=300))
    t.start()
    return t


@pytest.mark.parametrize("views_per_rank", [1, 2])
def test_deferred_and_immediate_sh_writes_train_the_same_cloud(hip_device, views_per_rank):
    """13 steps across two densify rebuilds, single-view (fused K17 + Adam) and batched (fp32 block + adam_repack_f32): cloud, SH rows and
    all optimizer state equal bit for bit with deferral on and off."""
    cfg = harness.small_config("c2", num_points=5000, width=160, height=128, sh_deg=2, s0=0.02)
    g, sh, cameras, images = _dataset(hip_device, cfg)
    out = []
    for deferred in (False, True):
        t = _trainer(hip_device, cfg, g, sh, cameras, images, deferred, views_per_rank=views_per_rank)
        for _ in range(13):
            t.step()
        t.drain()
        assert (t._dc_words is not None) == deferred
        st = t.optimizer.getStateBuffers()
        out.append(dict(n=t.getPointCount(), g=t.pointCloud.gaussian_3d_buffer.read(np.uint32), sh=t.pointCloud.sh_buffer.read(np.uint32),
                        **{k: st[k].read(np.uint32) for k in st}))
        t.destroy()
    assert out[0]["n"] == out[1]["n"] != 5000, "the run crosses densify rebuilds"
    for k in out[0]:
        if k != "n":
            assert_bits_equal(out[1][k], out[0][k], f"deferred vs immediate: {k}")


def test_every_hand_over_path_sees_current_sh_rows(hip_device):
    """After N steps with deferral on: (1) a host read of pointCloud.sh_buffer, (2) a PLY export, (3) a viewer's own forward pass after
    flushPointCloud(), (4) getStateBuffers().paramSH, (5) the cloud left behind by Trainer.destroy() -- all carry the trained DC values (the
    run with deferral off is the reference).  And BEFORE a flush the rows are indeed stale, so the test can tell."""
    cfg = harness.small_config("c2", num_points=4000, width=160, height=128, sh_deg=1, s0=0.02)
    g, sh, cameras, images = _dataset(hip_device, cfg)
    ref = _trainer(hip_device, cfg, g, sh, cameras, images, False)
    ref.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
    t = _trainer(hip_device, cfg, g, sh, cameras, images, True)
    t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
    for _ in range(6):
        ref.step(); t.step()
    ref.drain(); t.drain()
    want_sh = ref.pointCloud.sh_buffer.read(np.uint32).reshape(-1, 24)
    assert not np.array_equal(want_sh[:, 0], sh[:, 0]), "the DC rows were trained"
    # the rows on the device are stale until a hand-over (raw device copy that bypasses the read hook)
    raw = np.empty_like(want_sh)
    ops.check(hip_device.lib.wdgs_copy_to_host(hip_device.handle, raw.ctypes.data, t.pointCloud.sh_buffer.ptr, raw.nbytes))
    assert not np.array_equal(raw[:, 0], want_sh[:, 0]), "deferral leaves the 96-byte rows alone during training"
    # (3) viewer with its own forward pass over the same cloud
    t.flushPointCloud()
    frames = []
    for tr in (ref, t):
        v = Viewer(hip_device, cfg.width, cfg.height)
        v.setCamera(np.asarray(cameras[1]["camera"], np.float32))
        v.setPointCloud(tr.pointCloud)
        v.setRenderMode("gaussian")
        v.render()
        frames.append(v.readFrame().copy())
        v.destroy()
    assert frames[0][..., :3].max() > 0
    assert_bits_equal(frames[1], frames[0], "viewer image after flushPointCloud")
    # more steps make the rows stale again; (1) host read goes through the hook
    for _ in range(3):
        ref.step(); t.step()
    ref.drain(); t.drain()
    want_sh = ref.pointCloud.sh_buffer.read(np.uint32).reshape(-1, 24)
    assert_bits_equal(t.pointCloud.sh_buffer.read(np.uint32).reshape(-1, 24), want_sh, "host read of sh_buffer")
    # (2) export
    for _ in range(2):
        ref.step(); t.step()
    ref.drain(); t.drain()
    plys = [loaders.exportPly(tr.pointCloud.gaussian_3d_buffer.read(np.uint32).reshape(-1, 6), tr.pointCloud.sh_buffer.read(np.uint32).reshape(-1, 24), cfg.sh_deg)
            for tr in (ref, t)]
    assert plys[0] == plys[1]
    # (4) optimizer state
    for _ in range(2):
        ref.step(); t.step()
    ref.drain(); t.drain()
    assert_bits_equal(t.optimizer.getStateBuffers()["paramSH"].read(np.uint32), ref.optimizer.getStateBuffers()["paramSH"].read(np.uint32), "paramSH")
    # (5) destroy leaves the cloud current (the cloud outlives the trainer's optimizer)
    for _ in range(2):
        ref.step(); t.step()
    ref.drain(); t.drain()
    pc_ref, pc_t = ref.pointCloud, t.pointCloud
    want = pc_ref.sh_buffer.read(np.uint32)
    t.optimizer.destroy()
    raw = np.empty_like(want)
    ops.check(hip_device.lib.wdgs_copy_to_host(hip_device.handle, raw.ctypes.data, pc_t.sh_buffer.ptr, raw.nbytes))
    assert_bits_equal(raw, want, "SH rows after Optimizer.destroy()")
    t.optimizer = None
    t.destroy(); ref.destroy()
