"""Shared helpers for the parity tests: run the HIP path through the operator classes and collect every stage."""
import numpy as np

from webdgs_amd import ops, synth


def small_config(base="c1", num_points=None, width=None, height=None, sh_deg=None, s0=None, fy=None):
    c = synth.CONFIGS[base]
    return synth.SceneConfig(c.config_id, num_points or c.num_points, width or c.width, height or c.height,
                             c.sh_deg if sh_deg is None else sh_deg, fy or c.fy, s0 or c.s0, c.name + "-var")


def scene(cfg):
    g, sh = synth.make_gaussians(cfg)
    cam = synth.identity_camera(cfg)
    return g, sh, cam


class HipPipeline:
    """forward -> rasterize -> backward -> optimizer, wired as src/trainer.ts:568-660 wires them."""

    def __init__(self, dev, cfg, gaussians, sh, camera, max_tile_entries=0, compat_caps=False, training_config=None):
        self.dev, self.cfg = dev, cfg
        self.pc = ops.createPointCloud(dev, gaussians, sh, cfg.sh_deg)
        self.camera = dev.bufferFrom(camera)
        self.fwd = ops.TiledForwardPass(dev, self.pc, self.camera, dict(viewportWidth=cfg.width, viewportHeight=cfg.height, renderMode="gaussian",
                                                                        maxTileEntries=max_tile_entries, compatCaps=compat_caps))
        self.rast = ops.TiledRasterizer(dict(device=dev, forwardPass=self.fwd, format="rgba8unorm", compatCaps=compat_caps))
        self.bwd = ops.TiledBackwardPass(dev, self.pc, dict(viewportWidth=cfg.width, viewportHeight=cfg.height, trainingConfig=training_config or {}))
        self.opt = None

    def forward(self):
        enc = self.dev.createCommandEncoder()
        self.fwd.encode(enc)
        self.rast.encode(enc, self.cfg.width, self.cfg.height)
        self.dev.synchronize()

    def collect_forward(self):
        cfg, n = self.cfg, self.cfg.num_points
        r = self.fwd.getResources()
        stats = r["statsBuffer"].read(np.uint32)
        e = int(stats[0])
        out = dict(stats=stats, total_entries=e,
                   splats=r["splatBuffer"].read(np.uint32).reshape(-1, 6)[:n], depths=r["depthsBuffer"].read(np.uint32)[:n],
                   tile_counts=r["tileCountsBuffer"].read(np.uint32)[:n], tile_offsets=r["tileOffsetsBuffer"].read(np.uint32)[:n],
                   sorted_keys=r["tileKeysBuffer"].read(np.uint32, count=e), sorted_values=r["tileIndicesBuffer"].read(np.uint32, count=e),
                   tile_ranges=self.rast.getTileOffsetsBuffer().read(np.uint32),
                   rgba8=self.rast.getOutputTextureView().read(np.uint8).reshape(cfg.height, cfg.width, 4),
                   final_T=self.rast.getAlphaTextureView().read(np.float32).reshape(cfg.height, cfg.width),
                   n_contrib=self.rast.getNContribTextureView().read(np.uint32).reshape(cfg.height, cfg.width))
        return out

    def backward_resources(self):
        return dict(splatBuffer=self.fwd.getResources()["splatBuffer"], tileOffsetsBuffer=self.rast.getTileOffsetsBuffer(),
                    tileIndicesBuffer=self.fwd.getSortedIndicesBuffer(), cameraBuffer=self.camera,
                    alphaTexture=self.rast.getAlphaTextureView(), nContribTexture=self.rast.getNContribTextureView())

    def train_step(self, target_rgba8_buf):
        """One Trainer.step(): returns nothing; state lives on the device."""
        if self.opt is None:
            self.opt = ops.Optimizer(self.dev, self.pc)
        enc = self.dev.createCommandEncoder()
        self.fwd.encode(enc)
        self.rast.encode(enc, self.cfg.width, self.cfg.height)
        self.bwd.encode(enc, self.rast.getOutputTextureView(), target_rgba8_buf, self.backward_resources())
        self.opt.step(enc, self.pc, self.bwd.getGradientsBuffer(), self.fwd.getResources()["tileCountsBuffer"])

    def read_state(self):
        b = self.opt.getStateBuffers()
        n = self.cfg.num_points
        return dict(opt_pos=b["optPosBuffer"].read(np.float32).reshape(-1, 12)[:n], opt_rot=b["optRotBuffer"].read(np.float32).reshape(-1, 12)[:n],
                    opt_scale=b["optScaleBuffer"].read(np.float32).reshape(-1, 12)[:n], opt_opacity=b["optOpacityBuffer"].read(np.float32).reshape(-1, 3)[:n],
                    param_sh=b["paramSH"].read(np.float32).reshape(-1, 48)[:n], state_sh=b["stateSH"].read(np.float32).reshape(-1, 96)[:n])

    def destroy(self):
        for o in (self.opt, self.bwd, self.rast, self.fwd):
            if o is not None:
                o.destroy()


def acc_to_reference_layout(acc12, n):
    """INTERNAL i32[N,12] accumulators -> the reference's four arrays (means[2N], conics[4N], opacity[N], colors[3N])."""
    a = acc12.reshape(-1, 12)[:n]
    gm = np.ascontiguousarray(a[:, 0:2]).reshape(-1)
    gc = np.zeros((n, 4), np.int32)
    gc[:, 0], gc[:, 1], gc[:, 3] = a[:, 2], a[:, 3], a[:, 4]
    colors = np.empty((n, 3), np.int32)
    colors[:, 0:2] = a[:, 6:8]
    # blue is kept as four partial sums (words 8..11, one per 16-lane row of the producing waves); i32 sums wrap like atomicAdd
    colors[:, 2] = (np.ascontiguousarray(a[:, 8:12]).view(np.uint32).sum(axis=1, dtype=np.uint64) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
    return gm, gc.reshape(-1), np.ascontiguousarray(a[:, 5]), colors.reshape(-1)


def assert_bits_equal(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    av, bv = a.view(np.uint8).reshape(-1), b.view(np.uint8).reshape(-1)
    if not np.array_equal(av, bv):
        bad = np.flatnonzero(av != bv)
        item = a.dtype.itemsize
        idx = np.unique(bad // item)
        raise AssertionError(f"{what}: {idx.size} of {a.size} elements differ; first at flat index {idx[0]}: {a.reshape(-1)[idx[0]]!r} vs {b.reshape(-1)[idx[0]]!r}")
