"""GPU: degenerate and boundary inputs through the whole path against the oracle -- nothing visible, a single Gaussian, viewports
smaller than a tile and not multiples of 16, splats larger than the image with the radius cap lifted (incl. the > 2048-tile cull
of tiled-forward.wgsl), saturated pixels, and an empty point cloud."""
import numpy as np
import pytest

from webdgs_amd import ops, synth

import harness
from harness import assert_bits_equal

pytestmark = pytest.mark.gpu


def _full_step_matches_oracle(orc, dev, cfg, g, sh, cam, target, steps=2, max_radius=None):
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    if max_radius is not None:
        st[6] = max_radius
    ref_g, ref_sh = g.copy(), sh.copy()
    ref_state = orc.unpack(ref_g, ref_sh)
    pipe = harness.HipPipeline(dev, cfg, g, sh, cam) if max_radius is None else _pipeline_with_radius(dev, cfg, g, sh, cam, max_radius)
    tbuf = dev.bufferFrom(target)
    try:
        for step in range(steps):
            ref = orc.train_step(ref_g, ref_sh, ref_state, cam, st, ti, target)
            pipe.train_step(tbuf)
            dev.synchronize()
            got = pipe.collect_forward()
            n = cfg.num_points
            assert int(got["stats"][0]) == ref["total_entries"], f"E, step {step}"
            assert_bits_equal(got["rgba8"], ref["rgba8"], f"image, step {step}")
            assert_bits_equal(got["final_T"], ref["final_T"], f"final T, step {step}")
            assert_bits_equal(got["n_contrib"], ref["n_contrib"], f"n_contrib, step {step}")
            assert_bits_equal(pipe.bwd.getGradientsBuffer().read(np.uint32).reshape(-1, 8)[:n], ref["gradients"], f"packed gradients, step {step}")
            assert_bits_equal(pipe.pc.gaussian_3d_buffer.read(np.uint32).reshape(-1, 6)[:n], ref_g, f"re-packed Gaussians, step {step}")
            assert_bits_equal(pipe.pc.sh_buffer.read(np.uint32).reshape(-1, 24)[:n], ref_sh, f"re-packed SH, step {step}")
        return got
    finally:
        pipe.destroy()


def _pipeline_with_radius(dev, cfg, g, sh, cam, max_radius):
    p = harness.HipPipeline.__new__(harness.HipPipeline)
    p.dev, p.cfg = dev, cfg
    p.pc = ops.createPointCloud(dev, g, sh, cfg.sh_deg)
    p.camera = dev.bufferFrom(cam)
    p.fwd = ops.TiledForwardPass(dev, p.pc, p.camera, dict(viewportWidth=cfg.width, viewportHeight=cfg.height, renderMode="gaussian", maxSplatRadiusPx=max_radius))
    p.rast = ops.TiledRasterizer(dict(device=dev, forwardPass=p.fwd, format="rgba8unorm"))
    p.bwd = ops.TiledBackwardPass(dev, p.pc, dict(viewportWidth=cfg.width, viewportHeight=cfg.height, trainingConfig={}, maxSplatRadiusPx=max_radius))
    p.opt = None
    return p


def _f16(x):
    return np.float16(x).view(np.uint16)


def _gaussians(rows):
    """rows: (x, y, z, opacity_raw, log_sigma) -> packed Gaussian words + SH (DC only, grey 0.5 + 0.28*1.5)."""
    g = np.zeros((len(rows), 12), np.uint16)
    sh = np.zeros((len(rows), 48), np.uint16)
    for i, (x, y, z, o, ls) in enumerate(rows):
        g[i, 0:4] = [_f16(x), _f16(y), _f16(z), _f16(o)]
        g[i, 4] = _f16(1.0)  # identity rotation (w, x, y, z)
        g[i, 8:11] = _f16(ls)
        sh[i, 0:3] = [_f16(1.5), _f16(0.3), _f16(-0.7)]
    return g.view(np.uint32).reshape(len(rows), 6), sh.view(np.uint32).reshape(len(rows), 24)


def test_nothing_visible_leaves_everything_untouched(hip_device, orc):
    cfg = harness.small_config("c1", num_points=500, width=64, height=48)
    g, sh, cam = harness.scene(cfg)
    gh = g.view(np.uint16).reshape(-1, 12).copy()
    gh[:, 2] = _f16(-5.0)  # behind the camera
    g = gh.view(np.uint32).reshape(-1, 6)
    target = np.full((cfg.height, cfg.width, 4), 90, np.uint8)
    got = _full_step_matches_oracle(orc, hip_device, cfg, g, sh, cam, target)
    assert int(got["stats"][0]) == 0 and int(got["stats"][1]) == 0
    assert (got["rgba8"][..., :3] == 0).all() and (got["final_T"] == 1.0).all() and (got["n_contrib"] == 0).all()
    assert (got["tile_ranges"][:-1] == 0xFFFFFFFF).all() and got["tile_ranges"][-1] == 0


@pytest.mark.parametrize("w,h", [(1, 1), (7, 5), (16, 16), (17, 33)])
def test_single_gaussian_and_tiny_viewports(hip_device, orc, w, h):
    cfg = harness.small_config("c1", num_points=1, width=w, height=h, fy=20.0)
    g, sh = _gaussians([(0.0, 0.0, 3.0, 2.0, -1.5)])
    cam = synth.identity_camera(cfg)
    target = np.full((h, w, 4), 200, np.uint8)
    got = _full_step_matches_oracle(orc, hip_device, cfg, g, sh, cam, target)
    assert int(got["stats"][1]) == 1 and got["n_contrib"].max() >= 1


def test_huge_splats_without_the_radius_cap_and_the_2048_tile_cull(hip_device, orc):
    """max_splat_radius_px < 0 lifts the cap: a splat covering the whole 640x480 image is binned into all 1200 tiles; on a
    1280x720 grid (3600 tiles) the same splat exceeds 2048 tiles and is dropped by the projection, as in the reference."""
    rows = [(0.0, 0.0, 2.0, 3.0, 0.5), (0.3, -0.2, 2.5, 1.0, -2.0), (-0.4, 0.1, 3.0, 0.5, -0.5)]
    g, sh = _gaussians(rows)
    for (w, h, expect_big) in ((640, 480, True), (1280, 720, False)):
        cfg = harness.small_config("c2", num_points=len(rows), width=w, height=h, fy=400.0, sh_deg=0)
        cam = synth.identity_camera(cfg)
        target = np.full((h, w, 4), 128, np.uint8)
        got = _full_step_matches_oracle(orc, hip_device, cfg, g, sh, cam, target, steps=1, max_radius=-1.0)
        counts = got["tile_counts"]
        tiles = ((w + 15) // 16) * ((h + 15) // 16)
        assert (counts[0] == tiles) == expect_big, (counts, tiles)
        if not expect_big:
            assert counts[0] == 0  # > 2048 tiles: culled (tiled-forward.wgsl num_tiles guard)


def test_saturated_pixels_stop_early_but_match(hip_device, orc):
    """Hundreds of opaque splats stacked on the same pixels: every wave saturates (A > 0.99) long before its list ends."""
    rng = np.random.default_rng(4)
    rows = [(float(rng.uniform(-0.05, 0.05)), float(rng.uniform(-0.05, 0.05)), float(2.0 + 0.01 * i), 6.0, -2.2) for i in range(600)]
    g, sh = _gaussians(rows)
    cfg = harness.small_config("c1", num_points=len(rows), width=48, height=48, fy=120.0, sh_deg=0)
    cam = synth.identity_camera(cfg)
    target = np.zeros((48, 48, 4), np.uint8)
    got = _full_step_matches_oracle(orc, hip_device, cfg, g, sh, cam, target)
    centre = got["final_T"][20:28, 20:28]
    assert (centre < 0.011).all() and got["n_contrib"][24, 24] < 600


def test_empty_point_cloud(hip_device):
    dev = hip_device
    pc = ops.createPointCloud(dev, np.zeros((0, 6), np.uint32), np.zeros((0, 24), np.uint32), 0)
    cam = dev.bufferFrom(synth.identity_camera(harness.small_config("c1", num_points=1, width=40, height=24)))
    fwd = ops.TiledForwardPass(dev, pc, cam, dict(viewportWidth=40, viewportHeight=24, renderMode="gaussian"))
    rast = ops.TiledRasterizer(dict(device=dev, forwardPass=fwd, format="rgba8unorm"))
    try:
        fwd.encode(None)
        rast.encode(None, 40, 24)
        st = fwd.check()
        assert int(st[0]) == 0 and int(st[1]) == 0
        img = rast.getOutputTextureView().read(np.uint8).reshape(24, 40, 4)
        assert (img[..., :3] == 0).all() and (img[..., 3] == 255).all()
        assert (rast.getAlphaTextureView().read(np.float32) == 1.0).all()
    finally:
        rast.destroy()
        fwd.destroy()


@pytest.mark.parametrize("w,h,fy,kind", [(96, 64, 70.0, "oversized"), (320, 208, 230.0, "lds-two-pass")])
def test_sort_paths_wide_depth_span_and_oversized_tiles(hip_device, orc, w, h, fy, kind):
    """The tile-structured sort has three per-tile paths (sort.hip, segment_sort): one 10-bit pass when the tile's depth span is below
    1024 depth16 steps, two 8-bit passes when it is wider, and the same two passes through global memory for tiles with more than 2048
    entries.  The synthetic scenes (z in [2, 10]) only ever take the first: here every Gaussian is moved along its view ray by a
    factor in [0.03, 9] (same pixel footprint, depth16 spanning dozens of exponent steps); a small viewport packs thousands of
    entries into each tile ("oversized"), a larger one keeps tiles below 2048 entries ("lds-two-pass").  Keys, stable order, ranges,
    the image and a training step must equal the oracle."""
    dev = hip_device
    cfg = harness.small_config("c3", num_points=60_000, width=w, height=h, s0=0.004, fy=fy)
    g, sh, cam = harness.scene(cfg)
    rng = np.random.default_rng(11)
    hv = g.copy().view(np.float16).reshape(-1, 12)
    f = np.exp(rng.uniform(np.log(0.03), np.log(9.0), cfg.num_points)).astype(np.float32)
    hv[:, 0:3] = (hv[:, 0:3].astype(np.float32) * f[:, None]).astype(np.float16)          # slide along the ray through the origin
    hv[:, 8:11] = (hv[:, 8:11].astype(np.float32) + np.log(f)[:, None]).astype(np.float16)  # keep the projected size
    g = hv.view(np.uint32).reshape(-1, 6)
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    ref = orc.forward(g, sh, cam, st, ti)
    e = ref["total_entries"]
    keys = ref["sorted_keys"][:e]
    tiles = keys >> 16
    counts = np.bincount(tiles, minlength=ti[2] + 2)
    d16 = (keys & 0xFFFF).astype(np.int64)
    spans = np.array([d16[tiles == t].max() - d16[tiles == t].min() for t in np.unique(tiles)])
    if kind == "oversized":
        assert counts.max() > 2048, "tiles beyond the LDS capacity"
    else:
        big = counts[1:ti[2] + 1]
        assert big.max() <= 2048 and (spans >= 1024).any() and big.max() > 300, "LDS-resident tiles that need the two-pass digit split"
    tg, tsh = synth.make_target_scene(g, sh)
    target = orc.forward(tg, tsh, cam, st, ti)["rgba8"]
    pipe = harness.HipPipeline(dev, cfg, g, sh, cam)
    try:
        pipe.forward()
        got = pipe.collect_forward()
        assert got["total_entries"] == e
        assert_bits_equal(got["sorted_keys"], keys, "sorted keys (wide depth span)")
        assert_bits_equal(got["sorted_values"], ref["sorted_values"][:e], "stable order (wide depth span)")
        assert_bits_equal(got["tile_ranges"], ref["tile_ranges"], "tile ranges")
        assert_bits_equal(got["rgba8"], ref["rgba8"], "image")
    finally:
        pipe.destroy()
    _full_step_matches_oracle(orc, dev, cfg, g, sh, cam, target, steps=1)


def test_needle_splats_and_marginal_opacities(hip_device, orc):
    """Adversarial input for the block-level alpha test of backward_rasterize (a CONSERVATIVE cull: it may only drop a splat from an
    8x8 block when no pixel of the block can reach alpha >= 1/255): needles with axis ratios up to 300:1 at random orientations --
    their bounding boxes are mostly empty corners, exactly what the test culls -- large screen footprints, and opacities spread down
    to the 1/128 visibility threshold, where the 1/255 contour hugs the box.  Every accumulator must still equal the oracle."""
    cfg = harness.small_config("c3", num_points=12_000, width=208, height=144, s0=0.01, fy=160.0)
    g, sh, cam = harness.scene(cfg)
    rng = np.random.default_rng(5)
    hv = g.copy().view(np.float16).reshape(-1, 12)
    n = cfg.num_points
    axis = rng.integers(0, 3, n)
    stretch = np.where(rng.random(n) < 0.5, rng.uniform(np.log(10.0), np.log(30.0), n), 0.0).astype(np.float32)
    ls = hv[:, 8:11].astype(np.float32)
    ls[np.arange(n), axis] += stretch                       # one axis up to 30x longer (on top of the generator's 10:1 spread)
    hv[:, 8:11] = ls.astype(np.float16)
    hv[:, 3] = rng.uniform(-4.8, 3.0, n).astype(np.float16)  # sigmoid(-4.8) = 0.0082: just above the 1/128 = 0.0078 cull
    g = hv.view(np.uint32).reshape(-1, 6)
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    tg, tsh = synth.make_target_scene(g, sh)
    target = orc.forward(tg, tsh, cam, st, ti)["rgba8"]
    ref = orc.forward(g, sh, cam, st, ti)
    assert (ref["tile_counts"] > 0).sum() > 5000 and ref["total_entries"] > 100_000
    _full_step_matches_oracle(orc, hip_device, cfg, g, sh, cam, target, steps=2)
    _full_step_matches_oracle(orc, hip_device, cfg, g, sh, cam, target, steps=1, max_radius=-1.0)  # radius cap lifted: footprints of hundreds of pixels


def _forward_equals_oracle(orc, dev, cfg, g, sh, cam, what):
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    ref = orc.forward(g, sh, cam, st, ti)
    e = ref["total_entries"]
    pipe = harness.HipPipeline(dev, cfg, g, sh, cam)
    try:
        pipe.forward()
        got = pipe.collect_forward()
        assert got["total_entries"] == e, what
        assert_bits_equal(got["tile_offsets"], ref["tile_offsets"], f"{what}: per-Gaussian offsets")
        assert_bits_equal(got["sorted_keys"], ref["sorted_keys"][:e], f"{what}: sorted keys")
        assert_bits_equal(got["sorted_values"], ref["sorted_values"][:e], f"{what}: stable order")
        assert_bits_equal(got["tile_ranges"], ref["tile_ranges"], f"{what}: tile ranges")
        assert_bits_equal(got["rgba8"], ref["rgba8"], f"{what}: image")
        # encode(skipSort) right after a sorted encode: the reference's emission order, in the buffers the getters hand out
        pipe.fwd.encode(None, dict(skipSort=True))
        dev.synchronize()
        assert_bits_equal(pipe.fwd.getSortedKeysBuffer().read(np.uint32, count=e), ref["keys"][:e], f"{what}: emission-order keys after skipSort")
        assert_bits_equal(pipe.fwd.getSortedIndicesBuffer().read(np.uint32, count=e), ref["values"][:e], f"{what}: emission-order values after skipSort")
        pipe.forward()   # ... and sorted again
        assert_bits_equal(pipe.collect_forward()["sorted_values"], ref["sorted_values"][:e], f"{what}: stable order, second sorted encode")
    finally:
        pipe.destroy()
    return ref


@pytest.mark.parametrize("w,h", [(16, 400), (32, 400), (4096, 48), (4112, 48), (48, 4096), (48, 4112), (250, 130)],
                         ids=["1-column", "2-columns", "256-columns", "257-columns", "256-rows", "257-rows", "odd"])
def test_tile_grid_shapes_around_the_column_path(hip_device, orc, w, h):
    """The forward pass folds the sort's first pass into emit when the grid has 2..256 tile columns and <= 256 rows (emit_scatter: digit =
    tile column, then one pass on the tile row); outside that it keeps emit + two tile-bit passes.  Both sides of every limit, against the
    oracle: offsets, sorted keys, stable order, ranges, image; encode(skipSort) in between still yields the reference's emission order."""
    cfg = harness.small_config("c1", num_points=3000, width=w, height=h, fy=0.9 * max(w, h), s0=0.01)
    g, sh, cam = harness.scene(cfg)
    ref = _forward_equals_oracle(orc, hip_device, cfg, g, sh, cam, f"{w}x{h}")
    assert ref["total_entries"] > 1000


def test_workgroups_with_many_chunks_of_entries(hip_device, orc):
    """emit_scatter expands a workgroup's 256 Gaussians in chunks of 2048 entries: large splats (hundreds of tiles each, radius cap lifted)
    give workgroups tens of chunks, with Gaussians spanning chunk boundaries and columns receiving entries from every chunk."""
    cfg = harness.small_config("c2", num_points=900, width=640, height=480, fy=500.0, s0=0.12, sh_deg=0)
    g, sh, cam = harness.scene(cfg)
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    ref = _forward_equals_oracle(orc, hip_device, cfg, g, sh, cam, "many chunks")
    per_wg = np.add.reduceat(ref["tile_counts"].astype(np.int64), np.arange(0, cfg.num_points, 256))
    assert per_wg.max() > 6 * 2048, per_wg
    tg, tsh = synth.make_target_scene(g, sh)
    target = orc.forward(tg, tsh, cam, st, ti)["rgba8"]
    _full_step_matches_oracle(orc, hip_device, cfg, g, sh, cam, target, steps=1)
