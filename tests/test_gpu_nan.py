"""GPU: Gaussians whose fp16 fields hold NaN or infinity -- what a long run of the reference's default schedule produces (its K17 deviates from
the gradient of its own forward pass, DESIGN.md section 2, and Adam then drives some Gaussians out of the number range).  Such a Gaussian passes
every rejection test of K1 that is written as a comparison (a comparison with NaN is false), lands in tile 0 and is walked by every pixel of it:
the 10 600-entry list of profiles/r06z_timelines_late_regime.txt is made of them.  The whole step must still equal the oracle.

Equality here is bit for bit with ONE allowance: a NaN equals a NaN whatever its sign and payload.  Which NaN an operation returns is the one thing
two IEEE machines do not agree on (x86 returns the first operand's payload and generates the negative "indefinite", the GPU propagates by source
operand priority and generates the positive one; for a fused multiply-add the x86 choice even depends on the instruction form the compiler
picked), and nothing downstream can tell them apart: a NaN converts to the fixed-point 0, to the texel 0, and stays a NaN in every sum."""
import numpy as np
import pytest

from webdgs_amd import synth

import harness

pytestmark = pytest.mark.gpu

NAN16, INF16 = 0x7E00, 0x7C00


def differences(a, b, what, fp=None):
    """'' if a == b bit for bit (NaNs of the float type `fp` counting as equal), else a one-line description."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.shape != b.shape:
        return f"{what}: shape {a.shape} vs {b.shape}"
    if fp is not None:
        fa, fb = a.view(fp).reshape(-1), b.view(fp).reshape(-1)
        ua, ub = fa.view(np.uint16 if fp == np.float16 else np.uint32), fb.view(np.uint16 if fp == np.float16 else np.uint32)
        bad = np.flatnonzero((ua != ub) & ~(np.isnan(fa) & np.isnan(fb)))
        if bad.size:
            return f"{what}: {bad.size} of {fa.size} values differ; first at {bad[0]}: {fa[bad[0]]!r} ({ua[bad[0]]:#x}) vs {fb[bad[0]]!r} ({ub[bad[0]]:#x})"
        return ""
    av, bv = a.view(np.uint8).reshape(-1), b.view(np.uint8).reshape(-1)
    if not np.array_equal(av, bv):
        idx = np.unique(np.flatnonzero(av != bv) // a.dtype.itemsize)
        return f"{what}: {idx.size} of {a.size} elements differ; first at {idx[0]}: {a.reshape(-1)[idx[0]]!r} vs {b.reshape(-1)[idx[0]]!r}"
    return ""


def step_differences(orc, dev, cfg, g, sh, cam, target, steps=2, pipeline_factory=None):
    """Runs `steps` training steps through the operator classes and through the oracle; returns the list of stages that differ (all of them, in order)."""
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    ref_g, ref_sh = g.copy(), sh.copy()
    ref_state = orc.unpack(ref_g, ref_sh)
    pipe = (pipeline_factory or harness.HipPipeline)(dev, cfg, g, sh, cam)
    tbuf = dev.bufferFrom(target)
    out = []
    try:
        for step in range(steps):
            ref = orc.train_step(ref_g, ref_sh, ref_state, cam, st, ti, target)
            pipe.train_step(tbuf)
            dev.synchronize()
            got = pipe.collect_forward()
            n = cfg.num_points
            e = ref["total_entries"]
            acc = harness.acc_to_reference_layout(pipe.bwd.getAccumulatorsBuffer().read(np.int32), n)
            state = pipe.read_state()
            checks = [(np.array([int(got["stats"][0])]), np.array([e]), "E", None),
                      (got["splats"], ref["splats"], "splats", np.float16), (got["tile_counts"], ref["tile_counts"], "tile counts", None),
                      (got["sorted_keys"], ref["sorted_keys"][:e], "sorted keys", None), (got["sorted_values"], ref["sorted_values"][:e], "sorted values", None),
                      (got["tile_ranges"], ref["tile_ranges"], "tile ranges", None),
                      (got["rgba8"], ref["rgba8"], "image", None), (got["final_T"], ref["final_T"], "final T", np.float32), (got["n_contrib"], ref["n_contrib"], "n_contrib", None),
                      (pipe.bwd.getLossTextureView().read(np.float32), ref["loss_grad"].reshape(-1), "loss gradient", np.float32)]
            checks += [(acc[i], ref[k], k, None) for i, k in enumerate(("grad_means", "grad_conics", "grad_opacity", "grad_colors"))]
            checks += [(pipe.bwd.getGradientsBuffer().read(np.uint32).reshape(-1, 8)[:n], ref["gradients"], "packed gradients", np.float16),
                       (pipe.pc.gaussian_3d_buffer.read(np.uint32).reshape(-1, 6)[:n], ref_g, "re-packed Gaussians", np.float16),
                       (pipe.pc.sh_buffer.read(np.uint32).reshape(-1, 24)[:n], ref_sh, "re-packed SH", np.float16)]
            checks += [(state[k], ref_state[k], f"optimizer state {k}", np.float32) for k in ("opt_pos", "opt_rot", "opt_scale", "opt_opacity", "param_sh", "state_sh")]
            for a, b, what, fp in checks:
                d = differences(a, b, f"step {step}: {what}", fp)
                if d:
                    out.append(d)
        return out
    finally:
        pipe.destroy()


def poisoned(cfg, field, value, every=7):
    g, sh, cam = harness.scene(cfg)
    gh = g.view(np.uint16).reshape(-1, 12).copy()
    shh = sh.view(np.uint16).reshape(-1, 48).copy()
    rows = np.arange(0, cfg.num_points, every)
    if field == "sh":
        shh[rows, 1] = value
    else:
        for c in dict(position=[0, 1, 2], x=[0], z=[2], opacity=[3], rotation=[4, 5, 6, 7], scale=[8, 9, 10], one_scale=[9])[field]:
            gh[rows, c] = value
    return gh.view(np.uint32).reshape(-1, 6), shh.view(np.uint32).reshape(-1, 24), cam


@pytest.mark.parametrize("value", [NAN16, INF16, 0xFE00, 0xFC00], ids=["nan", "inf", "-nan", "-inf"])
@pytest.mark.parametrize("field", ["position", "x", "z", "opacity", "rotation", "scale", "one_scale", "sh"])
def test_step_with_non_finite_gaussians(hip_device, orc, field, value):
    cfg = harness.small_config("c1", num_points=700, width=64, height=48)
    g, sh, cam = poisoned(cfg, field, value)
    rng = np.random.default_rng(5)
    target = rng.integers(0, 255, (cfg.height, cfg.width, 4), dtype=np.uint8)
    diffs = step_differences(orc, hip_device, cfg, g, sh, cam, target, steps=2)
    assert not diffs, "\n".join(diffs)
