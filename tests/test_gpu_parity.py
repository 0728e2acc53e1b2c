"""GPU parity: the HIP path, called through the C ABI via the operator classes, against the CPU oracle.

Everything is compared BIT FOR BIT (integers, fp16/fp32 bit patterns): the kernels' arithmetic is pinned
operation by operation (DESIGN.md "dmath"), so there is no tolerance to state -- except where a comment says so.
"""
import numpy as np
import pytest

from webdgs_amd import ops, synth

import harness
from harness import assert_bits_equal

pytestmark = pytest.mark.gpu

CASES = [
    ("c1", dict()),                                                       # BASELINE config 1: 10k / 256^2 / SH0
    ("c2", dict()),                                                       # BASELINE config 2 at its stated size: 100k / 640x480 / SH1
    ("c3", dict(num_points=30_000, width=500, height=300)),               # SH deg 3, ragged viewport (500 = 31.25 tiles)
    ("c1", dict(num_points=3_000, width=97, height=61, sh_deg=2, s0=0.05)),  # big splats, odd viewport, deg 2
]


def _forward_pair(orc, hip_device, base, kw):
    cfg = harness.small_config(base, **kw)
    g, sh, cam = harness.scene(cfg)
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    ref = orc.forward(g, sh, cam, st, ti)
    pipe = harness.HipPipeline(hip_device, cfg, g, sh, cam)
    pipe.forward()
    got = pipe.collect_forward()
    return cfg, g, sh, cam, st, ti, ref, pipe, got


@pytest.mark.parametrize("base,kw", CASES)
def test_forward_stages_bit_exact(orc, hip_device, base, kw):
    cfg, g, sh, cam, st, ti, ref, pipe, got = _forward_pair(orc, hip_device, base, kw)
    try:
        vis = ref["tile_counts"] > 0
        assert vis.sum() > 0
        assert_bits_equal(got["tile_counts"], ref["tile_counts"], "tile_counts (K1)")
        assert_bits_equal(got["splats"][vis], ref["splats"][vis], "splats of visible Gaussians (K1)")
        assert_bits_equal(got["depths"][vis], ref["depths"][vis], "depths (K1)")
        assert int(got["stats"][1]) == int(vis.sum()), "visible_gaussians"
        assert_bits_equal(got["tile_offsets"], ref["tile_offsets"], "per-Gaussian offsets (scan)")
        assert got["total_entries"] == ref["total_entries"]
        assert int(got["stats"][2]) == 0, "overflow flag"
        assert_bits_equal(got["sorted_keys"], ref["sorted_keys"][:ref["total_entries"]], "sorted keys")
        assert_bits_equal(got["sorted_values"], ref["sorted_values"][:ref["total_entries"]], "sorted values (stable order)")
        assert_bits_equal(got["tile_ranges"], ref["tile_ranges"], "tile ranges")
        assert_bits_equal(got["n_contrib"], ref["n_contrib"], "n_contrib (K14)")
        assert_bits_equal(got["final_T"], ref["final_T"], "final T (K14)")
        assert_bits_equal(got["rgba8"], ref["rgba8"], "rgba8 (K14)")
    finally:
        pipe.destroy()


@pytest.mark.parametrize("base,kw", CASES[:3])
def test_train_step_bit_exact(orc, hip_device, base, kw):
    """fwd + loss + backward + Adam + re-pack, two consecutive steps (the second sees non-zero moments)."""
    cfg = harness.small_config(base, **kw)
    g, sh, cam = harness.scene(cfg)
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    tg, tsh = synth.make_target_scene(g, sh)
    target = orc.forward(tg, tsh, cam, st, ti)["rgba8"]
    ref_g, ref_sh = g.copy(), sh.copy()
    ref_state = orc.unpack(ref_g, ref_sh)
    pipe = harness.HipPipeline(hip_device, cfg, g, sh, cam)
    tbuf = hip_device.bufferFrom(target)
    try:
        for step in range(2):
            ref = orc.train_step(ref_g, ref_sh, ref_state, cam, st, ti, target)
            pipe.train_step(tbuf)
            hip_device.synchronize()
            n = cfg.num_points
            assert_bits_equal(pipe.bwd.getLossTextureView().read(np.float32).reshape(cfg.height, cfg.width, 4), ref["loss_grad"], f"loss gradient image, step {step}")
            gm, gc, go, gcol = harness.acc_to_reference_layout(pipe.bwd.getAccumulatorsBuffer().read(np.int32), n)
            assert_bits_equal(gm, ref["grad_means"], f"grad_means_2d accumulators, step {step}")
            assert_bits_equal(gc, ref["grad_conics"], f"grad_conics accumulators, step {step}")
            assert_bits_equal(go, ref["grad_opacity"], f"grad_opacity accumulators, step {step}")
            assert_bits_equal(gcol, ref["grad_colors"], f"grad_colors accumulators, step {step}")
            assert_bits_equal(pipe.bwd.getGradientsBuffer().read(np.uint32).reshape(-1, 8)[:n], ref["gradients"], f"packed gradients (K17), step {step}")
            got_state = pipe.read_state()
            for k in ref_state:
                assert_bits_equal(got_state[k], ref_state[k], f"optimizer state {k}, step {step}")
            assert_bits_equal(pipe.pc.gaussian_3d_buffer.read(np.uint32).reshape(-1, 6), ref_g, f"re-packed Gaussians, step {step}")
            assert_bits_equal(pipe.pc.sh_buffer.read(np.uint32).reshape(-1, 24), ref_sh, f"re-packed SH, step {step}")
        assert pipe.opt.getIteration() == 2
    finally:
        pipe.destroy()


@pytest.mark.parametrize("env", [dict(WDGS_BWR_SUMS="butterfly"), dict(WDGS_BWR_WPW="4", WDGS_RASTER_WPW="1"), dict(WDGS_BWR_PRIO="0", WDGS_FWR_PRIO="0"),
                                 dict(WDGS_BWR_LONG="32", WDGS_FWR_LONG="32")],
                         ids=["register-butterfly", "other-workgroup-shapes", "no-issue-priorities", "long-list-passes"])
def test_alternative_kernel_forms_stay_bit_exact(env):
    """The forms kept for same-box A/B measurements -- backward_rasterize's register-only reduction (round 2) and the other
    waves-per-workgroup shapes of the two rasterization kernels, and the rasterization kernels without the issue priorities they set on grids
    that fit the chip (every case of this file does), and the opt-in passes of both rasterization kernels over long tile lists (here: every list
    above 32 entries, i.e. nearly all of c1's) -- are selected by environment variables that the library reads once per
    process: the train-step parity case runs again in a child process under each."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_parity.py"), "-x", "-q", "-m", "gpu", "-k", "test_train_step_bit_exact and c1"],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
