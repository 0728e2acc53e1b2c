"""GPU: the TypeScript-side host -- bindings/ts/trainer.js (the reference's ``Trainer`` rewritten over the N-API addon) driven by
node -- against the Python host on the same dataset and the same view draws: steps, a densify/prune rebuild in the middle, more
steps.  The trained cloud and all six optimizer-state arrays must be byte-identical (both hosts call the same C ABI; this pins the
binding layer, the recorded command buffers and the sequencing of the JS trainer).  Four forms of the step: the reference's one-view
step (fused K17 + Adam); a BATCHED step of two views on two lanes at pipeline depth 2; the same batched step through the sliced
exchange of the library's RCCL communicator in a world of one (what a rank of BASELINE config c4 runs); and the one-view step with
the packed gradient kept."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from webdgs_amd import ops, synth
from webdgs_amd.trainer import Trainer

import harness
from harness import assert_bits_equal
from test_gpu_trainer_oracle import _FixedViews

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


FORMS = {
    "one_view": dict(),
    "batched_two_lanes_depth2": dict(views_per_step=2, lanes=2, pipeline_depth=2),
    "batched_sliced_over_rccl": dict(views_per_step=2, lanes=2, comm="capi"),
    "one_view_gradients_kept": dict(keep_gradients=True),
    # long tile lists (csrc/longlist.h) for every tile above 40 entries, from scratch too small for them: no tile takes them until the densify event at
    # iteration 6 has enlarged it (growLongLists / _grow_long_lists), every long tile from then on -- the same frames in both hosts, the same bits either way
    "one_view_long_lists_grown_at_the_event": dict(pipeline_depth=2, long_lists=dict(threshold=40, maxItems=64, maxRows=65536)),
}


@pytest.mark.parametrize("form", list(FORMS))
def test_node_trainer_equals_python_trainer(hip_device, orc, tmp_path, form):
    opts = FORMS[form]
    vps = opts.get("views_per_step", 1)
    node = shutil.which("node")
    addon = os.path.join(ROOT, "bindings", "napi", "webdgs_napi.node")
    if not node or not os.path.exists(addon):
        pytest.skip("node or the N-API addon is not available")
    dev = hip_device
    cfg = harness.small_config("c2", num_points=5000, width=128, height=96, s0=0.01)
    g, sh, _ = harness.scene(cfg)
    tg, tsh = synth.make_target_scene(g, sh)
    cams = synth.circle_cameras(cfg, 4)
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    imgs = [orc.forward(tg, tsh, cams[i], st, ti)["rgba8"] for i in range(4)]
    dens = dict(schedule=dict(enabled=True, warmupIterations=6, interval=50, stopIterations=40), metricViews=3, metricDownscale=2, metricThreshold=0.5,
                cloneThresholdCount=5, splitScaleThreshold=0.03, pruneOpacity=0.2, maxNewPointsPerStep=300, maxBufferBytes=128 * 1024 * 1024)
    steps = 11
    train_views = [2, 0, 3, 1, 1, 2, 0, 3, 2, 2, 1]
    metric_views = {6: [1, 3, 0]}
    second = [1, 3, 0, 2, 3, 3, 1, 0, 0, 1, 2]   # the second view of a batched step
    draws = []
    for i in range(steps):
        draws.append(train_views[i])
        if vps == 2:
            draws.append(second[i])
        draws += metric_views.get(i + 1, [])
    g.tofile(tmp_path / "gaussians.bin")
    sh.tofile(tmp_path / "sh.bin")
    np.ascontiguousarray(cams, np.float32).tofile(tmp_path / "cameras.bin")
    np.stack(imgs).tofile(tmp_path / "images.bin")
    (tmp_path / "meta.json").write_text(json.dumps(dict(num_points=cfg.num_points, sh_deg=cfg.sh_deg, width=cfg.width, height=cfg.height, views=4, steps=steps,
                                                        draws=draws, densify=dens, **opts)))
    r = subprocess.run([node, os.path.join(ROOT, "bindings", "napi", "trainer_run.js"), str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "TRAINER_RUN_OK" in r.stdout, f"exit code {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    out = json.loads((tmp_path / "out_meta.json").read_text())
    assert out["scan_ok"] and out["sort_ok"], "get_prefix_scanner / get_dynamic_sorter through the addon"
    assert out["recorded_views"] >= 2, "the JS trainer replays recorded command buffers"
    assert out["stale_rows_seen"], "deferred SH writes: the rows are stale between hand-overs and a host read through the buffer brings them up to date"
    ov = out["overflow"]   # a step whose tile-entry list does not fit: same report, same guarantees as the Python host (test_gpu_pipeline.py)
    assert ov["code"] == "WDGS_E_CAPACITY" and ov["untouched"] and ov["usable_after"], ov
    assert ov["steps_until_error"] == (0 if opts.get("pipeline_depth", 1) == 1 else 1), "depth 1: the step itself throws; depth 2: the next one does"
    if vps == 2:
        assert out["lanes"] == 2 and out["op_sets"] == 2 and any(k.startswith("viewp/") for k in out["recorded_keys"]) and "adam" in out["recorded_keys"]
    if opts.get("comm") == "capi":
        assert "RCCL" in out["exchange"] and "apply" in out["recorded_keys"], "the sliced step ran through the library's communicator"

    from webdgs_amd import parallel
    exchange = parallel.CapiExchange(dev) if opts.get("comm") == "capi" else None
    t = Trainer(dev, seed=0, views_per_rank=vps, overlap_views=opts.get("lanes"), pipeline_depth=opts.get("pipeline_depth", 1), exchange=exchange)
    t.keep_gradients = bool(opts.get("keep_gradients"))
    if opts.get("long_lists"):
        t.longLists = dict(opts["long_lists"])
    t.setDensifyPruneConfig(dens)
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset([dict(camera=cams[i], width=cfg.width, height=cfg.height) for i in range(4)],
                 [dict(texture=dev.bufferFrom(imgs[i]), width=cfg.width, height=cfg.height) for i in range(4)])
    t.start()
    t._rng = _FixedViews(draws)
    sizes = [t.getPointCount()]
    try:
        for _ in range(steps):
            t.step()
            sizes.append(t.getPointCount())
        dev.synchronize()
        n = t.getPointCount()
        assert out["sizes"] == sizes and out["num_points"] == n and sizes[6] != sizes[5], (out["sizes"], sizes)
        assert out["iteration"] == t.getIteration() == steps and out["optimizer_iteration"] == t.optimizer.getIteration()
        assert out["last_densify"] == t.getLastDensifyPruneIteration() == 6 and out["next_densify"] == t.getNextDensifyPruneIteration()
        assert_bits_equal(np.fromfile(tmp_path / "out_gaussians.bin", np.uint32), t.pointCloud.gaussian_3d_buffer.read(np.uint32)[: n * 6], "node vs python: gaussians")
        assert_bits_equal(np.fromfile(tmp_path / "out_sh.bin", np.uint32), t.pointCloud.sh_buffer.read(np.uint32)[: n * 24], "node vs python: sh")
        words = dict(optPosBuffer=12, optRotBuffer=12, optScaleBuffer=12, optOpacityBuffer=3, paramSH=48, stateSH=96)
        for k, b in t.optimizer.getStateBuffers().items():
            assert_bits_equal(np.fromfile(tmp_path / f"out_state_{k}.bin", np.uint32), b.read(np.uint32)[: n * words[k]], f"node vs python: state {k}")
        if opts.get("long_lists"):
            mine, theirs = t.forwardPass.longListStats(), out["long_lists"]["stats"]
            assert {k: theirs.get(k) for k in mine} == mine, ("the last frame's long-list work, node vs python", theirs, mine)
            assert mine["maxItems"] > 64 and mine["itemsWanted"] <= mine["maxItems"] and mine["blocksWanted"] >= 4 and not mine["stalled"], mine
            assert mine["forwardQueue"] >= 2 * (mine["itemsWanted"] + mine["blocksWanted"]), ("the tasks of the last frame were all drawn", mine)
            assert out["long_lists"]["settings"]["maxItems"] == t.longLists["maxItems"] and out["long_lists"]["settings"]["maxRows"] == t.longLists["maxRows"]
        if opts.get("keep_gradients"):
            grads = np.fromfile(tmp_path / "out_gradients.bin", np.uint32)
            assert_bits_equal(grads, t.backwardPass.getGradientsBuffer().read(np.uint32)[: n * 8], "node vs python: packed gradients of the last step")
            assert grads.any(), "the fused step wrote the packed gradient"
    finally:
        t.destroy()
        if exchange is not None:
            exchange.destroy()
