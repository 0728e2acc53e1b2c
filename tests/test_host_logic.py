"""CPU: host-side logic that needs no GPU -- synthetic scene generator contract, camera block, densify schedule."""
import math
import os

import pytest

import numpy as np

from webdgs_amd import synth


def test_generator_is_deterministic_and_prefix_stable():
    cfg = synth.CONFIGS["c1"]
    g1, s1 = synth.make_gaussians(cfg, 500)
    g2, s2 = synth.make_gaussians(cfg, 800)
    assert np.array_equal(g1, g2[:500]) and np.array_equal(s1, s2[:500]), "first N Gaussians do not depend on N"
    assert g1.dtype == np.uint32 and g1.shape == (500, 6) and s1.shape == (500, 24)
    h = g1.view(np.float16).astype(np.float32).reshape(-1, 12)
    assert np.all((h[:, 2] >= 2.0) & (h[:, 2] <= 10.0)), "z ~ U[2,10]"
    assert np.allclose(np.linalg.norm(h[:, 4:8], axis=1), 1.0, atol=2e-3), "unit quaternions (fp16)"
    assert np.all(h[:, 11] == 0), "pad half is zero"
    assert np.all(s1.view(np.float16).astype(np.float32).reshape(-1, 48)[:, 3:] == 0), "SH deg 0: only DC is non-zero"


def test_splitmix64_known_answers():
    # first outputs of splitmix64 with seed 0 (public reference values)
    out = synth._splitmix64(0, 3)
    assert [int(x) for x in out] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


def test_camera_block_matches_reference_conventions():
    cfg = synth.CONFIGS["c2"]
    cam = synth.identity_camera(cfg)
    assert cam.shape == (68,) and cam.dtype == np.float32
    proj = cam[32:48].reshape(4, 4).T  # row-major view of the column-major block
    p = proj @ np.array([0.3, -0.2, 5.0, 1.0])
    assert math.isclose(p[3], 5.0, rel_tol=1e-6), "clip.w = view z (camera.ts:40-49)"
    assert p[1] > 0, "ndc y is flipped"
    assert math.isclose(p[2] / p[3], 100 / (100 - 0.01) * (1 - 0.01 / 5.0), rel_tol=1e-5)
    assert tuple(cam[64:68]) == (cfg.width, cfg.height, cfg.fy, cfg.fy)
    assert np.allclose(cam[0:16].reshape(4, 4), np.eye(4)) and np.allclose(cam[16:32].reshape(4, 4), np.eye(4))
    cams = synth.circle_cameras(cfg, 8)
    for c in cams:
        view = c[0:16].reshape(4, 4).T
        assert np.allclose(view[:3, :3] @ view[:3, :3].T, np.eye(3), atol=1e-6)
        assert np.allclose(view @ c[16:32].reshape(4, 4).T, np.eye(4), atol=1e-5)
        centre = c[16:32].reshape(4, 4).T[:3, 3]
        assert math.isclose(float(np.linalg.norm(centre)), 1.0, rel_tol=1e-5)
        t = view @ np.array([0, 0, 6.0, 1.0])
        assert abs(t[0]) < 1e-5 and abs(t[1]) < 1e-5 and t[2] > 0, "looks at (0,0,6)"


def test_densify_schedule_matches_reference_formula():
    """trainer.ts:593-601 (shouldDensify) and 550-565 (getNextDensifyPruneIteration), restated on plain integers."""
    from webdgs_amd.trainer import Trainer

    class T(Trainer):  # bypass device construction: only the schedule arithmetic is exercised
        def __init__(self):
            self.iteration = 0
            self.densifyPruneConfig = dict(schedule=dict(enabled=True, warmupIterations=500, interval=100, stopIterations=15_000))

    t = T()
    fired = []
    for it in range(0, 1300):
        t.iteration = it
        nxt = it + 1
        s = t.densifyPruneConfig["schedule"]
        if s["enabled"] and s["warmupIterations"] <= nxt <= s["stopIterations"] and (nxt == 500 or (nxt - 500) % 100 == 0):
            fired.append(nxt)
        n = t.getNextDensifyPruneIteration()
        assert n is not None and n > it - 0 and (n == 500 or (n - 500) % 100 == 0)
        assert n == (500 if it < 500 else 500 + math.ceil((it + 1 - 500) / 100) * 100)
    assert fired == list(range(500, 1301, 100))
    t.iteration = 15_000
    assert t.getNextDensifyPruneIteration() is None
    t.densifyPruneConfig["schedule"]["enabled"] = False
    t.iteration = 10
    assert t.getNextDensifyPruneIteration() is None


def test_js_host_modules_load_and_mirror_the_reference_surface():
    """bindings/ts/*.js (the TypeScript-side host: operator classes + Trainer over the N-API addon) load on a CPU-only box and export
    the reference's names; the JS metrics camera equals the Python one bit for bit (pure host math, no GPU)."""
    import json
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    node = shutil.which("node")
    if not node or not os.path.exists(os.path.join(root, "bindings", "napi", "webdgs_napi.node")):
        pytest.skip("node or the N-API addon is not available")
    js = ("const hip=require('./bindings/ts/webdgs_hip.js'); const tr=require('./bindings/ts/trainer.js');"
          "const own=(c)=>Object.getOwnPropertyNames(c.prototype);"
          "console.log(JSON.stringify({hip:Object.keys(hip), trainer:own(tr.Trainer), fwd:own(hip.TiledForwardPass), dens:own(hip.DensifyPrunePass),"
          " cam:Array.from(new Uint32Array(tr.cameraBlockFor(new Float32Array(JSON.parse(process.argv[1])),320,240).buffer))}))")
    from webdgs_amd import synth
    from webdgs_amd.trainer import Trainer
    cam = synth.circle_cameras(synth.CONFIGS["c2"], 4)[1]
    out = json.loads(subprocess.check_output([node, "-e", js, json.dumps([float(x) for x in cam])], text=True, cwd=root))
    for name in ("TiledForwardPass", "TiledRasterizer", "TiledBackwardPass", "Optimizer", "DensifyPrunePass", "get_prefix_scanner", "get_dynamic_sorter",
                 "allocatePointCloudLike", "allocateOptimizerStateBuffers"):
        assert name in out["hip"], name
    for m in ("setPointCloud", "requestPointCloudSwap", "consumePointCloudSwapRequest", "requestResizeTo", "applyPointCloudSwap", "setDataset", "getTrainingConfig",
              "setTrainingConfig", "getOptimizerHyperparameters", "setOptimizerHyperparameters", "setDensifyPruneConfig", "start", "stop", "getIsTraining",
              "setMaxIterations", "getMaxIterations", "getIteration", "getPointCount", "getLastStepMs", "getItersPerSec", "getLastDensifyPruneIteration",
              "getNextDensifyPruneIteration", "step"):   # trainer.ts:177-566
        assert m in out["trainer"], m
    for m in ("encodeDecision", "encodePrefixSum", "encodeCapToMax", "encodeTotalOut", "encodePrepare", "encodeScatter", "ensureSize", "setConfig", "getConfig"):
        assert m in out["dens"], m
    assert np.array_equal(np.array(out["cam"], np.uint32), Trainer.metrics_camera(cam, 320, 240).view(np.uint32))
