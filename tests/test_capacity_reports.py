"""CPU: the mailbox for capacity reports that reached the wrong owner (ops.CapacityReports; bindings/ts/webdgs_hip.js mirrors it).  The library reports
a truncated tile-entry list device-wide, to whoever waits first (csrc/api.hip: deferred_checks); a Trainer and a Viewer on one device each keep the
reports about their own passes and leave the others for the owner (dropped, such a report was lost to the owner)."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _report(*passes):
    return RuntimeError("[wdgs -3] tile entries overflow: " + "; ".join(f"{n} entries needed, max_tile_entries = 1048576 (forward pass {h:#x})" for n, h in passes)
                        + " (raise wdgs_tiled_forward_config.max_tile_entries)")


def test_reports_wait_for_the_owner_of_the_passes_they_name():
    from webdgs_amd.ops import CapacityReports
    box = CapacityReports(keep=3)
    trainer_passes, viewer_pass = [0x5000, 0x5100, 0x5200], [0x7000]
    assert CapacityReports.passes_named(_report((2_000_000, 0x5100), (1_500_000, 0x7000))) == {0x5100, 0x7000}
    assert box.take(trainer_passes) is None
    # the viewer's read consumed a report about the trainer's pass: left for the trainer, not for the viewer
    box.post(_report((2_000_000, 0x5100)))
    assert box.take(viewer_pass) is None
    e = box.take(trainer_passes)
    assert e is not None and "0x5100" in str(e) and box.take(trainer_passes) is None
    # a report that names passes of both is the first owner's who looks; the oldest matching one comes first; only `keep` are kept
    for n in (1, 2, 3, 4):
        box.post(_report((n, 0x7000)))
    assert len(box.pending) == 3
    assert "1 entries" not in str(box.pending[0]) and "2 entries" in str(box.take(viewer_pass))
    # a step skipped on every rank names no pass: nobody's
    box.post(RuntimeError("an optimizer step was skipped on every rank"))
    assert box.take(trainer_passes) is None


def test_the_node_host_keeps_the_same_mailbox():
    node = shutil.which("node")
    if not node or not os.path.exists(os.path.join(ROOT, "bindings", "napi", "webdgs_napi.node")):
        pytest.skip("node or the N-API addon is not available")
    src = """
      const { CapacityReports } = require(process.argv[1]);
      const rep = (n, h) => new Error(`tile entries overflow: ${n} entries needed, max_tile_entries = 1048576 (forward pass 0x${h.toString(16)})`);
      const box = new CapacityReports(3), out = [];
      box.post(rep(2000000, 0x5100));
      out.push(box.take([0x7000n]) === null, /0x5100/.test(box.take([0x5000n, 0x5100n]).message), box.take([0x5100n]) === null);
      for (const n of [1, 2, 3, 4]) box.post(rep(n, 0x7000));
      out.push(box.pending.length === 3, /^tile entries overflow: 2 /.test(box.take([0x7000]).message));
      console.log(JSON.stringify(out));
    """
    r = subprocess.run([node, "-e", src, os.path.join(ROOT, "bindings", "ts", "webdgs_hip.js")], capture_output=True, text=True, timeout=120)
    if r.returncode != 0 and "libamdhip64" in r.stderr + r.stdout:
        pytest.skip("the addon needs the HIP runtime to load")
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1]) == [True] * 5
