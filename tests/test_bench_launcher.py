"""`python bench.py --gpus N` from a bare shell starts its N ranks itself: the parent spawns
`python -m torch.distributed.run ... bench.py` as a child BEFORE it imports torch or touches a GPU, relays rank 0's JSON line and
exits with the child's return code.  Here the ranks run the launcher's self-test body (a gloo all-reduce of ones) instead of the GPU
workload, so the launch path itself -- argument forwarding, rendezvous on 127.0.0.1, stdout relay, exit code, stderr tail -- is
covered on the CPU.  The GPU workload behind the same launcher is covered by tests/test_gpu_dp.py::test_bench_self_launches_two_ranks."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(mode: str, *flags: str):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["WDGS_BENCH_SELFTEST"] = mode
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, capture_output=True, text=True, timeout=300)


def test_bare_invocation_launches_its_ranks_and_relays_one_json_line():
    r = _run("1", "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["self_launched"] is True
    assert out["argv"] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]  # the child sees the parent's flags unchanged


def test_a_failing_rank_fails_the_parent_with_its_stderr_tail():
    r = _run("fail", "--gpus", "2")
    assert r.returncode != 0
    assert "this rank fails on purpose" in r.stderr and "exited with code" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_the_parent_never_imports_torch_before_launching():
    """The launch decision is taken from argv and the environment alone: importing bench and running the launcher must not pull in
    torch (any torch.cuda call in the parent would make it a GPU process that then starts GPU programs)."""
    code = ("import sys, os; sys.argv = ['bench.py', '--gpus', '2']; os.environ['WDGS_BENCH_SELFTEST'] = '1'; os.environ.pop('WORLD_SIZE', None)\n"
            "import bench\n"
            "bench.self_launch = lambda n, argv: (print('torch' in sys.modules, n), 0)[1]\n"
            "try:\n    bench.main()\nexcept SystemExit as e:\n    assert e.code == 0\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split() == ["False", "2"]


def test_single_rank_does_not_launch():
    r = _run("1", "--gpus", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 1 and out["self_launched"] is False


def test_the_newest_counter_profile_describes_the_kernel_sources_in_the_tree():
    """``bench.py`` takes the dominant kernel's VALU count and HBM traffic from the newest ``profiles/*_pmc.json`` whose recorded hash
    of that kernel's source file equals the working tree's, and reports ``frac`` null otherwise.  This is the
    reminder to collect the counter passes again (scripts/collect_profiles.sh) after the last edit of backward_raster.hip."""
    import glob
    import json
    import bench
    files = sorted(glob.glob(os.path.join(bench.ROOT, "profiles", "*_pmc.json")), key=os.path.basename, reverse=True)
    assert files, "no counter profile under profiles/"
    newest = json.load(open(files[0]))
    for kernel in ("backward_rasterize", "rasterize"):
        assert (newest.get("source_sha") or {}).get(kernel) == bench.source_sha(kernel), \
            f"{os.path.basename(files[0])} was collected for another version of {kernel}'s source: run scripts/collect_profiles.sh on the GPU and commit its profiles"
        assert newest["kernels"][kernel].get("SQ_INSTS_VALU", 0) > 0


def test_node_bench_starts_its_own_ranks_and_relays_rank_zero():
    """`node bindings/napi/bench.js --gpus N` from a bare shell: N child processes (RANK / WORLD_SIZE / LOCAL_RANK, one rendezvous path), rank 0's line
    relayed, the worst child's exit code returned -- before the parent has loaded the addon (the CPU selftest body stands in for the GPU workload)."""
    import shutil
    node = shutil.which("node")
    if not node:
        import pytest
        pytest.skip("node is not installed")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "WDGS_RENDEZVOUS")}
    bench = os.path.join(ROOT, "bindings", "napi", "bench.js")
    r = subprocess.run([node, bench, "--gpus", "3", "--config", "c2"], env=dict(env, WDGS_BENCH_SELFTEST="1"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["selftest"] and out["n_gpus"] == 3 and out["n_ranks_seen"] == 3 and out["self_launched"] is True and out["argv"] == ["--gpus", "3", "--config", "c2"]
    r = subprocess.run([node, bench, "--gpus", "2"], env=dict(env, WDGS_BENCH_SELFTEST="fail"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "rank 1 exited with 7" in r.stderr, (r.returncode, r.stderr[-1000:])
