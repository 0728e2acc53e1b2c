"""CPU: the oracle reproduces the committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py)
and satisfies properties that do not depend on a second implementation."""
import os

import numpy as np
import pytest

from webdgs_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "train_step.npz"))


@pytest.fixture(scope="module")
def gold_densify():
    return np.load(os.path.join(HERE, "golden", "densify.npz"))


def test_oracle_matches_golden_train_step(orc, gold):
    g, sh = gold["in_gaussians"].copy(), gold["in_sh"].copy()
    state = orc.unpack(g, sh)
    for k in state:
        assert np.array_equal(state[k].view(np.uint32), gold["state0_" + k].view(np.uint32)), k
    r = orc.train_step(g, sh, state, gold["camera"], gold["settings"], gold["tile_info"], gold["target"])
    e = r["total_entries"]
    vis = gold["tile_counts"] > 0
    assert np.array_equal(r["tile_counts"], gold["tile_counts"])
    assert np.array_equal(r["splats"][vis], gold["splats"][vis])
    assert np.array_equal(r["sorted_keys"][:e], gold["sorted_keys"]) and np.array_equal(r["sorted_values"][:e], gold["sorted_values"])
    for k in ("tile_ranges", "rgba8", "n_contrib", "grad_means", "grad_conics", "grad_opacity", "grad_colors", "gradients"):
        assert np.array_equal(r[k], gold[k]), k
    for k in ("final_T", "loss_grad"):
        assert np.array_equal(r[k].view(np.uint32), gold[k].view(np.uint32)), k
    assert np.array_equal(g, gold["out_gaussians"]) and np.array_equal(sh, gold["out_sh"])
    for k in state:
        assert np.array_equal(state[k].view(np.uint32), gold["state1_" + k].view(np.uint32)), k


def test_forward_structure_properties(orc, gold):
    cfg_ti = gold["tile_info"]
    tiles = int(cfg_ti[2])
    counts, offsets = gold["tile_counts"], gold["tile_offsets"]
    keys, vals, ranges = gold["sorted_keys"], gold["sorted_values"], gold["tile_ranges"]
    e = keys.shape[0]
    assert int(counts.astype(np.uint64).sum()) == e == int(gold["stats"][0])
    assert np.array_equal(offsets, np.concatenate([[0], np.cumsum(counts.astype(np.uint64))[:-1]]).astype(np.uint32))
    assert np.all(keys[:-1] <= keys[1:]), "sorted keys are non-decreasing"
    # stable: equal keys keep ascending Gaussian index
    same = keys[:-1] == keys[1:]
    assert np.all(vals[:-1][same] < vals[1:][same])
    # permutation: every Gaussian appears exactly tile_counts times
    assert np.array_equal(np.bincount(vals, minlength=counts.shape[0]).astype(np.uint32), counts)
    # ranges partition [0, E): first index of each present tile, MAX for empty, E at the end
    tile_of = (keys >> 16) - 1
    assert ranges[tiles] == e
    present = np.unique(tile_of)
    for t in range(tiles):
        if t in present:
            assert ranges[t] == np.flatnonzero(tile_of == t)[0]
        else:
            assert ranges[t] == 0xFFFFFFFF
    # depth half of the key is the top 16 bits of the ordered-uint view depth (Q5)
    assert np.array_equal(keys & 0xFFFF, gold["depths"][vals] >> 16)


def test_final_T_round_trip_fp64(orc, gold):
    """Recompute the composite in float64 from the fp16 splats and the sorted lists: colour within 1 LSB, T within 1e-5,
    n_contrib exact away from the alpha threshold."""
    st = gold["settings"]
    W, H = int(st[2]), int(st[3])
    ntx = int(gold["tile_info"][0])
    sp = gold["splats"].view(np.float16).astype(np.float64).reshape(-1, 12)
    keys, vals, ranges = gold["sorted_keys"], gold["sorted_values"], gold["tile_ranges"]
    tile_of = (keys >> 16).astype(np.int64) - 1
    worst_t, worst_c = 0.0, 0
    rng = np.random.default_rng(0)
    for _ in range(300):
        x, y = int(rng.integers(0, W)), int(rng.integers(0, H))
        t = (y // 16) * ntx + x // 16
        if ranges[t] == 0xFFFFFFFF:
            assert gold["n_contrib"][y, x] == 0 and gold["final_T"][y, x] == 1.0
            continue
        idx = np.flatnonzero(tile_of == t)
        A, C = 0.0, np.zeros(3)
        for e in idx:
            s = sp[vals[e]]
            cx, cy = (s[0] * 0.5 + 0.5) * W, (s[1] * -0.5 + 0.5) * H
            dx, dy = x + 0.5 - cx, y + 0.5 - cy
            if abs(dx) > min(s[2], 128.0) or abs(dy) > min(s[3], 128.0) or A > 0.99:
                continue
            q = s[4] * dx * dx + 2 * s[5] * dx * dy + s[6] * dy * dy
            a = min(max(np.exp(-0.5 * q) * s[11], 0.0), 0.99)
            C += s[8:11] * a * (1 - A)
            A += a * (1 - A)
        worst_t = max(worst_t, abs((1 - A) - float(gold["final_T"][y, x])))
        worst_c = max(worst_c, int(np.abs(np.floor(np.clip(C, 0, 1) * 255 + 0.5) - gold["rgba8"][y, x, :3].astype(np.float64)).max()))
    assert worst_t < 1e-5 and worst_c <= 1, (worst_t, worst_c)


def test_backward_matches_finite_differences_fp64(orc):
    """K16's per-splat gradients against central differences of a float64 composite over one tile (self-consistent up to
    SURVEY Q7: backward skips alpha < 1/255 and recovers T by division).  Tolerance 2% of the largest gradient + 2e-4."""
    rng = np.random.default_rng(5)
    n, W, H = 12, 16, 16
    ndc = rng.uniform(-0.7, 0.7, (n, 2))
    ext = rng.uniform(6.0, 14.0, (n, 2))
    sx, sy, rho = rng.uniform(2.0, 5.0, n), rng.uniform(2.0, 5.0, n), rng.uniform(-0.5, 0.5, n)
    cov = np.stack([sx * sx, rho * sx * sy, sy * sy], 1)
    det = cov[:, 0] * cov[:, 2] - cov[:, 1] ** 2
    conic = np.stack([cov[:, 2] / det, -cov[:, 1] / det, cov[:, 0] / det], 1)
    col, op = rng.uniform(0.1, 0.9, (n, 3)), rng.uniform(0.3, 0.8, n)
    h = np.zeros((n, 12), np.float32)
    h[:, 0:2], h[:, 2:4], h[:, 4:6], h[:, 6], h[:, 8:11], h[:, 11] = ndc, ext, conic[:, :2], conic[:, 2], col, op
    splats = np.ascontiguousarray(synth.f32_to_f16_bits(h)).view(np.uint32).reshape(n, 6)
    p = splats.view(np.float16).astype(np.float64).reshape(n, 12)
    gl = rng.uniform(-1, 1, (H, W, 4)).astype(np.float32)
    settings = np.array([1, 0, W, H, 3, 1, 128], np.float32)
    tinfo = synth.tile_info(W, H, 0)
    keys = np.full(n, 1 << 16, np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    ranges = np.array([0, n], np.uint32)
    rgba, T, nc = orc.rasterize(settings, tinfo, splats, ranges, keys, vals, n)
    bs = settings.copy(); bs[5] = 0
    gm, gc, go, gcol = orc.backward_rasterize(bs, n, ranges, vals, splats, T, nc, gl)

    def loss(pp):
        L = 0.0
        for y in range(H):
            for x in range(W):
                A, C = 0.0, np.zeros(3)
                for s in pp:
                    dx, dy = x + 0.5 - (s[0] * 0.5 + 0.5) * W, y + 0.5 - (s[1] * -0.5 + 0.5) * H
                    if abs(dx) > s[2] or abs(dy) > s[3]:
                        continue
                    a = min(np.exp(-0.5 * (s[4] * dx * dx + 2 * s[5] * dx * dy + s[6] * dy * dy)) * s[11], 0.99)
                    C += s[8:11] * a * (1 - A)
                    A += a * (1 - A)
                L += float((C * gl[y, x, :3].astype(np.float64)).sum())
        return L

    def fd(i, j, eps):
        a, b = p.copy(), p.copy()
        a[i, j] += eps; b[i, j] -= eps
        return (loss(a) - loss(b)) / (2 * eps)

    got = dict(conic_x=gc.reshape(n, 4)[:, 0] / 1e6, conic_y=gc.reshape(n, 4)[:, 1] / 1e6, conic_z=gc.reshape(n, 4)[:, 3] / 1e6, opacity=go / 1e6,
               r=gcol.reshape(n, 3)[:, 0] / 1e6, mean_x=gm.reshape(n, 2)[:, 0] / 1e6, mean_y=gm.reshape(n, 2)[:, 1] / 1e6)
    cols = dict(conic_x=(4, 1e-4, 1.0), conic_y=(5, 1e-4, 1.0), conic_z=(6, 1e-4, 1.0), opacity=(11, 1e-4, 1.0), r=(8, 1e-4, 1.0),
                mean_x=(0, 1e-5, 1.0 / (0.5 * W)), mean_y=(1, 1e-5, 1.0 / (-0.5 * H)))  # d/d(px) = d/d(ndc) / (+-0.5*viewport)
    for name, (j, eps, scale) in cols.items():
        ref = np.array([fd(i, j, eps) * scale for i in range(n)])
        tol = 0.02 * np.abs(ref).max() + 2e-4
        assert np.abs(got[name] - ref).max() < tol, (name, np.abs(got[name] - ref).max(), tol)


def test_oracle_matches_golden_densify(orc, gold_densify):
    d = gold_densify
    g, sh = d["in_gaussians"], d["in_sh"]
    st = {k: d["in_" + k].copy() for k in ("opt_pos", "opt_rot", "opt_scale", "opt_opacity", "param_sh", "state_sh")}
    err, mm, flags = orc.metric_map(d["metric_rgba8"], d["gt_small"], 0.5)
    assert np.array_equal(err, d["metric_err"]) and np.array_equal(mm, d["metric_minmax"]) and np.array_equal(flags, d["metric_flags"])
    counts = d["metric_counts"].copy()
    orc.metric_normalize(counts, 1)
    prep = orc.densify_prepare(g, counts, int(d["max_out"][0]), clone_threshold=6, prune_opacity=0.15, split_scale=0.05)
    assert np.array_equal(prep["actions"], d["actions"]) and np.array_equal(prep["counts"], d["out_counts"]) and np.array_equal(prep["offsets"], d["out_offsets"])
    assert prep["total"] == int(d["total"][0])
    out_n = min(prep["total"], int(d["max_out"][0]))
    og, osh, ost = orc.densify_scatter(g, sh, st, prep, out_n)
    assert np.array_equal(og, d["out_gaussians"]) and np.array_equal(osh, d["out_sh"])
    for k in ost:
        assert np.array_equal(ost[k].view(np.uint32), d["out_" + k].view(np.uint32)), k


def test_densify_properties(gold_densify):
    d = gold_densify
    a, c, off = d["actions"], d["out_counts"], d["out_offsets"]
    assert set(np.unique(a)) <= {0, 1, 2, 3}
    assert np.all(c[a == 3] == 0) and np.all(c[a == 0] == 1) and np.all(c[(a == 1) | (a == 2)] == 2)
    assert int(c.sum()) == int(d["total"][0]) <= int(d["max_out"][0])
    assert np.array_equal(off, np.concatenate([[0], np.cumsum(c)[:-1]]).astype(np.uint32))
    # survivors' opacity is clamped to sigmoid^-1(0.8) = ln 4 at most; opacity moments are zero for everyone (Q16)
    assert np.all(d["out_opt_opacity"][:, 1:] == 0)
    assert d["out_opt_opacity"][:, 0].max() <= np.float32(1.38629436112)
    # split children: log-scale reduced by ln 1.6 in the masters
    src = np.repeat(np.arange(a.shape[0]), c)
    split_rows = a[src] == 2
    assert np.allclose(d["out_opt_scale"][split_rows, :3], d["in_opt_scale"][src[split_rows], :3] - np.float32(0.4700036292457356), atol=1e-6)
    # keep rows are verbatim copies (unless opacity was clamped)
    keep_rows = np.flatnonzero(a[src] == 0)
    same = np.all(d["out_gaussians"][keep_rows] == d["in_gaussians"][src[keep_rows]], axis=1)
    assert same.mean() > 0.5
