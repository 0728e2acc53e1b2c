#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the ORACLE (not from the reference: it ships no fixtures and cannot run here --
DESIGN.md "Oracle").  The fixtures freeze the oracle's outputs so that (a) an accidental change to the restatement is
caught on CPU and (b) the HIP path is compared against committed data, not only against a freshly built oracle.

    python tests/golden/make_golden.py        # rewrites the .npz files next to this script
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle as orc  # noqa: E402
from webdgs_amd import synth  # noqa: E402

GOLDEN_CFG = synth.SceneConfig(7, 700, 80, 56, 3, 90.0, 0.02, "golden")


def build():
    cfg = GOLDEN_CFG
    g, sh = synth.make_gaussians(cfg)
    cam = synth.circle_cameras(cfg, 3)[1]
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    tg, tsh = synth.make_target_scene(g, sh)
    target = orc.forward(tg, tsh, cam, st, ti)["rgba8"]
    g1, sh1 = g.copy(), sh.copy()
    state = orc.unpack(g1, sh1)
    state0 = {k: v.copy() for k, v in state.items()}
    r = orc.train_step(g1, sh1, state, cam, st, ti, target)
    fw = dict(in_gaussians=g, in_sh=sh, camera=cam, settings=st, tile_info=ti, target=target,
              splats=r["splats"], depths=r["depths"], tile_counts=r["tile_counts"], tile_offsets=r["tile_offsets"], stats=r["stats"],
              sorted_keys=r["sorted_keys"][:r["total_entries"]], sorted_values=r["sorted_values"][:r["total_entries"]], tile_ranges=r["tile_ranges"],
              rgba8=r["rgba8"], final_T=r["final_T"], n_contrib=r["n_contrib"])
    bw = dict(loss_grad=r["loss_grad"], grad_means=r["grad_means"], grad_conics=r["grad_conics"], grad_opacity=r["grad_opacity"], grad_colors=r["grad_colors"],
              gradients=r["gradients"], out_gaussians=g1, out_sh=sh1, **{"state0_" + k: v for k, v in state0.items()}, **{"state1_" + k: v for k, v in state.items()})
    np.savez_compressed(os.path.join(HERE, "train_step.npz"), **fw, **bw)

    # densify: metric map on a half-resolution view, counts, decide/cap/scan/total, scatter
    mw, mh = cfg.width // 2, cfg.height // 2
    mst, mti = synth.render_settings(cfg, mw, mh), synth.tile_info(mw, mh, 0)
    mcam = synth.camera_block(cam[0:16].reshape(4, 4).T.astype(np.float64), mw, mh, cfg.fy * mh / cfg.height)
    mfw = orc.forward(g1, sh1, mcam, mst, mti)
    gt_small = orc.downsample_bilinear(target, mw, mh)
    err, mm, flags = orc.metric_map(mfw["rgba8"], gt_small, 0.5)
    counts = np.zeros(cfg.num_points, np.uint32)
    bst = mst.copy(); bst[5] = 0.0
    cap = max(int(mfw["total_entries"]), 1)
    orc.metric_count(bst, mfw["tile_ranges"], mfw["sorted_values"][:cap].copy(), cap, mfw["splats"], flags, mfw["n_contrib"], counts)
    counts_raw = counts.copy()
    orc.metric_normalize(counts, 1)
    max_out = cfg.num_points + 40
    prep = orc.densify_prepare(g1, counts, max_out, clone_threshold=6, prune_opacity=0.15, split_scale=0.05)
    out_n = min(prep["total"], max_out)
    og, osh, ost = orc.densify_scatter(g1, sh1, state, prep, out_n)
    np.savez_compressed(os.path.join(HERE, "densify.npz"), in_gaussians=g1, in_sh=sh1, metrics_camera=mcam, metrics_settings=mst, gt_small=gt_small,
                        metric_rgba8=mfw["rgba8"], metric_err=err, metric_minmax=mm, metric_flags=flags, metric_counts=counts_raw, actions=prep["actions"],
                        out_counts=prep["counts"], out_offsets=prep["offsets"], total=np.array([prep["total"]], np.uint32), max_out=np.array([max_out], np.uint32),
                        out_gaussians=og, out_sh=osh, **{"in_" + k: v for k, v in state.items()}, **{"out_" + k: v for k, v in ost.items()})
    return prep


if __name__ == "__main__":
    p = build()
    a = p["actions"]
    print("golden written: keep/clone/split/prune =", [(a == i).sum() for i in range(4)], "total", p["total"])
