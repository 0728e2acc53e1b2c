#!/usr/bin/env python3
"""Dev probe for backward_rasterize (K16): how full are its waves?   python scripts/bwr_lane_utilisation.py [config] [tiles]

Takes the buffers of one forward pass (splats, sorted tile lists, ranges, n_contrib), evaluates on the host, for a sample of tiles,
which (pixel, entry) pairs contribute (entry position < n_contrib[pixel], inside the extent box, alpha >= 1/255), and reports for several
pixel-region shapes a wave could own how many (region, entry) iterations have at least one contributing pixel and how many of the region's
pixels contribute in them on average.  The kernel's cost is iterations x (shared + per-pixel-slot work); the pairs are fixed."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from webdgs_amd import ops, synth  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    n_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    cfg = synth.CONFIGS[name]
    dev = ops.HipDevice(0)
    g, sh = synth.make_gaussians(cfg)
    cam = synth.circle_cameras(cfg, 4)[0]
    pc = ops.createPointCloud(dev, g, sh, cfg.sh_deg)
    cb = dev.bufferFrom(cam)
    fw = ops.TiledForwardPass(dev, pc, cb, dict(viewportWidth=cfg.width, viewportHeight=cfg.height))
    rs = ops.TiledRasterizer(dict(device=dev, forwardPass=fw))
    fw.encode(None)
    rs.encode(None, cfg.width, cfg.height)
    dev.synchronize()
    E = int(fw.check()[0])
    W, H = cfg.width, cfg.height
    tx, ty = (W + 15) // 16, (H + 15) // 16
    ranges = rs.getTileOffsetsBuffer().read(np.uint32, count=tx * ty + 1)
    inst = fw.getSortedIndicesBuffer().read(np.uint32, count=E)
    splats = fw.getResources()["splatBuffer"].read(np.uint32).reshape(-1, 6)
    ncon = rs.getNContribTextureView().read(np.uint32, count=W * H).reshape(H, W)

    def lo(w):
        return (w & 0xFFFF).astype(np.uint16).view(np.float16).astype(np.float32)

    def hi(w):
        return (w >> 16).astype(np.uint16).view(np.float16).astype(np.float32)

    rng = np.random.default_rng(0)
    tiles = rng.choice(tx * ty, size=min(n_tiles, tx * ty), replace=False)
    shapes = {"8x8 (now)": (8, 8), "16x8": (16, 8), "8x16": (8, 16), "16x16": (16, 16), "4x4": (4, 4)}
    it = {k: 0 for k in shapes}
    pairs = 0
    entries_total = 0
    hist = np.zeros(65, np.int64)
    rows_it = rows_cull_it = same_all = 0  # 4 independent 4x4 quadrants per wave: iterations = per chunk, the longest quadrant list
    # lane-per-splat systolic form: the block's KEPT splats sit on lanes (up to 64 at a time, compacted across the
    # block's chunks), the 64 pixel states rotate through them back to front; a group of k splats takes 64 + k - 1 steps to fill and drain
    sys_steps = sys_steps_ideal = sys_groups = 0
    chunks = walked = in_box = reach = 0  # per 8x8 block: 64-entry chunks walked, entries walked, entries passing the box test, entries with alpha >= 1/255 somewhere
    for t in tiles:
        a, b = int(ranges[t]), int(ranges[t + 1])
        if b <= a:
            continue
        x0, y0 = (t % tx) * 16, (t // tx) * 16
        s = splats[inst[a:b]]
        cx = (lo(s[:, 0]) * 0.5 + 0.5) * W
        cy = (hi(s[:, 0]) * -0.5 + 0.5) * H
        ex, ey = np.minimum(lo(s[:, 1]), 128.0), np.minimum(hi(s[:, 1]), 128.0)
        A, B, C = lo(s[:, 2]), hi(s[:, 2]), lo(s[:, 3])
        op = hi(s[:, 5])
        px = (x0 + np.arange(16) + 0.5)[None, :, None]   # [1, x, 1]
        py = (y0 + np.arange(16) + 0.5)[:, None, None]   # [y, 1, 1]
        dx, dy = px - cx[None, None, :], py - cy[None, None, :]
        q = A * dx * dx + 2 * B * dx * dy + C * dy * dy
        alpha = np.minimum(0.99, op * np.exp(-0.5 * q))
        nc = np.zeros((16, 16), np.uint32)
        hh, ww = min(16, H - y0), min(16, W - x0)
        nc[:hh, :ww] = ncon[y0:y0 + hh, x0:x0 + ww]
        act = (np.arange(b - a)[None, None, :] < nc[:, :, None]) & (np.abs(dx) <= ex) & (np.abs(dy) <= ey) & (alpha >= 1.0 / 255.0)
        pairs += int(act.sum())
        entries_total += b - a
        for k, (sw, sh_) in shapes.items():
            r = act.reshape(16 // sh_, sh_, 16 // sw, sw, -1).any(axis=(1, 3))
            it[k] += int(r.sum())
        ent = np.arange(b - a)
        for by in range(2):
            for bx in range(2):
                wmax = min(int(nc[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8].max()), b - a)
                chunks += (wmax + 63) // 64
                walked += wmax
                X0, X1, Y0, Y1 = x0 + bx * 8 + 0.5, x0 + bx * 8 + 7.5, y0 + by * 8 + 0.5, y0 + by * 8 + 7.5
                box = (ent < wmax) & ~((X0 - cx > ex) | (cx - X1 > ex) | (Y0 - cy > ey) | (cy - Y1 > ey))
                in_box += int(box.sum())
                reach += int((box & (alpha[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8, :] >= 1.0 / 255.0).any(axis=(0, 1))).sum())
                kept = int((box & (alpha[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8, :] >= 1.0 / 255.0).any(axis=(0, 1))).sum())
                full, rest = divmod(kept, 64)
                sys_groups += full + (1 if rest else 0)
                sys_steps += full * (64 + 63) + ((64 + rest - 1) if rest else 0)   # every group fills and drains on its own
                sys_steps_ideal += kept + (63 if kept else 0)                       # groups chained without a bubble (two groups resident): one drain per block
                blk_act = act[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8, :]
                blk_alpha = alpha[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8, :] >= 1.0 / 255.0
                quad = blk_act.reshape(2, 4, 2, 4, -1).any(axis=(1, 3)).reshape(4, -1)          # [quadrant, entry]: a contributing pixel
                quad_cull = (blk_alpha.reshape(2, 4, 2, 4, -1).any(axis=(1, 3)).reshape(4, -1)) & box[None, :]  # what a per-quadrant cull keeps
                hi_ = wmax
                while hi_ > 0:
                    lo_ = max(0, hi_ - 64)
                    rows_it += int(quad[:, lo_:hi_].sum(axis=1).max())
                    rows_cull_it += int(quad_cull[:, lo_:hi_].sum(axis=1).max())
                    same_all += int(quad[:, lo_:hi_].all(axis=0).sum())
                    hi_ = lo_
        per = act.reshape(2, 8, 2, 8, -1).sum(axis=(1, 3)).ravel()
        hist += np.bincount(per[per > 0], minlength=65)[:65]
    print(f"{name}: E={E}, {len(tiles)} tiles sampled, {entries_total} entries, {pairs} contributing (pixel, entry) pairs")
    for k, (sw, sh_) in shapes.items():
        print(f"  region {k:10s}: iterations with a contributing pixel {it[k]:9d}  ({it[k] / max(1, entries_total * (256 // (sw * sh_))):.3f} of region x entry),"
              f" pixels contributing per iteration {pairs / max(1, it[k]):6.1f} of {sw * sh_}")
    print(f"  8x8 blocks: {chunks} chunks of 64 entries walked ({walked} entries), {in_box} pass the extent-box test, {reach} reach alpha >= 1/255 in the block, "
          f"{it['8x8 (now)']} have a contributing pixel; per chunk: {in_box / max(1, chunks):.1f} in box, {reach / max(1, chunks):.1f} kept, {it['8x8 (now)'] / max(1, chunks):.1f} iterations")
    print(f"  four independent 4x4 quadrants per wave (per chunk: the longest quadrant list): {rows_it} iterations ({rows_it / max(1, it['8x8 (now)']):.2f} of now), "
          f"{rows_cull_it} when lists hold what a per-quadrant alpha cull keeps; entries contributing in all four quadrants: {same_all}")
    now_instr = it["8x8 (now)"] * 125                      # ~125 VALU wave-instructions per (wave, splat) iteration today, 22 of them the butterfly
    per_step = 125 - 22 + 10 + 9                            # no butterfly; + rotating ~10 registers of pixel state by DPP; + nine register accumulations
    print(f"  lane-per-splat systolic form: {sys_groups} groups of <= 64 kept splats, {sys_steps} steps ({sys_steps_ideal} if groups of a block chain without a bubble) "
          f"x ~{per_step} VALU = {sys_steps * per_step / 1e6:.2f} M ({sys_steps_ideal * per_step / 1e6:.2f} M) wave-instructions vs {now_instr / 1e6:.2f} M today: "
          f"{sys_steps * per_step / max(1, now_instr):.2f}x ({sys_steps_ideal * per_step / max(1, now_instr):.2f}x)")
    c = np.cumsum(hist) / max(1, hist.sum())
    print("  8x8: share of iterations with <= k contributing pixels: " + ", ".join(f"k={k}: {c[k]:.2f}" for k in (1, 2, 4, 8, 16, 32, 48, 63)))


if __name__ == "__main__":
    main()
