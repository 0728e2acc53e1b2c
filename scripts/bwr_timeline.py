#!/usr/bin/env python3
"""Per-wave timeline of backward_rasterize -- and of rasterize, whose TIMELINE form writes the same records (WDGS_FWR_TIMELINE) -- (dev tool).

    WDGS_BWR_TIMELINE=/tmp/tl.bin WDGS_PROFILE_FROZEN=1 python scripts/profile_step.py c2 4     # eager steps; every launch appends its records
    python scripts/bwr_timeline.py /tmp/tl.bin                                                  # reads the LAST launch in the file

The kernel's TIMELINE form (csrc/backward_raster.hip) leaves per wave {start, end} of the 100 MHz wall clock, the hardware ids it ran on and the
number of splats it iterated over.  Printed: the launch's span, how many waves are resident over time, how evenly the work falls on the XCDs and
CUs, the time per iterated splat as a function of the residency the wave saw, and how much of the span is the tail after the last wave started.
"""
import struct
import sys

import numpy as np


def launches(path):
    data = open(path, "rb").read()
    at = 0
    while at < len(data):
        slots, tiles = struct.unpack_from("<II", data, at)
        at += 8
        rec = np.frombuffer(data, dtype=np.uint64, count=slots * 4, offset=at).reshape(slots, 4)
        at += slots * 32
        yield slots, tiles, rec


def main():
    path = sys.argv[1]
    slots, tiles, rec = list(launches(path))[-1]
    ran = rec[:, 1] != 0
    t0 = rec[ran, 0].astype(np.int64)
    t1 = rec[ran, 1].astype(np.int64)
    ids = rec[ran, 2]
    it = (rec[ran, 3] & np.uint64(0xFFFF)).astype(np.int64)
    ff = (rec[ran, 3] >> np.uint64(16)).astype(np.int64)  # SPLIT form: iterations of the state recurrence alone, over the chunks behind the wave's segment
    base = t0.min()
    t0 -= base
    t1 -= base
    tick_us = 0.01
    span = t1.max()
    hw = (ids & np.uint64(0xFFFFFFFF)).astype(np.int64)
    xcc = ((ids >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    where = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print(f"slots={slots} tiles={tiles} waves that recorded={ran.sum()} (the others had no work: empty tile or surplus slot)")
    print(f"span first start -> last end: {span * tick_us:.2f} us;  last wave start at {t0.max() * tick_us:.2f} us;  iterated splats total={it.sum()}")
    dur = t1 - t0
    if ff.any():
        print(f"state-only iterations total={ff.sum()} (per wave that has any: mean={ff[ff > 0].mean():.1f} max={ff.max()})")
    print(f"wave life us: mean={dur.mean() * tick_us:.2f} p50={np.percentile(dur, 50) * tick_us:.2f} p90={np.percentile(dur, 90) * tick_us:.2f} "
          f"p99={np.percentile(dur, 99) * tick_us:.2f} max={dur.max() * tick_us:.2f}")
    work = it > 0
    print(f"waves with iterations: {work.sum()};  ns per iterated splat (life / iterations): mean={(dur[work] / it[work]).mean() * 10:.1f} "
          f"p50={np.percentile(dur[work] / it[work], 50) * 10:.1f};  sum of lives = {dur.sum() * tick_us:.0f} wave-us = {dur.sum() / max(span, 1):.0f} waves resident on average")
    # residency over time
    grid = np.arange(0, span + 1)
    delta = np.zeros(span + 2, dtype=np.int64)
    np.add.at(delta, t0, 1)
    np.add.at(delta, t1, -1)
    resident = np.cumsum(delta)[: span + 1]
    print("resident waves over time (tenths of the span):", " ".join(str(int(resident[int(span * k / 10): max(int(span * (k + 1) / 10), int(span * k / 10) + 1)].mean())) for k in range(10)))
    # places
    places = np.unique(where)
    print(f"distinct (xcc,se,sh,cu) places used: {len(places)};  xcc ids seen: {sorted(set(xcc.tolist()))};  simd ids seen: {sorted(set(simd.tolist()))}")
    busy = np.zeros(places.max() + 1, dtype=np.int64)
    np.add.at(busy, where, dur)
    its = np.zeros(places.max() + 1, dtype=np.int64)
    np.add.at(its, where, it)
    b = busy[places] / max(span, 1)
    print(f"waves resident per place (time average): mean={b.mean():.2f} min={b.min():.2f} p10={np.percentile(b, 10):.2f} p90={np.percentile(b, 90):.2f} max={b.max():.2f}")
    i = its[places]
    print(f"iterated splats per place: mean={i.mean():.0f} min={i.min()} p10={np.percentile(i, 10):.0f} p90={np.percentile(i, 90):.0f} max={i.max()}   (max / mean = {i.max() / i.mean():.2f})")
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        print(f"  xcc {x}: waves={m.sum():5d} iterations={it[m].sum():8d} first start={t0[m].min() * tick_us:6.2f} last end={t1[m].max() * tick_us:6.2f} us  places={len(np.unique(where[m]))}")
    # when does each place finish
    last_end = np.zeros(places.max() + 1, dtype=np.int64)
    np.maximum.at(last_end, where, t1)
    le = last_end[places] * tick_us
    print(f"a place's last wave ends at: p10={np.percentile(le, 10):.2f} p50={np.percentile(le, 50):.2f} p90={np.percentile(le, 90):.2f} max={le.max():.2f} us")
    # start order vs slot id: how the dispatcher hands out workgroups
    order = np.argsort(rec[ran, 0], kind="stable")
    slot_ids = np.nonzero(ran)[0][order]
    q = len(slot_ids) // 4
    print("mean slot id of the waves by start order (quarters):", " ".join(f"{slot_ids[k * q:(k + 1) * q].mean():.0f}" for k in range(4)))
    # rate a wave achieves against the residency of its place while it lived
    rate = dur[work] / it[work] * 10
    mid = ((t0[work] + t1[work]) // 2).clip(0, span)
    res_at = resident[mid]
    for lo, hi in ((0, 1000), (1000, 2000), (2000, 3000), (3000, 4000), (4000, 9999)):
        m = (res_at >= lo) & (res_at < hi)
        if m.any():
            print(f"  waves living at chip residency [{lo},{hi}): {m.sum():5d} waves, ns per iterated splat p50={np.percentile(rate[m], 50):.1f}")
    # ---- per SIMD: all waves of a launch that fits the chip start together, so a SIMD's waves share its issue slots from t = 0 and leave one
    # by one.  Between two consecutive departures k waves are resident; under fair sharing each advances (n[j+1] - n[j]) iterations in
    # (t[j+1] - t[j]): the time one iteration of one wave takes when k waves share the SIMD.
    simd_key = where * 4 + simd
    per_k = {}
    simd_iters, simd_end, simd_waves = [], [], []
    for key in np.unique(simd_key):
        m = simd_key == key
        n = it[m]; e = t1[m]; s0 = t0[m]
        simd_iters.append(n.sum()); simd_end.append(e.max()); simd_waves.append(m.sum())
        if s0.max() > 300:  # (3 us) a wave of this SIMD started late: the model does not hold
            continue
        o = np.argsort(e)
        n = n[o]; e = e[o]
        k = len(n)
        prev_n, prev_t = 0, 0
        for j in range(len(n)):
            dn, dt = n[j] - prev_n, e[j] - prev_t
            if dn > 0 and dt > 0:
                per_k.setdefault(k - j, []).append((dn, dt))
            prev_n, prev_t = max(prev_n, n[j]), e[j]
    simd_iters = np.array(simd_iters); simd_end = np.array(simd_end); simd_waves = np.array(simd_waves)
    print(f"SIMDs used: {len(simd_iters)}; waves per SIMD: mean={simd_waves.mean():.2f} min={simd_waves.min()} max={simd_waves.max()}; "
          f"iterations per SIMD: mean={simd_iters.mean():.0f} p10={np.percentile(simd_iters, 10):.0f} p90={np.percentile(simd_iters, 90):.0f} max={simd_iters.max()} (max / mean = {simd_iters.max() / simd_iters.mean():.2f})")
    print(f"a SIMD's last wave ends at: p10={np.percentile(simd_end, 10) * tick_us:.2f} p50={np.percentile(simd_end, 50) * tick_us:.2f} p90={np.percentile(simd_end, 90) * tick_us:.2f} max={simd_end.max() * tick_us:.2f} us;  "
          f"correlation(iterations of the SIMD, its end) = {np.corrcoef(simd_iters, simd_end)[0, 1]:.2f}")
    print(f"iterations per wave: mean={it[work].mean():.1f} p50={np.percentile(it[work], 50):.0f} p90={np.percentile(it[work], 90):.0f} p99={np.percentile(it[work], 99):.0f} max={it.max()};  "
          f"correlation(iterations, life) = {np.corrcoef(it[work], dur[work])[0, 1]:.2f}")
    print("time of one iteration of one wave while k waves share its SIMD (fair-sharing estimate), and the SIMD's iterations per us at that k:")
    for k in sorted(per_k):
        a = np.array(per_k[k], dtype=np.float64)
        per_iter_ns = a[:, 1].sum() / a[:, 0].sum() * 10
        print(f"  k={k}: {len(a):5d} intervals, {per_iter_ns:7.1f} ns per iteration per wave -> {k * 1000 / per_iter_ns:6.2f} iterations / us / SIMD")
    longest = np.argsort(-dur)[:5]
    for w in longest:
        print(f"  long wave: life={dur[w] * tick_us:.2f} us iterations={it[w]} ({dur[w] * 10 / max(it[w], 1):.0f} ns each) on a SIMD with {int((simd_key == simd_key[w]).sum())} waves, {int(it[simd_key == simd_key[w]].sum())} iterations")


if __name__ == "__main__":
    main()
