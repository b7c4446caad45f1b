#!/usr/bin/env python3
"""VERDICT r3 item 7, costed before building: what would WAVE-ALIGNED multi-tile units need?

configs[2]'s reads (bench.make_long_reads' length law: LogNormal(8.8903, 0.8), clamped to [200, 500000], 1.5 Gbp) cut into
256-window tiles as plan.hip does.  For units of <= 64 tiles placed so that none straddles two scan waves:
  * lane utilisation of phase A, (a) in stream order (a wave is closed when the next unit does not fit), (b) best-fit
    decreasing inside each planning block of 2048 reads (what plan_kernel could do with one more sort);
  * what in-wave resolution of such a wave would have to hold: emitted minimizers per lane (the per-lane list takes 40 before a
    mid-scan flush: DCN_LCAP) and hits per wave at the workload's hit rates (the LDS ring compares a unit's hits within the
    last 192: DCN_RCAP - 64).
Pure arithmetic on the length distribution; no GPU.  Output goes to profiles/r04_ab.txt."""
import numpy as np

K, W, TILE = 31, 15, 256
L = K + W - 1
rng = np.random.default_rng(6)
lens, tot = [], 0
while tot < 1_500_000_000:
    ln = int(min(500_000, max(200, rng.lognormal(8.8903, 0.8))))
    lens.append(ln)
    tot += ln
lens = np.array(lens)
tiles = np.maximum(1, -(-(lens - L + 1) // TILE))
print(f"{len(lens):,} reads, {tot / 1e9:.2f} Gbp, {int(tiles.sum()):,} tiles; mean {tiles.mean():.1f} tiles per read")
fits = tiles <= 64
print(f"reads of <= 64 tiles: {fits.mean() * 100:.1f} % of the reads, {lens[fits].sum() / tot * 100:.1f} % of the bases "
      f"(the rest still straddles waves whatever is done)")


def stream_order(t):
    waves, fill = 0, 0
    for x in t:
        if x > 64:            # spans waves anyway: fills whole waves, its tail shares one
            if fill:
                waves += 1
            waves += x // 64
            fill = x % 64
            continue
        if fill + x > 64:
            waves += 1
            fill = 0
        fill += x
    return waves + (1 if fill else 0)


def best_fit_blocks(t, block=2048):
    waves = 0
    for a in range(0, len(t), block):
        blk = np.sort(t[a:a + block])[::-1]
        big = blk[blk > 64]
        waves += int((big // 64).sum())
        items = list(big % 64) + list(blk[blk <= 64])
        bins = []
        for x in sorted((int(i) for i in items if i), reverse=True):
            best = -1
            for j, free in enumerate(bins):
                if free >= x and (best < 0 or free < bins[best]):
                    best = j
            if best < 0:
                bins.append(64 - x)
            else:
                bins[best] -= x
        waves += len(bins)
    return waves


n_tiles = int(tiles.sum())
dense = -(-n_tiles // 64)
for name, w in (("today (tiles dense, units straddle waves)", dense), ("wave-aligned, stream order", stream_order(tiles)),
                ("wave-aligned, best-fit decreasing per 2048-read block", best_fit_blocks(tiles))):
    print(f"{name:58s} {w:8,} waves, lane utilisation {n_tiles / (64 * w) * 100:5.1f} %, phase A x{w / dense:.3f}")
# what a wave of whole units would have to resolve in LDS
per_tile = TILE * 2 / (W + 1)
print(f"emitted minimizers per full tile ~ {per_tile:.0f} (list capacity before a mid-scan flush: 40 -> a full tile flushes about once, "
      f"and a unit's hits are contiguous in the ring only within ONE flush)")
for rate, what in ((0.20, "host read, 5 % substitutions"), (0.85, "host read, 0.5 % substitutions"), (1.0, "exact host read")):
    print(f"hits per wave of 64 full tiles at hit rate {rate:.2f} ({what}): {64 * per_tile * rate:6.0f}  (the ring dedups a unit within its last 192 hits)")
