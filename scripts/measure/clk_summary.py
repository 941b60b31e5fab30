#!/usr/bin/env python
"""Summarise a phase-clock dump of hpk_stencil (libhpk built with -DHPK_PHASE_CLOCK, HPK_CLK_DUMP=<file>):
u64 [workgroups][16 waves][8] = cycles in (0) wait for the prefetched rows + phase 1, (1) prefetch issue + column
totals + two barriers, (2) scans + SAT stores, (3) barrier, (4) gap rows + candidate lists, (5) candidate batches,
(6) end-of-tile barrier, and (7) the number of batches.  Ticks are s_memtime's; their rate against wall time did not calibrate the same way twice (2.1 per ns in scripts/measure/ubench/valu_cost.hip,
~0.6 per ns of this kernel's duration with all 256 workgroups alive from start to end): read the shares, not the ticks.  A mark costs its own s_memtime and the wait for it, sixteen waves at a time behind a barrier: shares below ~5 %
of a tile, and whatever lies between two marks close together, are the marks' own (profiles/r06_stencil_band_switch.txt)."""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16, 8)
raw = raw[raw[:, :, :7].sum(axis=(1, 2)) > 0]            # (the dump holds 1024 workgroup slots)
per = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0    # chromosomes per launch: figures per chromosome
risky = (raw[:, :, 6] >> np.uint64(40)).astype(np.float64)      # hpk_stencil_s: candidates redone exactly, in the high bits
raw[:, :, 6] &= np.uint64((1 << 40) - 1)
explicit = (raw[:, :, 7] >> np.uint64(40)).astype(np.float64)   # ... of those, added cell by cell by the whole wave
raw[:, :, 7] &= np.uint64((1 << 40) - 1)
a = raw.astype(np.float64) / per
names = ['wait+phase1', 'prefetch+coltot', 'scan+write', 'barrier(SAT)', 'gap+lists', 'batches', 'barrier(end)']
tot = a[:, :, :7].sum(axis=2)
print('workgroups %d; ticks per wave: mean %.0f  min %.0f  max %.0f' % (a.shape[0], tot.mean(), tot.min(), tot.max()))
for i, nm in enumerate(names):
    v = a[:, :, i]
    print('%-16s mean %8.0f (%4.1f%%)   per-WG mean min %8.0f max %8.0f' % (nm, v.mean(), 100 * v.mean() / tot.mean(),
                                                                         v.mean(axis=1).min(), v.mean(axis=1).max()))
nb = a[:, :, 7]
print('batches per wave: mean %.1f; per-WG sum min %d max %d mean %.0f' % (nb.mean(), nb.sum(axis=1).min(), nb.sum(axis=1).max(),
                                                                          nb.sum(axis=1).mean()))
wg = a[:, :, 5].mean(axis=1)
print('batch-phase ticks per WG, deciles:', np.percentile(wg, [0, 10, 25, 50, 75, 90, 100]).round(0).tolist())
print('ticks per batch (sum batches-phase / sum batches): %.1f' % (a[:, :, 5].sum() / max(nb.sum(), 1)))
print('sums below the risk threshold (hpk_stencil_s): %d, per-WG max %d; added cell by cell: %d, per-WG max %d' % (
    risky.sum(), risky.sum(axis=1).max(), explicit.sum(), explicit.sum(axis=1).max()))
if len(sys.argv) > 3:       # per wave index
    np.set_printoptions(linewidth=200, suppress=True)
    for i, nm in enumerate(names):
        print('%-16s' % nm, a[:, :, i].mean(axis=0).round(0))
