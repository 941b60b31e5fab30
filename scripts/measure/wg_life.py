"""When the workgroups of a stencil launch start and end (libhpk built with -DHPK_PHASE_CLOCK -DHPK_WG_LIFE, HPK_CLK_DUMP=<file>: a wave's first and last
s_memrealtime in slots 6 / 7, no marks in the tile loop): how even the static split of the tiles over the persistent workgroups comes out."""
import sys, numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16, 8)
raw = raw[raw[:, 0, 0] > 0]
s = raw[:, :, 6].min(axis=1).astype(np.int64); e = raw[:, :, 7].max(axis=1).astype(np.int64)
t0, t1 = s.min(), e.max()
life = (e - s).astype(float)
print('workgroups %d; last start %+d, first end %.3f, mean end %.3f of the launch\'s span (%d ticks); life / span: mean %.3f min %.3f max %.3f' % (
    len(s), s.max() - t0, (e.min() - t0) / (t1 - t0), (e.mean() - t0) / (t1 - t0), t1 - t0, (life / (t1 - t0)).mean(), (life / (t1 - t0)).min(), (life / (t1 - t0)).max()))
print('ends, deciles of the span:', np.round(np.percentile((e - t0) / (t1 - t0), [0, 10, 25, 50, 75, 90, 100]), 3).tolist())
by_xcd = [(e[np.arange(len(e)) % 8 == x] - t0).mean() / (t1 - t0) for x in range(8)]
print('mean end by XCD (workgroup index mod 8):', np.round(by_xcd, 3).tolist())
