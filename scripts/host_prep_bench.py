#!/usr/bin/env python
"""Host band preparation at whole-genome 5 kb scale: time of the O(nnz) band builder
(`hpk_band_from_coo`, C, one pass) on ~10^8 upper-triangle pixels, next to numpy's `np.add.at` scatter it replaced
and to the reference's way (`[H.diagonal(i) for i in range(num)]` on a scipy COO, scripts/pyHICCUPS:147 - O(num * nnz),
timed on a slice and extrapolated).  No GPU needed.  usage: host_prep_bench.py [npixels]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hicpeaks_amd import band

npx = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
n, num = 49792, 2011                    # hg38 chr1 at 5 kb, 10 Mb band (BASELINE configs[3], largest work item)
rng = np.random.default_rng(0)
t0 = time.perf_counter()
i = rng.integers(0, n, npx, dtype=np.int64)
k = np.minimum((rng.exponential(300.0, npx)).astype(np.int64), num + 200)       # a few pixels beyond the band
j = np.minimum(i + k, n - 1)
v = rng.integers(1, 20, npx, dtype=np.int32)
print('generated %d pixels in %.1f s' % (npx, time.perf_counter() - t0))
t0 = time.perf_counter()
raw = band.band_from_coo(i, j, v, n, num)
t1 = time.perf_counter()
print('hpk_band_from_coo        : %.2f s  (%.1f M pixels/s, one core), band %d x %d f32 = %.0f MB' % (
    t1 - t0, npx / (t1 - t0) / 1e6, n, num, raw.nbytes / 1e6))
m = min(npx, 10_000_000)
t0 = time.perf_counter()
ref = np.zeros((n, num), dtype=np.float32)
kk = j[:m] - i[:m]
keep = kk < num
np.add.at(ref, (i[:m][keep], kk[keep]), v[:m][keep])
t1 = time.perf_counter()
print('np.add.at (round 1)      : %.2f s for %d pixels -> %.1f s for all (%.1f M pixels/s)' % (t1 - t0, m, (t1 - t0) * npx / m, m / (t1 - t0) / 1e6))
chk = band.band_from_coo(i[:m], j[:m], v[:m], n, num)
assert np.array_equal(chk, ref)
from scipy import sparse
m2 = min(npx, 2_000_000)
H = sparse.coo_matrix((v[:m2], (i[:m2], j[:m2])), shape=(n, n))
t0 = time.perf_counter()
for d in range(40):
    H.diagonal(d)
t1 = time.perf_counter()
per = (t1 - t0) / 40 / m2
print('H.diagonal(i) x num (ref): %.3f s per diagonal at %d pixels -> %.0f s for %d diagonals x %d pixels' % (
    (t1 - t0) / 40, m2, per * num * npx, num, npx))
