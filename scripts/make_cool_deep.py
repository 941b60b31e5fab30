#!/opt/conda/bin/python3.9
"""A DEEP whole-genome .mcool for scripts/host_e2e.py --deep (the end-to-end figure of round 3 stood on a map
with 73 M pixels; a deep 5 kb map holds twenty times that and more).  Streams the synthetic genome into a cooler-format file
(schema version 3, the storage `cooler` itself uses: 2^18-row chunks, shuffle + gzip-6) without ever holding it in memory:

  * per chromosome and slab of rows: Poisson(depth / (1 + k)) on the band's stored diagonals, 3x3 planted enrichments, plus -
    `--far` - the pixels BEYOND the band that a real map carries (same 1/(1 + k) law out to the chromosome's end, sampled
    sparsely): a per-chromosome fetch has to read and skip them, and in a deep map they are most of the file;
  * pixels in cooler's order (bin1, bin2); the three pixel columns are cut into chunks, byte-shuffled and deflated on a thread
    pool (zlib releases the GIL) and handed to HDF5 already compressed (H5Dwrite_chunk), since h5py's own filter pipeline is
    one thread: 1.6 G pixels would take it half an hour.

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 scripts/make_cool_deep.py /tmp/deep.mcool --res 5000 --num 2011 --depth 500 --far
"""
import argparse
import importlib.util
import os
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import h5py
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('synthetic', os.path.join(REPO, 'hicpeaks_amd', 'synthetic.py'))
synthetic = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synthetic)

CHUNK = 1 << 18
SLAB = 2048


def slab_pixels(n, num, depth, r0, r1, seed, loops, far):
    """pixels of rows [r0, r1) of one chromosome in (row, column) order -> (row, col, count), band row sums, and what the
    slab adds to the column sums"""
    rng = np.random.default_rng(seed)
    k = np.arange(num, dtype=np.float64)
    lam = np.broadcast_to(depth / (1.0 + k), (r1 - r0, num)).copy()
    for (r, d) in loops:
        for dr in (-1, 0, 1):
            for dc in (-1, 0, 1):
                rr, kk = r + dr, d + dc - dr
                if r0 <= rr < r1 and 0 <= kk < num:
                    lam[rr - r0, kk] *= 8.0
    blk = rng.poisson(lam).astype(np.int32)
    rows = np.arange(r0, r1)[:, None]
    blk[(rows + np.arange(num)[None, :]) >= n] = 0
    rr, kk = np.nonzero(blk)
    v = blk[rr, kk]
    rr = rr + r0
    cc = rr + kk
    rowsum = np.bincount(rr - r0, weights=v, minlength=r1 - r0)
    off = kk >= 1
    colidx, colval = cc[off], v[off].astype(np.float64)
    if far and n > num + 1:
        # beyond the band: expected pixels per row = depth (ln(n - r) - ln(num)), distances log-uniform on [num, n - r)
        per_row = depth * np.maximum(np.log(np.maximum(n - np.arange(r0, r1), num + 1) / float(num)), 0.0)
        m = rng.poisson(per_row.sum())
        if m:
            fr = rng.choice(r1 - r0, size=m, p=per_row / per_row.sum()) + r0
            fk = np.floor(num * np.exp(rng.random(m) * np.log(np.maximum(n - fr, num + 1) / float(num)))).astype(np.int64)
            ok = fr + fk < n
            key, cnt = np.unique(fr[ok] * np.int64(n) + (fr[ok] + fk[ok]), return_counts=True)
            rr = np.concatenate([rr, key // n])
            cc = np.concatenate([cc, key % n])
            v = np.concatenate([v, cnt.astype(np.int32)])
            order = np.lexsort((cc, rr))
            rr, cc, v = rr[order], cc[order], v[order]
    return rr.astype(np.int64), cc.astype(np.int64), v.astype(np.int32), rowsum, colidx, colval


def deflate(col):
    """one chunk as HDF5's shuffle + deflate filters would store it"""
    a = np.ascontiguousarray(col)
    sh = a.view(np.uint8).reshape(a.size, a.itemsize).T.copy()
    return zlib.compress(sh.tobytes(), 6)


class ChunkWriter(object):
    """three resizable columns written chunk by chunk, compressed on a pool"""

    def __init__(self, grp, pool):
        kw = dict(maxshape=(None,), chunks=(CHUNK,), compression='gzip', compression_opts=6, shuffle=True)
        self.d = [grp.create_dataset('bin1_id', shape=(0,), dtype=np.int64, **kw),
                  grp.create_dataset('bin2_id', shape=(0,), dtype=np.int64, **kw),
                  grp.create_dataset('count', shape=(0,), dtype=np.int32, **kw)]
        self.buf = [[], [], []]
        self.nbuf = 0
        self.rows = 0
        self.pool = pool
        self.jobs = []

    def add(self, b1, b2, c):
        for t, a in enumerate((b1, b2, c)):
            self.buf[t].append(a)
        self.nbuf += b1.size
        if self.nbuf >= 8 * CHUNK:
            self._flush(False)

    def _flush(self, last):
        cols = [np.concatenate(b) if b else np.zeros(0, d.dtype) for b, d in zip(self.buf, self.d)]
        full = cols[0].size if last else cols[0].size // CHUNK * CHUNK
        for s in range(0, full, CHUNK):
            parts = []
            for col, d in zip(cols, self.d):
                piece = col[s:s + CHUNK]
                if piece.size < CHUNK:
                    piece = np.concatenate([piece, np.zeros(CHUNK - piece.size, piece.dtype)])
                parts.append(self.pool.submit(deflate, piece))
            self.jobs.append((self.rows + s, parts))
        self.rows += full
        self.buf = [[col[full:]] for col in cols]
        self.nbuf = cols[0].size - full
        self._drain(16 if not last else 0)

    def _drain(self, keep):
        while len(self.jobs) > keep:
            row, parts = self.jobs.pop(0)
            for d, f in zip(self.d, parts):
                if d.shape[0] < row + CHUNK:
                    d.resize((row + CHUNK,))
                d.id.write_direct_chunk((row,), f.result(), 0)

    def close(self):
        self._flush(True)
        for d in self.d:
            d.resize((self.rows,))
        return self.rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('out')
    ap.add_argument('--res', type=int, default=5000)
    ap.add_argument('--num', type=int, default=2011)
    ap.add_argument('--depth', type=float, default=500.0)
    ap.add_argument('--far', action='store_true', help='also the pixels beyond the band, out to the chromosome ends')
    ap.add_argument('--chroms', nargs='*', default=None)
    ap.add_argument('--threads', type=int, default=min(32, os.cpu_count() or 1))
    a = ap.parse_args()
    if os.path.exists(a.out):
        os.remove(a.out)
    sizes = synthetic.hg38_bins(a.res)
    names = a.chroms or list(sizes)
    chroms = [('chr' + c, sizes[c]) for c in names]
    nb = np.array([n for _, n in chroms], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(nb)])
    nbins = int(off[-1])
    kw = dict(compression='gzip', compression_opts=6, shuffle=True)
    t0 = time.perf_counter()
    with h5py.File(a.out, 'w') as f, ThreadPoolExecutor(a.threads) as pool:
        g = f.require_group('/resolutions/%d' % a.res)
        gc = g.create_group('chroms')
        gc.create_dataset('name', data=np.array([c for c, _ in chroms], dtype='S32'), **kw)
        gc.create_dataset('length', data=(nb * a.res).astype(np.int32), **kw)
        gb = g.create_group('bins')
        enum = h5py.special_dtype(enum=('i', {c: i for i, (c, _) in enumerate(chroms)}))
        gb.create_dataset('chrom', data=np.repeat(np.arange(len(chroms), dtype='i'), nb), dtype=enum, **kw)
        start = np.concatenate([np.arange(n, dtype=np.int64) * a.res for n in nb])
        gb.create_dataset('start', data=start.astype(np.int32), **kw)
        gb.create_dataset('end', data=(start + a.res).astype(np.int32), **kw)
        wr = ChunkWriter(g.create_group('pixels'), pool)
        per_bin = np.zeros(nbins + 1, dtype=np.int64)
        weights = []
        total = 0
        for ci, (c, n) in enumerate(chroms):
            rng = np.random.default_rng(100 + ci)
            loops = [(int(rng.integers(2, max(3, n - 40))), int(rng.integers(10, a.num - 15))) for _ in range(max(1, n // 60))]
            slabs = [(r0, min(n, r0 + SLAB)) for r0 in range(0, n, SLAB)]
            futs = [pool.submit(slab_pixels, n, a.num, a.depth, r0, r1, (100 + ci) * 100003 + r0, loops, a.far) for r0, r1 in slabs]
            rowsum, colsum, npx = np.zeros(n), np.zeros(n), 0
            for (r0, r1), fu in zip(slabs, futs):
                rr, cc, v, rs, cidx, cval = fu.result()
                rowsum[r0:r1] = rs
                colsum += np.bincount(cidx, weights=cval, minlength=n)
                per_bin[:nbins] += np.bincount(rr + off[ci], minlength=nbins)
                wr.add(rr + off[ci], cc + off[ci], v)
                npx += rr.size
            w = 1.0 / np.sqrt(rowsum + colsum + 1.0)
            nbad = int(round(n * 0.025))
            s0 = int(rng.integers(n // 3, max(n // 3 + 1, 2 * n // 3 - nbad)))
            w[s0:s0 + (2 * nbad) // 3] = np.nan
            w[rng.choice(n, size=nbad - (2 * nbad) // 3, replace=False)] = np.nan
            weights.append(w)
            total += npx
            print('%s %d bins, %d pixels (%.0f s)' % (c, n, npx, time.perf_counter() - t0), file=sys.stderr, flush=True)
        nnz = wr.close()
        assert nnz == total
        wd = gb.create_dataset('weight', data=np.concatenate(weights), **kw)
        wd.attrs['ignore_diags'] = 2
        wd.attrs['converged'] = True
        gi = g.create_group('indexes')
        gi.create_dataset('chrom_offset', data=off.astype(np.int64), **kw)
        gi.create_dataset('bin1_offset', data=np.concatenate([[0], np.cumsum(per_bin[:nbins])]).astype(np.int64), **kw)
        g.attrs['format'] = 'HDF5::Cooler'
        g.attrs['format-version'] = 3
        g.attrs['bin-type'] = 'fixed'
        g.attrs['bin-size'] = int(a.res)
        g.attrs['storage-mode'] = 'symmetric-upper'
        g.attrs['nchroms'] = len(chroms)
        g.attrs['nbins'] = nbins
        g.attrs['nnz'] = int(nnz)
        g.attrs['generated-by'] = 'hicpeaks_amd/scripts/make_cool_deep.py (h5py %s)' % h5py.__version__
    print('wrote %s: %d pixels, %.1f MB, %.0f s' % (a.out, nnz, os.path.getsize(a.out) / 1e6, time.perf_counter() - t0))


if __name__ == '__main__':
    main()
