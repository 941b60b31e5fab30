#!/opt/conda/bin/python3.9
"""Write a cooler-format (.cool, schema version 3) file from synthetic bands with plain h5py - the `cooler` package is
not installed in this image, so this follows its published schema (https://cooler.readthedocs.io/en/latest/schema.html)
and `cooler.create_cooler`'s storage choices (chunked, gzip-6, shuffle; bins/chrom as an HDF5 enum; fixed-length ASCII
chromosome names; weight column with the attributes `cooler balance` leaves).

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 scripts/make_cool.py tests/golden/tiny.cool           (the test fixture)
    ... scripts/make_cool.py /tmp/wg.mcool --genome hg38 --res 5000 --num 2011 --group /resolutions/5000      (scripts/host_e2e.py)

The fixture: three chromosomes at 10 kb (chrA 400 bins, chrB 57 bins - shorter than the 61-diagonal band -, chrC 260 bins),
NaN weights on masked bins, a second weight column 'KR' holding the reciprocals (a divisive column), trans pixels between
the chromosomes (which a per-chromosome fetch must leave out)."""
import argparse
import importlib.util
import os
import sys

import h5py
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('synthetic', os.path.join(REPO, 'hicpeaks_amd', 'synthetic.py'))
synthetic = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synthetic)


def write_cool(path, group, res, chroms, bands, weights, trans=None, extra_cols=None):
    """chroms: [(name, nbins)], bands: {name: raw int [n, num]}, weights: {name: f64 [n]} -> one cooler under `group`"""
    names = [c for c, _ in chroms]
    nb = np.array([n for _, n in chroms], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(nb)])
    nbins = int(off[-1])
    b1, b2, cnt = [], [], []
    for ci, (c, n) in enumerate(chroms):
        r, k = np.nonzero(bands[c])
        keep = r + k < n
        r, k = r[keep], k[keep]
        b1.append(r + off[ci]); b2.append(r + k + off[ci]); cnt.append(bands[c][r, k])
    if trans is not None:
        b1.append(trans[0]); b2.append(trans[1]); cnt.append(trans[2])
    b1, b2, cnt = np.concatenate(b1), np.concatenate(b2), np.concatenate(cnt)
    order = np.lexsort((b2, b1))
    b1, b2, cnt = b1[order].astype(np.int64), b2[order].astype(np.int64), cnt[order].astype(np.int32)
    kw = dict(compression='gzip', compression_opts=6, shuffle=True)
    with h5py.File(path, 'a') as f:
        g = f.require_group(group) if group != '/' else f
        gc = g.create_group('chroms')
        gc.create_dataset('name', data=np.array(names, dtype='S32'), **kw)
        gc.create_dataset('length', data=(nb * res).astype(np.int32), **kw)
        gb = g.create_group('bins')
        enum = h5py.special_dtype(enum=('i', {c: i for i, c in enumerate(names)}))
        chrom_ids = np.repeat(np.arange(len(names), dtype='i'), nb)
        gb.create_dataset('chrom', data=chrom_ids, dtype=enum, **kw)
        start = np.concatenate([np.arange(n, dtype=np.int64) * res for n in nb])
        gb.create_dataset('start', data=start.astype(np.int32), **kw)
        gb.create_dataset('end', data=(start + res).astype(np.int32), **kw)
        w = gb.create_dataset('weight', data=np.concatenate([weights[c] for c in names]), **kw)
        w.attrs['ignore_diags'] = 2
        w.attrs['converged'] = True
        for nm, (col, attrs) in (extra_cols or {}).items():
            d = gb.create_dataset(nm, data=np.concatenate([col[c] for c in names]), **kw)
            for k, v in attrs.items():
                d.attrs[k] = v
        gp = g.create_group('pixels')
        gp.create_dataset('bin1_id', data=b1, chunks=(min(len(b1), 1 << 18),), **kw)
        gp.create_dataset('bin2_id', data=b2, chunks=(min(len(b2), 1 << 18),), **kw)
        gp.create_dataset('count', data=cnt, chunks=(min(len(cnt), 1 << 18),), **kw)
        gi = g.create_group('indexes')
        gi.create_dataset('chrom_offset', data=off.astype(np.int64), **kw)
        gi.create_dataset('bin1_offset', data=np.searchsorted(b1, np.arange(nbins + 1)).astype(np.int64), **kw)
        g.attrs['format'] = 'HDF5::Cooler'
        g.attrs['format-version'] = 3
        g.attrs['bin-type'] = 'fixed'
        g.attrs['bin-size'] = int(res)
        g.attrs['storage-mode'] = 'symmetric-upper'
        g.attrs['nchroms'] = len(names)
        g.attrs['nbins'] = nbins
        g.attrs['nnz'] = int(len(cnt))
        g.attrs['sum'] = int(cnt.sum())
        g.attrs['generated-by'] = 'hicpeaks_amd/scripts/make_cool.py (h5py %s)' % h5py.__version__
    return len(cnt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('out')
    ap.add_argument('--genome', default=None, help="'hg38': chr1-22,X at --res with --num stored diagonals (large)")
    ap.add_argument('--res', type=int, default=10000)
    ap.add_argument('--num', type=int, default=61)
    ap.add_argument('--depth', type=float, default=40.0)
    ap.add_argument('--group', default='/')
    ap.add_argument('--chroms', nargs='*', default=None)
    ap.add_argument('--trans', type=int, default=0, help="--genome: this many trans pixels per chromosome (but the last), as a real map's rows carry")
    a = ap.parse_args()
    if os.path.exists(a.out):
        os.remove(a.out)
    if a.genome:
        sizes = synthetic.hg38_bins(a.res)
        names = a.chroms or list(sizes)
        chroms = [('chr' + c, sizes[c]) for c in names]
        bands, weights = {}, {}
        for i, (c, n) in enumerate(chroms):
            raw, w, _ = synthetic.synth_band(n, a.num, depth=a.depth, nloops=max(1, n // 60), seed=100 + i)
            bands[c], weights[c] = raw, w
            print(c, n, int((raw != 0).sum()), file=sys.stderr)
        trans = None
        if a.trans > 0 and len(chroms) > 1:        # trans pixels: rows in every chromosome but the last, columns in a later one
            rng = np.random.default_rng(10)
            offs = np.concatenate([[0], np.cumsum([n for _, n in chroms])])
            keys = []
            for i in range(len(chroms) - 1):
                t1 = rng.integers(offs[i], offs[i + 1], a.trans).astype(np.int64)
                t2 = rng.integers(offs[i + 1], offs[-1], a.trans).astype(np.int64)
                keys.append(t1 * (1 << 32) + t2)
            key = np.unique(np.concatenate(keys))
            trans = (key >> 32, key & ((1 << 32) - 1), rng.integers(1, 5, key.size))
        nnz = write_cool(a.out, a.group, a.res, chroms, bands, weights, trans=trans)
    else:
        chroms = [('chrA', 400), ('chrB', 57), ('chrC', 260)]
        bands, weights, kr = {}, {}, {}
        for i, (c, n) in enumerate(chroms):
            raw, w, _ = synthetic.synth_band(n, a.num, depth=a.depth, nloops=max(1, n // 40), seed=50 + i,
                                             loop_dist=(10, min(a.num - 15, max(n - 8, 12))))
            bands[c], weights[c] = raw, w
            kr[c] = 1.0 / w                     # the same balancing as a divisive column (4DN style: count / (w1 w2))
        rng = np.random.default_rng(9)
        t1 = rng.integers(0, 400, 300)
        t2 = rng.integers(400, 717, 300)
        key = np.unique(t1.astype(np.int64) * 100000 + t2)
        trans = (key // 100000, key % 100000, rng.integers(1, 5, key.size))
        nnz = write_cool(a.out, a.group, a.res, chroms, bands, weights, trans=trans,
                         extra_cols={'KR': (kr, {'divisive_weights': True})})
    print('wrote', a.out, 'pixels', nnz, 'bytes', os.path.getsize(a.out))


if __name__ == '__main__':
    main()
