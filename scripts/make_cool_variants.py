#!/opt/conda/bin/python3.9
"""Legal variations of the cooler schema that real files show and tests/golden/tiny.cool (cooler's own storage choices) does
not: the same small map written in other ways.  -> tests/golden/cool_variants/*.cool (a few KB each), read by
tests/test_cool_cpu.py through hicpeaks_amd/cool.py.  Verified against the published schema only - the `cooler` package is
not installed in this image.

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 scripts/make_cool_variants.py

  plain        int16 counts, bins/chrom as plain int32, chroms/name as variable-length strings, contiguous (unchunked,
               uncompressed) data sets, no storage-mode attribute, bin-size stored as int32
  merged       int64 counts, every third pixel split into two rows with the same (bin1, bin2) - an unsummed merge
  floats       float64 counts (whole numbers), float32 weights, pixel columns in 64-row gzip-1 chunks without the shuffle filter
  square       storage-mode 'square': both triangles in the pixel table
  noweight     no balancing column at all"""
import importlib.util
import os

import h5py
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('synthetic', os.path.join(REPO, 'hicpeaks_amd', 'synthetic.py'))
synthetic = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synthetic)

RES, NUM = 10000, 31
CHROMS = [('chr1', 120), ('chrX', 64)]


def base():
    bands, weights = {}, {}
    for i, (c, n) in enumerate(CHROMS):
        bands[c], weights[c], _ = synthetic.synth_band(n, NUM, depth=30.0, nloops=3, seed=70 + i, loop_dist=(8, 14))
    return bands, weights


def pixels(bands):
    nb = np.array([n for _, n in CHROMS]); off = np.concatenate([[0], np.cumsum(nb)])
    b1, b2, cnt = [], [], []
    for ci, (c, n) in enumerate(CHROMS):
        r, k = np.nonzero(bands[c])
        keep = r + k < n
        r, k = r[keep], k[keep]
        b1.append(r + off[ci]); b2.append(r + k + off[ci]); cnt.append(bands[c][r, k])
    # a few trans pixels
    b1.append(np.array([3, 50, 100])); b2.append(np.array([130, 150, 183])); cnt.append(np.array([2, 1, 4]))
    b1, b2, cnt = np.concatenate(b1), np.concatenate(b2), np.concatenate(cnt)
    o = np.lexsort((b2, b1))
    return b1[o].astype(np.int64), b2[o].astype(np.int64), cnt[o].astype(np.int64), off


def write(path, bands, weights, kind):
    if os.path.exists(path):
        os.remove(path)
    b1, b2, cnt, off = pixels(bands)
    nbins = int(off[-1])
    if kind == 'merged':            # every third pixel as two rows
        rep = np.ones(b1.size, int); rep[::3] = 2
        half = cnt // 2
        idx = np.repeat(np.arange(b1.size), rep)
        first = np.r_[True, idx[1:] != idx[:-1]]
        c2 = np.where(rep[idx] == 1, cnt[idx], np.where(first, half[idx], cnt[idx] - half[idx]))
        keep = c2 > 0
        b1, b2, cnt = b1[idx][keep], b2[idx][keep], c2[keep]
    if kind == 'square':
        offd = b1 != b2
        b1, b2, cnt = np.r_[b1, b2[offd]], np.r_[b2, b1[offd]], np.r_[cnt, cnt[offd]]
        o = np.lexsort((b2, b1)); b1, b2, cnt = b1[o], b2[o], cnt[o]
    with h5py.File(path, 'w') as f:
        names = [c for c, _ in CHROMS]
        nb = np.array([n for _, n in CHROMS])
        gc = f.create_group('chroms')
        if kind == 'plain':
            gc.create_dataset('name', data=np.array(names, dtype=object), dtype=h5py.string_dtype())
        else:
            gc.create_dataset('name', data=np.array(names, dtype='S8'))
        gc.create_dataset('length', data=(nb * RES).astype(np.int32))
        gb = f.create_group('bins')
        ids = np.repeat(np.arange(len(names), dtype=np.int32), nb)
        if kind == 'plain':
            gb.create_dataset('chrom', data=ids)
        else:
            gb.create_dataset('chrom', data=ids, dtype=h5py.special_dtype(enum=('i', {c: i for i, c in enumerate(names)})))
        start = np.concatenate([np.arange(n, dtype=np.int64) * RES for n in nb])
        gb.create_dataset('start', data=start.astype(np.int32))
        gb.create_dataset('end', data=(start + RES).astype(np.int32))
        if kind != 'noweight':
            w = np.concatenate([weights[c] for c in names])
            gb.create_dataset('weight', data=w.astype(np.float32) if kind == 'floats' else w)
        gp = f.create_group('pixels')
        kw = dict(compression='gzip', compression_opts=1, chunks=(64,)) if kind == 'floats' else {}
        gp.create_dataset('bin1_id', data=b1, **kw)
        gp.create_dataset('bin2_id', data=b2, **kw)
        cdt = {'plain': np.int16, 'merged': np.int64, 'floats': np.float64}.get(kind, np.int32)
        gp.create_dataset('count', data=cnt.astype(cdt), **kw)
        gi = f.create_group('indexes')
        gi.create_dataset('chrom_offset', data=off.astype(np.int64))
        gi.create_dataset('bin1_offset', data=np.searchsorted(b1, np.arange(nbins + 1)).astype(np.int64))
        f.attrs['format'] = 'HDF5::Cooler'
        f.attrs['format-version'] = 3
        f.attrs['bin-type'] = 'fixed'
        f.attrs['bin-size'] = np.int32(RES) if kind == 'plain' else int(RES)
        if kind != 'plain':
            f.attrs['storage-mode'] = 'square' if kind == 'square' else 'symmetric-upper'
        f.attrs['nbins'] = nbins
        f.attrs['nnz'] = int(len(cnt))


def main():
    out = os.path.join(REPO, 'tests', 'golden', 'cool_variants')
    os.makedirs(out, exist_ok=True)
    bands, weights = base()
    for kind in ('plain', 'merged', 'floats', 'square', 'noweight'):
        write(os.path.join(out, kind + '.cool'), bands, weights, kind)
        print(kind, os.path.getsize(os.path.join(out, kind + '.cool')))


if __name__ == '__main__':
    main()
