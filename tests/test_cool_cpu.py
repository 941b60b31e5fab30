"""The cooler-reading half of scripts/pyHICCUPS:142-166 on a real cooler-format file (HDF5, schema version 3:
tests/golden/tiny.cool, written by scripts/make_cool.py with h5py - gzip-compressed chunked data sets, bins/chrom as an
HDF5 enum, fixed-length ASCII names, a NaN-masked weight column and a divisive 'KR' column, trans pixels between the
chromosomes).  The `cooler` package is not installed in this image; the file goes through the package's own reader
(hicpeaks_amd/cool.py on libhdf5 or h5py).  What is pinned: the band, the weights and - with the reference's own rule -
IR / cDiags / biases of every chromosome equal what worker() builds from `Lib.matrix(...).fetch(key)`."""
import importlib.util
import os

import numpy as np
import pytest

from conftest import REPO, GOLDEN_DIR
from hicpeaks_amd import band as hband, io, synthetic
from oracle import hiccups_oracle as orc

COOL = os.path.join(GOLDEN_DIR, 'tiny.cool')
CHROMS = [('chrA', 400), ('chrB', 57), ('chrC', 260)]          # scripts/make_cool.py
NUM = 61


def _reader_available():
    try:
        from hicpeaks_amd import cool
        cool.CoolFile(COOL).close()
        return True
    except ImportError:
        return False


pytestmark = pytest.mark.skipif(not _reader_available(), reason='neither h5py nor libhdf5 here')


def _expected(i, n):
    raw, w, _ = synthetic.synth_band(n, NUM, depth=40.0, nloops=max(1, n // 40), seed=50 + i,
                                     loop_dist=(10, min(NUM - 15, max(n - 8, 12))))
    rr = np.arange(n)[:, None]
    raw[(rr + np.arange(NUM)[None, :]) >= n] = 0
    return raw, w


def test_cooler_file_through_the_own_reader():
    src = io.open_source(COOL)
    assert isinstance(src, io.CoolerSource) and src.clr is None           # no cooler package: hicpeaks_amd.cool
    assert src.binsize == 10000 and src.chromnames == [c for c, _ in CHROMS]
    for i, (c, n) in enumerate(CHROMS):
        assert src.nbins(c) == n
        want_raw, want_w = _expected(i, n)
        raw, w, b = src.fetch(c, NUM)
        assert b is None and raw.dtype == np.float32 and raw.shape == (n, NUM)
        np.testing.assert_array_equal(raw, want_raw)                      # trans pixels are not part of the chromosome
        np.testing.assert_array_equal(w, want_w)                          # NaN = masked bins, bit for bit
        assert np.isnan(w).sum() == np.isnan(want_w).sum() > 0
        narrow, _, _ = src.fetch(c, 31)                                   # a narrower band: diagonals beyond it are dropped
        np.testing.assert_array_equal(narrow, want_raw[:, :31])


@pytest.mark.parametrize('ci', [0, 2])
def test_prep_from_the_cooler_equals_the_reference_rule(ci):
    """IR[d] with the NaN-denominator rule, cDiags (NaN -> 0) and biases (scripts/pyHICCUPS:149-166, restated by the oracle
    and pinned to the reference by the fixtures) from what the reader hands over."""
    c, n = CHROMS[ci]
    src = io.open_source(COOL)
    raw, w, _ = src.fetch(c, NUM)
    for mw in (3, 5):
        IR, cband, biases = orc.prep_from_band(raw.astype(np.int64), w, mw)
        IRh, bh = hband.expected_and_biases(raw, w, mw)
        np.testing.assert_allclose(IRh, IR, rtol=1e-13, atol=0)
        np.testing.assert_array_equal(bh, biases)
        np.testing.assert_array_equal(synthetic.balanced_band(raw.astype(np.int64), w, mw), cband)
        assert (biases == 0).sum() == np.isnan(w).sum()


def test_divisive_weight_column():
    """'KR' holds 1 / weight with divisive_weights = True: the balanced values are the multiplicative column's (count /
    (KR1 KR2) = count w1 w2), while the biases are what the reference forms from the column as stored (1 / KR)."""
    src = io.open_source(COOL)
    for c, n in CHROMS:
        raw, w, b = src.fetch(c, NUM)
        raw2, wkr, bkr = src.fetch(c, NUM, weight_name='KR')
        np.testing.assert_array_equal(raw2, raw)
        ok = ~np.isnan(w)
        np.testing.assert_array_equal(np.isnan(wkr), ~ok)
        np.testing.assert_allclose(wkr[ok], w[ok], rtol=4e-16, atol=0)     # 1 / (1 / w)
        np.testing.assert_allclose(bkr[ok], w[ok], rtol=4e-16, atol=0)     # 1 / KR, as scripts/pyHICCUPS:163-166 would
        assert np.all(bkr[~ok] == 0)
        np.testing.assert_allclose(synthetic.balanced_band(raw.astype(np.int64), wkr, 5),
                                   synthetic.balanced_band(raw.astype(np.int64), w, 5), rtol=1e-15, atol=0)


def test_mcool_uri_and_errors(tmp_path):
    from hicpeaks_amd import cool
    assert cool.parse_uri('a.mcool::/resolutions/5000') == ('a.mcool', '/resolutions/5000')
    assert cool.parse_uri('a.mcool::resolutions/5000') == ('a.mcool', '/resolutions/5000')
    assert cool.parse_uri('b.cool') == ('b.cool', '/')
    with pytest.raises(IOError):
        cool.CoolFile(str(tmp_path / 'missing.cool'))
    with pytest.raises(IOError):
        cool.CoolFile(COOL + '::/resolutions/5000')                      # no such group
    f = cool.CoolFile(COOL)
    with pytest.raises(KeyError):
        f.weights('chrA', 'no_such_column')
    # pixel table of the short chromosome: bins relative to its first bin, upper triangle, sorted by (bin1, bin2)
    b1, b2, cnt = f.pixels('chrB')
    assert b1.min() >= 0 and b2.max() < 57 and np.all(b2 >= b1) and np.all(cnt > 0)
    assert np.all(np.diff(b1 * 1000 + b2) > 0)
    f.close()


def test_chunkwise_reading_equals_h5dread():
    """The pixel table of a big chromosome is read chunk by chunk (H5Dread_chunk) and inflated / un-shuffled on a thread pool
    (cool._H5C.read_big); on the fixture every column is a single chunk, forced through that path here - whole and
    partial ranges, integer and float columns - and compared with the library's own H5Dread."""
    from hicpeaks_amd import cool
    f = cool.CoolFile(COOL)
    h = f.h
    if not isinstance(h, cool._H5C) or not h.have_chunks:
        pytest.skip('h5py backend or an HDF5 library without H5Dread_chunk')
    old = h.PARALLEL_MIN
    h.PARALLEL_MIN = 0
    try:
        n = h.shape('pixels/bin1_id')[0]
        for name, kind in (('pixels/bin1_id', 'i'), ('pixels/bin2_id', 'i'), ('pixels/count', None), ('bins/weight', 'f')):
            m = h.shape(name)[0]
            for a, b in ((0, m), (3, m - 5), (m // 2, m // 2 + 1)):
                np.testing.assert_array_equal(h.read_big(name, a, b, kind, threads=2), h.read(name, a, b, kind))
                os.environ['HPK_READ_PYTHON'] = '1'         # (the Python pool, what a host without the library falls back to)
                try:
                    np.testing.assert_array_equal(h.read_big(name, a, b, kind, threads=2), h.read(name, a, b, kind))
                finally:
                    del os.environ['HPK_READ_PYTHON']
        i, j, c = f.pixels('chrA')
        h.PARALLEL_MIN = 1 << 60
        i2, j2, c2 = f.pixels('chrA')
        np.testing.assert_array_equal(i, i2); np.testing.assert_array_equal(j, j2); np.testing.assert_array_equal(c, c2)
        assert n > 0
        # the same through a pool of result arrays (what the command lines' reader thread does): the columns of one chromosome
        # serve the next once they are handed back, and the counts arrive as int32 (the file stores them so)
        h.PARALLEL_MIN = 0
        f.pool = cool.ArrayPool()
        bases = set()
        for rnd in range(2):
            for ch in f.chromnames:
                f.pool = None
                ref = f.pixels(ch)
                f.pool = pool = getattr(f, '_test_pool', None) or cool.ArrayPool()
                f._test_pool = pool
                got = f.pixels(ch)
                for a, b in zip(got, ref):
                    np.testing.assert_array_equal(a, b)
                assert got[2].dtype == np.int32
                for a in got:
                    if a.base is not None:
                        bases.add(id(a.base))
                f.release(*got)
        assert sum(len(v) for v in pool.free.values()) <= 3 and len(bases) <= 3 + len(f.chromnames)      # (arrays come back: no pile of them)
    finally:
        h.PARALLEL_MIN = old
        f.close()


def test_chunkwise_reading_across_chunk_borders(tmp_path):
    """A pixel table of several chunks (chr21 @5 kb of the synthetic genome, written by scripts/make_cool.py under the image's
    conda python, which has h5py): ranges that start and end inside chunks, decoded on four threads, against H5Dread."""
    import subprocess
    from hicpeaks_amd import cool
    conda = '/opt/conda/bin/python3.9'
    if not os.path.exists(conda):
        pytest.skip('no interpreter with h5py to write the file')
    path = str(tmp_path / 'c21.mcool')
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rc = subprocess.call([conda, os.path.join(repo, 'scripts', 'make_cool.py'), path, '--genome', 'hg38', '--res', '5000', '--num', '2011',
                          '--group', '/resolutions/5000', '--depth', '25', '--chroms', '21'], env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'),
                         stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if rc != 0:
        pytest.skip('make_cool.py could not write the file here')
    f = cool.CoolFile(path + '::/resolutions/5000')
    h = f.h
    if not isinstance(h, cool._H5C) or not h.have_chunks:
        pytest.skip('h5py backend or an HDF5 library without H5Dread_chunk')
    try:
        n = h.shape('pixels/bin2_id')[0]
        assert n > 3 * (1 << 18)                        # more than three chunks of 2^18 pixels
        old, h.PARALLEL_MIN = h.PARALLEL_MIN, 0
        for a, b in ((0, n), (5, 300000), ((1 << 18) - 3, (1 << 18) + (1 << 19) + 7), (n - 10, n)):
            for name, kind in (('pixels/bin1_id', 'i'), ('pixels/bin2_id', 'i'), ('pixels/count', None)):
                np.testing.assert_array_equal(h.read_big(name, a, b, kind, threads=4), h.read(name, a, b, kind))
        # (above: the chunks pread and decoded by libhpk's threads; here: read by H5Dread_chunk and decoded by libhpk | by the Python pool)
        assert h.fd_ok is True
        for var in ('HPK_READ_NO_PREAD', 'HPK_READ_PYTHON'):
            os.environ[var] = '1'
            try:
                np.testing.assert_array_equal(h.read_big('pixels/bin2_id', 5, n - 7, 'i', threads=4), h.read('pixels/bin2_id', 5, n - 7, 'i'))
            finally:
                del os.environ[var]
        h.PARALLEL_MIN = old
        i, j, c = f.pixels('chr21')
        assert i.size == n and (j >= i).all() and c.min() >= 1
    finally:
        f.close()


# ---------------------------------------------------------------------------------------------------------------------
# Legal variations of the schema (scripts/make_cool_variants.py -> tests/golden/cool_variants/): the same small map written
# the ways real files are - other integer / float widths, plain-integer bins/chrom, variable-length names, contiguous and
# differently chunked data sets, an unsummed merge (repeated pixels), both triangles stored, no balancing column.
VARIANTS = os.path.join(GOLDEN_DIR, 'cool_variants')
V_CHROMS = [('chr1', 120), ('chrX', 64)]
V_NUM = 31


def _v_expected(i, n):
    raw, w, _ = synthetic.synth_band(n, V_NUM, depth=30.0, nloops=3, seed=70 + i, loop_dist=(8, 14))
    rr = np.arange(n)[:, None]
    raw[(rr + np.arange(V_NUM)[None, :]) >= n] = 0
    return raw, w


@pytest.mark.parametrize('kind', ['plain', 'merged', 'floats', 'square'])
def test_schema_variants_read_the_same_map(kind):
    src = io.open_source(os.path.join(VARIANTS, kind + '.cool'))
    assert src.binsize == 10000 and src.chromnames == [c for c, _ in V_CHROMS]
    for i, (c, n) in enumerate(V_CHROMS):
        want_raw, want_w = _v_expected(i, n)
        raw, w, b = src.fetch(c, V_NUM)
        assert raw.shape == (n, V_NUM) and b is None
        np.testing.assert_array_equal(raw, want_raw)          # repeats summed, the lower triangle and the trans pixels left out
        if kind == 'floats':                                   # (a float32 weight column: the stored values, widened)
            np.testing.assert_array_equal(w, want_w.astype(np.float32).astype(np.float64))
        else:
            np.testing.assert_array_equal(w, want_w)
        # the pixel table itself (what the device-side band builder takes) gives the same band
        pi, pj, pc, pn, pw_, pb = src.fetch_pixels(c)
        assert pn == n and pb is None
        np.testing.assert_array_equal(hband.band_from_coo(pi, pj, pc, n, V_NUM), want_raw)


def test_missing_weight_column_is_a_clear_error():
    """scripts/pyHICCUPS:143 fails inside cooler with a KeyError on an unbalanced file; here the message says what to do"""
    src = io.open_source(os.path.join(VARIANTS, 'noweight.cool'))
    with pytest.raises(KeyError) as ei:
        src.fetch('chr1', V_NUM)
    assert 'weight' in str(ei.value) and 'balance' in str(ei.value)
    with pytest.raises(KeyError):
        src.fetch_pixels('chrX', 'KR')


def test_cooler_package_backend_drops_the_lower_triangle_of_square_files(monkeypatch):
    """The `cooler`-package backend (not installable here: a stand-in `cooler` module with the calls io.CoolerSource makes) on a
    file in storage mode 'square': `clr.matrix(as_pixels=True).fetch()` hands back both triangles, and the band builders would
    fold (i, j) and (j, i) onto one cell - every off-diagonal count doubled.  Both backends give the same band."""
    import sys
    import types
    n = V_CHROMS[0][1]
    want_raw, want_w = _v_expected(0, n)
    r, k = np.nonzero(want_raw)
    up = (r.astype(np.int64) + 100, (r + k).astype(np.int64) + 100, want_raw[r, k].astype(np.int64))     # (bins offset: a second chromosome)
    off = up[0] != up[1]
    both = tuple(np.concatenate([a, b[off]]) for a, b in zip(up, (up[1], up[0], up[2])))

    class Col(dict):
        def __getitem__(self, key):
            return types.SimpleNamespace(values=dict.__getitem__(self, key))

    class Fetcher:
        def __init__(self, table):
            self.table = table

        def fetch(self, chrom):
            return self.table

    class FakeCooler:
        binsize = 10000
        chromnames = ['chr1']

        def __init__(self, uri, mode):
            self.storage_mode = mode

        def extent(self, chrom):
            return 100, 100 + n

        def matrix(self, balance, as_pixels, join):
            assert balance is False and as_pixels and not join
            px = both if self.storage_mode == 'square' else up
            return Fetcher(Col(bin1_id=px[0], bin2_id=px[1], count=px[2]))

        def bins(self):
            return Fetcher(Col(weight=want_w))

    for mode in ('square', 'symmetric-upper'):
        fake = types.ModuleType('cooler')
        fake.Cooler = lambda uri, mode=mode: FakeCooler(uri, mode)
        monkeypatch.setitem(sys.modules, 'cooler', fake)
        src = io.CoolerSource('whatever.cool')
        assert src.clr is not None
        raw, w, b = src.fetch('chr1', V_NUM)
        np.testing.assert_array_equal(raw, want_raw)
        np.testing.assert_array_equal(w, want_w)
        pi, pj, pc, pn, pw_, pb = src.fetch_pixels('chr1')
        np.testing.assert_array_equal(hband.band_from_coo(pi, pj, pc, n, V_NUM), want_raw)


def test_hpk_decode_chunks_against_zlib_and_numpy():
    """libhpk's host-side chunk decoder (include/hpk.h: hpk_decode_chunks - inflate, un-shuffle, widen on threads) against zlib +
    numpy on chunks written the way HDF5's deflate / shuffle filters store them: every element width and kind, with and
    without shuffle, ranges that start and end inside chunks, both output types; a damaged chunk is an error."""
    import ctypes as C
    import zlib
    from hicpeaks_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    cs, n = 1000, 3500
    for dt, kind in ((np.int32, 0), (np.uint16, 1), (np.float32, 2), (np.int64, 0), (np.float64, 2), (np.uint8, 1), (np.int8, 0)):
        v = rng.integers(0, 100, n).astype(dt)
        size = np.dtype(dt).itemsize
        for shuffle in (0, 1):
            chunks = []
            for c in range(0, n, cs):
                a = np.zeros(cs, dt)
                a[:min(cs, n - c)] = v[c:c + cs]
                b = a.view(np.uint8)
                if shuffle and size > 1:
                    b = np.ascontiguousarray(b.reshape(-1, size).T).reshape(-1)
                chunks.append(zlib.compress(b.tobytes(), 6))
            for start, stop in ((0, n), (7, 2999), (1000, 2000), (999, 1001)):
                c0, c1 = start // cs, (stop - 1) // cs + 1
                bufs = [np.frombuffer(chunks[i], dtype=np.uint8) for i in range(c0, c1)]
                src = (C.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
                sl = (C.c_uint64 * len(bufs))(*[b.size for b in bufs])
                for of in (0, 1):
                    out = np.empty(stop - start, np.float64 if of else np.int64)
                    assert lib.hpk_decode_chunks(src, sl, len(bufs), c0, cs, size, kind, shuffle, start, stop, out.ctypes.data, of, 7, 3) == 0
                    np.testing.assert_array_equal(out, v[start:stop].astype(out.dtype) - 7)
                # int32 output: the integer columns that fit (signed up to four bytes, unsigned up to two), refused for the others
                out = np.empty(stop - start, np.int32)
                rc = lib.hpk_decode_chunks(src, sl, len(bufs), c0, cs, size, kind, shuffle, start, stop, out.ctypes.data, 2, 7, 3)
                if (kind == 0 and size <= 4) or (kind == 1 and size <= 2):
                    assert rc == 0
                    np.testing.assert_array_equal(out, v[start:stop].astype(np.int32) - 7)
                else:
                    assert rc == _lib.ERR_INVALID
    bad = np.frombuffer(b'not a deflate stream at all', dtype=np.uint8)
    out = np.empty(10, np.int64)
    assert lib.hpk_decode_chunks((C.c_void_p * 1)(bad.ctypes.data), (C.c_uint64 * 1)(bad.size), 1, 0, cs, 4, 0, 1, 0, 10,
                                 out.ctypes.data, 0, 0, 1) == _lib.ERR_INVALID
    # a valid stream that inflates to less than a whole chunk (HDF5 stores every chunk whole): refused - the caller then reads
    # through H5Dread - instead of leaving part of a (recycled) result array as it was
    short = np.frombuffer(zlib.compress(np.arange(cs - 8, dtype=np.int32).tobytes(), 6), dtype=np.uint8)
    full = np.frombuffer(zlib.compress(np.arange(cs, dtype=np.int32).tobytes(), 6), dtype=np.uint8)
    for bufs in ([short], [full, short, full]):
        out = np.full(cs * len(bufs), -1, np.int64)
        src = (C.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
        sl = (C.c_uint64 * len(bufs))(*[b.size for b in bufs])
        assert lib.hpk_decode_chunks(src, sl, len(bufs), 0, cs, 4, 0, 0, 0, cs * len(bufs), out.ctypes.data, 0, 0, 1) == _lib.ERR_INVALID


def test_array_pool_hands_arrays_back_by_identity():
    """cool.ArrayPool: the smallest free array that is large enough, arrays compared by identity (several of one dtype, other
    sizes, lie in the list), one that is given twice is kept once, none large enough -> a fresh one and the smallest one dropped."""
    from hicpeaks_amd import cool
    p = cool.ArrayPool()
    a, b, c = p.take(np.int64, 100), p.take(np.int64, 50), p.take(np.int64, 80)
    f = p.take(np.float64, 10)
    p.give(a, b, c)
    p.give(a)
    p.give(np.zeros(5))                                     # (not the pool's: ignored)
    assert len(p.free[np.dtype(np.int64)]) == 3 and not p.free[np.dtype(np.float64)]
    d = p.take(np.int64, 60)
    assert d.base is c.base and d.size == 60
    e = p.take(np.int64, 200)
    assert e.size == 200 and len(p.free[np.dtype(np.int64)]) == 1
    p.give(f)
    assert p.take(np.float64, 3).base is f.base
    # an array the pool dropped is forgotten (its id may come back with another object), and one of a dtype the pool never
    # handed out is ignored, not an error
    assert len(p.owned) == 4                                # a, c (= d's), e, f - not b, which the take of 200 dropped
    assert id(b.base) not in p.owned
    p.give(b)
    assert all(x is not b.base for x in p.free[np.dtype(np.int64)])
    p.give(np.zeros(4, np.int16).reshape(2, 2)[0])
    assert np.dtype(np.int16) not in p.free


def test_hpk_compact_pixels_against_numpy():
    """libhpk's threaded selection of a chromosome's own pixels (hpk_compact_pixels: bin2 inside the chromosome, upper triangle of
    files that store both) against the numpy mask it replaces - int32 / int64 / f64 counts, nothing to drop, everything to
    drop, the count-only call - and through CoolFile.pixels on a table long enough to take that path."""
    import ctypes as C
    from hicpeaks_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    n, nbins = 300000, 1000
    b1 = np.sort(rng.integers(0, nbins, n)).astype(np.int64)
    for square in (0, 1):
        for cdt in (np.int32, np.int64, np.float64):
            for hi in (nbins, 3 * nbins, 1):
                b2 = rng.integers(0, hi, n).astype(np.int64) if square else (b1 + rng.integers(0, hi, n)).astype(np.int64)
                cnt = rng.integers(1, 50, n).astype(cdt)
                keep = (b2 < nbins) & (b2 >= 0)
                if square:
                    keep &= b2 >= b1
                want = int(keep.sum())
                args = (b1.ctypes.data, b2.ctypes.data, cnt.ctypes.data, cnt.dtype.itemsize, n, nbins, square)
                assert lib.hpk_compact_pixels(*args, None, None, None, 5) == want
                o1, o2, oc = np.full(n, -1, np.int64), np.full(n, -1, np.int64), np.zeros(n, cdt)
                assert lib.hpk_compact_pixels(*args, o1.ctypes.data, o2.ctypes.data, oc.ctypes.data, 5) == want
                if want < n:
                    np.testing.assert_array_equal(o1[:want], b1[keep]); np.testing.assert_array_equal(o2[:want], b2[keep])
                    np.testing.assert_array_equal(oc[:want], cnt[keep])
                    assert (o1[want:] == -1).all()
                else:
                    assert (o1 == -1).all()                 # (everything kept: nothing written)
    assert lib.hpk_compact_pixels(b1.ctypes.data, b1.ctypes.data, b1.ctypes.data, 2, n, nbins, 0, None, None, None, 1) == _lib.ERR_INVALID


def test_long_pixel_table_with_trans_pixels(tmp_path):
    """A map whose rows carry trans pixels (as every real one does), long enough for the chunk-wise reader and the threaded
    selection (hpk_compact_pixels): the chromosome's own pixels, with and without the pool of arrays, equal what numpy selects
    from the table read in one piece."""
    import subprocess
    from hicpeaks_amd import cool
    conda = '/opt/conda/bin/python3.9'
    if not os.path.exists(conda):
        pytest.skip('no interpreter with h5py to write the file')
    path = str(tmp_path / 'trans.mcool')
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rc = subprocess.call([conda, os.path.join(repo, 'scripts', 'make_cool.py'), path, '--genome', 'hg38', '--res', '5000', '--num', '1200',
                          '--group', '/resolutions/5000', '--depth', '25', '--chroms', '21', '22', '--trans', '400000'],
                         env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if rc != 0:
        pytest.skip('make_cool.py could not write the file here')
    f = cool.CoolFile(path + '::/resolutions/5000')
    h = f.h
    if not isinstance(h, cool._H5C) or not h.have_chunks:
        pytest.skip('h5py backend or an HDF5 library without H5Dread_chunk')
    try:
        lo, hi = f.extent('chr21')
        off = h.read('indexes/bin1_offset', lo, hi + 1, kind='i')
        p0, p1 = int(off[0]), int(off[-1])
        assert p1 - p0 > h.PARALLEL_MIN
        r1, r2, rc_ = (h.read('pixels/' + nm, p0, p1, kind='i') for nm in ('bin1_id', 'bin2_id', 'count'))
        keep = r2 < hi
        assert 0 < keep.sum() < keep.size                       # trans pixels there, and dropped
        for pooled in (False, True):
            f.pool = cool.ArrayPool() if pooled else None
            for rnd in range(2):
                i, j, c = f.pixels('chr21')
                np.testing.assert_array_equal(i, r1[keep] - lo); np.testing.assert_array_equal(j, r2[keep] - lo)
                np.testing.assert_array_equal(c, rc_[keep])
                f.release(i, j, c)
            if pooled:
                assert sum(len(v) for v in f.pool.free.values()) <= 6       # (full and selected columns: handed back, not piled up)
        i, j, c = f.pixels('chr22')                             # the last chromosome: nothing to drop
        assert i.size == h.shape('pixels/bin1_id')[0] - p1 and (j >= i).all()
    finally:
        f.close()
