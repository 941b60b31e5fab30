import numpy as np
import pytest

from conftest import golden_names, load_golden
from hicpeaks_amd import band, synthetic


@pytest.mark.parametrize('name', golden_names()[:6])
def test_expected_and_biases_match_reference_prep(name):
    g = load_golden(name)
    if 'IR' not in g:
        pytest.skip('reference prep raised')
    num = g.meta['num']
    IR, biases = band.expected_and_biases(g['raw'][:, :num], g['weight'], g.mw)
    np.testing.assert_allclose(IR[g.mw:], g['IR'], rtol=1e-13, atol=0)
    np.testing.assert_array_equal(biases, g['biases'])


def test_band_from_coo_roundtrip():
    raw, w, _ = synthetic.synth_band(200, 31, depth=5.0, seed=3)
    i, j, v = synthetic.band_to_coo(raw)
    # feed both triangles in scrambled order
    rng = np.random.default_rng(0)
    o = rng.permutation(i.size)
    back = band.band_from_coo(j[o], i[o], v[o], 200, 31, dtype=np.int64)
    np.testing.assert_array_equal(back, raw)
    assert band.band_pixels(200, 31, 5, 20) == sum(200 - d for d in range(5, 21))


def test_bench_configurations_match_the_survey_table():
    """Band pixel counts of bench.py's configurations = SURVEY.md section 8 size table (d in [min(ww), D])."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location('bench', os.path.join(os.path.dirname(__file__), '..', 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from hicpeaks_amd import band, synthetic

    def px(cfg, n):
        D = cfg['maxapart'] // cfg['res']
        return band.band_pixels(n, D + cfg['maxww'] + 1, min(cfg['ww']), D)
    c = bench.CONFIGS
    assert px(c['chr1_10kb'], c['chr1_10kb']['n']) == 12223176
    assert px(c['deep_1kb'], c['deep_1kb']['n']) == 496999010
    assert px(c['chr1_5kb'], c['chr1_5kb']['n']) == 97284269
    g10 = sum(px(c['wg_10kb_union'], n) for n in synthetic.hg38_bins(10000).values())
    g5 = sum(px(c['wg_5kb'], n) for n in synthetic.hg38_bins(5000).values())
    assert g10 == 148070091 and g10 * 3 == 444210273
    assert g5 == 1162778169
    assert bench.BYTES_PER_PX == 20.0 and bench.HBM_PEAK_GBS == 8000.0


class _FakeSelector(object):
    def __init__(self, fn):
        self.fetch = fn


class _FakeCooler(object):
    """Duck-typed stand-in for cooler.Cooler (the package is not installed here): two chromosomes in one genome-wide bin
    table, upper-triangle pixel table, a weight column with NaN bins - what scripts/pyHICCUPS:142-163 reads."""
    def __init__(self, uri):
        import pandas as pd
        self.uri = uri
        self.binsize = 10000
        self.chromnames = ['chr1', 'chr2']
        self._n = {'chr1': 180, 'chr2': 130}
        self._lo = {'chr1': 0, 'chr2': 180}
        self._bands, self._w = {}, {}
        frames = []
        for s, c in enumerate(self.chromnames):
            raw, w, _ = synthetic.synth_band(self._n[c], 41, depth=8.0, seed=40 + s)
            self._bands[c], self._w[c] = raw, w
            i, j, v = synthetic.band_to_coo(raw)
            frames.append(pd.DataFrame({'bin1_id': i + self._lo[c], 'bin2_id': j + self._lo[c], 'count': v.astype(np.int32)}))
        self._px = pd.concat(frames, ignore_index=True)
        self._bins = pd.DataFrame({'chrom': np.repeat(self.chromnames, [180, 130]),
                                   'weight': np.r_[self._w['chr1'], self._w['chr2']],
                                   'w2': np.r_[self._w['chr1'], self._w['chr2']] * 2.0,
                                   'KR': 1.0 / np.r_[self._w['chr1'], self._w['chr2']]})

    def extent(self, chrom):
        return self._lo[chrom], self._lo[chrom] + self._n[chrom]

    def matrix(self, balance=False, as_pixels=False, join=False, sparse=False):
        assert balance is False and as_pixels and not join
        def fetch(chrom):
            lo, hi = self.extent(chrom)
            p = self._px
            return p[(p.bin1_id >= lo) & (p.bin1_id < hi) & (p.bin2_id >= lo) & (p.bin2_id < hi)]
        return _FakeSelector(fetch)

    def bins(self):
        return _FakeSelector(lambda chrom: self._bins[self._bins.chrom == chrom])


def test_cooler_source_against_a_fake_cooler(monkeypatch):
    """io.CoolerSource (counterpart of the cooler reads in scripts/pyHICCUPS:142-143, 163): bins relative to the
    chromosome's extent, band by one O(nnz) scatter, weights from the chosen column."""
    import sys, types
    from hicpeaks_amd import io
    mod = types.ModuleType('cooler')
    mod.Cooler = _FakeCooler
    monkeypatch.setitem(sys.modules, 'cooler', mod)
    src = io.open_source('fake.cool::resolutions/10000')
    assert isinstance(src, io.CoolerSource) and src.binsize == 10000 and src.chromnames == ['chr1', 'chr2']
    assert src.nbins('chr2') == 130
    for c in src.chromnames:
        raw, w, b = src.fetch(c, 31)
        assert b is None and raw.dtype == np.float32 and raw.shape == (src.nbins(c), 31)
        np.testing.assert_array_equal(raw, src.clr._bands[c][:, :31])         # diagonals beyond num are dropped
        np.testing.assert_array_equal(w, src.clr._w[c])
        raw50, _, _ = src.fetch(c, 50)                                         # wider than what is stored: zeros
        assert not raw50[:, 41:].any()
        np.testing.assert_array_equal(raw50[:, :41], src.clr._bands[c])
    _, w2, b2 = src.fetch('chr1', 31, weight_name='w2')
    np.testing.assert_array_equal(w2, src.clr._w['chr1'] * 2.0)
    assert b2 is None
    # a divisive column (cooler: balanced = count / (KR1 KR2)): 1 / column as the weights, and the biases the reference
    # forms from the column as stored (scripts/pyHICCUPS:163-166)
    _, wkr, bkr = src.fetch('chr1', 31, weight_name='KR')
    kr = 1.0 / src.clr._w['chr1']
    ok = ~np.isnan(kr)
    np.testing.assert_array_equal(np.isnan(wkr), ~ok)
    np.testing.assert_array_equal(wkr[ok], 1.0 / kr[ok])
    np.testing.assert_array_equal(bkr[ok], 1.0 / kr[ok])
    assert np.all(bkr[~ok] == 0)


def test_band_from_coo_wide_integer_counts():
    """int64 / uint32 counts do not wrap on their way into the band; counts the f32 band cannot hold exactly
    are refused."""
    i = np.array([0, 1, 2]); j = np.array([1, 3, 2])
    for dt in (np.int64, np.uint32, np.int32, np.float64):
        raw = band.band_from_coo(i, j, np.array([5, 70000, 16777215], dtype=dt), 4, 3)
        assert raw[0, 1] == 5 and raw[1, 2] == 70000 and raw[2, 0] == 16777215
    with pytest.raises(ValueError):
        band.band_from_coo(i, j, np.array([1, 2, 1 << 33], dtype=np.int64), 4, 3)
    with pytest.raises(ValueError):
        band.band_from_coo(i, j, np.array([1, 2, 1 << 24], dtype=np.uint32), 4, 3)


def test_band_from_coo_threaded_path(monkeypatch):
    """The threaded scatter of hpk_band_from_coo (large pixel tables in row order are cut at row changes, one stretch per thread;
    the threshold per thread is lowered through HPK_COO_MIN_PER_THREAD): a sorted table gives the serial result; a shuffled one
    is recognised by the read-only pass and scattered by one thread, repeats and all; a bin outside the chromosome is an error."""
    n, num = 900, 61
    raw, _, _ = synthetic.synth_band(n, num, depth=30.0, nloops=5, seed=5)
    i, j, v = synthetic.band_to_coo(raw)                 # row-major: in row order
    i, j, v = np.concatenate([i, i[:500]]), np.concatenate([j, j[:500]]), np.concatenate([v, v[:500]])      # repeats ...
    o = np.argsort(i, kind='stable')
    i, j, v = i[o], j[o], v[o]                           # ... kept in row order
    monkeypatch.delenv('HPK_COO_MIN_PER_THREAD', raising=False)
    serial = band.band_from_coo(i, j, v, n, num)
    monkeypatch.setenv('HPK_COO_MIN_PER_THREAD', '1000')
    assert i.size // 1000 >= 2
    np.testing.assert_array_equal(band.band_from_coo(i, j, v, n, num), serial)
    np.testing.assert_array_equal(band.band_from_coo(j, i, v, n, num), serial)        # either orientation
    p = np.random.default_rng(3).permutation(i.size)
    np.testing.assert_array_equal(band.band_from_coo(i[p], j[p], v[p], n, num), serial)
    bad = j.copy()
    bad[i.size // 2] = n
    from hicpeaks_amd import _lib
    with pytest.raises(_lib.HpkError):
        band.band_from_coo(i, bad, v, n, num)
    monkeypatch.delenv('HPK_COO_MIN_PER_THREAD')
    with pytest.raises(_lib.HpkError):
        band.band_from_coo(i, bad, v, n, num)
