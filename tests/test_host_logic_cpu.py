"""Host half of the drop-in (gap filter, donut/LL combination, union over pairs, clustering, result dicts;
hicpeaks_amd/callers.py + clustering.py) fed with the per-set survivors recorded from the real reference:
must reproduce the reference's final table and text lines.  No GPU involved."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from hicpeaks_amd import callers, clustering
from hicpeaks_amd.cli import format_hiccups, format_bhfdr


class FakeResult(object):
    pass


def _fake_hiccups_result(g):
    p = g.params
    sig = p['sig']
    IRfull = np.r_[np.zeros(g.mw), g['IR']]
    cband = g['cband']
    R = FakeResult()
    R.steps = [(int(a), int(b), int(c), True) for a, b, c in g['steps']]
    R.gap = cband.sum(axis=1) == 0
    R.sets = []
    for t in range(g.meta['nsets']):
        vx, vy = g['s%d_vx' % t].astype(np.int64), g['s%d_vy' % t].astype(np.int64)
        q = g['s%d_q' % t]
        k = q <= sig
        s = dict(pair=t // 2, fl='KY'[t % 2], nvalid=int(vx.size), numbin=int(g['s%d_chunk' % t].max(initial=0)),
                 x=vx[k], y=vy[k], O=g['s%d_O' % t][k], E=g['s%d_E' % t][k], p=g['s%d_p' % t][k], q=q[k],
                 bal=cband[vx[k], vy[k] - vx[k]])
        if t % 2 == 0:     # K set: is the lower-left corrected expected zero there?  (callers.py:330)
            yx, yy, yr = g['s%d_x' % (t + 1)], g['s%d_y' % (t + 1)], g['s%d_ratio' % (t + 1)]
            cE = dict(zip(zip(yx.tolist(), yy.tolist()), (IRfull[yy - yx] * yr).tolist()))
            s['other_zero'] = np.array([cE.get(key, 0.0) == 0 for key in zip(s['x'].tolist(), s['y'].tolist())], dtype=bool)
        else:
            s['other_zero'] = np.zeros(s['x'].size, dtype=bool)
        R.sets.append(s)
    return R


def _arr(table):
    keys = sorted(table)
    if not keys:
        return np.zeros((0, 2), np.int64), np.zeros((0, 0))
    return np.array(keys, dtype=np.int64), np.array([[float(v) for v in table[k]] for k in keys])


@pytest.mark.parametrize('name', [n for n in golden_names('hiccups')])
def test_finish_hiccups_reproduces_reference_table(name):
    g = load_golden(name)
    if 'exception' in g.meta or 'prep_exception' in g.meta:
        pytest.skip('reference raised')
    p = g.params
    R = _fake_hiccups_result(g)
    n = g['raw'].shape[0]
    final, table = callers._finish_hiccups(R, n, 'T', p['pw'], p['ww'], p['sig'], p['sumq'], p['double_fold'],
                                           p['single_fold'], p['res'], p['use_raw'], p['min_marginal_peaks'],
                                           p['onlyanchor'])
    k, v = _arr(final)
    np.testing.assert_array_equal(k, g['final_keys'])
    if k.size:
        np.testing.assert_allclose(v, g['final_vals'], rtol=1e-12, atol=0)
    assert format_hiccups('T', final, p['res'], sort=True) == g.meta['lines']
    # pre-clustering table
    pre = {(a // p['res'], b // p['res']): table[(a, b)][3:8] for a, b in table}
    k, v = _arr(pre)
    np.testing.assert_array_equal(k, g['pre_keys'])
    if k.size:
        np.testing.assert_allclose(v, g['pre_donut'], rtol=1e-12, atol=0)


@pytest.mark.parametrize('name', golden_names())
def test_clustering_matches_reference(name):
    g = load_golden(name)
    if 'pre_keys' not in g or g['pre_keys'].shape[0] == 0:
        pytest.skip('nothing to cluster')
    p = g.params
    res = p['res']
    keys = [tuple(int(v) for v in k) for k in g['pre_keys']]
    Donuts = {k: tuple(v) for k, v in zip(keys, g['pre_donut'].tolist())}
    if g.mode == 'hiccups':
        LL = {k: tuple(v) for k, v in zip(keys, g['pre_ll'].tolist())}
        peaks = clustering.local_clustering(Donuts, LL, res, min_count=p['min_marginal_peaks'], r=2 * res,
                                            sumq=p['sumq'], onlysummit=p['onlyanchor'])
        got = {(px[0] * res, px[1] * res): (cen[0] * res, cen[1] * res, rad * res) for px, cen, rad in peaks}
    else:
        peaks = clustering.local_clustering(Donuts, None, res, min_count=p['min_marginal_peaks'], r=2 * res,
                                            onlysummit=p['onlyanchor'])
        got = {(px[0] * res, px[1] * res): (cen[0] * res, cen[1] * res, rad * res) for px, cen, rad in peaks
               if Donuts[px][1] > 2}
    want = {tuple(int(v) for v in k): tuple(int(x) for x in v[:3]) for k, v in zip(g['final_keys'], g['final_vals'])}
    assert got == want


def test_clustering_rectangles_dealt_once_equal_the_walk_over_all_cells(monkeypatch):
    """local_clustering deals the pixels to their (x anchor, y anchor) rectangles once instead of testing every cell of every
    pair of anchors (callers.py:700-706: quadratic in the anchors, a second of Python per large chromosome at depth); the same
    list, in the same order, as the walk - random clustered pixel sets, three parameter sets, with and without the LL table."""
    import warnings
    rng = np.random.default_rng(0)
    for trial in range(25):
        n, nclu = int(rng.integers(200, 3000)), int(rng.integers(1, 80))
        D, L = {}, {}
        for c in range(nclu):
            cx = int(rng.integers(0, n)); cy = cx + int(rng.integers(5, 200))
            for _ in range(int(rng.integers(1, 60))):
                i, j = cx + int(rng.integers(-4, 5)), cy + int(rng.integers(-4, 5))
                if i < 0 or j <= i:
                    continue
                D[(i, j)] = (float(rng.integers(1, 400)) / 7.0, float(rng.random()), float(rng.random() * 0.02))
                L[(i, j)] = (float(rng.integers(1, 400)) / 7.0, float(rng.random()), float(rng.random() * 0.02))
        for k in range(int(rng.integers(0, 30))):
            i = int(rng.integers(0, n)); j = i + int(rng.integers(3, 300))
            D[(i, j)] = L[(i, j)] = (float(rng.integers(1, 50)), 0.1, float(rng.random() * 0.02))
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            for res, r, mc, onlys, ll in ((10000, 20000, 3, False, True), (5000, 20000, 2, True, True), (25000, 50000, 3, False, False)):
                kw = dict(onlysummit=onlys, min_count=mc, r=r, sumq=0.01)
                got = clustering.local_clustering(D, L if ll else None, res, **kw)
                monkeypatch.setattr(clustering, '_WALK_ALL_CELLS', True)
                want = clustering.local_clustering(D, L if ll else None, res, **kw)
                monkeypatch.setattr(clustering, '_WALK_ALL_CELLS', False)
                assert got == want and len(want) > 0


def test_clustering_components_equal_dbscan():
    from sklearn.cluster import dbscan
    rng = np.random.default_rng(5)
    for trial in range(20):
        pts = [tuple(int(v) for v in p) for p in rng.integers(0, 25, size=(int(rng.integers(2, 60)), 2))]
        pts = list(dict.fromkeys(pts))
        if len(pts) < 2:
            continue
        lab = clustering._components(pts, 2)
        _, ref = dbscan(np.array(pts), eps=2, min_samples=2)
        assert np.array_equal(lab == -1, ref == -1)
        for i in range(len(pts)):
            for j in range(len(pts)):
                if ref[i] != -1 and ref[j] != -1:
                    assert (lab[i] == lab[j]) == (ref[i] == ref[j])


def test_gap_keep_matches_loop():
    rng = np.random.default_rng(3)
    n, mw = 200, 5
    gap = rng.random(n) < 0.05
    gap[-mw:] = True
    x = rng.integers(0, n - 10, 300)
    y = np.minimum(x + rng.integers(5, 50, 300), n - 1)
    keep = callers._gap_keep(x, y, gap, mw, n)
    gaps = set(np.where(gap)[0].tolist())
    for t in range(x.size):
        reg = set()
        for v in (x[t], y[t]):
            lo = (v - mw) if v > mw else 0
            hi = (v + mw) if (v + mw) < n else (n - 1)
            reg |= set(range(lo, hi))
        assert keep[t] == (not (reg & gaps))


def test_default_context_runs_under_the_own_frozen_width_layout(monkeypatch):
    """The process-wide context the drop-in functions and the command lines score on is switched to spec_halo = 2 when it is made
    (values that depend on the chromosome alone); HPK_SPEC_HALO in the environment is left standing; one context per device."""
    from hicpeaks_amd import _lib
    calls = []

    class Fake(object):
        def __init__(self, device):
            self.device = device

        def set_option(self, k, v):
            calls.append((self.device, k, v))

    monkeypatch.setattr(_lib, 'Context', Fake)
    monkeypatch.setattr(_lib, '_default_ctx', {})
    monkeypatch.delenv('HPK_SPEC_HALO', raising=False)
    c = _lib.default_context(3)
    assert calls == [(3, 'spec_halo', 2)] and _lib.default_context(3) is c and len(calls) == 1
    monkeypatch.setenv('HPK_SPEC_HALO', '1')
    assert _lib.default_context(4).device == 4 and len(calls) == 1
