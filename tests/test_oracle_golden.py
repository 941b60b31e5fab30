"""Pin the CPU oracle (oracle/hiccups_oracle.py) against fixtures made by the real reference."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from oracle import hiccups_oracle as orc

RTOL = 1e-12   # the oracle repeats the reference's add order; what is left is scipy/numpy version noise


def _table_arrays(table):
    keys = sorted(table)
    if not keys:
        return np.zeros((0, 2), np.int64), np.zeros((0, 0))
    return np.array(keys, dtype=np.int64), np.array([[float(v) for v in table[k]] for k in keys])


def _run(g, detail):
    p = g.params
    raw = g['raw']
    n, num = raw.shape[0], g.meta.get('num', raw.shape[1])      # (no 'num' where the reference's prep raised)
    raw = raw[:, :num]
    IR, cband, biases = orc.prep_from_band(raw, g['weight'], g.mw)
    if g.mode == 'hiccups':
        return orc.hiccups(raw, cband, biases, biases, IR, n, num, pw=p['pw'], ww=p['ww'], maxww=p['maxww'],
                           sig=p['sig'], sumq=p['sumq'], double_fold=p['double_fold'],
                           single_fold=p['single_fold'], maxapart=p['maxapart'], res=p['res'],
                           use_raw=p['use_raw'], min_marginal_peaks=p['min_marginal_peaks'],
                           onlyanchor=p['onlyanchor'], min_local_reads=p['min_local_reads'], detail=detail)
    return orc.bhfdr(raw, cband, biases, biases, IR, n, num, pw=p['pw'], ww=p['ww'], sig=p['sig'],
                     maxww=p['maxww'], maxapart=p['maxapart'], res=p['res'],
                     min_marginal_peaks=p['min_marginal_peaks'], onlyanchor=p['onlyanchor'], detail=detail)


@pytest.mark.parametrize('name', golden_names())
def test_prep_matches_reference(name):
    g = load_golden(name)
    if 'IR' not in g:
        pytest.skip('reference prep raised')
    num = g.meta['num']
    IR, cband, biases = orc.prep_from_band(g['raw'][:, :num], g['weight'], g.mw)
    np.testing.assert_allclose(IR[g.mw:], g['IR'], rtol=1e-13, atol=0)
    assert np.all(IR[:g.mw] == 0)
    np.testing.assert_array_equal(cband, g['cband'])
    np.testing.assert_array_equal(biases, g['biases'])


@pytest.mark.parametrize('name', golden_names())
def test_oracle_matches_reference(name):
    g = load_golden(name)
    detail = {}
    if 'prep_exception' in g.meta:          # e.g. a contig shorter than the band: sparse.diags refuses (scripts/pyHICCUPS:148)
        with pytest.raises(ValueError) as ei:
            _run(g, detail)
        assert type(ei.value).__name__ == g.meta['prep_exception'] and 'out of bounds' in str(ei.value)
        return
    if 'exception' in g.meta:
        with pytest.raises((ValueError, ZeroDivisionError)):
            _run(g, detail)
        return
    final = _run(g, detail)

    # G4: resolve counts per executed step
    steps = detail['loc']['steps'] if g.mode == 'hiccups' else detail['steps']
    assert [tuple(int(v) for v in s) for s in g['steps']] == [tuple(s) for s in steps]

    # G5: scoring intermediates
    if g.mode == 'hiccups':
        assert g.meta['nsets'] == len(detail['sets'])
        for t, s in enumerate(detail['sets']):
            np.testing.assert_array_equal(s['x'], g['s%d_x' % t])
            np.testing.assert_array_equal(s['y'], g['s%d_y' % t])
            np.testing.assert_allclose(s['ratio'], g['s%d_ratio' % t], rtol=RTOL, atol=0)
            np.testing.assert_array_equal(s['vx'], g['s%d_vx' % t])
            np.testing.assert_array_equal(s['vy'], g['s%d_vy' % t])
            np.testing.assert_allclose(s['E'], g['s%d_E' % t], rtol=RTOL, atol=0)
            np.testing.assert_array_equal(s['O'], g['s%d_O' % t])
            np.testing.assert_array_equal(s['chunk'], g['s%d_chunk' % t])
            np.testing.assert_allclose(s['p'], g['s%d_p' % t], rtol=0, atol=1e-14)
            np.testing.assert_allclose(s['q'], g['s%d_q' % t], rtol=0, atol=1e-9)
    else:
        np.testing.assert_array_equal(detail['vx'], g['s0_vx'])
        np.testing.assert_array_equal(detail['vy'], g['s0_vy'])
        np.testing.assert_allclose(detail['E'], g['s0_E'], rtol=RTOL, atol=0)
        np.testing.assert_allclose(detail['p'], g['s0_p'], rtol=0, atol=1e-14)
        np.testing.assert_allclose(detail['q'], g['s0_q'], rtol=0, atol=1e-9)
        np.testing.assert_array_equal(detail['reject'], g['s0_reject'])

    # G6: pre-clustering table
    if 'pre_keys' in g:
        k, v = _table_arrays(detail['Donuts'])
        np.testing.assert_array_equal(k, g['pre_keys'])
        np.testing.assert_allclose(v, g['pre_donut'], rtol=1e-9, atol=1e-12)
        if g.mode == 'hiccups':
            k2, v2 = _table_arrays(detail['LL'])
            np.testing.assert_allclose(v2, g['pre_ll'], rtol=1e-9, atol=1e-12)

    # G7: final table and text lines
    k, v = _table_arrays(final)
    np.testing.assert_array_equal(k, g['final_keys'])
    if k.size:
        np.testing.assert_allclose(v, g['final_vals'], rtol=1e-9, atol=1e-12)
    lines = orc.hiccups_lines('T', final, g.params['res']) if g.mode == 'hiccups' else \
        orc.bhfdr_lines('T', final, g.params['res'])
    assert lines == g.meta['lines']


def test_pw_ww_pairs_known_answer():
    # SURVEY §8-A2
    assert orc.pw_ww_pairs([1, 2, 4], [3, 5, 7], 10)[:9] == [(1, 3), (1, 4), (1, 5), (2, 5), (1, 6), (2, 6), (1, 7),
                                                             (2, 7), (4, 7)]
    assert len(orc.pw_ww_pairs([1, 2, 4], [3, 5, 7], 10)) == 18
    assert orc.pw_ww_pairs([2], [12], 10) == []


@pytest.mark.parametrize('name', ['hiccups_p4w7', 'hiccups_union_g3'])
def test_per_step_accumulators_match_reference(name):
    """G3: what the reference read out of its CSR accumulators at every executed step (Reads at the still unresolved
    candidates, callers.py:205; bS / bE of both filters at the candidates the step resolves, callers.py:212-213),
    captured by oracle/gen_golden.py, against the oracle's dense-band accumulators at the same step."""
    g = load_golden(name)
    p = g.params
    num = g.meta['num']
    raw = g['raw'][:, :num]
    n = raw.shape[0]
    IR, cband, biases = orc.prep_from_band(raw, g['weight'], g.mw)
    seen = []

    def trace(k, pi, wi, bS, bE, Reads):
        assert (pi, wi) == tuple(int(v) for v in g['steps'][k][:2])
        ux, uy = g['g3_%d_ux' % k].astype(np.int64), g['g3_%d_uy' % k].astype(np.int64)
        np.testing.assert_array_equal(Reads[ux, uy - ux], g['g3_%d_reads' % k])
        ex, ey = g['g3_%d_ex' % k].astype(np.int64), g['g3_%d_ey' % k].astype(np.int64)
        for fl, a, b in (('K', 'bSK', 'bEK'), ('Y', 'bSY', 'bEY')):
            np.testing.assert_allclose(bS[fl][ex, ey - ex], g['g3_%d_%s' % (k, a)], rtol=1e-13, atol=0)
            np.testing.assert_allclose(bE[fl][ex, ey - ex], g['g3_%d_%s' % (k, b)], rtol=1e-13, atol=0)
        seen.append(k)

    orc.hiccups_local_sums(raw, cband, IR, n, num, p['pw'], p['ww'], p['maxww'], p['maxapart'], p['res'],
                           p['min_local_reads'], trace=trace)
    assert seen == list(range(g.meta['g3_steps'])) and len(seen) >= 4


def test_expected_exactly_on_a_chunk_boundary():
    """Fixture hiccups_E_on_boundary: the reference leaves pixels whose corrected expected is exactly 1.0 (the boundary
    of the first two lambda chunks) in no chunk, p = q = 1 (callers.py:38, 259-260)."""
    g = load_golden('hiccups_E_on_boundary')
    for t in range(g.meta['nsets']):
        E, ch = g['s%d_E' % t], g['s%d_chunk' % t]
        on = E == 1.0
        assert on.sum() >= 20 and np.all(ch[on] == 0) and np.all(g['s%d_p' % t][on] == 1) and np.all(g['s%d_q' % t][on] == 1)
        assert np.all(ch[~on] > 0)
