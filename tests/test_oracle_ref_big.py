"""Pin the CPU oracle against the real reference ABOVE fixture size (tests/golden/ref_*.npz, oracle/gen_golden_big.py):
BASELINE configs[0]'s own shape, chr21 @10 kb (one pair, the three-pair union, bhfdr), chr1 @10 kb at full size
(configs[1]) and a 2 011-diagonal band.  The band is regenerated from the fixture's seed; the oracle's whole per-pixel
population is compared with the reference's checksums, family sizes, survivors' neighbourhood and random sample."""
import numpy as np
import pytest

import refbig
from oracle import hiccups_oracle as orc

# seconds of one host core for the oracle: cfg1 1, chr21 2-6, chr1 @10 kb 12, wide 5 kb band 25
CASES = refbig.names()


@pytest.mark.parametrize('name', CASES)
def test_oracle_matches_reference_at_size(name):
    g = refbig.load(name)
    p = g.params
    raw, weight = refbig.band(g)
    n, num = raw.shape
    assert (n, num) == (g.meta['chromLen'], g.meta['num'])
    IR, cband, biases = orc.prep_from_band(raw, weight, g.mw)
    np.testing.assert_allclose(IR[g.mw:], g['IR'], rtol=1e-13, atol=0)
    det = {}
    if g.mode == 'hiccups':
        final = orc.hiccups(raw, cband, biases, biases, IR, n, num, pw=p['pw'], ww=p['ww'], maxww=p['maxww'], sig=p['sig'],
                            sumq=p['sumq'], double_fold=p['double_fold'], single_fold=p['single_fold'], maxapart=p['maxapart'],
                            res=p['res'], use_raw=p['use_raw'], min_marginal_peaks=p['min_marginal_peaks'],
                            onlyanchor=p['onlyanchor'], min_local_reads=p['min_local_reads'], detail=det)
        assert [tuple(int(v) for v in s) for s in g['steps']] == [tuple(s) for s in det['loc']['steps']]
        assert det['loc']['vx'].size == g.meta['ncand']
        assert len(det['sets']) == g.meta['nsets']
        for t, s in enumerate(det['sets']):
            refbig.check_population(g, t, s['vx'], s['vy'], s['E'], s['O'], s['p'], s['q'], chunk=s['chunk'])
        k, v = refbig.table_arrays(det['Donuts'])
        np.testing.assert_array_equal(k, g['pre_keys'])
        np.testing.assert_allclose(v, g['pre_donut'], rtol=1e-9, atol=1e-12)
        refbig.check_final(g, final, orc.hiccups_lines('T', final, p['res']))
    else:
        final = orc.bhfdr(raw, cband, biases, biases, IR, n, num, pw=p['pw'], ww=p['ww'], sig=p['sig'], maxww=p['maxww'],
                          maxapart=p['maxapart'], res=p['res'], min_marginal_peaks=p['min_marginal_peaks'],
                          onlyanchor=p['onlyanchor'], detail=det)
        assert [tuple(int(v) for v in s) for s in g['steps']] == [tuple(s) for s in det['steps']]
        refbig.check_population(g, 0, det['vx'], det['vy'], det['E'], det['O'], det['p'], det['q'], reject=det['reject'])
        refbig.check_final(g, final, orc.bhfdr_lines('T', final, p['res']))
