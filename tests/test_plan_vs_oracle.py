"""The ring-multiplicity plan of libhpk (hpk_plan_rings, host only) against the oracle's cell-by-cell
accumulators: after every executed step, sum_rho m_rho * ring_rho must reproduce bS / bE / Reads."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from hicpeaks_amd import _lib
from oracle import hiccups_oracle as orc


def ring_sum(A, W, rho, ll_only):
    S = orc._Shifter(A, W)
    out = np.zeros_like(A, dtype=np.float64)
    for di in range(-rho, rho + 1):
        for dj in range(-rho, rho + 1):
            if max(abs(di), abs(dj)) != rho or di == 0 or dj == 0:
                continue
            if ll_only and not (di > 0 and dj < 0):
                continue
            out = out + S(di, dj)
    return out


@pytest.mark.parametrize('name', ['hiccups_union_shallow', 'hiccups_swapped_pairs', 'hiccups_p2w5_shallow',
                                  'hiccups_w8_pairdrop'])
def test_ring_plan_reproduces_oracle_accumulators(name):
    g = load_golden(name)
    p = g.params
    num = g.meta['num']
    raw = g['raw'][:, :num]
    n = raw.shape[0]
    IR, cband, biases = orc.prep_from_band(raw, g['weight'], g.mw)
    W = p['maxww']
    prm = _lib.make_params(_lib.MODE_HICCUPS, p['pw'], p['ww'], W, p['sig'], p['maxapart'], p['res'],
                           p['min_local_reads'])
    steps, mk, mr = _lib.plan_rings(prm)
    X = orc.expected_band(IR, n, num, g.mw)
    rawf = raw.astype(np.float64)
    rings = {}
    for rho in range(1, W + 1):
        rings[rho] = (ring_sum(cband, W, rho, False), ring_sum(cband, W, rho, True),
                      ring_sum(X, W, rho, False), ring_sum(X, W, rho, True), ring_sum(rawf, W, rho, True))
    seen = []

    def trace(si, pi, wi, bS, bE, Reads):
        # the oracle only runs executed steps; they form a prefix of the plan
        assert steps[si] == (pi, wi)
        m, r = mk[si], mr[si]
        want = [sum(m[rho] * rings[rho][t] for rho in range(1, W + 1)) for t in range(4)]
        wantR = sum(r[rho] * rings[rho][4] for rho in range(1, W + 1))
        np.testing.assert_allclose(bS['K'], want[0], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(bS['Y'], want[1], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(bE['K'], want[2], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(bE['Y'], want[3], rtol=1e-12, atol=1e-12)
        np.testing.assert_array_equal(Reads, wantR)
        seen.append(si)

    orc.hiccups_local_sums(raw, cband, IR, n, num, p['pw'], p['ww'], W, p['maxapart'], p['res'],
                           p['min_local_reads'], trace=trace)
    assert len(seen) >= 4
