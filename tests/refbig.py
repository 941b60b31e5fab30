"""Reference-pinned fixtures above fixture size (tests/golden/ref_*.npz, made by oracle/gen_golden_big.py from the real
reference): loader and the comparisons shared by the oracle test (CPU) and the GPU test.

A fixture stores outputs only; the band is regenerated from its seed (`meta['gen']`).  Per (pair, filter) set it holds the
population's size and checksums, the family sizes, every pixel with q <= 2 sig, and a seeded random sample."""
import glob
import json
import os

import numpy as np

from conftest import GOLDEN_DIR, Golden

M63 = (1 << 63) - 1
E_RTOL, P_ATOL, Q_ATOL = 1e-9, 1e-12, 1e-9


def names():
    return sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, 'ref_*.npz')))


def load(name):
    return Golden(os.path.join(GOLDEN_DIR, 'ref_' + name + '.npz'))


def band(g):
    from hicpeaks_amd import synthetic
    raw, weight, _ = synthetic.synth_band(**g.meta['gen'])
    return raw, weight


def _isum(a):
    return int(np.sum(a.astype(object)) & M63) if a.size else 0


def _lookup(vx, vy, x, y):
    """positions of (x, y) in the row-major sorted (vx, vy)"""
    key = vx.astype(np.int64) * (1 << 32) + vy.astype(np.int64)
    ii = np.searchsorted(key, x.astype(np.int64) * (1 << 32) + y.astype(np.int64))
    assert np.all(ii < key.size) and np.array_equal(vx[ii], x) and np.array_equal(vy[ii], y)
    return ii


def check_population(g, t, vx, vy, E, O, p, q, chunk=None, reject=None):
    """A full per-pixel population (the oracle's) against set t of the fixture."""
    pre = 's%d_' % t
    sig = g.params['sig']
    assert vx.size == int(g[pre + 'nvalid'])
    assert _isum(vx.astype(np.int64)) == int(g[pre + 'hx']) and _isum(vy.astype(np.int64)) == int(g[pre + 'hy'])
    assert _isum(vx.astype(np.int64) * vy.astype(np.int64)) == int(g[pre + 'hxy'])
    np.testing.assert_allclose(E.max() if E.size else 0.0, float(g[pre + 'Emax']), rtol=1e-12)
    np.testing.assert_allclose(E.sum(), float(g[pre + 'sumE']), rtol=1e-11)
    np.testing.assert_allclose(p.sum(), float(g[pre + 'sump']), rtol=1e-10)
    if chunk is not None:
        tests = np.bincount(chunk, minlength=g[pre + 'chunk_tests'].size)
        np.testing.assert_array_equal(tests[1:], g[pre + 'chunk_tests'][1:])
        below = np.bincount(chunk[p <= sig], minlength=tests.size)
        np.testing.assert_array_equal(below[1:], g[pre + 'chunk_below'][1:])
    # the survivors' neighbourhood: every pixel with q <= 2 sig, as a set and value by value
    near = q <= 2 * sig
    if reject is not None:
        near = near | reject
    firm = np.abs(q - 2 * sig) > 1e-8
    assert int((~firm).sum()) <= 3, int((~firm).sum())
    kx, ky = g[pre + 'kx'].astype(np.int64), g[pre + 'ky'].astype(np.int64)
    got = set(zip(vx[near & firm].tolist(), vy[near & firm].tolist()))
    maybe = set(zip(vx[~firm].tolist(), vy[~firm].tolist()))
    want = set(zip(kx.tolist(), ky.tolist()))
    assert got <= want and want <= (got | maybe), (len(got - want), len(want - got))
    for tag in ('k', 'r'):
        x, y = g[pre + tag + 'x'].astype(np.int64), g[pre + tag + 'y'].astype(np.int64)
        ii = _lookup(vx, vy, x, y)
        np.testing.assert_array_equal(O[ii], g[pre + tag + 'O'])
        np.testing.assert_allclose(E[ii], g[pre + tag + 'E'], rtol=1e-12, atol=0)
        np.testing.assert_allclose(p[ii], g[pre + tag + 'p'], rtol=0, atol=1e-14)
        np.testing.assert_allclose(q[ii], g[pre + tag + 'q'], rtol=0, atol=Q_ATOL)
        if tag == 'r' and chunk is not None:
            np.testing.assert_array_equal(chunk[ii], g[pre + 'rchunk'])
    if reject is not None:
        ii = _lookup(vx, vy, kx, ky)
        np.testing.assert_array_equal(reject[ii], g[pre + 'kreject'])


def check_survivors(g, t, s, bh=False):
    """What the library returns for set t (pixels with q <= sig, family sizes) against the fixture."""
    pre = 's%d_' % t
    sig = g.params['sig']
    assert s['nvalid'] == int(g[pre + 'nvalid'])
    kx, ky, kq = g[pre + 'kx'].astype(np.int64), g[pre + 'ky'].astype(np.int64), g[pre + 'kq']
    sel = g[pre + 'kreject'] if bh else (kq <= sig)
    firm = np.abs(kq - sig) > 1e-8
    # (pixels within 1e-8 of the threshold may fall on either side: there must be next to none of them - a drift of the
    # q-values towards the threshold would otherwise hide behind the mask)
    assert int((~firm).sum()) <= 3, int((~firm).sum())
    want = set(zip(kx[sel & firm].tolist(), ky[sel & firm].tolist()))
    maybe = set(zip(kx[~firm].tolist(), ky[~firm].tolist()))
    got = set(zip(s['x'].tolist(), s['y'].tolist()))
    assert want <= got and got <= (want | maybe), (len(want - got), len(got - want))
    order = np.lexsort((ky, kx))
    ii = order[_lookup(kx[order], ky[order], s['x'].astype(np.int64), s['y'].astype(np.int64))]
    np.testing.assert_array_equal(s['O'], g[pre + 'kO'][ii])
    np.testing.assert_allclose(s['E'], g[pre + 'kE'][ii], rtol=E_RTOL, atol=0)
    np.testing.assert_allclose(s['p'], g[pre + 'kp'][ii], rtol=0, atol=P_ATOL)
    np.testing.assert_allclose(s['q'], g[pre + 'kq'][ii], rtol=0, atol=Q_ATOL)
    tests = g[pre + 'chunk_tests']
    if bh:
        assert s['chunk_tests'][1] == tests[1]
    else:
        np.testing.assert_array_equal(s['chunk_tests'][1:tests.size], tests[1:])
        assert s['chunk_tests'][tests.size:].sum() == 0
        np.testing.assert_array_equal(s['chunk_below'][1:tests.size], g[pre + 'chunk_below'][1:])


def table_arrays(table):
    keys = sorted(table)
    if not keys:
        return np.zeros((0, 2), np.int64), np.zeros((0, 0))
    return np.array(keys, dtype=np.int64), np.array([[float(v) for v in table[k]] for k in keys])


def check_final(g, final, lines=None):
    k, v = table_arrays(final)
    np.testing.assert_array_equal(k, g['final_keys'])
    if k.size:
        np.testing.assert_allclose(v, g['final_vals'], rtol=1e-9, atol=1e-9)
    if lines is not None:
        assert lines == g.meta['lines']


def _json(meta):
    return json.dumps(meta)
