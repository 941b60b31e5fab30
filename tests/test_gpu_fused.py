"""The fused kernel (hpk_stencil_s<., ., true>, DESIGN 4.11: bounded single-pair hiccups launches score inside the stencil -
no candidate records, no hpk_score) against the two-kernel path it replaces: same tile geometry, same box sums, same
scoring rules, so the results must be identical bit for bit - survivors, family sizes, Emax, widening log, gap rows."""
import numpy as np
import pytest

from hicpeaks_amd import _lib, synthetic

pytestmark = pytest.mark.gpu

RES, MAXAPART, MAXWW, SIG = 10000, 2000000, 10, 0.05
NUM = MAXAPART // RES + MAXWW + 1


def _bands():
    """chromosomes of one sample-to-sample spread: the widening freezes at 5 ... 8, one band is shorter than the band is wide
    plus twice the edge depth (windows clipped by both matrix ends), one is almost empty"""
    out = []
    for i, (n, depth) in enumerate([(3000, 150.0), (2500, 15.0), (2800, 60.0), (215, 60.0), (3000, 40.0), (2600, 150.0), (900, 2.0),
                                    (3100, 400.0)]):
        raw, w, _ = synthetic.synth_band(n, NUM, depth=depth, nloops=20, seed=900 + i)
        out.append(dict(raw=raw.astype(np.float32), weight=w, num=NUM))
    return out


def _run(fuse, pw, ww, bands, options=()):
    c = _lib.Context(0)
    try:
        c.set_option('fuse', fuse)
        prm = _lib.make_params(_lib.MODE_HICCUPS, pw, ww, MAXWW, SIG, MAXAPART, RES, 16, 0)
        first = c.submit_batch_host(bands[:1], prm).results()           # learns the record bound (not fused: no bound yet)
        for k, v in options:
            c.set_option(k, v)
        return first, c.submit_batch_host(bands, prm).results()
    finally:
        c.close()


def _same(a, b):
    """identical results; bit for bit where both ran under the same tile geometry (hpk_result::halo_w), else to rounding"""
    exact = a.halo_w == b.halo_w
    assert a.steps == b.steps and a.frozen_w == b.frozen_w and a.ncand == b.ncand
    np.testing.assert_array_equal(a.gap, b.gap)
    assert len(a.sets) == len(b.sets)
    for sa, sb in zip(a.sets, b.sets):
        assert sa['nvalid'] == sb['nvalid'] and sa['numbin'] == sb['numbin']
        np.testing.assert_array_equal(sa['chunk_tests'], sb['chunk_tests'])
        np.testing.assert_array_equal(sa['chunk_below'], sb['chunk_below'])
        for k in ('x', 'y', 'O', 'other_zero'):
            np.testing.assert_array_equal(sa[k], sb[k])
        if exact:
            assert sa['emax'] == sb['emax']
            for k in ('E', 'p', 'q', 'bal'):
                np.testing.assert_array_equal(sa[k], sb[k])
        else:
            # (box sums from other tile corners: the parity tolerances of tests/test_gpu_parity.py)
            np.testing.assert_allclose(sa['emax'], sb['emax'], rtol=1e-9)
            np.testing.assert_allclose(sa['E'], sb['E'], rtol=1e-9, atol=0)
            np.testing.assert_allclose(sa['p'], sb['p'], rtol=0, atol=1e-12)
            np.testing.assert_allclose(sa['q'], sb['q'], rtol=0, atol=1e-9)
            np.testing.assert_array_equal(sa['bal'], sb['bal'])


@pytest.mark.parametrize('pw,ww', [([2], [5]), ([1], [3]), ([4], [7]), ([0], [6])])
def test_fused_equals_two_kernel_path(pw, ww):
    bands = _bands()
    bound = min(ww) + 2                                    # the record bound of the batch (test hook): two widths above the narrowest
    f1, fused = _run(1, pw, ww, bands, (('spec_force', bound),))
    t1, two = _run(0, pw, ww, bands, (('spec_force', bound),))
    assert all(r.stencil_kernel == 2 for r in two)
    nf = sum(1 for r in fused if r.stencil_kernel == 3)
    assert nf >= 3, [r.frozen_w for r in fused]            # the rest froze beyond the bound and were computed once more
    # (a band is scored inside the stencil unless it froze beyond the bound - or its survivors' bound proved too narrow, or a
    # survivor region overflowed: all three are computed once more through the two-kernel path)
    assert all(r.stencil_kernel == (2 if r.redone else 3) for r in fused)
    assert all(r.redone & 1 for r in fused if r.frozen_w > bound) and all(r.redone & 1 for r in two if r.frozen_w > bound)
    assert len(set(r.frozen_w for r in fused if r.stencil_kernel == 3)) >= 2      # bands that freeze before the bound and at it
    for a, b in zip(fused, two):
        _same(a, b)
    assert sum(s['x'].size for r in fused for s in r.sets) > 50


def test_fused_reruns():
    """The survivors' bound forced too narrow (every family's cut lies above it) and survivor regions forced to overflow:
    a fused band has no candidate records to score again - it is computed once more through the two-kernel path."""
    bands = _bands()
    _, want = _run(0, [2], [5], bands)
    for options in ((('spec_surv_force', 15),), (('surv_cap', 256),)):
        _, got = _run(1, [2], [5], bands, options)
        assert any(r.redone for r in got)
        for a, b in zip(got, want):
            _same(a, b)
