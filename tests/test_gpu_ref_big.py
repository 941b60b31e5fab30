"""The HIP path against the REAL reference above fixture size (tests/golden/ref_*.npz, oracle/gen_golden_big.py): BASELINE
configs[0]'s own shape (n = 1 869, num = 411, (1,3), 10 Mb @25 kb), chr21 @10 kb (n = 4 671: one pair, the three-pair
union, bhfdr), chr1 @10 kb at full size (configs[1]: n = 24 896, num = 511) and a 2 011-diagonal band ((4,7) @5 kb) -
every case as a single call (the plan's tile geometry) and inside a batch under the bounds inherited from that call - and
the wide band ((2,5) and (4,7), num = 2011) against the oracle."""
import numpy as np
import pytest

import refbig
from hicpeaks_amd import _lib, callers, synthetic
from hicpeaks_amd.cli import format_hiccups, format_bhfdr
from oracle import hiccups_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def _kw(g):
    p = g.params
    if g.mode == 'hiccups':
        return dict(pw=p['pw'], ww=p['ww'], maxww=p['maxww'], sig=p['sig'], sumq=p['sumq'], double_fold=p['double_fold'],
                    single_fold=p['single_fold'], maxapart=p['maxapart'], res=p['res'], use_raw=p['use_raw'],
                    min_marginal_peaks=p['min_marginal_peaks'], onlyanchor=p['onlyanchor'], min_local_reads=p['min_local_reads'])
    return dict(pw=p['pw'], ww=p['ww'], sig=p['sig'], maxww=p['maxww'], maxapart=p['maxapart'], res=p['res'],
                min_marginal_peaks=p['min_marginal_peaks'], onlyanchor=p['onlyanchor'])


def _check_result(g, R, final):
    assert [(a, b, c) for a, b, c, ex in R.steps if ex] == [tuple(int(v) for v in s) for s in g['steps']]
    assert R.ncand == g.meta['ncand'] and R.band_px == g.meta['band_px']
    if g.mode == 'hiccups':
        assert len(R.sets) == g.meta['nsets']
        for t, s in enumerate(R.sets):
            refbig.check_survivors(g, t, s)
        lines = ''.join(format_hiccups('T', final, g.params['res']))
    else:
        refbig.check_survivors(g, 0, R.sets[0], bh=True)
        lines = ''.join(format_bhfdr('T', final, g.params['res']))
    refbig.check_final(g, final)
    # the text the command lines write (scripts/pyHICCUPS:200-210, pyBHFDR:169-176), field by field: coordinates and the
    # count exact, the six / three statistics as printed by the reference within one unit of the third significant digit
    want = g.meta['lines'].splitlines()
    got = sorted(lines.splitlines(), key=lambda l: (int(l.split('\t')[1]), int(l.split('\t')[4])))
    assert len(got) == len(want)
    # (the rule of tests/test_gpu_cli.py: a '%.3g' field may differ from the reference's string only where the value sits on a
    # rounding edge of its third digit - by one unit of that digit -, and next to none may: at most 1 field in 500)
    fields = tipped = 0
    for a, b in zip(got, want):
        fa, fb = a.split('\t'), b.split('\t')
        assert fa[:10] == fb[:10]
        for u, v in zip(fa[10:], fb[10:]):
            fields += 1
            if u == v:
                continue
            tipped += 1
            fv = float(v)
            ulp3 = 10.0 ** (np.floor(np.log10(abs(fv))) - 2) if fv != 0.0 else 0.0
            assert abs(float(u) - fv) <= 1.01 * ulp3, (a, b)
    assert tipped <= max(1, fields // 500), (tipped, fields)


@pytest.mark.parametrize('name', refbig.names())
def test_reference_at_size(name, ctx):
    g = refbig.load(name)
    raw, weight = refbig.band(g)
    n, num = raw.shape
    IR, cband, biases = orc.prep_from_band(raw, weight, g.mw)
    np.testing.assert_allclose(IR[g.mw:], g['IR'], rtol=1e-13, atol=0)
    rawf = raw.astype(np.float32)
    kw = _kw(g)
    call = callers.hiccups_band if g.mode == 'hiccups' else callers.bhfdr_band
    # (i) single call: no bound known, the plan's own halo
    d1 = {}
    final = call(rawf, IR, biases, biases, chrom='T', weight=weight, ctx=ctx, detail=d1, **kw)
    R1 = d1['result']
    assert R1.stencil_kernel == 2
    assert R1.halo_w == g.params['maxww'] and not R1.redone
    _check_result(g, R1, final)
    # (ii) inside a batch, between two other chromosomes, IR / biases derived on the device, record bound / halo / survivor
    # bound inherited from (i)
    other, ow, _ = synthetic.synth_band(max(num + 40, 700), num, depth=g.meta['gen']['depth'], nloops=10, seed=91)
    items = [('a', other.astype(np.float32), ow), ('T', rawf, weight), ('b', other.astype(np.float32), ow)]
    sub = callers.hiccups_batch_submit if g.mode == 'hiccups' else callers.bhfdr_batch_submit
    pending = sub(items, ctx=ctx, **kw)
    Rs = pending._job.results()
    assert Rs[1].batch_bands == 3 and Rs[1].record_bound == R1.frozen_w
    fin2 = pending._finishers[1](Rs[1])
    _check_result(g, Rs[1], fin2)
    # (iii) the balanced band handed over as f64 (what the drop-in hiccups() / bhfdr() receive as cDiags)
    if n * num <= 3000000:
        d3 = {}
        fin3 = call(rawf, IR, biases, biases, chrom='T', balanced=cband, ctx=ctx, detail=d3, **kw)
        _check_result(g, d3['result'], fin3)


@pytest.mark.parametrize('name', refbig.names())
def test_reference_at_size_under_the_layout_of_the_own_frozen_width(name):
    """The same fixtures under option spec_halo = 2 (what the command lines and the drop-in functions run under), in a context that
    scored a deep and a shallow chromosome before: the chromosome ends up under the tile layout of its own frozen width - through
    a second pass where the inherited one did not match, lean tiles included on the wide band - and equals the reference."""
    g = refbig.load(name)
    raw, weight = refbig.band(g)
    n, num = raw.shape
    IR, cband, biases = orc.prep_from_band(raw, weight, g.mw)
    kw = _kw(g)
    call = callers.hiccups_band if g.mode == 'hiccups' else callers.bhfdr_band
    c = _lib.Context(0)
    c.set_option('spec_halo', 2)
    try:
        for depth, seed in ((150.0, 5), (8.0, 6)):
            other, ow, _ = synthetic.synth_band(max(num + 40, 900), num, depth=depth, nloops=10, seed=seed)
            oIR, _, ob = orc.prep_from_band(other, ow, g.mw)
            call(other.astype(np.float32), oIR, ob, ob, chrom='o', weight=ow, ctx=c, **kw)
        d = {}
        final = call(raw.astype(np.float32), IR, biases, biases, chrom='T', weight=weight, ctx=c, detail=d, **kw)
        R = d['result']
        ww = g.params['ww']
        mw = min(ww) if isinstance(ww, (list, tuple)) else ww
        assert R.halo_w == min(g.params['maxww'], max(R.frozen_w, mw, 4)), (R.halo_w, R.frozen_w)
        _check_result(g, R, final)
        if name.startswith('wide'):
            assert R.redone and R.lean_tiles > R.tiles // 2
    finally:
        c.close()


# ---------------------------------------------------------------------------------------------------------------------
# The wide band (num = 2011: ~20 column chunks per row block, 2 001-entry expected tables) against the oracle, in the
# default suite (the driver-run suite never compared a wide band with the oracle): (2,5) and (4,7), single call
# and inside a batch under an inherited bound.  The oracle takes 10-25 s of a host core per case.
WIDE = {
    'wide_p2w5': dict(n=5600, res=5000, maxapart=10000000, pw=[2], ww=[5], depth=40.0, nloops=60, seed=11),
    'wide_p4w7': dict(n=6100, res=5000, maxapart=10000000, pw=[4], ww=[7], depth=18.0, nloops=60, seed=12),
}


@pytest.mark.parametrize('name', sorted(WIDE))
def test_wide_band_vs_oracle(name, ctx):
    from test_gpu_fullsize import _check_against_oracle
    cfg = WIDE[name]
    n, res, maxww, sig = cfg['n'], cfg['res'], 10, 0.05
    pw, ww = cfg['pw'], cfg['ww']
    num = cfg['maxapart'] // res + maxww + 1
    assert num == 2011
    raw, weight, _ = synthetic.synth_band(n, num, depth=cfg['depth'], nloops=cfg['nloops'], seed=cfg['seed'])
    IR, cband, biases = orc.prep_from_band(raw, weight, min(ww))
    kw = dict(pw=pw, ww=ww, maxww=maxww, sig=sig, maxapart=cfg['maxapart'], res=res, min_local_reads=16,
              min_marginal_peaks=2, onlyanchor=False)
    det = {}
    want = orc.hiccups(raw, cband, biases, biases, IR, n, num, detail=det, **kw)
    rawf = raw.astype(np.float32)
    d1 = {}
    final = callers.hiccups_band(rawf, IR, biases, biases, chrom='1', weight=weight, ctx=ctx, detail=d1, **kw)
    R1 = d1['result']
    assert R1.stencil_kernel == 2 and R1.halo_w == maxww
    _check_against_oracle(R1, final, det, want, pw, ww, sig, min_sig=5, min_final=1)
    prm = _lib.make_params(_lib.MODE_HICCUPS, pw, ww, maxww, sig, cfg['maxapart'], res, 16, 0)
    other, ow, _ = synthetic.synth_band(2500, num, depth=cfg['depth'], nloops=20, seed=78)
    items = [dict(raw=other.astype(np.float32), weight=ow, num=num), dict(raw=rawf, weight=weight, num=num)]
    Rs = ctx.submit_batch_host(items, prm).results()
    assert Rs[1].record_bound == R1.frozen_w and not Rs[1].redone
    assert Rs[1].halo_w == max(Rs[1].record_bound, min(ww), 4)
    # a band this wide is mostly far field: most of its column chunks are lean by the library's own prediction, and go
    # through hpk_stencil_lean (no option forced here) - with few tiles handed back, if any
    assert Rs[1].lean_tiles > Rs[1].tiles // 2 and Rs[1].lean_redone <= Rs[1].lean_tiles // 20, (Rs[1].lean_tiles, Rs[1].lean_redone, Rs[1].tiles)
    fin2, _ = callers._finish_hiccups(Rs[1], n, '1', pw, ww, sig, 0.01, 1.75, 2, res, False, 2, False)
    _check_against_oracle(Rs[1], fin2, det, want, pw, ww, sig, min_sig=5, min_final=1)
