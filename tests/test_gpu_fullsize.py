"""BASELINE.json configs[3] and configs[4] at full size on the GPU: the wide-band code paths (num = 2011: ~20 column
chunks per row block, 2001-entry expected tables, 32-bit table offsets, the last-tile gap-row rule) that no fixture
reaches.  The reference cannot run at these sizes in a test (hours), so the checks are size-independent properties:

  * counting identities of the widening log (candidates, band pixels, executed prefix, resolve counts),
  * gap rows against an independent reduction of the band,
  * the production kernels' sums and resolving widths at ~25 000 sampled pixels (`hpk_probe_sums`) against the
    explicit-window kernel `hpk_brute` (no summed-area table, no tiles) - the sample covers the first and the last
    column chunk and both chromosome ends,
  * p <= q <= sig for every reported pixel.
"""
import numpy as np
import pytest

from hicpeaks_amd import _lib, band as hband, callers, synthetic
from oracle import hiccups_oracle as orc

pytestmark = pytest.mark.gpu

SIG, MIN_READS = 0.05, 16
CASES = {
    # configs[3] largest work item: hg38 chr1 at 5 kb, (p, w) = (4, 7), 10 Mb band, weights-only input
    'chr1_5kb_p4w7': dict(n=49792, res=5000, maxapart=10000000, pw=[4], ww=[7], maxww=10, depth=25.0, nloops=800, seed=5),
    # configs[4]: synthetic 1 kb deep Hi-C, 2 Mb band, 250 000 bins
    'deep_1kb_p2w5': dict(n=250000, res=1000, maxapart=2000000, pw=[2], ww=[5], maxww=10, depth=8.0, nloops=2000, seed=6),
}


@pytest.fixture(scope='module')
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def _sample(raw_d, n, num, mw, D, W, rng):
    """Candidate pixels (non-zero count, mw <= d <= D): random ones plus the corners of the band."""
    import torch
    dev = raw_d.device
    picks = []
    r = torch.from_numpy(rng.integers(0, n, 600000)).to(dev)
    k = torch.from_numpy(rng.integers(mw, D + 1, 600000)).to(dev)
    ok = (r + k < n) & (raw_d[r, k] != 0)
    picks.append(torch.stack([r[ok], k[ok]], 1)[:20000])

    def block(rsl, ksl, cap):
        sub = raw_d[rsl, ksl]
        rr, kk = torch.nonzero(sub, as_tuple=True)
        rr = rr + (rsl.start or 0)
        kk = kk + (ksl.start or 0)
        ok = rr + kk < n
        sel = torch.stack([rr[ok], kk[ok]], 1)
        if sel.shape[0] > cap:
            sel = sel[torch.from_numpy(rng.choice(sel.shape[0], cap, replace=False)).to(dev)]
        return sel
    picks.append(block(slice(0, W + 2), slice(mw, D + 1), 3000))                      # first rows: window clipped above
    picks.append(block(slice(n - D - 5, n), slice(mw, D + 1), 200000))                 # last columns: filtered below
    picks.append(block(slice(0, n), slice(mw, mw + 4), 2000))                          # first column chunk, d = mw ..
    picks.append(block(slice(0, n), slice(D - 3, D + 1), 2000))                        # last column chunk, d = .. D
    picks.append(block(slice(0, 300), slice(mw, D + 1), 1500))
    p = torch.cat(picks).cpu().numpy()
    rows, ks = p[:, 0], p[:, 1]
    cols = rows + ks
    keep = np.ones(rows.size, bool)
    far = rows >= n - D - 5
    keep[far] = (cols[far] >= n - W - 2) | (rng.random(int(far.sum())) < 0.002)     # right end: window clipped at the side
    rows, cols = rows[keep], cols[keep]
    _, first = np.unique(rows.astype(np.int64) * n + cols, return_index=True)
    return rows[first].astype(np.int32), cols[first].astype(np.int32)


@pytest.mark.parametrize('name', sorted(CASES))
def test_full_size_wide_band(name, ctx):
    import torch
    from hicpeaks_amd import bandgen
    cfg = CASES[name]
    n, res, W = cfg['n'], cfg['res'], cfg['maxww']
    mw, D = min(cfg['ww']), cfg['maxapart'] // res
    num = D + W + 1
    ld = (num + 63) // 64 * 64
    dev = torch.device('cuda', 0)
    raw_d, w_d, ir_d, b_d = bandgen.device_band(n, num, ld, mw, depth=cfg['depth'], nloops=cfg['nloops'], seed=cfg['seed'],
                                                device=dev)
    torch.cuda.synchronize()
    prm = _lib.make_params(_lib.MODE_HICCUPS, cfg['pw'], cfg['ww'], W, SIG, cfg['maxapart'], res, MIN_READS, 0)
    bd = ctx._band(n, num, ld, raw_d.data_ptr(), None, w_d.data_ptr(), ir_d.data_ptr(), b_d.data_ptr(), b_d.data_ptr(), True)
    R = ctx.score(bd, prm, n)

    # ---- counting identities
    kk = torch.arange(ld, device=dev)
    rr = torch.arange(n, device=dev)
    inband = (kk[None, :] >= mw) & (kk[None, :] <= D) & ((rr[:, None] + kk[None, :]) < n)
    assert R.ncand == int(((raw_d != 0) & inband).sum().item())
    assert R.band_px == hband.band_pixels(n, num, mw, D)
    TR, TC = 64 - 2 * W - 1, 160 - 2 * W - 1            # table of 64 x 160 cells less the halo (hpk_kernels.h)
    assert R.tiles == -(-n // TR) * -(-(TR + D - mw) // TC) and -(-(TR + D - mw) // TC) >= 14     # ~15 column chunks
    ex = [e for _, _, _, e in R.steps]
    assert ex == sorted(ex, reverse=True) and ex[0]
    assert all(wi <= R.frozen_w for _, wi, _, e in R.steps if e)
    assert sum(c for _, _, c, e in R.steps if e) <= R.ncand
    assert [wi for _, wi, _, _ in R.steps] == list(range(max(cfg['ww']), W + 1))

    # ---- gap rows: rows of the balanced upper band (diagonals mw .. num-1) that sum to 0 (callers.py:238)
    wc = torch.where(rr[:, None] + kk[None, :] < n, w_d[(rr[:, None] + kk[None, :]).clamp(max=n - 1)], torch.zeros((), dtype=w_d.dtype, device=dev))
    live = torch.zeros(n, dtype=torch.bool, device=dev)
    for r0 in range(0, n, 16384):                        # in slabs: the f64 product of the whole band would be 4 GB
        sl = slice(r0, min(n, r0 + 16384))
        b = (raw_d[sl].to(torch.float64) * w_d[sl, None]) * wc[sl]
        b = torch.nan_to_num(b, nan=0.0)
        b[:, :mw] = 0
        b[:, num:] = 0
        live[sl] = (b != 0).any(dim=1)
    np.testing.assert_array_equal(R.gap, ~live.cpu().numpy())
    del wc, inband

    # ---- sampled pixels: production records vs explicit windows
    rng = np.random.default_rng(cfg['seed'])
    rows, cols = _sample(raw_d, n, num, mw, D, W, rng)
    assert rows.size > 15000
    assert (rows < W).sum() > 200 and (cols >= n - W).sum() > 20                       # both chromosome ends
    d = cols - rows
    assert (d < mw + 4).sum() > 500 and (d > D - 4).sum() > 50                          # first and last column chunk
    pr = ctx.probe_sums(bd, prm, rows, cols, 1)[:, 0, :]
    width = pr[:, 4].astype(np.int64)
    assert np.all(width >= 0)                                                           # every sampled pixel is a candidate
    steps = [(pi, wi) for pi, wi, _, _ in R.steps]
    reads_at = {}
    for si, (pi, wi) in enumerate(steps):
        bf = ctx.bruteforce_band(bd, prm, si, rows, cols)
        reads_at[wi] = bf[:, 4]
        sel = width == wi
        if sel.any():
            np.testing.assert_allclose(pr[sel, :4], bf[sel, :4], rtol=1e-11, atol=0)
            assert np.array_equal(pr[sel, :4] == 0, bf[sel, :4] == 0)
            assert np.all(bf[sel, 4] >= MIN_READS)
            if si > 0:                                                                  # first sufficient width
                assert np.all(reads_at[steps[si - 1][1]][sel] < MIN_READS)
    never = width == 0
    assert np.all(reads_at[steps[-1][1]][never] < MIN_READS)
    assert len(set(width.tolist())) >= 3                                                # the sample exercises the widening
    # a zero-count pixel is not a candidate
    zr = torch.nonzero(raw_d[1000:1064, mw:D + 1] == 0)[:50].cpu().numpy()
    pz = ctx.probe_sums(bd, prm, zr[:, 0] + 1000, zr[:, 0] + 1000 + zr[:, 1] + mw, 1)
    assert np.all(pz[:, 0, 4] == -1)

    # ---- reported pixels
    for s in R.sets:
        assert s['x'].size > 0
        assert np.all(s['p'] <= s['q'] + 1e-18) and np.all(s['q'] <= SIG)
        assert np.all((s['y'] - s['x'] >= max(cfg['ww'])) & (s['y'] - s['x'] <= D))
        o = raw_d[torch.from_numpy(s['x']).to(dev), torch.from_numpy(s['y'] - s['x']).to(dev)].cpu().numpy()
        np.testing.assert_array_equal(s['O'], o)


# ---------------------------------------------------------------------------------------------------------------------
# The headline configurations against the ORACLE at full size (above n = 640 the production kernels had only
# ever met the builder's own explicit-window kernel).  The oracle - the numpy restatement of hicpeaks/callers.py:44-362
# that the fixtures pin to the reference - takes 8-30 s per case on a host core; the GPU side runs the production path
# (no dense outputs: packed records, record bound, scoring kernel, device-side cut), once as a single call and once as a
# batch of three chromosomes.
FULL = {
    # BASELINE configs[1]: hg38 chr1 @10 kb, (p, w) = (2, 5), 5 Mb band: n = 24 896, num = 511
    'chr1_10kb_p2w5': dict(n=24896, res=10000, maxapart=5000000, pw=[2], ww=[5], depth=60.0, nloops=400, seed=0),
    # BASELINE configs[2]'s plan on its largest chromosome: union of (1,3) / (2,5) / (4,7): 18 widening steps, 3 slots
    'chr1_10kb_union': dict(n=24896, res=10000, maxapart=5000000, pw=[1, 2, 4], ww=[3, 5, 7], depth=60.0, nloops=400, seed=1),
}


def _arrays(table):
    keys = sorted(table)
    return np.array(keys, dtype=np.int64).reshape(-1, 2), np.array([[float(v) for v in table[k]] for k in keys])


def _check_against_oracle(R, final, det, want, pw, ww, sig, min_sig=100, min_final=10):
    loc = det['loc']
    # widening log: candidates, resolve counts of the executed steps, the width the widening froze at
    assert R.ncand == loc['vx'].size
    got_steps = [(a, b, c) for a, b, c, ex in R.steps if ex]
    assert got_steps == [tuple(int(v) for v in s) for s in loc['steps']]
    assert R.frozen_w == loc['frozen_w']
    # every (pair, filter) set: the pixels with q <= sig, coordinates exact, E / p / q within the parity tolerances
    assert len(R.sets) == 2 * len(pw)
    nsig = 0
    for t, (s, o) in enumerate(zip(R.sets, det['sets'])):
        assert (s['pair'], s['fl']) == (t // 2, o['fl']) and (pw[t // 2], ww[t // 2]) == (o['pi'], o['wi'])
        assert s['nvalid'] == o['vx'].size
        q, vx, vy = o['q'], o['vx'], o['vy']
        firm = np.abs(q - sig) > 1e-8                       # (pixels sitting on the threshold may fall either way)
        assert int((~firm).sum()) <= 3, int((~firm).sum())  # (next to none of them)
        sel = q <= sig
        wantpx = set(zip(vx[sel & firm].tolist(), vy[sel & firm].tolist()))
        maybe = set(zip(vx[~firm].tolist(), vy[~firm].tolist()))
        got = set(zip(s['x'].tolist(), s['y'].tolist()))
        assert wantpx <= got and got <= (wantpx | maybe), (len(wantpx - got), len(got - wantpx))
        order = np.lexsort((vy, vx))
        key = vx[order] * (1 << 32) + vy[order]
        ii = order[np.searchsorted(key, s['x'] * (1 << 32) + s['y'])]
        np.testing.assert_array_equal(vx[ii], s['x'])
        np.testing.assert_array_equal(vy[ii], s['y'])
        np.testing.assert_array_equal(s['O'], o['O'][ii])
        np.testing.assert_allclose(s['E'], o['E'][ii], rtol=1e-9, atol=0)
        np.testing.assert_allclose(s['p'], o['p'][ii], rtol=0, atol=1e-12)
        np.testing.assert_allclose(s['q'], o['q'][ii], rtol=0, atol=1e-9)
        # family sizes of the Benjamini-Hochberg step = the oracle's chunk memberships
        tests = np.bincount(o['chunk'], minlength=s['chunk_tests'].size)
        np.testing.assert_array_equal(s['chunk_tests'][1:], tests[1:s['chunk_tests'].size])
        nsig += s['x'].size
    assert nsig > min_sig
    # gap rows and the final table (after gap filter, donut / lower-left combination, clustering)
    np.testing.assert_array_equal(R.gap, np.isin(np.arange(R.gap.size), sorted(det['gaps'])))
    k, v = _arrays(final)
    kw, vw = _arrays(want)
    np.testing.assert_array_equal(k, kw)
    np.testing.assert_allclose(v, vw, rtol=1e-9, atol=1e-9)
    assert len(want) >= min_final


# BASELINE configs[3]'s largest chromosome, chr1 @5 kb, (4,7), 10 Mb band (n = 49 792, num = 2011): the oracle needs ~10 GB
# of host memory and 1-4 minutes of a core - run with HPK_SLOW=1 (scripts/measure/gpu_slow_tests.sh, profiles/r03_slow_tests.txt)
SLOW = {'chr1_5kb_p4w7': dict(n=49792, res=5000, maxapart=10000000, pw=[4], ww=[7], depth=25.0, nloops=800, seed=0)}


@pytest.mark.parametrize('name', sorted(FULL) + [pytest.param(k, marks=pytest.mark.slow) for k in sorted(SLOW)])
def test_full_size_chr1_vs_oracle(name, ctx):
    import os
    if name in SLOW and not os.environ.get('HPK_SLOW'):
        pytest.skip('slow: set HPK_SLOW=1')
    cfg = FULL.get(name) or SLOW[name]
    n, res, maxww, sig = cfg['n'], cfg['res'], 10, 0.05
    pw, ww = cfg['pw'], cfg['ww']
    num = cfg['maxapart'] // res + maxww + 1
    raw, weight, _ = synthetic.synth_band(n, num, depth=cfg['depth'], nloops=cfg['nloops'], seed=cfg['seed'])
    IR, cband, biases = orc.prep_from_band(raw, weight, min(ww))
    kw = dict(pw=pw, ww=ww, maxww=maxww, sig=sig, maxapart=cfg['maxapart'], res=res, min_local_reads=16,
              min_marginal_peaks=2, onlyanchor=False)
    det = {}
    want = orc.hiccups(raw, cband, biases, biases, IR, n, num, detail=det, **kw)
    rawf = raw.astype(np.float32)
    # (i) one call, IR / biases given, weights on the chip
    d1 = {}
    final = callers.hiccups_band(rawf, IR, biases, biases, chrom='1', weight=weight, ctx=ctx, detail=d1, **kw)
    TR, TC = 64 - 2 * maxww - 1, 160 - 2 * maxww - 1    # table of 64 x 160 cells less the halo (hpk_kernels.h)
    assert d1['result'].stencil_kernel == 2 and d1['result'].tiles == -(-n // TR) * -(-(TR + cfg['maxapart'] // res - min(ww)) // TC)
    _check_against_oracle(d1['result'], final, det, want, pw, ww, sig)
    # (ii) the same chromosome between two others in one batch, IR / biases derived on the device
    prm = _lib.make_params(_lib.MODE_HICCUPS, pw, ww, maxww, sig, cfg['maxapart'], res, 16, 0)
    other, ow, _ = synthetic.synth_band(9000, num, depth=cfg['depth'], nloops=100, seed=77)
    items = [dict(raw=other.astype(np.float32), weight=ow, num=num), dict(raw=rawf, weight=weight, num=num),
             dict(raw=other.astype(np.float32), weight=ow, num=num)]
    Rs = ctx.submit_batch_host(items, prm).results()
    assert Rs[1].batch_bands == 3
    # (the batch took its record bound from call (i), and with it the halo of its tiles: this comparison runs the
    # bound's tile geometry - 64 x (127 - 2 halo) output tiles, search stopped at the bound - against the oracle)
    assert Rs[1].record_bound == d1['result'].frozen_w and not Rs[1].redone
    assert d1['result'].halo_w == maxww and Rs[1].halo_w == max(Rs[1].record_bound, min(ww), 4) < maxww
    fin2, _ = callers._finish_hiccups(Rs[1], n, '1', pw, ww, sig, 0.01, 1.75, 2, res, False, 2, False)
    _check_against_oracle(Rs[1], fin2, det, want, pw, ww, sig)
    # (the chromosome before and after it are the same band: same result)
    assert Rs[0].ncand == Rs[2].ncand and [s['x'].size for s in Rs[0].sets] == [s['x'].size for s in Rs[2].sets]


def test_full_size_chr1_bhfdr_vs_oracle(ctx):
    """The sibling path at full size: pyBHFDR's defaults ((2,5), 2 Mb band) on chr1 @10 kb against the oracle's bhfdr() -
    one Benjamini-Hochberg family of ~2.5 million tests per chromosome, whose cut the device brackets with 64 fine p-value
    bins - as a first call (no bounds), again (record bound, halo and survivor bound taken over) and in a batch."""
    n, res, maxww, maxapart, sig = 24896, 10000, 10, 2000000, 0.05
    num = maxapart // res + maxww + 1
    raw, weight, _ = synthetic.synth_band(n, num, depth=60.0, nloops=400, seed=2)
    IR, cband, biases = orc.prep_from_band(raw, weight, 5)
    kw = dict(pw=2, ww=5, sig=sig, maxww=maxww, maxapart=maxapart, res=res, min_marginal_peaks=2, onlyanchor=False)
    want = orc.bhfdr(raw, cband, biases, biases, IR, n, num, **kw)
    assert len(want) >= 10
    kwant, vwant = _arrays(want)
    rawf = raw.astype(np.float32)
    c = _lib.Context(0)
    try:
        for rnd in range(2):
            d = {}
            got = callers.bhfdr_band(rawf, IR, biases, biases, chrom='1', weight=weight, ctx=c, detail=d, **kw)
            k, v = _arrays(got)
            np.testing.assert_array_equal(k, kwant)
            np.testing.assert_allclose(v, vwant, rtol=1e-9, atol=1e-9)
            R = d['result']
            assert R.halo_w == (maxww if rnd == 0 else max(R.record_bound, 5)) and not R.redone
            # the cut comes within the bins' resolution of the pixels reported: a few hundred records too many, not tens of thousands
            assert R.nsurv_cut < 2 * R.nsig + 1000, (R.nsurv_cut, R.nsig)
        prm = _lib.make_params(_lib.MODE_BHFDR, [2], [5], maxww, sig, maxapart, res, 16, 0)
        Rs = c.submit_batch_host([dict(raw=rawf, weight=weight, num=num)] * 3, prm).results()
        for R in Rs:
            fin = callers._bhfdr_finisher(n, '1', 5, res, 2, False, None)(R)
            k, v = _arrays(fin)
            np.testing.assert_array_equal(k, kwant)
            np.testing.assert_allclose(v, vwant, rtol=1e-9, atol=1e-9)
    finally:
        c.close()
