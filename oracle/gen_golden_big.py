#!/opt/conda/bin/python3.9
"""Reference-pinned fixtures above the small-chromosome sizes of gen_golden.py ("close the parity ladder").

TEST INFRASTRUCTURE - runs only in the build container, where /root/reference exists:

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 oracle/gen_golden_big.py [case ...]

Runs the *real* reference (hicpeaks 0.3.9, imported unmodified from /root/reference through gen_golden.py's
observers) on whole synthetic chromosomes - chr21 @10 kb (n = 4 671), BASELINE configs[0]'s own shape (n = 1 869,
num = 411, (1,3), 10 Mb @25 kb), chr1 @10 kb at full size (n = 24 896, num = 511: BASELINE configs[1]) and a
2 011-diagonal band ((4,7) @5 kb) - and stores **outputs only**: the band is regenerated from its seed by
`hicpeaks_amd/synthetic.synth_band(**meta['gen'])` wherever the fixture is used.

Per fixture (tests/golden/ref_<case>.npz): parameters and versions; the widening log (per executed step: pi, wi,
resolved); the candidate count; IR (the reference's prep); per (pair, filter) set t:
  s<t>_nvalid, s<t>_Emax, s<t>_sumE, s<t>_sump   size and checksums of the E > 0 population (callers.py:250-253)
  s<t>_hx, s<t>_hy, s<t>_hxy                     integer checksums of its coordinates (sums mod 2^63)
  s<t>_chunk_tests / _chunk_below                family size and #p <= sig per lambda chunk (callers.py:263-275)
  s<t>_kx, _ky, _kE, _kO, _kp, _kq               every pixel with q <= 2 sig (the survivors and their neighbours)
  s<t>_rx, _ry, _rE, _rO, _rp, _rq, _rchunk      a seeded random sample of up to 20 000 of the population
and the pre-clustering table, the final table and the text lines.  (`bhfdr`: one set, `_kreject` = statsmodels'
step-up mask.)
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import gen_golden as gg  # noqa: E402  (imports the reference and installs nothing until asked)

SAMPLE = 20000

CASES = {
    # name: (mode, gen, params)
    'cfg1': ('hiccups', dict(n=1869, num=411, depth=150.0, nloops=30, seed=0),
             gg.P(pw=[1], ww=[3], maxapart=10000000, res=25000)),
    'chr21_10kb_p2w5': ('hiccups', dict(n=4671, num=211, depth=60.0, nloops=60, seed=0),
                        gg.P(pw=[2], ww=[5], maxapart=2000000)),
    'chr21_10kb_union': ('hiccups', dict(n=4671, num=511, depth=60.0, nloops=60, seed=0),
                         gg.P(pw=[1, 2, 4], ww=[3, 5, 7], maxapart=5000000)),
    'chr21_10kb_bhfdr': ('bhfdr', dict(n=4671, num=211, depth=60.0, nloops=60, seed=0),
                         gg.P(pw=2, ww=5, maxapart=2000000, min_marginal_peaks=3)),
    'chr1_10kb_p2w5': ('hiccups', dict(n=24896, num=511, depth=60.0, nloops=300, seed=0),
                       gg.P(pw=[2], ww=[5], maxapart=5000000)),
    'wide_5kb_p4w7': ('hiccups', dict(n=6000, num=2011, depth=25.0, nloops=80, seed=0),
                      gg.P(pw=[4], ww=[7], maxapart=10000000, res=5000)),
    # round 5: inputs with structure (synthetic.structure_fields: TAD blocks, a compartment checkerboard, dense far-field
    # patches) - what the speculations of the HIP path (record bounds, depth classes, lean column chunks) are NOT tuned on
    'struct_chr21_10kb_p2w5': ('hiccups', dict(n=4671, num=211, depth=40.0, nloops=60, seed=5, structure={}),
                               gg.P(pw=[2], ww=[5], maxapart=2000000)),
    'struct_chr21_10kb_union': ('hiccups', dict(n=4671, num=511, depth=40.0, nloops=60, seed=6, structure={}),
                                gg.P(pw=[1, 2, 4], ww=[3, 5, 7], maxapart=5000000)),
    'struct_chr21_10kb_bhfdr': ('bhfdr', dict(n=4671, num=211, depth=40.0, nloops=60, seed=7, structure={}),
                                gg.P(pw=2, ww=5, maxapart=2000000, min_marginal_peaks=3)),
}

M63 = (1 << 63) - 1


def isum(a):
    return int(np.sum(a.astype(object)) & M63) if a.size else 0


def run(name):
    mode, gen, params = CASES[name]
    t0 = time.perf_counter()
    raw, weight, loops = gg.synthetic.synth_band(**gen)
    res, maxapart, maxww = params['res'], params['maxapart'], params['maxww']
    pw, ww = params['pw'], params['ww']
    mw = min(ww) if mode == 'hiccups' else ww
    assert gen['num'] == maxapart // res + maxww + 1
    H, cH = gg.cooler_like(raw, weight)
    M, cM, biases, IR, chromLen, Diags, cDiags, num = gg.worker_prep(H, cH, weight, mw, maxapart, res, maxww)
    t1 = time.perf_counter()
    tap = gg.Tap()
    tap.capture_gets = False
    saved, h = gg.install(tap)
    try:
        if mode == 'hiccups':
            table = gg.ref.hiccups(M, cM, biases, biases, IR, chromLen, Diags, cDiags, num, 'T', pw=pw, ww=ww, maxww=maxww,
                                   sig=params['sig'], sumq=params['sumq'], double_fold=params['double_fold'],
                                   single_fold=params['single_fold'], maxapart=maxapart, res=res, use_raw=params['use_raw'],
                                   min_marginal_peaks=params['min_marginal_peaks'], onlyanchor=params['onlyanchor'],
                                   min_local_reads=params['min_local_reads'])
        else:
            table = gg.ref.bhfdr(M, cM, biases, biases, IR, chromLen, Diags, cDiags, num, 'T', pw=pw, ww=ww, sig=params['sig'],
                                 maxww=maxww, maxapart=maxapart, res=res, min_marginal_peaks=params['min_marginal_peaks'],
                                 onlyanchor=params['onlyanchor'])
    finally:
        gg.uninstall(saved, h)
    t2 = time.perf_counter()

    out = dict(IR=np.array([IR[i] for i in sorted(IR)]))
    steps = []
    for msg in tap.log:
        m = gg.STEP_RE.search(msg)
        if m:
            steps.append((int(m.group(1)), int(m.group(2)), int(m.group(3))))
        elif mode == 'bhfdr':
            m = gg.BH_STEP_RE.search(msg)
            if m:
                steps.append((pw, ww + len(steps), int(m.group(1))))
    out['steps'] = np.array(steps, dtype=np.int64).reshape(-1, 3)
    import re
    ncand = [int(re.search(r'Observed Contact Number: (\d+)', m).group(1)) for m in tap.log if 'Observed Contact Number' in m]
    D = maxapart // res
    meta = dict(name='ref_' + name, mode=mode, gen=gen, params=params, num=int(num), chromLen=int(chromLen),
                ncand=ncand[0] if ncand else None, band_px=int(sum(max(chromLen - d, 0) for d in range(mw, D + 1))),
                seconds=dict(prep=round(t1 - t0, 1), call=round(t2 - t1, 1)),
                versions=dict(python=sys.version.split()[0], numpy=np.__version__, scipy=__import__('scipy').__version__,
                              statsmodels=__import__('statsmodels').__version__, sklearn=__import__('sklearn').__version__,
                              hicpeaks=__import__('hicpeaks').__version__))
    sig = params['sig']
    rng = np.random.default_rng(12345)
    nsets = len(tap.lil)
    bi = pi_ = 0
    IRarr = np.zeros(num)
    for i in IR:
        IRarr[i] = IR[i]
    for t in range(nsets):
        x, y, ratio = tap.lil[t]
        Eall = (IRarr[y - x] * ratio) * biases[x] * biases[y]
        keep = Eall > 0
        order = np.lexsort((y[keep], x[keep]))
        vx, vy = x[keep][order].astype(np.int64), y[keep][order].astype(np.int64)
        O = raw[vx, vy - vx].astype(np.float64)
        if mode == 'hiccups':
            E = tap.E[t]
            assert np.array_equal(Eall[keep][order], E), 'E reconstruction mismatch'
            p = np.ones(E.size)
            q = np.ones(E.size)
            chunk_id = np.zeros(E.size, dtype=np.int32)
            chunks = saved['lambdachunk'](E)
            tests = np.zeros(len(chunks) + 1, np.int64)
            below = np.zeros(len(chunks) + 1, np.int64)
            for ci, (lv, rv, idx) in enumerate(chunks):
                if idx.size:
                    mu, Oc, cdf = tap.pois[pi_]
                    pi_ += 1
                    pp, qq, _ = tap.bh[bi]
                    bi += 1
                    assert float(mu) == float(rv) and np.array_equal(Oc, O[idx])
                    p[idx] = pp
                    q[idx] = qq
                    chunk_id[idx] = ci + 1
                    tests[ci + 1] = idx.size
                    below[ci + 1] = int((pp <= sig).sum())
            rej = None
        else:
            mu, Oc, cdf = tap.pois[0]
            assert np.array_equal(Eall[keep][order], mu)
            E = mu
            p, q, rej = tap.bh[0]
            chunk_id = np.ones(E.size, dtype=np.int32)
            tests = np.array([0, E.size], np.int64)
            below = np.array([0, int((p <= sig).sum())], np.int64)
        pre = 's%d_' % t
        out[pre + 'nvalid'] = np.int64(E.size)
        out[pre + 'Emax'] = np.float64(E.max() if E.size else 0.0)
        out[pre + 'sumE'] = np.float64(E.sum())
        out[pre + 'sump'] = np.float64(p.sum())
        out[pre + 'hx'] = np.int64(isum(vx))
        out[pre + 'hy'] = np.int64(isum(vy))
        out[pre + 'hxy'] = np.int64(isum(vx * vy))
        out[pre + 'chunk_tests'] = tests
        out[pre + 'chunk_below'] = below
        near = q <= 2 * sig
        if rej is not None:
            near |= rej
            out[pre + 'kreject'] = rej[near]
        for key, arr in (('x', vx.astype(np.int32)), ('y', vy.astype(np.int32)), ('E', E), ('O', O), ('p', p), ('q', q)):
            out[pre + 'k' + key] = arr[near]
        pick = np.sort(rng.choice(E.size, size=min(SAMPLE, E.size), replace=False)) if E.size else np.zeros(0, np.int64)
        for key, arr in (('x', vx.astype(np.int32)), ('y', vy.astype(np.int32)), ('E', E), ('O', O), ('p', p), ('q', q),
                         ('chunk', chunk_id)):
            out[pre + 'r' + key] = arr[pick]
    meta['nsets'] = nsets
    if tap.pre is not None:
        Dn, L = tap.pre
        k, v = gg.table_to_array(Dn)
        out['pre_keys'] = k
        out['pre_donut'] = v
        if L is not None:
            k2, v2 = gg.table_to_array(L)
            assert np.array_equal(k, k2)
            out['pre_ll'] = v2
    k, v = gg.table_to_array(table)
    out['final_keys'] = k
    out['final_vals'] = v
    meta['lines'] = gg.hiccups_lines('T', table, res) if mode == 'hiccups' else gg.bhfdr_lines('T', table, res)
    meta['nfinal'] = len(table)
    path = os.path.join(REPO, 'tests', 'golden', 'ref_' + name + '.npz')
    np.savez_compressed(path, meta=json.dumps(meta), **out)
    print('%-20s n=%d num=%d cand=%s steps=%s final=%d  prep %.1f s call %.1f s -> %.0f band px/s  (%d KB)' % (
        name, chromLen, num, meta['ncand'], [tuple(s) for s in steps], len(table), t1 - t0, t2 - t1,
        meta['band_px'] * (len(pw) if mode == 'hiccups' else 1) / (t2 - t1), os.path.getsize(path) // 1024), flush=True)


if __name__ == '__main__':
    for nm in (sys.argv[1:] or ['cfg1', 'chr21_10kb_p2w5', 'chr21_10kb_bhfdr', 'chr21_10kb_union']):
        run(nm)
