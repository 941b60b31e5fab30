#!/usr/bin/env python
"""Randomised parity run: HIP path (through the C ABI) against the numpy oracle on random small chromosomes and random
parameter sets (maxww 3..20, one to three pairs in any order, thresholds, band widths, NaN runs).  Test infrastructure.
usage: gpu_fuzz.py [ncases] [first_seed]"""
import os, sys, time, traceback
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hicpeaks_amd import _lib, callers, synthetic
from oracle import hiccups_oracle as orc


def table_arrays(t):
    k = sorted(t)
    return np.array(k, dtype=np.int64).reshape(-1, 2), np.array([[float(v) for v in t[x]] for x in k])


def one_case(seed, ctx, cpu=None):
    rng = np.random.default_rng(seed)
    maxww = int(rng.integers(3, 21))
    if os.environ.get('HPK_FUZZ_NARROW'):       # bands of 8-14 diagonals: a band row is shorter than the stencil's wide loads
        maxww = 3 + maxww % 2
    npairs = int(rng.integers(1, 4))
    ww = sorted(set(int(v) for v in rng.integers(2, maxww + 3, npairs)))     # a pair may be wider than maxww
    pw = [int(rng.integers(0, max(1, w))) for w in ww]
    if rng.random() < 0.25:
        order = rng.permutation(len(ww))
        pw, ww = [pw[i] for i in order], [ww[i] for i in order]
    res = 10000
    n = int(rng.integers(150, 1400)) if rng.random() < 0.8 else int(rng.integers(25, 150))     # some shorter than the band
    D = int(rng.integers(max(ww) + 2, 160))
    if os.environ.get('HPK_FUZZ_NARROW'):
        D = int(rng.integers(max(ww) + 2, max(ww) + 4))
    if os.environ.get('HPK_FUZZ_BIG'):          # many tiles: several row blocks per XCD chunk, 3-5 column chunks
        n, D = int(rng.integers(2000, 5000)), int(rng.integers(200, 520))
    if os.environ.get('HPK_FUZZ_WIDE'):         # bands as wide as the 5 kb / 1 kb configurations: 12-20 column chunks
        D = int(rng.integers(1200, 2001))       # (the oracle needs a minute or two per case at this width)
        n = D + maxww + 1 + int(rng.integers(60, 500))
        if len(ww) > 1:
            pw, ww = pw[:1], ww[:1]
    maxapart = D * res
    num = D + maxww + 1
    depth = float(rng.choice([2.0, 8.0, 25.0, 60.0, 400.0, 5000.0]))
    min_reads = int(rng.choice([1, 8, 16, 25, 200, 1000]))
    sig = float(rng.choice([0.01, 0.05, 0.1, 0.3]))
    # a quarter of the cases (all of them under HPK_FUZZ_STRUCT): TAD blocks, a compartment checkerboard and dense far-field patches
    # on top of the distance decay (synthetic.structure_fields)
    struct = bool(os.environ.get('HPK_FUZZ_STRUCT')) or seed % 4 == 3
    raw, weight, _ = synthetic.synth_band(n, num, depth=depth, nloops=int(rng.integers(0, 25)), seed=seed,
                                          structure={} if struct else None)
    if raw.min() < 0 or raw.max() >= (1 << 24):      # outside the domain (counts, exact in f32): a generator artefact, not a case
        return 'invalid-input', dict(seed=seed), 'counts outside [0, 2^24)'
    mode = 'bhfdr' if rng.random() < 0.2 else 'hiccups'
    inp = str(rng.choice(['weight', 'balanced', 'derive']))       # weights + IR | f64 balanced band + IR | weights only
    desc = dict(seed=seed, mode=mode, inp=inp, n=n, D=D, maxww=maxww, pw=pw, ww=ww, depth=depth, min_reads=min_reads, sig=sig, struct=struct)
    mw = min(ww) if mode == 'hiccups' else ww[0]
    try:
        IR, cband, biases = orc.prep_from_band(raw, weight, mw)
    except ValueError as e:          # chromosome shorter than the band: the reference's own prep fails (pyHICCUPS:148)
        return 'reference-prep-raises', desc, str(e)[:60]
    rawf = raw.astype(np.float32)
    want = werr = None
    try:
        if mode == 'hiccups':
            want = orc.hiccups(raw, cband, biases, biases, IR, n, num, pw=pw, ww=ww, maxww=maxww, sig=sig, maxapart=maxapart,
                               res=res, min_local_reads=min_reads, min_marginal_peaks=2, onlyanchor=False)
        else:
            want = orc.bhfdr(raw, cband, biases, biases, IR, n, num, pw=pw[0], ww=ww[0], sig=sig, maxww=maxww,
                             maxapart=maxapart, res=res, min_marginal_peaks=2, onlyanchor=False)
    except (ValueError, ZeroDivisionError, IndexError) as e:
        werr = e
    got = gerr = None
    detail = dict(dense=True)
    kw = dict(balanced=cband) if inp == 'balanced' else dict(weight=weight)
    gIR, gb = (None, None) if inp == 'derive' else (IR, biases)
    try:
        if mode == 'hiccups':
            got = callers.hiccups_band(rawf, gIR, gb, gb, chrom='T', pw=pw, ww=ww, maxww=maxww, sig=sig,
                                       maxapart=maxapart, res=res, min_local_reads=min_reads, min_marginal_peaks=2,
                                       onlyanchor=False, ctx=ctx, detail=detail, **kw)
        else:
            got = callers.bhfdr_band(rawf, gIR, gb, gb, chrom='T', pw=pw[0], ww=ww[0], sig=sig,
                                     maxww=maxww, maxapart=maxapart, res=res, min_marginal_peaks=2, onlyanchor=False,
                                     ctx=ctx, detail=detail, **kw)
    except (ValueError, ZeroDivisionError, _lib.HpkError) as e:
        gerr = e
    # back-end #0 (hpk_create(-1): the path on host threads) on the same case: the oracle's table, or its exception
    if cpu is not None:
        cgot = cerr = None
        try:
            if mode == 'hiccups':
                cgot = callers.hiccups_band(rawf, gIR, gb, gb, chrom='T', pw=pw, ww=ww, maxww=maxww, sig=sig,
                                            maxapart=maxapart, res=res, min_local_reads=min_reads, min_marginal_peaks=2,
                                            onlyanchor=False, ctx=cpu, **kw)
            else:
                cgot = callers.bhfdr_band(rawf, gIR, gb, gb, chrom='T', pw=pw[0], ww=ww[0], sig=sig,
                                          maxww=maxww, maxapart=maxapart, res=res, min_marginal_peaks=2, onlyanchor=False,
                                          ctx=cpu, **kw)
        except (ValueError, ZeroDivisionError, _lib.HpkError) as e:
            cerr = e
        if isinstance(cerr, _lib.HpkError) and not isinstance(cerr, _lib.EmptyStepError):
            if not isinstance(gerr, _lib.HpkError) or isinstance(gerr, _lib.EmptyStepError):
                return 'MISMATCH-cpu-unsupported', desc, str(cerr)[:100]
        elif (werr is None) != (cerr is None):
            return 'MISMATCH-cpu-exception', desc, 'oracle %r vs back-end #0 %r' % (werr, cerr)
        elif cerr is None:
            kc, vc = table_arrays(cgot)
            kwant, vwant = table_arrays(want)
            if kc.shape != kwant.shape or not np.array_equal(kc, kwant):
                return 'MISMATCH-cpu-keys', desc, '%d vs %d pixels' % (len(kc), len(kwant))
            if kc.size and not np.allclose(vc, vwant, rtol=1e-9, atol=1e-9):
                return 'MISMATCH-cpu-values', desc, 'max abs diff %g' % np.abs(vc - vwant).max()
    if werr is not None or gerr is not None:
        unsupported = isinstance(gerr, _lib.HpkError) and not isinstance(gerr, _lib.EmptyStepError)
        if unsupported:
            return 'unsupported', desc, str(gerr)[:100]
        if (werr is None) != (gerr is None):
            return 'MISMATCH-exception', desc, 'oracle %r vs hip %r' % (werr, gerr)
        return 'both-raise', desc, ''
    k, v = table_arrays(got)
    kw, vw = table_arrays(want)
    if k.shape != kw.shape or not np.array_equal(k, kw):
        return 'MISMATCH-keys', desc, '%d vs %d pixels' % (len(k), len(kw))
    if k.size and not np.allclose(v, vw, rtol=1e-9, atol=1e-9):
        return 'MISMATCH-values', desc, 'max abs diff %g' % np.abs(v - vw).max()
    if mode == 'hiccups':          # resolving widths of every candidate
        loc = orc.hiccups_local_sums(raw, cband, IR, n, num, pw, ww, maxww, maxapart, res, min_reads)
        R = detail['result']
        vx, vy = loc['vx'], loc['vy']
        for slot, pi in enumerate(R.slot_pi):
            w = R.dense_w[slot][vx, vy - vx].astype(np.int64)
            w = np.where(w > R.frozen_w, 0, w)
            if not np.array_equal(w, loc['wres'][pi]):
                bad = np.nonzero(w != loc['wres'][pi])[0][:4]
                where = ', '.join('(r %d, c %d): %d, oracle %d' % (vx[i], vy[i], w[i], loc['wres'][pi][i]) for i in bad)
                return 'MISMATCH-widths', desc, 'slot %d: %d differ; %s' % (slot, int((w != loc['wres'][pi]).sum()), where)
    # The production path once more: no dense outputs, so the stencil writes records up to a width bound only - first
    # the width this very case froze at (taken over from the call above), then a bound forced to the narrowest width
    # (option spec_force: the widening freezes later, the library notices and computes the case again in full).
    # ... and (weight input) with every column chunk declared lean: hpk_stencil_lean over all tiles, the candidates that count
    # summed cell by cell (lean_max 4096) or nearly every tile handed back to hpk_stencil_s through the redo queue (lean_max 2)
    passes = [(None, None), (mw, None)] + ([(None, 4096 if seed % 2 else 2)] if inp != 'balanced' else [])
    for force, lean_max in passes:
        if force is not None:
            ctx.set_option('spec_force', force)
        if lean_max is not None:
            ctx.set_option('lean_frac_pct', 100000)
            ctx.set_option('lean_max', lean_max)
        try:
            d2 = dict()
            if mode == 'hiccups':
                again = callers.hiccups_band(rawf, gIR, gb, gb, chrom='T', pw=pw, ww=ww, maxww=maxww, sig=sig,
                                             maxapart=maxapart, res=res, min_local_reads=min_reads, min_marginal_peaks=2,
                                             onlyanchor=False, ctx=ctx, detail=d2, **(dict(balanced=cband) if inp == 'balanced' else dict(weight=weight)))
            else:
                again = callers.bhfdr_band(rawf, gIR, gb, gb, chrom='T', pw=pw[0], ww=ww[0], sig=sig,
                                           maxww=maxww, maxapart=maxapart, res=res, min_marginal_peaks=2, onlyanchor=False,
                                           ctx=ctx, detail=d2, **(dict(balanced=cband) if inp == 'balanced' else dict(weight=weight)))
        finally:
            ctx.set_option('spec_force', -1)
            if lean_max is not None:
                ctx.set_option('lean_frac_pct', 35)
                ctx.set_option('lean_max', 24)
        k2, v2 = table_arrays(again)
        # (the bounded run's tiles carry the bound's halo: sums from other tile corners, equal to rounding)
        if not (np.array_equal(k2, k) and np.allclose(v2, v, rtol=1e-10, atol=1e-12)):
            return 'MISMATCH-record-bound', desc, 'bound %s lean_max %s' % (force or 'own', lean_max)
    return 'ok', desc, '%d pixels' % len(k)


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    ctx = _lib.Context(0)
    cpu = None
    if os.environ.get('HPK_FUZZ_CPU'):          # ... every case through back-end #0 as well
        cpu = _lib.Context(-1)
        cpu.set_option('cpu_threads', int(os.environ.get('HPK_FUZZ_CPU')))
    tally = {}
    t0 = time.time()
    for seed in range(first, first + ncases):
        try:
            status, desc, note = one_case(seed, ctx, cpu)
        except Exception:
            status, desc, note = 'CRASH', dict(seed=seed), traceback.format_exc()[-600:]
        tally[status] = tally.get(status, 0) + 1
        if status.startswith('MISMATCH') or status == 'CRASH' or os.environ.get('HPK_FUZZ_VERBOSE'):
            print(status, desc, note, flush=True)
    mode = 'wide' if os.environ.get('HPK_FUZZ_WIDE') else 'big' if os.environ.get('HPK_FUZZ_BIG') else 'small'
    print('fuzz[%s]: seeds %d..%d, %d cases in %.0f s: %s' % (mode, first, first + ncases - 1, ncases, time.time() - t0, tally))
    return 1 if any(k.startswith('MISMATCH') or k == 'CRASH' for k in tally) else 0


if __name__ == '__main__':
    sys.exit(main())
