"""Drop-in counterparts of `hicpeaks.callers.hiccups` / `hicpeaks.callers.bhfdr` (hicpeaks 0.3.9).

Same names, positional arguments, keyword defaults, return dictionaries and failure behaviour as
hicpeaks/callers.py:44-46 and 364-365, so the `worker()` bodies of scripts/pyHICCUPS:139-175 and
scripts/pyBHFDR:112-146 run unchanged with ``from hicpeaks_amd.callers import hiccups``.

Division of labour (BASELINE.json north_star): everything per band pixel - zero-padded band, donut /
lower-left box sums, adaptive widening, corrected expected, lambda-chunk Poisson p-values and the
Benjamini-Hochberg q-values - runs on the MI355X behind the C ABI of include/hpk.h.  What is left here
is what the reference does to the few thousand pixels with q <= sig: the gap filter
(callers.py:289-312, 555-577), the donut / lower-left combination and fold thresholds (319-349), the union
over (pw, ww) pairs, clustering (353-360, 579-588) and the result dictionaries.

There is no CPU implementation of the scoring path in this package: without libhpk.so and a gfx950
device the functions raise `hicpeaks_amd.HpkError`.
"""
import logging

import numpy as np

from . import _lib
from .clustering import local_clustering

logger = logging.getLogger(__name__)

__all__ = ['pw_ww_pairs', 'lambdachunk', 'hiccups', 'bhfdr', 'hiccups_band', 'bhfdr_band', 'hiccups_batch_submit',
           'bhfdr_batch_submit', 'local_clustering']


def pw_ww_pairs(pw, ww, maxww):
    """callers.py:15-23: the widening plan [(p, w') for w' in w..maxww] ordered by (w', p)."""
    pool = sorted((i, p) for p, w in zip(pw, ww) for i in range(w, maxww + 1))
    return [(p, i) for i, p in pool]


def lambdachunk(E):
    """callers.py:25-41: [(lv, rv, idx)] with the strict-inequality membership of the reference.  The device
    kernel applies the same boundaries (handed over by `_lib.Context`); this host version exists for API parity."""
    if E.size == 0:
        return []
    numbin = int(np.ceil(np.log(E.max()) / np.log(2) * 3 + 1))
    pool = []
    for i in range(1, numbin + 1):
        lv, rv = (0, 1) if i == 1 else (np.power(2, ((i - 2) / 3.)), np.power(2, ((i - 1) / 3.)))
        pool.append((lv, rv, np.where((E > lv) & (E < rv))[0]))
    return pool


# ----------------------------------------------------------------------------- band assembly
def _bands_from_diags(Diags, cDiags, IR, chromLen, num, mw):
    """(Diags, cDiags, IR) of scripts/pyHICCUPS:147-158 -> dense bands raw f32 [n, num], balanced f64 [n, num],
    IR f64 [num]."""
    n = int(chromLen)
    raw = np.zeros((n, num), dtype=np.float32)
    for i in range(num):
        d = np.asarray(Diags[i])
        raw[:d.size, i] = d
    bal = np.zeros((n, num), dtype=np.float64)
    ks = sorted(IR)
    IRa = np.zeros(num, dtype=np.float64)
    for t, k in enumerate(ks):
        IRa[k] = IR[k]
        d = np.asarray(cDiags[t], dtype=np.float64)
        bal[:d.size, k] = d
    return raw, bal, IRa


def _gap_keep(x, y, gap, mw, n):
    """callers.py:291-303: keep a pixel unless range(lo, hi) around x or y holds a gap row."""
    if x.size == 0 or not gap.any():
        return np.ones(x.size, dtype=bool)
    G = np.concatenate([[0], np.cumsum(gap)])

    def hit(v):
        lo = np.where(v > mw, v - mw, 0)
        hi = np.where(v + mw < n, v + mw, n - 1)
        hi = np.maximum(hi, lo)
        return (G[hi] - G[lo]) > 0
    return ~(hit(x) | hit(y))


# ----------------------------------------------------------------------------- hiccups
def _finish_hiccups(R, n, chrom, pw, ww, sig, sumq, double_fold, single_fold, res, use_raw, min_marginal_peaks,
                    onlyanchor):
    mw = min(ww)
    for pi, wi, cnt, ex in R.steps:
        if ex:
            logger.info('Chrom:{0},    ({1},{2}) Valid Contact Number from This Loop: {3}'.format(chrom, pi, wi, cnt))
    pixel_table = {}
    for j, (pi, wi) in enumerate(zip(pw, ww)):
        pre = {}
        for fl in (0, 1):
            s = R.sets[2 * j + fl]
            logger.info('Chrom:{0},    ({1},{2}), Valid contact number: {3}'.format(chrom, pi, wi, s['nvalid']))
            logger.info('Chrom:{0},    ({1},{2}), Number of chunks: {3}'.format(chrom, pi, wi, s['numbin']))
            keep = _gap_keep(s['x'], s['y'], R.gap, mw, n)
            x, y = s['x'][keep], s['y'][keep]
            O, ICE, E, p, q = s['O'][keep], s['bal'][keep], s['E'][keep], s['p'][keep], s['q'][keep]
            fold = O / E
            first = O if (use_raw and fl == 0) else ICE                          # callers.py:321-325
            pre[fl] = dict(zip(zip(x.tolist(), y.tolist()),
                               zip(first.tolist(), O.tolist(), fold.tolist(), p.tolist(), q.tolist())))
            if fl == 0:
                ll_zero = dict(zip(zip(x.tolist(), y.tolist()), s['other_zero'][keep].tolist()))
        preD, preL = pre[0], pre[1]
        common = set(preD) & set(preL)
        for key in set(preD) - set(preL):                                        # callers.py:328-331
            if ll_zero[key]:
                common.add(key)
        for key in common:
            donut = preD[key]
            ll = preL[key] if key in preL else donut
            tk = (key[0] * res, key[1] * res)
            if (donut[2] > double_fold) and (ll[2] > double_fold) and ((donut[2] > single_fold) or (ll[2] > single_fold)):
                if tk not in pixel_table:
                    pixel_table[tk] = tk + (0,) + donut + ll[2:]
                elif (donut[-1] < pixel_table[tk][7]) and (ll[-1] < pixel_table[tk][10]):   # callers.py:348
                    pixel_table[tk] = tk + (0,) + donut + ll[2:]
    Donuts = {(k[0] // res, k[1] // res): pixel_table[k][3:8] for k in pixel_table}
    LL = {(k[0] // res, k[1] // res): pixel_table[k][8:] for k in pixel_table}
    peak_list = local_clustering(Donuts, LL, res, min_count=min_marginal_peaks, r=2 * res, sumq=sumq,
                                 onlysummit=onlyanchor)
    final_table = {}
    for pixel, cen, radius in peak_list:
        key = (pixel[0] * res, pixel[1] * res)
        final_table[key] = (cen[0] * res, cen[1] * res) + (radius * res,) + pixel_table[key][4:]
    return final_table, pixel_table


class PendingCall(object):
    """A chromosome whose kernels are in flight (hpk_submit_band); result() waits for it and runs the host half
    (gap filter, donut / lower-left combine, clustering).  Lets a loop over chromosomes stay one ahead."""

    def __init__(self, job, finish):
        self._job, self._finish = job, finish

    def result(self):
        return self._finish(self._job.result())


def hiccups_band_submit(raw, IR, B1, B2, chrom='', balanced=None, weight=None, pw=[2], ww=[5], maxww=20, sig=0.1,
                        sumq=0.01, double_fold=1.75, single_fold=2, maxapart=2000000, res=10000, use_raw=False,
                        min_marginal_peaks=3, onlyanchor=True, min_local_reads=25, device=0, detail=None, ctx=None):
    """Arguments of `hiccups_band`; returns a PendingCall."""
    ctx = ctx or _lib.default_context(device)
    flags = _lib.FLAG_NO_STENCIL_TIMING
    if detail is not None and detail.get('dense'):
        flags |= _lib.FLAG_DENSE_SUMS
    prm = _lib.make_params(_lib.MODE_HICCUPS, pw, ww, maxww, sig, maxapart, res, min_local_reads, flags)
    n = raw.shape[0]
    logger.info('Chrom:{0}, Two local neighborhoods, two expected matrices ...'.format(chrom))
    job = ctx.submit_host(raw, IR, B1, B2, prm, balanced=balanced, weight=weight)
    finish = _hiccups_finisher(prm, n, chrom, list(pw), list(ww), sig, sumq, double_fold, single_fold, res, use_raw,
                               min_marginal_peaks, onlyanchor, detail)
    return PendingCall(job, finish)


def _hiccups_finisher(prm, n, chrom, pw, ww, sig, sumq, double_fold, single_fold, res, use_raw, min_marginal_peaks,
                      onlyanchor, detail=None):
    """The host half of one chromosome (callers.py:289-362) as a function of its BandResult."""
    def finish(R):
        logger.info('Chrom:{0}, Observed Contact Number: {1}'.format(chrom, R.ncand))
        if R.redone:
            logger.info('Chrom:{0}, widening froze at {1}, beyond the width bound taken from the previous chromosomes: '
                        'computed once more in full'.format(chrom, R.frozen_w))
        npairs = prm.npairs
        final, table = _finish_hiccups(R, n, chrom, pw[:npairs], ww[:npairs], sig, sumq, double_fold,
                                       single_fold, res, use_raw, min_marginal_peaks, onlyanchor)
        if detail is not None:
            detail['result'] = R
            detail['pixel_table'] = table
        return final
    return finish


class PendingBatch(object):
    """A batch of chromosomes whose kernels are in flight (hpk_submit_batch); results() waits for it and runs the host half
    of every chromosome, in submission order."""

    def __init__(self, job, finishers):
        self._job, self._finishers = job, finishers

    def results(self):
        return self.collect()()

    def collect(self):
        """Waits for the kernels and the library's host half (hpk_collect_batch) -> a callable that runs the Python half of every
        chromosome (gap filter, donut / lower-left combination, clustering) and returns the tables; it touches no GPU state and
        may run on another thread while the caller submits the next batch."""
        fins, Rs = self._finishers, self._job.results()
        return lambda: [f(R) for f, R in zip(fins, Rs)]


def hiccups_batch_submit(items, pw=[2], ww=[5], maxww=20, sig=0.1, sumq=0.01, double_fold=1.75, single_fold=2,
                         maxapart=2000000, res=10000, use_raw=False, min_marginal_peaks=3, onlyanchor=True, min_local_reads=25,
                         device=0, ctx=None):
    """`hiccups_band` for several chromosomes at once - the loop of scripts/pyHICCUPS:192-198 as ONE set of kernel launches
    (hpk_submit_batch).  items: [(chrom, raw [n, num], weight [n][, biases [n] or None]), ...]; IR and - unless given -
    the biases are derived on the device (scripts/pyHICCUPS:149-166).  Returns a PendingBatch whose results() are the
    chromosomes' final tables."""
    ctx = ctx or _lib.default_context(device)
    prm = _lib.make_params(_lib.MODE_HICCUPS, pw, ww, maxww, sig, maxapart, res, min_local_reads, _lib.FLAG_NO_STENCIL_TIMING)
    fins = []
    for it in items:
        chrom, raw = it[0], it[1]
        logger.info('Chrom:{0}, Two local neighborhoods, two expected matrices ...'.format(chrom))
        fins.append(_hiccups_finisher(prm, raw.shape[0], chrom, list(pw), list(ww), sig, sumq, double_fold, single_fold, res,
                                      use_raw, min_marginal_peaks, onlyanchor))
    job = ctx.submit_batch_host([_batch_item(it) for it in items], prm)
    return PendingBatch(job, fins)


def _batch_item(it):
    b = it[3] if len(it) > 3 else None
    return dict(raw=it[1], weight=it[2], bias1=b, bias2=b)


def hiccups_band(raw, IR, B1, B2, chrom='', balanced=None, weight=None, pw=[2], ww=[5], maxww=20, sig=0.1,
                 sumq=0.01, double_fold=1.75, single_fold=2, maxapart=2000000, res=10000, use_raw=False,
                 min_marginal_peaks=3, onlyanchor=True, min_local_reads=25, device=0, detail=None, ctx=None):
    """`hiccups` on band inputs: raw [n, num] counts, IR [num], and either the balanced f64 band or the
    balancing weights (balanced is then formed on chip).  With IR = B1 = B2 = None and `weight` given, the 1-D
    expected and the biases are derived on the device too (scripts/pyHICCUPS:149-166).  Returns the reference's
    final_table."""
    return hiccups_band_submit(raw, IR, B1, B2, chrom=chrom, balanced=balanced, weight=weight, pw=pw, ww=ww, maxww=maxww,
                               sig=sig, sumq=sumq, double_fold=double_fold, single_fold=single_fold, maxapart=maxapart,
                               res=res, use_raw=use_raw, min_marginal_peaks=min_marginal_peaks, onlyanchor=onlyanchor,
                               min_local_reads=min_local_reads, device=device, detail=detail, ctx=ctx).result()


def hiccups(M, cM, B1, B2, IR, chromLen, Diags, cDiags, num, chrom, pw=[2], ww=[5],
            maxww=20, sig=0.1, sumq=0.01, double_fold=1.75, single_fold=2, maxapart=2000000,
            res=10000, use_raw=False, min_marginal_peaks=3, onlyanchor=True, min_local_reads=25, **kw):
    """Drop-in for hicpeaks.callers.hiccups (callers.py:44-362).

    `M` / `cM` are accepted for signature compatibility; the band is taken from `Diags` / `cDiags`, which hold
    the same numbers (scripts/pyHICCUPS:147-159).  Extra keywords: device=<ordinal>, detail=<dict>."""
    raw, bal, IRa = _bands_from_diags(Diags, cDiags, IR, chromLen, num, min(ww))
    return hiccups_band(raw, IRa, B1, B2, chrom=chrom, balanced=bal, pw=pw, ww=ww, maxww=maxww, sig=sig, sumq=sumq,
                        double_fold=double_fold, single_fold=single_fold, maxapart=maxapart, res=res,
                        use_raw=use_raw, min_marginal_peaks=min_marginal_peaks, onlyanchor=onlyanchor,
                        min_local_reads=min_local_reads, **kw)


# ----------------------------------------------------------------------------- bhfdr
def bhfdr_band_submit(raw, IR, B1, B2, chrom='', balanced=None, weight=None, pw=2, ww=5, sig=0.05, maxww=20,
                      maxapart=2000000, res=10000, min_marginal_peaks=3, onlyanchor=False, device=0, detail=None, ctx=None):
    """Arguments of `bhfdr_band`; returns a PendingCall."""
    ctx = ctx or _lib.default_context(device)
    flags = _lib.FLAG_NO_STENCIL_TIMING
    if detail is not None and detail.get('dense'):
        flags |= _lib.FLAG_DENSE_SUMS
    prm = _lib.make_params(_lib.MODE_BHFDR, [pw], [ww], maxww, sig, maxapart, res, 16, flags)
    n = raw.shape[0]
    logger.info('Chrom:{0}, Calculate the expected matrix ...'.format(chrom))
    job = ctx.submit_host(raw, IR, B1, B2, prm, balanced=balanced, weight=weight)
    return PendingCall(job, _bhfdr_finisher(n, chrom, ww, res, min_marginal_peaks, onlyanchor, detail))


def bhfdr_batch_submit(items, pw=2, ww=5, sig=0.05, maxww=20, maxapart=2000000, res=10000, min_marginal_peaks=3,
                       onlyanchor=False, device=0, ctx=None):
    """`bhfdr_band` for several chromosomes at once (see hiccups_batch_submit); items: [(chrom, raw, weight), ...]."""
    ctx = ctx or _lib.default_context(device)
    prm = _lib.make_params(_lib.MODE_BHFDR, [pw], [ww], maxww, sig, maxapart, res, 16, _lib.FLAG_NO_STENCIL_TIMING)
    fins = []
    for it in items:
        chrom, raw = it[0], it[1]
        logger.info('Chrom:{0}, Calculate the expected matrix ...'.format(chrom))
        fins.append(_bhfdr_finisher(raw.shape[0], chrom, ww, res, min_marginal_peaks, onlyanchor))
    job = ctx.submit_batch_host([_batch_item(it) for it in items], prm)
    return PendingBatch(job, fins)


def _bhfdr_finisher(n, chrom, ww, res, min_marginal_peaks, onlyanchor, detail=None):
    """The host half of one chromosome (callers.py:555-590) as a function of its BandResult."""
    def finish(R):
        logger.info('Chrom:{0}, Observed Contact Number: {1}'.format(chrom, R.ncand))
        if R.redone:
            logger.info('Chrom:{0}, widening froze at {1}, beyond the width bound taken from the previous chromosomes: '
                        'computed once more in full'.format(chrom, R.frozen_w))
        s = R.sets[0]
        logger.info('Chrom:{0}, Number of Poisson Models: {1}'.format(chrom, s['nvalid']))
        if s['nvalid'] == 0:
            # statsmodels' multipletests fails on an empty p-value array (callers.py:545)
            raise _lib.EmptyStepError(_lib.ERR_EMPTY_STEP, 'no pixel with a positive expected value; the reference '
                                      'fails in multipletests (hicpeaks/callers.py:545)')
        keep = _gap_keep(s['x'], s['y'], R.gap, ww, n)                               # callers.py:557-577
        x, y, O, E, p, q = s['x'][keep], s['y'][keep], s['O'][keep], s['E'][keep], s['p'][keep], s['q'][keep]
        fold = O / E
        Donuts = dict(zip(zip(x.tolist(), y.tolist()), zip(O.tolist(), fold.tolist(), p.tolist(), q.tolist())))
        pixel_list = local_clustering(Donuts, None, res, min_count=min_marginal_peaks, r=2 * res, onlysummit=onlyanchor)
        pixel_table = {}
        for pixel, cen, radius in pixel_list:
            donut = Donuts[pixel]
            if donut[1] > 2:                                                         # callers.py:587
                pixel_table[(pixel[0] * res, pixel[1] * res)] = (cen[0] * res, cen[1] * res) + (radius * res,) + donut
        if detail is not None:
            detail['result'] = R
            detail['Donuts'] = Donuts
        return pixel_table
    return finish


def bhfdr_band(raw, IR, B1, B2, chrom='', balanced=None, weight=None, pw=2, ww=5, sig=0.05, maxww=20,
               maxapart=2000000, res=10000, min_marginal_peaks=3, onlyanchor=False, device=0, detail=None, ctx=None):
    return bhfdr_band_submit(raw, IR, B1, B2, chrom=chrom, balanced=balanced, weight=weight, pw=pw, ww=ww, sig=sig,
                             maxww=maxww, maxapart=maxapart, res=res, min_marginal_peaks=min_marginal_peaks,
                             onlyanchor=onlyanchor, device=device, detail=detail, ctx=ctx).result()


def bhfdr(M, cM, B1, B2, IR, chromLen, Diags, cDiags, num, chrom, pw=2, ww=5, sig=0.05, maxww=20,
          maxapart=2000000, res=10000, min_marginal_peaks=3, onlyanchor=False, **kw):
    """Drop-in for hicpeaks.callers.bhfdr (callers.py:364-590)."""
    raw, bal, IRa = _bands_from_diags(Diags, cDiags, IR, chromLen, num, ww)
    return bhfdr_band(raw, IRa, B1, B2, chrom=chrom, balanced=bal, pw=pw, ww=ww, sig=sig, maxww=maxww,
                      maxapart=maxapart, res=res, min_marginal_peaks=min_marginal_peaks, onlyanchor=onlyanchor, **kw)
