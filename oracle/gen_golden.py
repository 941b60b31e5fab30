#!/opt/conda/bin/python3.9
"""Generate golden fixtures by running the *real* reference (hicpeaks 0.3.9) on synthetic input.

TEST INFRASTRUCTURE - runs only in the build container, where /root/reference exists:

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 oracle/gen_golden.py

Interpreter pinned by SURVEY.md §8-C1: python 3.9.7, numpy 1.26.4, scipy 1.7.1,
statsmodels 0.12.2, scikit-learn 0.24.2 (the only stack in the image that imports
`hicpeaks.callers` unmodified).  Nothing of the reference is copied: the module is imported
from where it lies and observed through wrappers installed on the names it calls
(`lambdachunk`, `poisson`, `multipletests`, `sparse.lil_matrix`, `local_clustering`) and
through its own log records.  The per-chromosome prep of scripts/pyHICCUPS:142-166 (a closure,
not importable, and it needs `cooler`) is re-stated here on scipy COO matrices.

Each fixture (tests/golden/<case>.npz) holds the inputs (raw band, weights, parameters) and what
the reference produced for them: IR / cDiags / biases (G1), per-step resolve counts (G4), per
(pair, filter) arrays x, y, bS/bE ratio, E, O, p, q before `reject` (G5), the pre-clustering
table (G6), the final table and the 16-/13-column text lines (G7), or the exception it raised.
"""
import io
import json
import logging
import os
import re
import sys
import types
import warnings

import numpy as np
from scipy import sparse

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, '/root/reference')
sys.dont_write_bytecode = True

# the generator is plain numpy and lives in the product package; load it by path so that
# importing it does not pull in the ctypes layer
import importlib.util
_spec = importlib.util.spec_from_file_location('synthetic', os.path.join(REPO, 'hicpeaks_amd', 'synthetic.py'))
synthetic = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synthetic)

warnings.filterwarnings('ignore')
import hicpeaks.callers as ref  # noqa: E402  (the real reference)


# ---------------------------------------------------------------- prep (pyHICCUPS:142-166 restated)
def cooler_like(raw, weight):
    """Symmetric COO count matrix H and balanced COO cH as `cooler` would hand them over:
    balanced = count * w_i * w_j at *stored* pixels, NaN where either bin is masked."""
    n = raw.shape[0]
    i, j, v = synthetic.band_to_coo(raw)
    off = i != j
    ii = np.r_[i, j[off]]
    jj = np.r_[j, i[off]]
    vv = np.r_[v, v[off]]
    H = sparse.coo_matrix((vv.astype(np.int64), (ii, jj)), shape=(n, n))
    lo = np.minimum(ii, jj)
    hi = np.maximum(ii, jj)
    bal = (vv.astype(np.float64) * weight[lo]) * weight[hi]
    cH = sparse.coo_matrix((bal, (ii, jj)), shape=(n, n))
    return H, cH


def worker_prep(H, cH, weight, mw, maxapart, res, maxww):
    chromLen = H.shape[0]
    num = maxapart // res + maxww + 1
    Diags = [H.diagonal(i) for i in np.arange(num)]
    M = sparse.diags(Diags, np.arange(num), format='csr')
    x = np.arange(mw, num)
    IR = {}
    cDiags = []
    for i in x:
        diag = cH.diagonal(i)
        mask = np.isnan(diag)
        notnan = diag[np.logical_not(mask)]
        IR[i] = notnan.mean()
        diag[mask] = 0
        cDiags.append(diag)
    cM = sparse.diags(cDiags, x, format='csr')
    tmp = weight.copy()
    mask = np.logical_not((tmp == 0) | np.isnan(tmp))
    biases = np.zeros_like(tmp)
    biases[mask] = 1 / tmp[mask]
    return M, cM, biases, IR, chromLen, Diags, cDiags, num


# ---------------------------------------------------------------- observers
class Tap(object):
    def __init__(self):
        self.lil = []        # (x, y, ratio) per (pair, fl)
        self.E = []          # Evalues handed to lambdachunk
        self.pois = []       # (rv, O, cdf)
        self.bh = []         # (p, q)
        self.pre = None      # (Donuts, LL) handed to local_clustering
        self.log = []
        self.gets = []       # (rows, cols, values) of every M[rows, cols] read of a CSR matrix, in call order (G3)


def install(tap):
    saved = {k: getattr(ref, k) for k in ('lambdachunk', 'poisson', 'multipletests', 'sparse', 'local_clustering')}

    real_lc = ref.lambdachunk

    def lambdachunk(E):
        tap.E.append(np.array(E, dtype=np.float64))
        return real_lc(E)

    real_pois = ref.poisson

    class PoisProxy(object):
        def __init__(self, mu):
            self.mu = mu
            self.rv = real_pois(mu)

        def cdf(self, O):
            out = self.rv.cdf(O)
            tap.pois.append((np.array(self.mu, dtype=np.float64), np.array(O, dtype=np.float64), np.array(out)))
            return out

    real_mt = ref.multipletests

    def multipletests(p, alpha=0.05, method='fdr_bh'):
        out = real_mt(p, alpha=alpha, method=method)
        tap.bh.append((np.array(p), np.array(out[1]), np.array(out[0])))
        return out

    class LilProxy(sparse.lil_matrix):
        def __setitem__(self, key, val):
            tap.lil.append((np.array(key[0]), np.array(key[1]), np.array(val, dtype=np.float64)))
            sparse.lil_matrix.__setitem__(self, key, val)

    sp = types.ModuleType('sparse_proxy')
    sp.__dict__.update({k: getattr(sparse, k) for k in dir(sparse) if not k.startswith('__')})
    sp.lil_matrix = LilProxy

    real_clu = ref.local_clustering

    def local_clustering(Donuts, LL, res, **kw):
        tap.pre = (dict(Donuts), None if LL is None else dict(LL))
        return real_clu(Donuts, LL, res, **kw)

    # G3: the widening loop reads its accumulators only through fancy indexing of CSR matrices
    # (Reads[Txi, Tyi], bS[fl][Exi, Eyi], bE[fl][Exi, Eyi]; callers.py:205, 212-213) - observe those reads
    real_getitem = sparse.csr_matrix.__getitem__

    def getitem(self, key):
        out = real_getitem(self, key)
        if tap.capture_gets and isinstance(key, tuple) and len(key) == 2 and isinstance(key[0], np.ndarray) \
                and isinstance(key[1], np.ndarray):
            tap.gets.append((np.array(key[0]), np.array(key[1]), np.array(out, dtype=np.float64).ravel()))
        return out
    sparse.csr_matrix.__getitem__ = getitem
    saved['__csr_getitem__'] = real_getitem

    ref.lambdachunk = lambdachunk
    ref.poisson = PoisProxy
    ref.multipletests = multipletests
    ref.sparse = sp
    ref.local_clustering = local_clustering

    class H(logging.Handler):
        def emit(self, record):
            tap.log.append(record.getMessage())
    h = H()
    ref.logger.addHandler(h)
    ref.logger.setLevel(logging.INFO)
    return saved, h


def uninstall(saved, h):
    sparse.csr_matrix.__getitem__ = saved.pop('__csr_getitem__')
    for k, v in saved.items():
        setattr(ref, k, v)
    ref.logger.removeHandler(h)


STEP_RE = re.compile(r'\((\d+),(\d+)\) Valid Contact Number from This Loop: (\d+)')
BH_STEP_RE = re.compile(r'Valid Contact Number from This Loop: (\d+)')


def hiccups_lines(chrom, table, res):
    """scripts/pyHICCUPS:200-210 formatting, sorted for stable comparison."""
    fmt = '{0}\t{1}\t{2}\t{3}\t{4}\t{5}\t{6}\t{7:.3g}\t{8}\t{9}\t{10:.3g}\t{11:.3g}\t{12:.3g}\t{13:.3g}\t{14:.3g}\t{15:.3g}\n'
    out = []
    for pixel in sorted(table):
        tmp = table[pixel]
        c = 'chr' + chrom.lstrip('chr')
        content = (c, pixel[0], pixel[0] + res, c, pixel[1], pixel[1] + res, '.', tmp[3], '.', '.') + tuple(tmp[4:])
        out.append(fmt.format(*content))
    return ''.join(out)


def bhfdr_lines(chrom, table, res):
    """scripts/pyBHFDR:169-176 formatting."""
    fmt = '{0}\t{1}\t{2}\t{3}\t{4}\t{5}\t{6}\t{7:.3g}\t{8}\t{9}\t{10:.3g}\t{11:.3g}\t{12:.3g}\n'
    out = []
    for pixel in sorted(table):
        tmp = table[pixel]
        c = 'chr' + chrom.lstrip('chr')
        content = (c, pixel[0], pixel[0] + res, c, pixel[1], pixel[1] + res, '.', tmp[3], '.', '.') + tuple(tmp[4:])
        out.append(fmt.format(*content))
    return ''.join(out)


def table_to_array(table):
    """dict {(x,y): tuple} -> (keys int64 [m,2], vals f64 [m,k]) sorted by key."""
    keys = sorted(table)
    if not keys:
        return np.zeros((0, 2), np.int64), np.zeros((0, 0))
    return np.array(keys, dtype=np.int64), np.array([[float(v) for v in table[k]] for k in keys], dtype=np.float64)


# ---------------------------------------------------------------- cases
def run_case(name, mode, gen, params, plant=None, mutate=None, g3=False):
    raw, weight, loops = synthetic.synth_band(**gen)
    if plant is not None:
        raw = plant(raw)
    if mutate is not None:
        raw, weight = mutate(raw, weight)
    res = params['res']
    maxapart = params['maxapart']
    maxww = params['maxww']
    pw = params['pw']
    ww = params['ww']
    mw = min(ww) if mode == 'hiccups' else ww
    H, cH = cooler_like(raw, weight)
    out = dict(raw=raw.astype(np.int32), weight=weight, loops=loops)
    meta = dict(name=name, mode=mode, gen=gen, params=params,
                versions=dict(python=sys.version.split()[0], numpy=np.__version__,
                              scipy=__import__('scipy').__version__,
                              statsmodels=__import__('statsmodels').__version__,
                              sklearn=__import__('sklearn').__version__,
                              hicpeaks=__import__('hicpeaks').__version__))
    try:
        M, cM, biases, IR, chromLen, Diags, cDiags, num = worker_prep(H, cH, weight, mw, maxapart, res, maxww)
    except Exception as e:  # contigs shorter than the band fail in sparse.diags
        meta['prep_exception'] = type(e).__name__
        np.savez_compressed(os.path.join(REPO, 'tests', 'golden', name + '.npz'),
                            meta=json.dumps(meta), **out)
        print(name, 'prep raised', type(e).__name__)
        return
    out['IR'] = np.array([IR[i] for i in sorted(IR)])
    out['cband'] = np.zeros((chromLen, num))
    for t, i in enumerate(range(mw, num)):
        out['cband'][:chromLen - i, i] = cDiags[t]
    out['biases'] = biases
    meta['num'] = int(num)
    meta['chromLen'] = int(chromLen)

    tap = Tap()
    tap.capture_gets = bool(g3)
    saved, h = install(tap)
    exc = None
    table = None
    try:
        if mode == 'hiccups':
            table = ref.hiccups(M, cM, biases, biases, IR, chromLen, Diags, cDiags, num, 'T',
                                pw=pw, ww=ww, maxww=maxww, sig=params['sig'], sumq=params['sumq'],
                                double_fold=params['double_fold'], single_fold=params['single_fold'],
                                maxapart=maxapart, res=res, use_raw=params['use_raw'],
                                min_marginal_peaks=params['min_marginal_peaks'],
                                onlyanchor=params['onlyanchor'], min_local_reads=params['min_local_reads'])
        else:
            table = ref.bhfdr(M, cM, biases, biases, IR, chromLen, Diags, cDiags, num, 'T',
                              pw=pw, ww=ww, sig=params['sig'], maxww=maxww, maxapart=maxapart, res=res,
                              min_marginal_peaks=params['min_marginal_peaks'], onlyanchor=params['onlyanchor'])
    except Exception as e:
        exc = e
    finally:
        uninstall(saved, h)

    steps = []
    for msg in tap.log:
        m = STEP_RE.search(msg)
        if m:
            steps.append((int(m.group(1)), int(m.group(2)), int(m.group(3))))
        elif mode == 'bhfdr':
            m = BH_STEP_RE.search(msg)
            if m:
                steps.append((pw, ww + len(steps), int(m.group(1))))
    out['steps'] = np.array(steps, dtype=np.int64).reshape(-1, 3)
    ncand = [int(re.search(r'Observed Contact Number: (\d+)', m).group(1)) for m in tap.log
             if 'Observed Contact Number' in m]
    meta['ncand'] = ncand[0] if ncand else None

    if g3 and exc is None and mode == 'hiccups':
        # per executed step: Reads at the candidates still unresolved for this peak width, then the four accumulators at
        # the candidates resolved by this step - exactly the reads of callers.py:205 and 212-213, in that order
        for k, (spi, swi, scnt) in enumerate(steps):
            rx, ry, rv = tap.gets[5 * k]
            out['g3_%d_ux' % k] = rx.astype(np.int32)
            out['g3_%d_uy' % k] = ry.astype(np.int32)
            out['g3_%d_reads' % k] = rv
            ex, ey, bsk = tap.gets[5 * k + 1]
            assert ex.size == scnt and int((rv >= params['min_local_reads']).sum()) == scnt
            out['g3_%d_ex' % k] = ex.astype(np.int32)
            out['g3_%d_ey' % k] = ey.astype(np.int32)
            out['g3_%d_bSK' % k] = bsk
            out['g3_%d_bEK' % k] = tap.gets[5 * k + 2][2]
            out['g3_%d_bSY' % k] = tap.gets[5 * k + 3][2]
            out['g3_%d_bEY' % k] = tap.gets[5 * k + 4][2]
            for t in (2, 3, 4):
                assert np.array_equal(tap.gets[5 * k + t][0], ex) and np.array_equal(tap.gets[5 * k + t][1], ey)
        meta['g3_steps'] = len(steps)

    if exc is not None:
        meta['exception'] = type(exc).__name__
        meta['exception_msg'] = str(exc)[:200]
    else:
        # per (pair, fl) scoring intermediates
        nsets = len(tap.lil)
        bi = 0
        pi_ = 0
        for t in range(nsets):
            x, y, ratio = tap.lil[t]
            out['s%d_x' % t] = x.astype(np.int32)
            out['s%d_y' % t] = y.astype(np.int32)
            out['s%d_ratio' % t] = ratio
            if mode == 'hiccups':
                E = tap.E[t]
                # which candidates survived E > 0, in the row-major order the reference uses
                Eall = (np.array([IR[int(d)] for d in (y - x)]) * ratio) * biases[x] * biases[y]
                keep = Eall > 0
                order = np.lexsort((y[keep], x[keep]))
                vx, vy = x[keep][order], y[keep][order]
                assert np.array_equal(Eall[keep][order], E), 'E reconstruction mismatch'
                p = np.ones(E.size)
                q = np.ones(E.size)
                O = raw[vx, vy - vx].astype(np.float64)
                chunks = ref.__dict__['lambdachunk'](E) if False else saved['lambdachunk'](E)
                chunk_id = np.zeros(E.size, dtype=np.int32)
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
                out['s%d_vx' % t] = vx.astype(np.int32)
                out['s%d_vy' % t] = vy.astype(np.int32)
                out['s%d_E' % t] = E
                out['s%d_O' % t] = O
                out['s%d_p' % t] = p
                out['s%d_q' % t] = q
                out['s%d_chunk' % t] = chunk_id
        if mode == 'bhfdr' and nsets:
            x, y, ratio = tap.lil[0]
            Eall = (np.array([IR[int(d)] for d in (y - x)]) * ratio) * biases[x] * biases[y]
            keep = Eall > 0
            order = np.lexsort((y[keep], x[keep]))
            vx, vy = x[keep][order], y[keep][order]
            mu, Oc, cdf = tap.pois[0]
            assert np.array_equal(Eall[keep][order], mu)
            pp, qq, rej = tap.bh[0]
            out['s0_vx'] = vx.astype(np.int32)
            out['s0_vy'] = vy.astype(np.int32)
            out['s0_E'] = mu
            out['s0_O'] = Oc
            out['s0_p'] = pp
            out['s0_q'] = qq
            out['s0_reject'] = rej
        meta['nsets'] = nsets
        if tap.pre is not None:
            D, L = tap.pre
            k, v = table_to_array(D)
            out['pre_keys'] = k
            out['pre_donut'] = v
            if L is not None:
                k2, v2 = table_to_array(L)
                assert np.array_equal(k, k2)
                out['pre_ll'] = v2
        k, v = table_to_array(table)
        out['final_keys'] = k
        out['final_vals'] = v
        meta['lines'] = hiccups_lines('T', table, res) if mode == 'hiccups' else bhfdr_lines('T', table, res)
        meta['nfinal'] = len(table)

    np.savez_compressed(os.path.join(REPO, 'tests', 'golden', name + '.npz'), meta=json.dumps(meta), **out)
    print('%-28s cand=%s steps=%s -> %s' % (
        name, meta.get('ncand'), [tuple(s) for s in steps][:20],
        ('EXC ' + meta['exception']) if exc is not None else
        ('pre=%d final=%d' % (0 if tap.pre is None else len(tap.pre[0]), len(table)))))


CLI = dict(res=10000, maxww=10, sig=0.05, sumq=0.01, double_fold=1.75, single_fold=2,
           use_raw=False, min_marginal_peaks=2, onlyanchor=False, min_local_reads=16)


def P(**kw):
    d = dict(CLI)
    d.update(kw)
    return d


def plant_short(raw):
    """strong 3x3 enrichments at d = 3..6 so that donut-only pixels (LL expected == 0, callers.py:328-331)
    reach the combine step (SURVEY §8-A11 'postcheck' branch)."""
    raw = raw.copy()
    n = raw.shape[0]
    rng = np.random.default_rng(99)
    for d in (3, 4, 5, 6, 4, 3, 5):
        r = int(rng.integers(20, n - 40))
        for dr in (-1, 0, 1):
            for dc in (-1, 0, 1):
                k = d + dc - dr
                if k >= 0:
                    raw[r + dr, k] = raw[r + dr, k] * 6 + 40
    return raw


def ones_with_a_block(raw, weight):
    """Every stored pixel = 1 and every weight = 1: IR[d] = 1, bS = bE (sums of ones are exact) and both biases are 1,
    so the corrected expected is *exactly* 1.0 = the boundary between the first two lambda chunks - such pixels
    belong to no chunk and keep p = q = 1 (callers.py:38, 259-260).  A block of larger counts makes the diagonals it
    touches differ (IR != 1 there), so that the same run also holds ordinary pixels and calls."""
    n, num = raw.shape
    raw = np.ones_like(raw)
    rr = np.arange(n)[:, None]
    raw[(rr + np.arange(num)[None, :]) >= n] = 0
    for (r, d) in ((60, 18), (150, 25)):
        for dr in (-1, 0, 1):
            for dc in (-1, 0, 1):
                raw[r + dr, d + dc - dr] = 30
    return raw, np.ones_like(weight)


def main():
    only = sys.argv[1:]
    cases = []
    g = dict(n=500, num=61, depth=60.0, nloops=14, seed=1)
    cases.append(('hiccups_p2w5', 'hiccups', g, P(pw=[2], ww=[5], maxapart=500000), None, None))
    g = dict(n=420, num=51, depth=25.0, nloops=12, seed=2)
    cases.append(('hiccups_union_shallow', 'hiccups', g, P(pw=[1, 2, 4], ww=[3, 5, 7], maxapart=400000), None, None))
    g = dict(n=420, num=51, depth=14.0, nloops=12, seed=7)
    cases.append(('hiccups_union_frozen', 'hiccups', g, P(pw=[1, 2, 4], ww=[3, 5, 7], maxapart=400000), None, None))
    g = dict(n=400, num=51, depth=150.0, nloops=10, seed=3)
    cases.append(('hiccups_p1w3_short', 'hiccups', g, P(pw=[1], ww=[3], maxapart=400000), plant_short, None))
    g = dict(n=400, num=49, depth=9.0, nloops=10, seed=4)
    cases.append(('hiccups_w8_pairdrop', 'hiccups', g, P(pw=[1, 2, 4], ww=[3, 5, 9], maxww=8, maxapart=400000), None, None))
    g = dict(n=360, num=51, depth=60.0, nloops=10, seed=5)
    cases.append(('hiccups_useraw_anchor', 'hiccups', g,
                  P(pw=[2], ww=[5], maxapart=400000, use_raw=True, onlyanchor=True, min_marginal_peaks=3), None, None))
    g = dict(n=360, num=51, depth=1.2, nloops=0, seed=6)
    cases.append(('hiccups_lowE', 'hiccups', g, P(pw=[1], ww=[3], maxapart=400000, min_local_reads=2), None, None))
    g = dict(n=300, num=41, depth=60.0, nloops=8, seed=8, nan_frac=0.0)
    cases.append(('hiccups_nonan', 'hiccups', g, P(pw=[2], ww=[5], maxapart=300000), None, None))
    g = dict(n=360, num=51, depth=60.0, nloops=10, seed=9)
    cases.append(('hiccups_defaults_kw', 'hiccups', g,
                  P(pw=[2], ww=[5], maxww=20, sig=0.1, maxapart=300000, min_marginal_peaks=3, onlyanchor=True,
                    min_local_reads=25), None, None))
    g = dict(n=420, num=51, depth=10.0, nloops=12, seed=12)
    cases.append(('hiccups_p2w5_shallow', 'hiccups', g, P(pw=[2], ww=[5], maxapart=400000), None, None))
    g = dict(n=420, num=51, depth=4.0, nloops=12, seed=13)
    cases.append(('hiccups_union_vshallow', 'hiccups', g, P(pw=[1, 2, 4], ww=[3, 5, 7], maxapart=400000), None, None))
    g = dict(n=300, num=46, depth=5.0, nloops=8, seed=14)
    cases.append(('hiccups_swapped_pairs', 'hiccups', g, P(pw=[2, 1], ww=[3, 5], maxapart=350000), None, None))
    # crash edges: dense multi-pair run where a pi runs out of unresolved candidates
    g = dict(n=200, num=31, depth=400.0, nloops=0, seed=10, nan_frac=0.0)
    cases.append(('hiccups_exhausted_pi', 'hiccups', g, P(pw=[1, 2], ww=[3, 5], maxapart=200000), None, None))
    g = dict(n=200, num=31, depth=60.0, nloops=0, seed=11)
    cases.append(('hiccups_empty', 'hiccups', g, P(pw=[2], ww=[5], maxapart=200000), None,
                  lambda raw, w: (raw * 0, w)))
    # bhfdr
    g = dict(n=500, num=61, depth=60.0, nloops=14, seed=21)
    cases.append(('bhfdr_p2w5', 'bhfdr', g, P(pw=2, ww=5, maxapart=500000, min_marginal_peaks=3), None, None))
    g = dict(n=420, num=51, depth=20.0, nloops=12, seed=22)
    cases.append(('bhfdr_shallow', 'bhfdr', g, P(pw=1, ww=3, maxapart=400000, min_marginal_peaks=3), None, None))
    g = dict(n=360, num=61, depth=60.0, nloops=10, seed=23)
    cases.append(('bhfdr_w20', 'bhfdr', g, P(pw=2, ww=5, maxww=20, maxapart=400000, min_marginal_peaks=3,
                                             onlyanchor=True), None, None))
    # ---- round 2: single (4,7) pair, wide bands (several column chunks of the stencil's tiling), per-step accumulators
    # captured from the reference (G3), E exactly on a chunk boundary
    g = dict(n=500, num=71, depth=12.0, nloops=14, seed=31)
    cases.append(('hiccups_p4w7', 'hiccups', g, P(pw=[4], ww=[7], maxapart=600000), None, None, True))
    g = dict(n=300, num=46, depth=12.0, nloops=8, seed=32)
    cases.append(('hiccups_union_g3', 'hiccups', g, P(pw=[1, 2, 4], ww=[3, 5, 7], maxapart=350000), None, None, True))
    g = dict(n=640, num=331, depth=60.0, nloops=30, seed=33)
    cases.append(('hiccups_wide_p2w5', 'hiccups', g, P(pw=[2], ww=[5], maxapart=3200000), None, None))
    g = dict(n=560, num=311, depth=25.0, nloops=30, seed=34)
    cases.append(('hiccups_wide_p4w7', 'hiccups', g, P(pw=[4], ww=[7], maxapart=3000000), None, None))
    g = dict(n=260, num=41, depth=5.0, nloops=0, seed=35, nan_frac=0.0)
    cases.append(('hiccups_E_on_boundary', 'hiccups', g, P(pw=[2], ww=[5], maxapart=300000), None, ones_with_a_block))
    # ---- round 3: a contig shorter than the band (n < num): the reference's worker() fails before hiccups() is reached,
    # `sparse.diags(Diags, ...)` refuses offsets beyond the matrix (scripts/pyHICCUPS:148) - the golden
    # outcome is that exception
    g = dict(n=45, num=61, depth=60.0, nloops=1, seed=41, loop_dist=(10, 20))
    cases.append(('hiccups_short_contig', 'hiccups', g, P(pw=[2], ww=[5], maxapart=500000), None, None))
    for c in cases:
        if only and c[0] not in only:
            continue
        run_case(*c)


if __name__ == '__main__':
    main()
