"""CPU oracle: dense-band numpy restatement of the HiCCUPS / BH-FDR core of hicpeaks 0.3.9.

TEST INFRASTRUCTURE.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this module; the product (`hicpeaks_amd`) never does.

Parity status: PINNED.  Every function below is checked in `tests/test_oracle_golden.py` against
fixtures produced by the real reference (`oracle/gen_golden.py`, run with /root/reference on
python 3.9.7 / numpy 1.26.4 / scipy 1.7.1 / statsmodels 0.12.2 / scikit-learn 0.24.2).

Representation.  The reference keeps n x n scipy CSR matrices and builds one shifted copy of the
whole band per window cell (hicpeaks/callers.py:143-198).  Here every matrix is a dense *band*
``A[r, k]`` = value of pixel ``(r, r + k)``, ``0 <= k < num``; the matrix the reference builds for
window cell ``(i, j)`` of a ``(2w+1)^2`` window is ``N[r, c] = A[r + di, c + dj]`` with
``(di, dj) = (i - w, j - w)``, i.e. the band shifted by ``di`` rows and ``dj - di`` diagonals, zero
outside the stored diagonals and outside the matrix (derivation: callers.py:157-173 slices the
zero-padded diagonals of callers.py:50-96).  Cells are visited and accumulated in the reference's
own order (i-major, j-minor; dict insertion order), so the sums are bit-identical to its CSR adds.

Third-party arithmetic on the path (none of it is pinned by the reference):
  * scipy.stats.poisson(mu).cdf(k)  == scipy.special.pdtr(floor(k), mu)   (callers.py:268-270, 536-540)
  * statsmodels multipletests(method='fdr_bh'), statsmodels 0.12.2 stats/multitest.py, restated in `fdr_bh`
  * sklearn.cluster.dbscan, scipy.signal.find_peaks / peak_widths (callers.py:593-678), called as is
"""
import numpy as np
from scipy.special import pdtr


class ReferenceCrash(ValueError, ZeroDivisionError):
    """The reference raises here (ValueError with scipy 1.7.1, ZeroDivisionError with newer scipy):
    a step was entered with no unresolved candidate for its peak width (callers.py:203-208)."""


# ----------------------------------------------------------------------------- small helpers
def pw_ww_pairs(pw, ww, maxww):
    """callers.py:15-23 - (p, w') for w' = w..maxww of every pair, ordered by (w', p)."""
    pool = sorted((wp, p) for p, w in zip(pw, ww) for wp in range(w, maxww + 1))
    return [(p, wp) for wp, p in pool]


def lambdachunk(E):
    """callers.py:25-41 - log-spaced lambda chunks, membership by strict inequalities."""
    if E.size == 0:
        return []
    numbin = int(np.ceil(np.log(E.max()) / np.log(2) * 3 + 1))
    out = []
    for i in range(1, numbin + 1):
        if i == 1:
            lv, rv = 0, 1
        else:
            lv = np.power(2, ((i - 2) / 3.))
            rv = np.power(2, ((i - 1) / 3.))
        out.append((lv, rv, np.where((E > lv) & (E < rv))[0]))
    return out


def fdr_bh(pvals, alpha):
    """statsmodels 0.12.2 stats/multitest.py multipletests(method='fdr_bh') -> (reject, q)."""
    pvals = np.asarray(pvals, dtype=np.float64)
    order = np.argsort(pvals)
    ps = pvals[order]
    m = ps.size
    ecdf = np.arange(1, m + 1) / float(m)
    reject = ps <= ecdf * alpha
    if reject.any():
        reject[:np.max(np.nonzero(reject)[0])] = True
    q = np.minimum.accumulate((ps / ecdf)[::-1])[::-1]
    q[q > 1] = 1
    q_out = np.empty_like(q)
    q_out[order] = q
    r_out = np.empty_like(reject)
    r_out[order] = reject
    return r_out, q_out


def poisson_sf_as_coded(O, mu):
    """``1 - poisson(mu).cdf(O)`` (callers.py:268-270 / 536-540): 1 - pdtr(floor(O), mu)."""
    return 1 - pdtr(np.floor(O), mu)


# ----------------------------------------------------------------------------- prep (A1)
def prep_from_band(raw, weight, mw):
    """scripts/pyHICCUPS:146-166 on the band.

    ``H.diagonal(i)`` of the COO is dense: 0 at unstored pixels and NaN only at *stored* pixels
    whose bins are masked, so IR[i] = sum(finite) / ((n - i) - #stored-in-masked-bins).
    Returns (IR f64[num] with 0 below mw, cband f64[n, num] with NaN -> 0, biases f64[n])."""
    n, num = raw.shape
    r = np.arange(n)
    IR = np.zeros(num)
    cband = np.zeros((n, num))
    for i in range(mw, num):
        m = n - i
        if m <= 0:
            raise ValueError('Offset %d (index %d) out of bounds' % (i, i))   # sparse.diags, pyHICCUPS:148
        cnt = raw[:m, i].astype(np.float64)
        diag = (cnt * weight[:m]) * weight[i:i + m]
        diag[cnt == 0] = 0.0                      # unstored pixel -> 0 even in a masked bin
        mask = np.isnan(diag)
        IR[i] = diag[~mask].mean()
        diag[mask] = 0
        cband[:m, i] = diag
    ok = ~((weight == 0) | np.isnan(weight))
    biases = np.zeros_like(weight)
    biases[ok] = 1 / weight[ok]
    return IR, cband, biases


def expected_band(IR, n, num, mw):
    """EDiags / EM of callers.py:66-72 as a band: IR[k] on diagonal k (k >= mw), 0 outside the matrix."""
    X = np.zeros((n, num))
    for k in range(mw, num):
        X[:max(n - k, 0), k] = IR[k]
    return X


class _Shifter(object):
    """N[r, k] = A[r + di, k + dj - di] with zero fill (the window-cell matrices of callers.py:175-178)."""

    def __init__(self, A, W):
        n, num = A.shape
        self.n, self.num, self.W = n, num, W
        self.P = np.zeros((n + 2 * W, num + 4 * W), dtype=A.dtype)
        self.P[W:W + n, 2 * W:2 * W + num] = A

    def __call__(self, di, dj):
        W = self.W
        return self.P[W + di:W + di + self.n, 2 * W + dj - di:2 * W + dj - di + self.num]


# ----------------------------------------------------------------------------- box sums + widening (A3-A6)
def hiccups_local_sums(raw, cband, IR, n, num, pw, ww, maxww, maxapart, res, min_local_reads, trace=None):
    """callers.py:98-232.  Returns dict with candidate coords and bSV/bEV[pi][fl], step log, frozen_w."""
    mw = min(ww)
    D = maxapart // res
    rawf = raw.astype(np.float64)
    X = expected_band(IR, n, num, mw)
    SR, SC, SX = _Shifter(rawf, maxww), _Shifter(cband, maxww), _Shifter(X, maxww)

    # candidates: nonzero(M) row-major with mw <= d <= D   (callers.py:101-104)
    vx, vk = np.nonzero(raw)
    keep = (vk >= mw) & (vk <= D)
    vx, vk = vx[keep], vk[keep]
    ncand = vx.size
    bSV = {p: {'K': np.zeros(ncand), 'Y': np.zeros(ncand)} for p in pw}
    bEV = {p: {'K': np.zeros(ncand), 'Y': np.zeros(ncand)} for p in pw}
    wres = {p: np.zeros(ncand, dtype=np.int32) for p in pw}
    RefIdx = {p: np.arange(ncand) for p in pw}
    iniNum = {p: ncand for p in pw}
    totalNum = ncand

    bS = {'K': np.zeros((n, num)), 'Y': np.zeros((n, num))}
    bE = {'K': np.zeros((n, num)), 'Y': np.zeros((n, num))}
    Reads = np.zeros((n, num))
    limit = False
    last_pi = last_wi = 0
    frozen_w = maxww
    steps = []
    for pi, wi in pw_ww_pairs(pw, ww, maxww):
        if wi > frozen_w:
            continue
        ps, ws = 2 * pi + 1, 2 * wi + 1
        P1 = set((i, j) for i in range(wi - pi, ps + wi - pi) for j in range(wi - pi, ps + wi - pi))
        P_1 = set((i, j) for i in range(wi + 1, ws) for j in range(wi))
        P_2 = set((i, j) for i in range(wi + 1, ps + wi - pi) for j in range(wi - pi, wi))
        P2 = P_1 - P_2
        cells = []
        for i in range(ws):
            for j in range(ws):
                bg = max(abs(i - wi), abs(j - wi))
                if limit and (((bg <= last_wi) and (bg > max(pi, last_pi))) or (bg <= min(pi, last_pi))):
                    continue                                            # callers.py:150-152
                cells.append((i, j, bg))
        for i, j, bg in cells:
            di, dj = i - wi, j - wi
            key = (i, j)
            plus = (not limit) or bg > last_wi or (bg > pi and bg <= last_pi)   # callers.py:180 / 187
            sgn = 1.0 if plus else -1.0
            if (i != wi) and (j != wi) and (key not in P1) and (key not in P2):
                bS['K'] = bS['K'] + sgn * SC(di, dj)
                bE['K'] = bE['K'] + sgn * SX(di, dj)
            if key in P2:
                bS['K'] = bS['K'] + sgn * SC(di, dj)
                bE['K'] = bE['K'] + sgn * SX(di, dj)
                bS['Y'] = bS['Y'] + sgn * SC(di, dj)
                bE['Y'] = bE['Y'] + sgn * SX(di, dj)
                if (not limit) or (pi == min(pw) and bg > last_wi):      # callers.py:197-198
                    Reads = Reads + SR(di, dj)
        limit = True
        last_pi, last_wi = pi, wi
        if trace is not None:
            trace(len(steps), pi, wi, bS, bE, Reads)

        idx = RefIdx[pi]
        if idx.size == 0:
            raise ReferenceCrash('step (%d,%d) entered with no unresolved candidate' % (pi, wi))
        RN = Reads[vx[idx], vk[idx]]
        ok = RN >= min_local_reads
        EIdx = idx[ok]
        valid_ratio = EIdx.size / float(iniNum[pi])
        for fl in ('K', 'Y'):
            bSV[pi][fl][EIdx] = bS[fl][vx[EIdx], vk[EIdx]]
            bEV[pi][fl][EIdx] = bE[fl][vx[EIdx], vk[EIdx]]
        wres[pi][EIdx] = wi
        RefIdx[pi] = idx[~ok]
        iniNum[pi] = RefIdx[pi].size
        left_ratio = iniNum[pi] / float(totalNum)
        steps.append((pi, wi, int(EIdx.size)))
        if (valid_ratio < 0.3) and (wi >= max(ww)):
            frozen_w = wi
        if (left_ratio < 0.03) and (wi >= max(ww)):
            frozen_w = wi
    return dict(vx=vx, vy=vx + vk, bSV=bSV, bEV=bEV, wres=wres, steps=steps, frozen_w=frozen_w)


def gap_rows(cband):
    """callers.py:238 - rows of the (upper-band) balanced matrix whose sum is 0."""
    return set(np.where(cband.sum(axis=1) == 0)[0].tolist())


def gap_filter(xi, yi, gaps, mw, n):
    """callers.py:291-303 (hiccups) / 558-570 (bhfdr): indices that survive."""
    keep = []
    for t in range(xi.size):
        lo = (xi[t] - mw) if (xi[t] > mw) else 0
        hi = (xi[t] + mw) if ((xi[t] + mw) < n) else (n - 1)
        reg = set(range(lo, hi))
        lo = (yi[t] - mw) if (yi[t] > mw) else 0
        hi = (yi[t] + mw) if ((yi[t] + mw) < n) else (n - 1)
        reg |= set(range(lo, hi))
        if not (reg & gaps):
            keep.append(t)
    return np.array(keep, dtype=np.int64)


# ----------------------------------------------------------------------------- hiccups (A7-A12)
def hiccups(raw, cband, B1, B2, IR, n, num, chrom='T', pw=[2], ww=[5], maxww=20, sig=0.1, sumq=0.01,
            double_fold=1.75, single_fold=2, maxapart=2000000, res=10000, use_raw=False,
            min_marginal_peaks=3, onlyanchor=True, min_local_reads=25, detail=None):
    """callers.py:44-362 on band inputs.  `detail` (a dict) receives the intermediates."""
    mw = min(ww)
    loc = hiccups_local_sums(raw, cband, IR, n, num, pw, ww, maxww, maxapart, res, min_local_reads)
    vx, vy = loc['vx'], loc['vy']
    gaps = gap_rows(cband)
    pixel_table = {}
    sets = []
    for pi, wi in zip(pw, ww):
        pos = {}
        stats = {}
        cE_last = None
        for fl in ('K', 'Y'):
            bEV, bSV = loc['bEV'][pi][fl], loc['bSV'][pi][fl]
            mask = (bEV != 0) & (vy - vx >= wi)                                  # callers.py:244
            x, y = vx[mask], vy[mask]
            ratio = bSV[mask] / bEV[mask]
            cE = IR[y - x] * ratio                                               # EM.multiply(tmp), 247
            Eall = cE * B1[x] * B2[y]                                            # 249
            nz = cE != 0                                                         # cEM.nonzero(), 248
            keep = nz & (Eall > 0)                                               # 250
            xi, yi, E = x[keep], y[keep], Eall[keep]
            O = raw[xi, yi - xi].astype(np.float64)                              # 254
            ICE = cband[xi, yi - xi]                                             # 255
            fold = O / E
            p = np.ones(xi.size)
            q = np.ones(xi.size)
            chunk_id = np.zeros(xi.size, dtype=np.int32)
            for ci, (lv, rv, idx) in enumerate(lambdachunk(E)):                  # 263-277
                if idx.size > 0:
                    cp = poisson_sf_as_coded(O[idx], rv)
                    p[idx] = cp
                    q[idx] = fdr_bh(cp, sig)[1]
                    chunk_id[idx] = ci + 1
            sets.append(dict(pi=pi, wi=wi, fl=fl, x=x, y=y, ratio=ratio, vx=xi, vy=yi, E=E, O=O, p=p, q=q,
                             chunk=chunk_id))
            rej = q <= sig                                                       # 279
            xi, yi, O, ICE, fold, p, q = xi[rej], yi[rej], O[rej], ICE[rej], fold[rej], p[rej], q[rej]
            if len(gaps) > 0:                                                    # 291
                k = gap_filter(xi, yi, gaps, mw, n)
                xi, yi, O, ICE, fold, p, q = xi[k], yi[k], O[k], ICE[k], fold[k], p[k], q[k]
            first = O if (use_raw and fl == 'K') else ICE                        # 321-325
            stats[fl] = dict(zip(zip(xi.tolist(), yi.tolist()),
                                 zip(first.tolist(), O.tolist(), fold.tolist(), p.tolist(), q.tolist())))
            if fl == 'Y':
                cE_last = dict(zip(zip(x.tolist(), y.tolist()), cE.tolist()))    # leftover cEM of the Y pass
        preD, preL = stats['K'], stats['Y']
        common = set(preD) & set(preL)
        for key in set(preD) - set(preL):                                        # 328-331
            if cE_last.get(key, 0.0) == 0:
                common.add(key)
        for key in common:
            donut = preD[key]
            ll = preL[key] if key in preL else preD[key]
            tk = (key[0] * res, key[1] * res)
            if (donut[2] > double_fold) and (ll[2] > double_fold) and ((donut[2] > single_fold) or (ll[2] > single_fold)):
                if tk not in pixel_table:
                    pixel_table[tk] = tk + (0,) + donut + ll[2:]
                elif (donut[-1] < pixel_table[tk][7]) and (ll[-1] < pixel_table[tk][10]):   # 348
                    pixel_table[tk] = tk + (0,) + donut + ll[2:]
    Donuts = {(k[0] // res, k[1] // res): pixel_table[k][3:8] for k in pixel_table}
    LL = {(k[0] // res, k[1] // res): pixel_table[k][8:] for k in pixel_table}
    if detail is not None:
        detail.update(loc=loc, sets=sets, Donuts=Donuts, LL=LL, gaps=gaps)
    peaks = local_clustering(Donuts, LL, res, min_count=min_marginal_peaks, r=2 * res, sumq=sumq,
                             onlysummit=onlyanchor)
    final = {}
    for pixel, cen, radius in peaks:
        key = (pixel[0] * res, pixel[1] * res)
        final[key] = (cen[0] * res, cen[1] * res) + (radius * res,) + pixel_table[key][4:]
    return final


# ----------------------------------------------------------------------------- bhfdr (A14)
def bhfdr(raw, cband, B1, B2, IR, n, num, chrom='T', pw=2, ww=5, sig=0.05, maxww=20, maxapart=2000000,
          res=10000, min_marginal_peaks=3, onlyanchor=False, detail=None):
    """callers.py:364-590 on band inputs."""
    D = maxapart // res
    rawf = raw.astype(np.float64)
    X = expected_band(IR, n, num, ww)
    SR, SC, SX = _Shifter(rawf, maxww), _Shifter(cband, maxww), _Shifter(X, maxww)
    vx, vk = np.nonzero(raw)
    keep = (vk >= ww) & (vk <= D)
    vx, vk = vx[keep], vk[keep]
    ncand = vx.size
    bSV, bEV = np.zeros(ncand), np.zeros(ncand)
    RefIdx = np.arange(ncand)
    RefMask = np.ones(ncand, dtype=bool)
    iniNum = totalNum = ncand
    bS, bE, Reads = np.zeros((n, num)), np.zeros((n, num)), np.zeros((n, num))
    limit = False
    ps = 2 * pw + 1
    steps = []
    for w in range(ww, maxww + 1):
        ws = 2 * w + 1
        P1 = set((i, j) for i in range(w - pw, ps + w - pw) for j in range(w - pw, ps + w - pw))
        P_1 = set((i, j) for i in range(w + 1, ws) for j in range(w))
        P_2 = set((i, j) for i in range(w + 1, ps + w - pw) for j in range(w - pw, w))
        P2 = P_1 - P_2
        for i in range(ws):
            for j in range(ws):
                bg = max(abs(i - w), abs(j - w))
                if limit and bg < w:                                             # 455
                    continue
                di, dj = i - w, j - w
                if (i != w) and (j != w) and ((i, j) not in P1):                 # 481-483
                    bS = bS + SC(di, dj)
                    bE = bE + SX(di, dj)
                if (i, j) in P2:                                                 # 484-485
                    Reads = Reads + SR(di, dj)
        limit = True
        if RefIdx.size == 0:
            raise ReferenceCrash('window %d entered with no unresolved candidate' % w)
        RN = Reads[vx[RefIdx], vk[RefIdx]]
        ok = RN >= 16                                                            # 490
        EIdx = RefIdx[ok]
        valid_ratio = EIdx.size / float(iniNum)
        bSV[EIdx] = bS[vx[EIdx], vk[EIdx]]
        bEV[EIdx] = bE[vx[EIdx], vk[EIdx]]
        RefIdx = RefIdx[~ok]
        iniNum = RefIdx.size
        left_ratio = iniNum / float(totalNum)
        steps.append((pw, w, int(EIdx.size)))
        if valid_ratio < 0.3 or left_ratio < 0.03:                               # 505-511
            break
    RefMask[RefIdx] = False
    mask = (bEV != 0) & RefMask
    x, y = vx[mask], (vx + vk)[mask]
    ratio = bSV[mask] / bEV[mask]
    cE = IR[y - x] * ratio
    Eall = cE * B1[x] * B2[y]
    keep = (cE != 0) & (Eall > 0)
    xi, yi, E = x[keep], y[keep], Eall[keep]
    O = raw[xi, yi - xi].astype(np.float64)
    p = poisson_sf_as_coded(O, E)                                                # 536-540
    fold = O / E
    if p.size:
        reject, q = fdr_bh(p, sig)                                               # 545
    else:
        raise ReferenceCrash('multipletests on an empty array')
    if detail is not None:
        detail.update(steps=steps, x=x, y=y, ratio=ratio, vx=xi, vy=yi, E=E, O=O, p=p, q=q, reject=reject)
    xi, yi, p, q, O, fold = xi[reject], yi[reject], p[reject], q[reject], O[reject], fold[reject]
    gaps = gap_rows(cband)
    if len(gaps) > 0:
        k = gap_filter(xi, yi, gaps, ww, n)
        xi, yi, p, q, O, fold = xi[k], yi[k], p[k], q[k], O[k], fold[k]
    Donuts = dict(zip(zip(xi.tolist(), yi.tolist()), zip(O.tolist(), fold.tolist(), p.tolist(), q.tolist())))
    if detail is not None:
        detail.update(Donuts=Donuts)
    table = {}
    for pixel, cen, radius in local_clustering(Donuts, None, res, min_count=min_marginal_peaks, r=2 * res,
                                               onlysummit=onlyanchor):
        donut = Donuts[pixel]
        if donut[1] > 2:                                                         # 587
            table[(pixel[0] * res, pixel[1] * res)] = (cen[0] * res, cen[1] * res) + (radius * res,) + donut
    return table


# ----------------------------------------------------------------------------- clustering (F1)
def find_anchors(pos, min_count=3, min_dis=20000, wlen=200000, res=10000):
    """callers.py:593-634 - anchors from marginal peak-pixel counts."""
    from collections import Counter
    from scipy.signal import find_peaks, peak_widths
    min_dis = max(min_dis // res, 1)
    wlen = min(wlen // res, 10)
    count = Counter(pos)
    refidx = range(min(count) - 1, max(count) + 2)
    signal = np.r_[[count[i] for i in refidx]]
    summits = find_peaks(signal, height=min_count, distance=min_dis)[0]
    ranked = sorted(((signal[i], i) for i in summits), reverse=True)
    peaks = set()
    records = {}
    for _, i in ranked:
        tmp = peak_widths(signal, [i], rel_height=1, wlen=wlen)[2:4]
        li, ri = int(np.round(tmp[0][0])), int(np.round(tmp[1][0]))
        lb, rb = refidx[li], refidx[ri]
        hit = None
        if peaks:
            for b in range(lb, rb + 1):
                if b in records:
                    hit = records[b]
                    break
        if hit is None:
            m_lb, m_rb, summit = lb, rb, refidx[i]
        else:
            m_lb, m_rb, summit = min(lb, hit[1]), max(rb, hit[2]), hit[0]
            peaks.remove(hit)
        peaks.add((summit, m_lb, m_rb))
        for b in range(m_lb, m_rb + 1):
            records[b] = (summit, m_lb, m_rb)
    return peaks


def _cluster_core(sort_list, r, visited, final_list):
    """callers.py:636-678 - DBSCAN partition, then greedy centroid growth in value order."""
    from sklearn.cluster import dbscan
    from scipy.spatial.distance import euclidean
    pos = np.r_[[i[1] for i in sort_list]]
    if len(pos) >= 2:
        _, labels = dbscan(pos, eps=r, min_samples=2)
        pool = set()
        for i, p in enumerate(sort_list):
            if p[1] in pool:
                continue
            c = labels[i]
            if c == -1:
                continue
            sub = pos[labels == c]
            cen = p[1]
            rad = r
            local = [p[1]]
            ini = -1
            while len(sub):
                out = []
                for q in sub:
                    if tuple(q) in pool:
                        continue
                    if euclidean(q, cen) <= rad:
                        local.append(tuple(q))
                    else:
                        out.append(tuple(q))
                if len(out) == ini:
                    break
                ini = len(out)
                tmp = np.r_[local]
                cen = tuple(tmp.mean(axis=0).round().astype(int))
                rad = np.int32(np.round(max([euclidean(cen, q) for q in local]))) + r
                sub = np.r_[out]
            for q in local:
                pool.add(q)
            final_list.append((p[1], cen, rad))
        visited.update(pool)


def local_clustering(Donuts, LL, res, onlysummit=False, min_count=3, r=20000, sumq=1):
    """callers.py:680-728."""
    final_list = []
    x = np.r_[[i[0] for i in Donuts]]
    y = np.r_[[i[1] for i in Donuts]]
    if x.size == 0:
        return final_list
    x_anchors = find_anchors(x, min_count=min_count, min_dis=r, res=res)
    y_anchors = find_anchors(y, min_count=min_count, min_dis=r, res=res)
    r = max(r // res, 1)
    visited = set()
    lookup = set(zip(x, y))
    for x_a in x_anchors:
        for y_a in y_anchors:
            sort_list = []
            for i in range(x_a[1], x_a[2] + 1):
                for j in range(y_a[1], y_a[2] + 1):
                    if (i, j) in lookup:
                        sort_list.append((Donuts[(i, j)][0], (i, j)))
            sort_list.sort(reverse=True)
            _cluster_core(sort_list, r, visited, final_list)
    sort_list = []
    for i, j in zip(x, y):
        if (i, j) in visited:
            continue
        sort_list.append((Donuts[(i, j)][0], (i, j)))
    sort_list.sort(reverse=True)
    _cluster_core(sort_list, r, visited, final_list)
    x_summits = set(i[0] for i in x_anchors)
    y_summits = set(i[0] for i in y_anchors)
    for i, j in zip(x, y):
        if (i, j) in visited:
            continue
        if LL is not None:
            qpass = (Donuts[(i, j)][-1] + LL[(i, j)][-1] <= sumq)
        else:
            qpass = (Donuts[(i, j)][-1] <= sumq / 2)
        if onlysummit:
            if qpass and ((i in x_summits) or (j in y_summits)):
                final_list.append(((i, j), (i, j), 0))
        elif qpass:
            final_list.append(((i, j), (i, j), 0))
    return final_list


# ----------------------------------------------------------------------------- text output (A13)
def hiccups_lines(chrom, table, res):
    """scripts/pyHICCUPS:200-210, rows sorted by pixel for set comparison."""
    fmt = ('{0}\t{1}\t{2}\t{3}\t{4}\t{5}\t{6}\t{7:.3g}\t{8}\t{9}\t{10:.3g}\t{11:.3g}\t{12:.3g}'
           '\t{13:.3g}\t{14:.3g}\t{15:.3g}\n')
    c = 'chr' + chrom.lstrip('chr')
    return ''.join(fmt.format(*((c, px[0], px[0] + res, c, px[1], px[1] + res, '.', table[px][3], '.', '.')
                                 + tuple(table[px][4:]))) for px in sorted(table))


def bhfdr_lines(chrom, table, res):
    """scripts/pyBHFDR:169-176."""
    fmt = '{0}\t{1}\t{2}\t{3}\t{4}\t{5}\t{6}\t{7:.3g}\t{8}\t{9}\t{10:.3g}\t{11:.3g}\t{12:.3g}\n'
    c = 'chr' + chrom.lstrip('chr')
    return ''.join(fmt.format(*((c, px[0], px[0] + res, c, px[1], px[1] + res, '.', table[px][3], '.', '.')
                                 + tuple(table[px][4:]))) for px in sorted(table))
