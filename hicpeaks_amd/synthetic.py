"""Synthetic banded Hi-C matrices (SURVEY.md §8-D2 recipe).

numpy-only and free of any device code so the same generator feeds the golden
fixture script (run under the reference's interpreter), the CPU tests and the
small-size GPU parity tests.  `bench.py` uses the device-side twin in
`hicpeaks_amd.bandgen` for full-size configs.

The output is the *upper band* of a symmetric contact matrix in the band layout
used everywhere in this package: ``raw[r, k]`` holds the count of pixel
``(r, r + k)`` for ``0 <= k < num`` (0 where ``r + k >= n``).
"""
import numpy as np


def structure_fields(n, num, seed, tads=True, compartments=True, patches=6):
    """What a real map has and a Poisson band does not (README.rst:129-233 runs on real maps; the sample data is not in the
    image): per-bin fields from which the rate of pixel (r, c) is multiplied by

      * a TAD gain where r and c lie in the same block of a random partition into blocks of 20 .. 200 bins (x 2 .. 4),
      * a compartment factor - bins carry a sign in runs of 50 .. 300 bins; same sign x 1.6, opposite x 0.6 - a checkerboard
        that reaches the far field,
      * the gain (x 6 .. 10) of a few dense far-field rectangles of 30 .. 120 bins a side, at distances beyond a third of the band.

    Returns a dict of small arrays (uploadable as they are: `hicpeaks_amd.bandgen` applies them on the device, slab by slab)."""
    rng = np.random.default_rng(1000003 * int(seed) + 17)
    f = {}
    if tads:
        ids, gains, pos = np.zeros(n, np.int32), [], 0
        while pos < n:
            size = int(rng.integers(20, 201))
            ids[pos:pos + size] = len(gains)
            gains.append(float(rng.uniform(2.0, 4.0)))
            pos += size
        f['tad_id'], f['tad_gain'] = ids, np.array(gains, np.float32)
    if compartments:
        comp, pos, sign = np.zeros(n, np.int8), 0, 1
        while pos < n:
            size = int(rng.integers(50, 301))
            comp[pos:pos + size] = sign
            sign = -sign
            pos += size
        f['comp'] = comp
    if patches:
        rects = []
        for _ in range(int(patches)):
            h, w = int(rng.integers(30, 121)), int(rng.integers(30, 121))
            d = int(rng.integers(max(1, num // 3), max(num // 3 + 1, num - 1)))
            r = int(rng.integers(0, max(1, n - d - w)))
            rects.append((r, r + h, r + d, r + d + w, float(rng.uniform(6.0, 10.0))))
        f['patches'] = np.array(rects, np.float64).reshape(-1, 5)
    return f


def structure_gain(fields, r, c, xp=np):
    """Rate multiplier of the pixels (r, c) - arrays of one shape, c clipped into the matrix by the caller - under
    `structure_fields`; `xp` = numpy, or torch with the fields as tensors on the pixels' device."""
    is_np = xp is np
    g = np.ones(r.shape, dtype=np.float32) if is_np else xp.ones(r.shape, dtype=xp.float32, device=r.device)

    def const(v):
        return np.float32(v) if is_np else xp.tensor(v, dtype=xp.float32, device=r.device)
    if 'tad_id' in fields:
        same = fields['tad_id'][r] == fields['tad_id'][c]
        g = xp.where(same, fields['tad_gain'][fields['tad_id'][r]], g)
    if 'comp' in fields:
        g = g * xp.where(fields['comp'][r] == fields['comp'][c], const(1.6), const(0.6))
    if 'patches' in fields:
        for r0, r1, c0, c1, gain in (fields['patches'].tolist() if hasattr(fields['patches'], 'tolist') else fields['patches']):
            inside = (r >= r0) & (r < r1) & (c >= c0) & (c < c1)
            g = xp.where(inside, g * float(gain), g)
    return g


def synth_band(n, num, depth=60.0, alpha=1.0, nloops=20, seed=0,
               nan_frac=0.025, enrich=8.0, loop_dist=None, dtype=np.int32, structure=None):
    """Return ``(raw, weight, loops)``.

    structure  None, or keyword arguments of `structure_fields` (e.g. ``{}`` for all three kinds): TAD blocks, a compartment
            checkerboard and dense far-field patches on top of the distance decay.

    raw     int array [n, num]: Poisson(depth * (1 + k) ** -alpha) per diagonal k, with
            `nloops` planted 3x3 enrichments (x `enrich` over the local rate).
    weight  f64 [n]: 1 / sqrt(rowsum + 1) balancing-like weights with NaN runs
            (one long "centromere" run plus isolated bins) covering ~nan_frac of
            the bins (pyHICCUPS:163-166 treats NaN / 0 weights as masked bins).
    loops   int array [nloops, 2] of planted (row, col) anchors.
    """
    rng = np.random.default_rng(seed)
    k = np.arange(num, dtype=np.float64)
    lam = depth * (1.0 + k) ** (-alpha)
    lam2d = np.broadcast_to(lam, (n, num)).copy()
    if structure is not None:
        fields = structure_fields(n, num, seed, **structure)
        rr_ = np.broadcast_to(np.arange(n)[:, None], (n, num))
        cc_ = np.minimum(rr_ + np.arange(num)[None, :], n - 1)
        lam2d *= structure_gain(fields, rr_, cc_)
    loops = []
    if nloops > 0:
        lo_d, hi_d = loop_dist if loop_dist is not None else (10, max(11, num - 15))
        hi_d = max(hi_d, lo_d + 1)
        for _ in range(nloops):
            d = int(rng.integers(lo_d, hi_d))
            r = int(rng.integers(2, max(3, n - d - 2)))
            loops.append((r, r + d))
            for dr in (-1, 0, 1):
                for dc in (-1, 0, 1):
                    rr, cc = r + dr, r + d + dc
                    kk = cc - rr
                    if 0 <= rr < n and 0 <= cc < n and 0 <= kk < num:
                        lam2d[rr, kk] *= enrich
    # (planted loops may pile up on one cell of a tiny chromosome: 8^k times the rate overflowed the integer type and left
    # negative "counts"; counts stay below 2^24, where f32 holds them exactly)
    np.minimum(lam2d, 4.0e6, out=lam2d)
    raw = rng.poisson(lam2d).astype(dtype)
    # zero the part of the band that falls outside the matrix
    rr = np.arange(n)[:, None]
    raw[(rr + np.arange(num)[None, :]) >= n] = 0

    # marginal sums of the symmetric matrix restricted to the band
    rowsum = raw.sum(axis=1).astype(np.float64)
    colsum = np.zeros(n)
    for kk in range(1, min(num, n)):          # diagonals beyond the matrix (num > n) hold nothing
        colsum[kk:] += raw[: n - kk, kk]
    weight = 1.0 / np.sqrt(rowsum + colsum + 1.0)

    nbad = int(round(n * nan_frac))
    if nbad > 0:
        run = max(1, (2 * nbad) // 3)
        start = int(rng.integers(n // 3, max(n // 3 + 1, 2 * n // 3 - run)))
        weight[start:start + run] = np.nan
        rest = nbad - run
        if rest > 0:
            idx = rng.choice(n, size=rest, replace=False)
            weight[idx] = np.nan
    return raw, weight, np.array(loops, dtype=np.int64).reshape(-1, 2)


def balanced_band(raw, weight, mw=0):
    """f64 band of balanced values ``(count * w_r) * w_c`` with NaN -> 0 and
    diagonals < mw zeroed (the `cDiags` of pyHICCUPS:149-158, as one array)."""
    n, num = raw.shape
    r = np.arange(n)[:, None]
    c = r + np.arange(num)[None, :]
    wc = np.where(c < n, weight[np.minimum(c, n - 1)], 0.0)
    bal = (raw.astype(np.float64) * weight[:, None]) * wc
    bal[np.isnan(bal)] = 0.0
    bal[:, :mw] = 0.0
    bal[c >= n] = 0.0
    return bal


def band_to_coo(raw):
    """Upper-triangle COO triplets (i, j, v) of the non-zero band pixels."""
    r, k = np.nonzero(raw)
    return r.astype(np.int64), (r + k).astype(np.int64), raw[r, k]


HG38_SIZES = {
    '1': 248956422, '2': 242193529, '3': 198295559, '4': 190214555, '5': 181538259,
    '6': 170805979, '7': 159345973, '8': 145138636, '9': 138394717, '10': 133797422,
    '11': 135086622, '12': 133275309, '13': 114364328, '14': 107043718, '15': 101991189,
    '16': 90338345, '17': 83257441, '18': 80373285, '19': 58617616, '20': 64444167,
    '21': 46709983, '22': 50818468, 'X': 156040895,
}


def hg38_bins(res):
    """Bin counts of chr1-22,X at `res` bp (CLI default --chroms '#' 'X')."""
    return {c: -(-s // res) for c, s in HG38_SIZES.items()}
