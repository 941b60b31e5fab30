"""Host-side band preparation - counterpart of the per-chromosome prep in scripts/pyHICCUPS:142-166
(`worker()`): from raw counts + balancing weights to the inputs of the scoring path.

The reference extracts `num` diagonals one by one from a COO matrix (O(num * nnz)); here the band is a dense
[n, num] array filled by one O(nnz) scatter, and the 1-D expected IR[d] is a column reduction of it.
`north_star` keeps this part on the host.
"""
import numpy as np


def band_from_coo(i, j, v, n, num, dtype=np.float32):
    """Pixels (i, j, count) of one chromosome, each listed once in either orientation (cooler's pixel table lists the
    upper triangle) -> dense band raw[r, k] = count of (r, r + k); pixels beyond the band are dropped, repeats add up.
    Replaces `Diags = [H.diagonal(i) for i in range(num)]` (pyHICCUPS:147): one O(nnz) pass in the C library
    (`hpk_band_from_coo`, host-only) instead of O(num * nnz)."""
    from . import _lib
    i = np.ascontiguousarray(i, dtype=np.int64)
    j = np.ascontiguousarray(j, dtype=np.int64)
    v = np.asarray(v)
    # counts travel as int32 or, when they are floats or do not fit, as f64 (exact up to 2^53); the band itself is f32,
    # in which counts are exact below 2^24 - the limit of the whole path (include/hpk.h), checked here on the way in
    f64 = v.dtype.kind == 'f' or (v.size > 0 and v.dtype.itemsize >= 4 and v.dtype != np.int32 and
                                  (int(v.max()) > 0x7fffffff or int(v.min()) < -0x80000000))
    if v.size and float(v.max()) >= float(1 << 24):
        raise ValueError('band_from_coo: a count of %g is beyond what the f32 band holds exactly (2^24)' % float(v.max()))
    v = np.ascontiguousarray(v, dtype=np.float64 if f64 else np.int32)
    raw = np.zeros((n, num), dtype=np.float32)
    rc = _lib.load().hpk_band_from_coo(i.ctypes.data, j.ctypes.data, v.ctypes.data, 1 if f64 else 0, i.size, n, num, num,
                                       raw.ctypes.data)
    if rc < 0:
        raise _lib.HpkError(int(rc), 'hpk_band_from_coo: bad arguments')
    return raw if dtype == np.float32 else raw.astype(dtype)


def expected_and_biases(raw, weight, mw):
    """IR[d] (0 below mw) and biases, with the reference's conventions (pyHICCUPS:149-166):

    * balanced = (count * w_r) * w_c; a *stored* (non-zero) pixel in a masked bin is NaN and is left out of the
      mean, an unstored pixel counts as 0 even in a masked bin: IR[d] = sum(finite) / ((n - d) - #NaN);
    * biases = 1 / weight, 0 where the weight is 0 or NaN.
    """
    n, num = raw.shape
    w = np.asarray(weight, dtype=np.float64)
    IR = np.zeros(num, dtype=np.float64)
    bad = np.isnan(w)
    for d in range(mw, min(num, n)):
        m = n - d
        cnt = raw[:m, d].astype(np.float64)
        diag = (cnt * w[:m]) * w[d:d + m]
        nan = np.isnan(diag) & (cnt != 0)
        diag[cnt == 0] = 0.0
        good = ~nan
        IR[d] = diag[good].mean() if good.any() else np.nan
    ok = ~((w == 0) | bad)
    biases = np.zeros_like(w)
    biases[ok] = 1 / w[ok]
    return IR, biases


def band_pixels(n, num, mw, D):
    """Number of pixels with mw <= d <= D inside the matrix (the unit of the throughput metric)."""
    return int(sum(max(n - d, 0) for d in range(mw, min(D, num - 1) + 1)))
