"""Host-side band preparation - counterpart of the per-chromosome prep in scripts/pyHICCUPS:142-166
(`worker()`): from raw counts + balancing weights to the inputs of the scoring path.

The reference extracts `num` diagonals one by one from a COO matrix (O(num * nnz)); here the band is a dense
[n, num] array filled by one O(nnz) scatter, and the 1-D expected IR[d] is a column reduction of it.
`north_star` keeps this part on the host.
"""
import numpy as np


def band_from_coo(i, j, v, n, num, dtype=np.float32):
    """Upper-triangle pixels (i <= j) -> dense band raw[r, k] = count of (r, r + k); pixels beyond the band are dropped.
    Replaces `Diags = [H.diagonal(i) for i in range(num)]` (pyHICCUPS:147)."""
    i = np.asarray(i, dtype=np.int64)
    j = np.asarray(j, dtype=np.int64)
    lo, hi = np.minimum(i, j), np.maximum(i, j)
    k = hi - lo
    keep = k < num
    raw = np.zeros((n, num), dtype=dtype)
    np.add.at(raw, (lo[keep], k[keep]), np.asarray(v)[keep])
    return raw


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
