"""Band sources for the command lines: a cooler file (when the `cooler` package is importable) or a band archive
(.npz) - the latter is what tests, benchmarks and machines without cooler / h5py use.

Counterpart of the data access in scripts/pyHICCUPS:142-166: instead of two sparse matrices per chromosome and
`num` calls of `H.diagonal(i)`, a source hands over the dense upper band of raw counts and the balancing weights;
the balanced values are formed on the GPU.
"""
import numpy as np

from . import band as _band


class BandSource(object):
    chromnames = ()
    binsize = None

    def nbins(self, chrom):
        raise NotImplementedError

    def fetch(self, chrom, num, weight_name='weight'):
        """-> (raw f32 [n, num], weight f64 [n])"""
        raise NotImplementedError


class NpzSource(BandSource):
    """Archive written by `save_band_archive`: res, chroms, and per chromosome `raw_<c>` [n, num_max] + `weight_<c>`."""

    def __init__(self, path):
        self.z = np.load(path, allow_pickle=False)
        self.binsize = int(self.z['res'])
        self.chromnames = [str(c) for c in self.z['chroms']]

    def nbins(self, chrom):
        return int(self.z['raw_' + chrom].shape[0])

    def fetch(self, chrom, num, weight_name='weight'):
        raw = self.z['raw_' + chrom]
        n = raw.shape[0]
        w = np.asarray(self.z[weight_name + '_' + chrom], dtype=np.float64)
        if raw.shape[1] == num and raw.dtype == np.float32 and raw.flags.c_contiguous:
            return raw, w                       # stored exactly as the library wants it: no second copy
        out = np.zeros((n, num), dtype=np.float32)
        k = min(num, raw.shape[1])
        out[:, :k] = raw[:, :k]
        return out, w


def save_band_archive(path, res, bands, compressed=True):
    """bands: {chrom: (raw [n, num], weight [n])}"""
    d = dict(res=np.int64(res), chroms=np.array(list(bands), dtype='U32'))
    for c, (raw, w) in bands.items():
        d['raw_' + c] = np.asarray(raw)
        d['weight_' + c] = np.asarray(w, dtype=np.float64)
    (np.savez_compressed if compressed else np.savez)(path, **d)


class CoolerSource(BandSource):
    """cooler URI (scripts/pyHICCUPS:178-179).  Needs the `cooler` package."""

    def __init__(self, uri):
        import cooler  # noqa: deferred, optional dependency
        self.clr = cooler.Cooler(uri)
        self.binsize = self.clr.binsize
        self.chromnames = list(self.clr.chromnames)

    def nbins(self, chrom):
        lo, hi = self.clr.extent(chrom)
        return hi - lo

    def fetch(self, chrom, num, weight_name='weight'):
        lo, hi = self.clr.extent(chrom)
        px = self.clr.matrix(balance=False, as_pixels=True, join=False).fetch(chrom)
        i = px['bin1_id'].values - lo
        j = px['bin2_id'].values - lo
        raw = _band.band_from_coo(i, j, px['count'].values, hi - lo, num, dtype=np.float32)
        w = self.clr.bins().fetch(chrom)[weight_name].values.astype(np.float64)
        return raw, w


def open_source(path):
    if str(path).endswith('.npz'):
        return NpzSource(path)
    try:
        return CoolerSource(path)
    except ImportError:
        raise SystemExit('reading %s needs the `cooler` package (not installed); band archives (.npz, see '
                         'hicpeaks_amd.io.save_band_archive) work without it' % path)
