"""Band sources for the command lines: a cooler file (through the `cooler` package when it is importable, else through
hicpeaks_amd/cool.py on h5py or libhdf5) or a band archive (.npz).

Counterpart of the data access in scripts/pyHICCUPS:142-166: instead of two sparse matrices per chromosome and
`num` calls of `H.diagonal(i)`, a source hands over the dense upper band of raw counts and the balancing weights;
the balanced values are formed on the GPU.
"""
import struct
import zipfile

import numpy as np

from . import band as _band


class BandSource(object):
    chromnames = ()
    binsize = None

    def nbins(self, chrom):
        raise NotImplementedError

    def fetch(self, chrom, num, weight_name='weight'):
        """-> (raw f32 [n, num], weight f64 [n], biases f64 [n] or None: see CoolerSource)"""
        raise NotImplementedError


class NpzSource(BandSource):
    """Archive written by `save_band_archive`: res, chroms, and per chromosome `raw_<c>` [n, num_max] + `weight_<c>`."""

    def __init__(self, path):
        self.path = str(path)
        self.z = np.load(path, allow_pickle=False)
        self.binsize = int(self.z['res'])
        self.chromnames = [str(c) for c in self.z['chroms']]
        self._zip = zipfile.ZipFile(self.path)

    def _member(self, name):
        """A member of the archive as an array.  Members stored without compression are mapped straight from the file
        (page cache -> upload, no copy and no CRC pass: numpy's own reader moves ~0.2-0.7 GB/s, which was 5 of the 7 s
        of a whole-genome run at 5 kb); compressed ones go through numpy."""
        try:
            info = self._zip.getinfo(name + '.npy')
            if info.compress_type != zipfile.ZIP_STORED:
                return self.z[name]
            with open(self.path, 'rb') as f:
                f.seek(info.header_offset)
                hdr = f.read(30)
                if hdr[:4] != b'PK\x03\x04':
                    return self.z[name]
                nlen, elen = struct.unpack('<HH', hdr[26:30])
                f.seek(info.header_offset + 30 + nlen + elen)
                version = np.lib.format.read_magic(f)
                if version == (1, 0):
                    shape, fortran, dtype = np.lib.format.read_array_header_1_0(f)
                elif version == (2, 0):
                    shape, fortran, dtype = np.lib.format.read_array_header_2_0(f)
                else:
                    return self.z[name]
                off = f.tell()
            if dtype.hasobject or fortran or int(np.prod(shape)) == 0:
                return self.z[name]
            return np.memmap(self.path, dtype=dtype, mode='r', offset=off, shape=tuple(shape), order='C')
        except (KeyError, ValueError, OSError):
            return self.z[name]

    def nbins(self, chrom):
        return int(self._member('raw_' + chrom).shape[0])

    def fetch(self, chrom, num, weight_name='weight'):
        raw = self._member('raw_' + chrom)
        n = raw.shape[0]
        w = np.asarray(self.z[weight_name + '_' + chrom], dtype=np.float64)
        if raw.shape[1] == num and raw.dtype == np.float32 and raw.flags.c_contiguous:
            return raw, w, None                 # stored exactly as the library wants it: no second copy
        out = np.zeros((n, num), dtype=np.float32)
        k = min(num, raw.shape[1])
        out[:, :k] = raw[:, :k]
        return out, w, None


def save_band_archive(path, res, bands, compressed=True):
    """bands: {chrom: (raw [n, num], weight [n])}"""
    d = dict(res=np.int64(res), chroms=np.array(list(bands), dtype='U32'))
    for c, (raw, w) in bands.items():
        d['raw_' + c] = np.asarray(raw)
        d['weight_' + c] = np.asarray(w, dtype=np.float64)
    (np.savez_compressed if compressed else np.savez)(path, **d)


class CoolerSource(BandSource):
    """cooler URI (scripts/pyHICCUPS:178-179): through the `cooler` package when it is importable, otherwise through the
    package's own reader of the format (hicpeaks_amd/cool.py: h5py or libhdf5).

    Balancing (scripts/pyHICCUPS:143 `Lib.matrix(balance=<column>)`): balanced = count * w[bin1] * w[bin2]; for a
    *divisive* column (by its name, as cooler.matrix decides: 'KR', 'VC', 'VC_SQRT' / 'SQRT_VC' - the same rule in both backends) count / (w[bin1] *
    w[bin2]) - `fetch` then hands over 1 / column as the weights, and, as third value, the biases the reference forms
    from the column as stored (1 / column, 0 where it is 0 / NaN: scripts/pyHICCUPS:163-166 does not know about
    divisive columns), so that the corrected expected comes out as the reference's does."""

    def __init__(self, uri):
        self.clr = None
        try:
            import cooler  # noqa: optional dependency
            self.clr = cooler.Cooler(uri)
            self.binsize = self.clr.binsize
            self.chromnames = list(self.clr.chromnames)
        except ImportError:
            from . import cool
            self.f = cool.CoolFile(uri)
            self.binsize = self.f.binsize
            self.chromnames = list(self.f.chromnames)

    def nbins(self, chrom):
        lo, hi = self.clr.extent(chrom) if self.clr is not None else self.f.extent(chrom)
        return hi - lo

    def close(self):
        f = getattr(self, 'f', None)
        if f is not None:
            f.close()
            self.f = None

    def enable_pool(self):
        """The pixel columns of `fetch_pixels` from a pool of arrays (cool.ArrayPool) from here on -> the callable that hands a
        chromosome's columns back once the band is built (thread-safe), or None (the `cooler`-package backend allocates its own)."""
        f = getattr(self, 'f', None)
        if f is None:
            return None
        from . import cool
        f.pool = cool.ArrayPool()
        return f.pool.give

    def _clr_pixels(self, chrom):
        """the chromosome's pixels through the `cooler` package, every pixel once: a file in storage mode 'square' lists both
        triangles ((i, j) and (j, i)), and the band builders fold them onto one cell - the lower one is dropped, as the
        package's own reader does (cool.CoolFile.pixels)."""
        lo, hi = self.clr.extent(chrom)
        px = self.clr.matrix(balance=False, as_pixels=True, join=False).fetch(chrom)
        i, j, cnt = px['bin1_id'].values - lo, px['bin2_id'].values - lo, px['count'].values
        if getattr(self.clr, 'storage_mode', 'symmetric-upper') == 'square':
            keep = j >= i
            i, j, cnt = i[keep], j[keep], cnt[keep]
        return lo, hi, i, j, cnt

    def fetch_pixels(self, chrom, weight_name='weight'):
        """-> (bin1, bin2, count, n, weight f64 [n], biases f64 [n] or None): the pixel table itself, for the device-side band
        builder (hpk_devband_create); same balancing conventions as `fetch`."""
        if self.clr is not None:
            from . import cool
            lo, hi, i, j, cnt = self._clr_pixels(chrom)
            w = self.clr.bins().fetch(chrom)[weight_name].values.astype(np.float64)
            divisive = weight_name in cool.DIVISIVE_NAMES
        else:
            lo, hi = self.f.extent(chrom)
            i, j, cnt = self.f.pixels(chrom)
            w, divisive = self.f.weights(chrom, weight_name)
        if not divisive:
            return i, j, cnt, hi - lo, w, None
        wm, biases = _divisive(w)
        return i, j, cnt, hi - lo, wm, biases

    def fetch(self, chrom, num, weight_name='weight'):
        """-> (raw f32 [n, num], weight f64 [n], biases f64 [n] or None)"""
        if self.clr is not None:
            from . import cool
            lo, hi, i, j, cnt = self._clr_pixels(chrom)
            col = self.clr.bins().fetch(chrom)[weight_name]
            w = col.values.astype(np.float64)
            divisive = weight_name in cool.DIVISIVE_NAMES
        else:
            lo, hi = self.f.extent(chrom)
            i, j, cnt = self.f.pixels(chrom)
            w, divisive = self.f.weights(chrom, weight_name)
        raw = _band.band_from_coo(i, j, cnt, hi - lo, num, dtype=np.float32)
        if not divisive:
            return raw, w, None
        wm, biases = _divisive(w)
        return raw, wm, biases


def _divisive(w):
    """a divisive weight column -> (the multiplicative weights 1 / column, the biases the reference forms from the column)"""
    with np.errstate(divide='ignore', invalid='ignore'):
        ok = ~((w == 0) | np.isnan(w))
        biases = np.zeros_like(w)
        biases[ok] = 1.0 / w[ok]                    # scripts/pyHICCUPS:163-166 on the column as stored
        wm = np.where(ok, 1.0 / np.where(ok, w, 1.0), np.nan)       # count / (w1 w2) as a product; masked bins stay NaN
    return wm, biases


def open_source(path):
    if str(path).endswith('.npz'):
        return NpzSource(path)
    try:
        return CoolerSource(path)
    except ImportError as e:
        raise SystemExit('reading %s needs the `cooler` package, h5py or libhdf5 (%s); band archives (.npz, see '
                         'hicpeaks_amd.io.save_band_archive) work without them' % (path, e))
