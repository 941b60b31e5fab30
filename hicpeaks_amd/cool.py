"""Minimal reader of the cooler file format (.cool / .mcool, HDF5) - what scripts/pyHICCUPS:142-166 asks of the
`cooler` package, without the package: the pixel table of one chromosome, the bin table's weight column, the bin size
and the chromosome list.

Schema read (cooler format version 3, https://cooler.readthedocs.io/en/latest/schema.html):

    chroms/name [S], chroms/length [i]            one row per chromosome
    bins/chrom [enum -> i], start, end [i]        one row per bin, genome order; bins/<weight> [f8], NaN = masked bin
    pixels/bin1_id, bin2_id [i8], count [i4|f8]   upper triangle, sorted by (bin1_id, bin2_id)
    indexes/chrom_offset [i8]                     first bin of every chromosome (+ total)
    indexes/bin1_offset [i8]                      first pixel of every bin1 (+ total)
    attrs: bin-size

Backends: `h5py` when importable; otherwise the HDF5 C library itself through ctypes (libhdf5.so - on this image under
/opt/conda/lib; HPK_LIBHDF5 names another one).  Neither h5py nor cooler is a dependency of the package.

Balancing conventions (cooler.Cooler.matrix, not under /root/reference and not installed here - SURVEY 8-C2 - so taken from
cooler's documentation, not verified against the package): balanced = count * w[bin1] * w[bin2]; the 4DN-style columns
"KR", "VC", "VC_SQRT" / "SQRT_VC" are *divisive* (balanced = count / (w[bin1] * w[bin2])) - decided by the column's name, as
cooler.matrix(balance=name) decides it; a bin whose divisive weight is 0 is treated as masked (cooler would form count / 0).
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

DIVISIVE_NAMES = ('KR', 'VC', 'VC_SQRT', 'SQRT_VC')      # (cooler's own list spells the last one SQRT_VC; both are taken)


def parse_uri(uri):
    """'file.mcool::/resolutions/10000' -> ('file.mcool', '/resolutions/10000'); a plain path -> (path, '/')."""
    uri = str(uri)
    if '::' in uri:
        path, group = uri.split('::', 1)
        group = '/' + group.strip('/')
        return path, group
    return uri, '/'


# ----------------------------------------------------------------------------- libhdf5 through ctypes
class ArrayPool(object):
    """Result arrays of `read_big` that are handed back (`give`) instead of freed: a chromosome's pixel columns are 0.4-0.8 GB
    each at depth, and an array of that size is a mapping of its own - every chromosome paid first-touch faults for its three
    columns and a `munmap` of as many when they were dropped, under the process' one address-space lock and the GIL (a fifth of
    the command line's wall time on a 10^9-pixel file, profiles/r05_cli_timeline.txt).  Largest-first order: what the first
    chromosomes allocate serves all the others.  Thread-safe (the reader thread takes, the scoring thread gives)."""

    def __init__(self):
        import threading
        self.lock = threading.Lock()
        self.free = {}              # dtype -> [arrays]
        self.owned = {}             # id -> the array itself, for every array this pool made and still keeps track of (a strong
                                    # reference: an id cannot be reused by another object while its array is alive)

    def take(self, dtype, n):
        dtype = np.dtype(dtype)
        with self.lock:
            fl = self.free.setdefault(dtype, [])
            fit = [k for k in range(len(fl)) if fl[k].size >= n]
            if fit:
                a = fl.pop(min(fit, key=lambda k: fl[k].size))
            else:
                if fl:                                  # (none large enough: the smallest one goes instead of piling up)
                    gone = fl.pop(min(range(len(fl)), key=lambda k: fl[k].size))
                    self.owned.pop(id(gone), None)      # ... and is forgotten: freed for good
                a = None
        if a is None:
            a = np.empty(max(int(n), 1), dtype=dtype)
            with self.lock:
                self.owned[id(a)] = a
        return a[:n]

    def give(self, *arrays):
        for v in arrays:
            a = v.base if isinstance(v, np.ndarray) and v.base is not None else v
            if isinstance(a, np.ndarray):
                with self.lock:
                    if self.owned.get(id(a)) is a and not any(a is b for b in self.free.get(a.dtype, [])):
                        self.free.setdefault(a.dtype, []).append(a)


def _h5_locked(fn):
    """libhdf5 is not thread-safe (and ctypes releases the GIL around every call into it): one call sequence at a time, whatever
    the file - `read_big` lets go of the lock while its chunks are decoded, which is when another column's look-ups run."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kw):
        with _H5C._h5lock:
            return fn(self, *args, **kw)
    return wrapper


class _H5C(object):
    """The dozen HDF5 C calls a cooler needs.  Data sets are read as native int64 / float64 / fixed strings (HDF5 converts
    whatever the file holds, enums included), whole or as a [start, stop) slice of the first dimension."""

    _lib = None
    _pool, _pool_n = None, 0            # decoding threads of read_big
    import threading as _threading
    _h5lock = _threading.RLock()

    @classmethod
    def lib(cls):
        if cls._lib is not None:
            return cls._lib
        names = [os.environ.get('HPK_LIBHDF5'), ctypes.util.find_library('hdf5'), 'libhdf5.so', '/opt/conda/lib/libhdf5.so',
                 '/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so']
        last = None
        for nm in names:
            if not nm:
                continue
            try:
                L = C.CDLL(nm)
                break
            except OSError as e:
                last = e
        else:
            raise ImportError('neither h5py nor libhdf5 is available (%s); set HPK_LIBHDF5' % last)
        hid, hs, he = C.c_int64, C.c_uint64, C.c_int
        sig = dict(H5open=(he, []), H5Fopen=(hid, [C.c_char_p, C.c_uint, hid]), H5Fclose=(he, [hid]),
                   H5Gopen2=(hid, [hid, C.c_char_p, hid]), H5Gclose=(he, [hid]),
                   H5Dopen2=(hid, [hid, C.c_char_p, hid]), H5Dclose=(he, [hid]), H5Dget_space=(hid, [hid]),
                   H5Dget_type=(hid, [hid]), H5Dread=(he, [hid, hid, hid, hid, hid, C.c_void_p]),
                   H5Sget_simple_extent_ndims=(C.c_int, [hid]), H5Sget_simple_extent_dims=(C.c_int, [hid, C.POINTER(hs), C.POINTER(hs)]),
                   H5Sselect_hyperslab=(he, [hid, C.c_int, C.POINTER(hs), C.POINTER(hs), C.POINTER(hs), C.POINTER(hs)]),
                   H5Screate_simple=(hid, [C.c_int, C.POINTER(hs), C.POINTER(hs)]), H5Sclose=(he, [hid]),
                   H5Tget_class=(C.c_int, [hid]), H5Tget_size=(C.c_size_t, [hid]), H5Tclose=(he, [hid]), H5Tcopy=(hid, [hid]),
                   H5Tset_size=(he, [hid, C.c_size_t]), H5Tis_variable_str=(C.c_int, [hid]),
                   H5Aexists_by_name=(C.c_int, [hid, C.c_char_p, C.c_char_p, hid]),
                   H5Aopen_by_name=(hid, [hid, C.c_char_p, C.c_char_p, hid, hid]), H5Aread=(he, [hid, hid, C.c_void_p]),
                   H5Aget_type=(hid, [hid]), H5Aclose=(he, [hid]), H5Lexists=(C.c_int, [hid, C.c_char_p, hid]),
                   H5Eset_auto2=(he, [hid, C.c_void_p, C.c_void_p]),
                   H5Dget_create_plist=(hid, [hid]), H5Pclose=(he, [hid]), H5Pget_layout=(C.c_int, [hid]),
                   H5Pget_chunk=(C.c_int, [hid, C.c_int, C.POINTER(hs)]), H5Pget_nfilters=(C.c_int, [hid]),
                   H5Pget_filter2=(C.c_int, [hid, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_size_t), C.POINTER(C.c_uint),
                                             C.c_size_t, C.c_char_p, C.POINTER(C.c_uint)]),
                   H5Tget_sign=(C.c_int, [hid]), H5Tget_order=(C.c_int, [hid]), H5free_memory=(he, [C.c_void_p]),
                   H5Tget_cset=(C.c_int, [hid]), H5Tset_cset=(he, [hid, C.c_int]))
        # raw chunk access (HDF5 >= 1.10.5): lets the gzip streams of a big read be inflated on several threads
        cls.have_chunks = all(hasattr(L, f) for f in ('H5Dget_chunk_info_by_coord', 'H5Dread_chunk'))
        if cls.have_chunks:
            L.H5Dget_chunk_info_by_coord.restype = he
            L.H5Dget_chunk_info_by_coord.argtypes = [hid, C.POINTER(hs), C.POINTER(C.c_uint), C.POINTER(hs), C.POINTER(hs)]
            L.H5Dread_chunk.restype = he
            L.H5Dread_chunk.argtypes = [hid, hid, C.POINTER(hs), C.POINTER(C.c_uint32), C.c_void_p]
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        if L.H5open() < 0:
            raise ImportError('H5open failed')
        L.H5Eset_auto2(0, None, None)               # no error stack printing: failures become Python exceptions here
        cls.T_I64 = C.c_int64.in_dll(L, 'H5T_NATIVE_INT64_g').value
        cls.T_F64 = C.c_int64.in_dll(L, 'H5T_NATIVE_DOUBLE_g').value
        cls.T_C_S1 = C.c_int64.in_dll(L, 'H5T_C_S1_g').value
        cls._lib = L
        return L

    def __init__(self, path, group):
        L = self.lib()
        self.f = L.H5Fopen(str(path).encode(), 0, 0)            # H5F_ACC_RDONLY, H5P_DEFAULT
        if self.f < 0:
            raise IOError('cannot open %s as an HDF5 file' % path)
        self.path = str(path)
        self.fd = -1                    # a descriptor of our own on the file (read_big: the chunks are pread by the decoding threads)
        self.fd_ok = None               # ... once a chunk read that way has been seen to equal H5Dread_chunk's (None: not checked yet)
        self.g = L.H5Gopen2(self.f, group.encode(), 0)
        if self.g < 0:
            L.H5Fclose(self.f)
            raise IOError('no group %s in %s' % (group, path))

    def close(self):
        L = self.lib()
        if getattr(self, 'g', -1) >= 0:
            L.H5Gclose(self.g)
            self.g = -1
        if getattr(self, 'f', -1) >= 0:
            L.H5Fclose(self.f)
            self.f = -1
        if getattr(self, 'fd', -1) >= 0:
            os.close(self.fd)
            self.fd = -1

    @_h5_locked
    def exists(self, name):
        L = self.lib()
        cur = ''
        for part in name.strip('/').split('/'):             # H5Lexists wants every intermediate link to exist
            cur = cur + '/' + part if cur else part
            if L.H5Lexists(self.g, cur.encode(), 0) <= 0:
                return False
        return True

    def _open(self, name):
        d = self.lib().H5Dopen2(self.g, name.encode(), 0)
        if d < 0:
            raise KeyError(name)
        return d

    @_h5_locked
    def shape(self, name):
        L = self.lib()
        d = self._open(name)
        sp = L.H5Dget_space(d)
        dims = (C.c_uint64 * 4)()
        nd = L.H5Sget_simple_extent_dims(sp, dims, None)
        L.H5Sclose(sp)
        L.H5Dclose(d)
        return tuple(int(dims[i]) for i in range(nd))

    @_h5_locked
    def read(self, name, start=None, stop=None, kind=None):
        """kind: 'i' -> int64, 'f' -> float64, 's' -> list of str, None -> by the data set's class"""
        L = self.lib()
        d = self._open(name)
        try:
            ft = L.H5Dget_type(d)
            cls, size = L.H5Tget_class(ft), L.H5Tget_size(ft)
            varstr = cls == 3 and L.H5Tis_variable_str(ft) > 0
            cset = L.H5Tget_cset(ft) if cls == 3 else 0          # (ASCII | UTF-8: HDF5 does not convert between the two)
            L.H5Tclose(ft)
            if kind is None:
                kind = {0: 'i', 1: 'f', 3: 's', 8: 'i'}.get(cls)        # H5T_INTEGER, FLOAT, STRING, ENUM
            if kind is None:
                raise TypeError('%s: unsupported HDF5 type class %d' % (name, cls))
            sp = L.H5Dget_space(d)
            dims = (C.c_uint64 * 4)()
            nd = L.H5Sget_simple_extent_dims(sp, dims, None)
            if nd != 1:
                L.H5Sclose(sp)
                raise TypeError('%s: one-dimensional data sets only' % name)
            n = int(dims[0])
            a = 0 if start is None else max(0, min(int(start), n))
            b = n if stop is None else max(a, min(int(stop), n))
            cnt = b - a
            if kind == 's' and varstr:          # variable-length strings (h5py's default for str data): an array of char*
                mt = L.H5Tcopy(self.T_C_S1)
                L.H5Tset_size(mt, C.c_size_t(-1).value)                 # H5T_VARIABLE
                L.H5Tset_cset(mt, cset)
                out = np.zeros(cnt, dtype=np.uint64)
            elif kind == 's':
                mt = L.H5Tcopy(self.T_C_S1)
                L.H5Tset_size(mt, size)
                L.H5Tset_cset(mt, cset)
                out = np.zeros(cnt, dtype='S%d' % size)
            else:
                mt = self.T_I64 if kind == 'i' else self.T_F64
                out = np.empty(cnt, dtype=np.int64 if kind == 'i' else np.float64)
            if cnt:
                st, ct = (C.c_uint64 * 1)(a), (C.c_uint64 * 1)(cnt)
                L.H5Sselect_hyperslab(sp, 0, st, None, ct, None)       # H5S_SELECT_SET
                ms = L.H5Screate_simple(1, ct, None)
                rc = L.H5Dread(d, mt, ms, sp, 0, out.ctypes.data)
                L.H5Sclose(ms)
                if rc < 0:
                    raise IOError('H5Dread failed on %s (a compression filter this HDF5 library does not have? cooler itself writes gzip)' % name)
            L.H5Sclose(sp)
            if kind == 's':
                L.H5Tclose(mt)
                if varstr:
                    strs = [C.string_at(int(p)).decode() if p else '' for p in out]
                    for p in out:
                        if p:
                            L.H5free_memory(C.c_void_p(int(p)))
                    return strs
                return [v.decode() for v in out]
            return out
        finally:
            L.H5Dclose(d)

    PARALLEL_MIN = 1 << 20          # elements from which a read goes chunk by chunk (below: one H5Dread)

    @staticmethod
    def _minus(a, bias):
        if bias:
            a -= bias
        return a

    @_h5_locked
    def read_big(self, name, start, stop, kind, threads=None, bias=0, pool=None):
        """(kind 'i4': an integer column as int32 where its stored type fits - the counts; `pool`: result arrays from an ArrayPool)
        `read` for the long slices of the pixel table: the data set's chunks are fetched as stored (H5Dread_chunk: the
        library itself is not thread-safe, so this part stays serial - it is a copy out of the page cache) and inflated /
        un-shuffled / widened by libhpk's host threads (hpk_decode_chunks; HPK_READ_PYTHON=1 or no library: a Python thread pool
        - zlib and numpy release the GIL).  cooler writes its columns gzip-compressed with the shuffle filter in chunks; the
        decoding is 90 % of what reading a chromosome costs.  Anything unexpected - another filter, big-endian data, an old
        libhdf5 - falls back to `read`."""
        import zlib
        from concurrent.futures import ThreadPoolExecutor
        L = self.lib()
        want32 = kind == 'i4'
        if want32:
            kind = None                 # (a float column stays what it is)
        if not self.have_chunks or stop - start < self.PARALLEL_MIN:
            return self._minus(self.read(name, start, stop, kind), bias)
        d = self._open(name)
        try:
            pl = L.H5Dget_create_plist(d)
            cs = (C.c_uint64 * 4)()
            ok = pl >= 0 and L.H5Pget_layout(pl) == 2 and L.H5Pget_chunk(pl, 4, cs) == 1        # H5D_CHUNKED, rank 1
            filt = []
            if ok:
                for i in range(L.H5Pget_nfilters(pl)):
                    flags, ne, cfg = C.c_uint(0), C.c_size_t(0), C.c_uint(0)
                    filt.append(L.H5Pget_filter2(pl, i, C.byref(flags), C.byref(ne), None, 0, None, C.byref(cfg)))
            if pl >= 0:
                L.H5Pclose(pl)
            ft = L.H5Dget_type(d)
            tcls, size = L.H5Tget_class(ft), int(L.H5Tget_size(ft))
            signed = L.H5Tget_sign(ft) != 0 if tcls in (0, 8) else True
            little = L.H5Tget_order(ft) == 0
            L.H5Tclose(ft)
            ok = ok and filt in ([1], [2, 1]) and little and tcls in (0, 1, 8) and size in (1, 2, 4, 8)     # deflate | shuffle + deflate
            if not ok:
                return self._minus(self.read(name, start, stop, kind), bias)
            if kind is None:
                kind = 'f' if tcls == 1 else 'i'
            dt = np.dtype(('<f%d' % size) if tcls == 1 else ('<%s%d' % ('i' if signed else 'u', size)))
            cs = int(cs[0])
            shuffle = filt[0] == 2
            # Where the chunks are (H5Dget_chunk_info_by_coord: the library is not thread-safe, this loop stays serial - it reads
            # the chunk index only).  The chunks themselves are then pread, inflated, un-shuffled and widened by libhpk's host threads
            # straight into the result (hpk_decode_chunks_fd: no Python object, no allocation per chunk); the first chunk a file is
            # read this way is compared with H5Dread_chunk's copy once (a user block or a driver that moves the addresses would show).
            c0, c1 = start // cs, (stop - 1) // cs + 1
            lens, addrs = [], []
            for ci in range(c0, c1):
                off = (C.c_uint64 * 1)(ci * cs)
                mask, addr, nbytes = C.c_uint(0), C.c_uint64(0), C.c_uint64(0)
                if L.H5Dget_chunk_info_by_coord(d, off, C.byref(mask), C.byref(addr), C.byref(nbytes)) < 0 or mask.value != 0 \
                        or nbytes.value == 0:
                    return self._minus(self.read(name, start, stop, kind), bias)
                lens.append(int(nbytes.value))
                addrs.append(int(addr.value))
            # (16 per column, three columns side by side: 2.1 s end to end on the 10^9-pixel file against 2.35 s with 64 - profiles/r05_host_e2e_deep.txt)
            nthr = threads or int(os.environ.get('HPK_READ_THREADS', 0)) or min(16, os.cpu_count() or 1)
            lib = None
            if not os.environ.get('HPK_READ_PYTHON'):
                try:
                    from . import _lib
                    lib = _lib.load()
                except Exception:          # (the reader also serves hosts where the library is not built: the Python pool below)
                    lib = None
            kd = 2 if tcls == 1 else (0 if signed else 1)
            want32 = want32 and lib is not None and kind == 'i' and tcls != 1 and size <= (4 if signed else 2)
            odt = np.int32 if want32 else (np.int64 if kind == 'i' else np.float64)
            omode = 2 if want32 else (0 if kind == 'i' else 1)
            out = pool.take(odt, stop - start) if pool is not None else np.empty(stop - start, dtype=odt)

            def read_chunk(k):
                buf = np.empty(lens[k], dtype=np.uint8)
                fm = C.c_uint32(0)
                if L.H5Dread_chunk(d, 0, (C.c_uint64 * 1)((c0 + k) * cs), C.byref(fm), C.c_void_p(buf.ctypes.data)) < 0 or fm.value != 0:
                    return None
                return buf

            if lib is not None and not os.environ.get('HPK_READ_NO_PREAD') and self.fd_ok is not False:
                if self.fd < 0:
                    try:
                        self.fd = os.open(self.path, os.O_RDONLY)
                    except OSError:
                        self.fd_ok = False
                if self.fd >= 0 and self.fd_ok is None:
                    ref = read_chunk(0)
                    self.fd_ok = ref is not None and os.pread(self.fd, lens[0], addrs[0]) == ref.tobytes()
                if self.fd_ok:
                    self._h5lock.release()          # (no HDF5 call in here: the chunks are pread and decoded by libhpk's threads)
                    try:
                        rc = lib.hpk_decode_chunks_fd(self.fd, (C.c_uint64 * len(addrs))(*addrs), (C.c_uint64 * len(lens))(*lens), len(lens), c0, cs,
                                                      size, kd, 1 if shuffle else 0, start, stop, out.ctypes.data, omode, int(bias), nthr)
                    finally:
                        self._h5lock.acquire()
                    if rc == 0:
                        return out
                    if pool is not None:
                        pool.give(out)
                    return self._minus(self.read(name, start, stop, kind), bias)
            # the chunks as stored, into one arena (H5Dread_chunk, serial: a copy out of the page cache)
            arena = np.empty(sum(lens) + 64, dtype=np.uint8)
            base, pos, offs = arena.ctypes.data, 0, []
            for k in range(len(lens)):
                fm = C.c_uint32(0)
                if L.H5Dread_chunk(d, 0, (C.c_uint64 * 1)((c0 + k) * cs), C.byref(fm), C.c_void_p(base + pos)) < 0 or fm.value != 0:
                    return self._minus(self.read(name, start, stop, kind), bias)
                offs.append(base + pos)
                pos += lens[k]
        finally:
            L.H5Dclose(d)
        if lib is not None:
            self._h5lock.release()
            try:
                rc = lib.hpk_decode_chunks((C.c_void_p * len(offs))(*offs), (C.c_uint64 * len(lens))(*lens), len(offs), c0, cs, size, kd,
                                           1 if shuffle else 0, start, stop, out.ctypes.data, omode, int(bias), nthr)
            finally:
                self._h5lock.acquire()
            if rc == 0:
                return out
            if pool is not None:
                pool.give(out)
            return self._minus(self.read(name, start, stop, kind), bias)
        raws = [(c0 + k, arena[o - base:o - base + n_]) for k, (o, n_) in enumerate(zip(offs, lens))]

        def decode(item):
            ci, buf = item
            b = zlib.decompress(buf)
            a = np.frombuffer(b, dtype=np.uint8)
            if shuffle and size > 1:
                a = np.ascontiguousarray(a.reshape(size, -1).T).reshape(-1)
            v = a.view(dt)
            lo, hi = max(start, ci * cs), min(stop, ci * cs + cs)
            out[lo - start:hi - start] = v[lo - ci * cs:hi - ci * cs]

        if nthr > 1 and len(raws) > 1:
            if _H5C._pool is None or _H5C._pool_n != nthr:
                _H5C._pool, _H5C._pool_n = ThreadPoolExecutor(nthr), nthr
            list(_H5C._pool.map(decode, raws))
        else:
            for it in raws:
                decode(it)
        return self._minus(out, bias)

    @_h5_locked
    def attr(self, obj, name, default=None):
        """Scalar integer / float / boolean-like attribute of the group (obj = '.') or of a data set."""
        L = self.lib()
        if L.H5Aexists_by_name(self.g, obj.encode(), name.encode(), 0) <= 0:
            return default
        a = L.H5Aopen_by_name(self.g, obj.encode(), name.encode(), 0, 0)
        if a < 0:
            return default
        try:
            t = L.H5Aget_type(a)
            cls = L.H5Tget_class(t)
            L.H5Tclose(t)
            if cls in (0, 8):
                v = C.c_int64(0)
                return int(v.value) if L.H5Aread(a, self.T_I64, C.byref(v)) >= 0 else default
            if cls == 1:
                v = C.c_double(0)
                return float(v.value) if L.H5Aread(a, self.T_F64, C.byref(v)) >= 0 else default
            if cls == 3:                            # string (fixed or variable length)
                t = L.H5Aget_type(a)
                var, size, cset = L.H5Tis_variable_str(t) > 0, int(L.H5Tget_size(t)), L.H5Tget_cset(t)
                L.H5Tclose(t)
                mt = L.H5Tcopy(self.T_C_S1)
                L.H5Tset_cset(mt, cset)
                try:
                    if var:
                        L.H5Tset_size(mt, C.c_size_t(-1).value)
                        p = C.c_void_p(0)
                        if L.H5Aread(a, mt, C.byref(p)) < 0 or not p.value:
                            return default
                        v = C.string_at(p.value).decode()
                        L.H5free_memory(p)
                        return v
                    L.H5Tset_size(mt, size + 1)
                    buf = C.create_string_buffer(size + 1)
                    return buf.value.decode() if L.H5Aread(a, mt, buf) >= 0 else default
                finally:
                    L.H5Tclose(mt)
            return default
        finally:
            L.H5Aclose(a)


class _H5Py(object):
    def __init__(self, path, group):
        import h5py
        self.f = h5py.File(path, 'r')
        self.g = self.f[group]

    def close(self):
        self.f.close()

    def exists(self, name):
        return name in self.g

    def shape(self, name):
        return tuple(self.g[name].shape)

    def read(self, name, start=None, stop=None, kind=None):
        d = self.g[name]
        v = d[slice(start, stop)]
        if v.dtype.kind in 'SO':
            return [x.decode() if isinstance(x, bytes) else str(x) for x in v]
        return v.astype(np.float64 if (kind == 'f' or (kind is None and v.dtype.kind == 'f')) else np.int64)

    def read_big(self, name, start, stop, kind, threads=None, bias=0, pool=None):
        v = self.read(name, start, stop, None if kind == 'i4' else kind)
        if bias:
            v -= bias
        return v

    def attr(self, obj, name, default=None):
        o = self.g if obj == '.' else self.g[obj]
        if name not in o.attrs:
            return default
        v = o.attrs[name]
        return v.item() if hasattr(v, 'item') else v


def _backend(path, group):
    try:
        import h5py  # noqa: F401
        return _H5Py(path, group)
    except ImportError:
        return _H5C(path, group)


# ----------------------------------------------------------------------------- the cooler itself
class CoolFile(object):
    """One cooler (a .cool file, or one resolution of an .mcool: 'file.mcool::/resolutions/5000')."""

    def __init__(self, uri):
        self.path, self.group = parse_uri(uri)
        self.h = _backend(self.path, self.group)
        for need in ('chroms/name', 'bins/chrom', 'pixels/bin1_id', 'indexes/chrom_offset', 'indexes/bin1_offset'):
            if not self.h.exists(need):
                self.h.close()
                raise IOError('%s%s is not a cooler: %s is missing' % (self.path, '::' + self.group if self.group != '/' else '', need))
        self.binsize = self.h.attr('.', 'bin-size')
        if self.binsize is None:
            self.h.close()
            raise IOError('cooler without a fixed bin-size (variable-size bins are not supported)')
        self.binsize = int(self.binsize)
        # 'symmetric-upper' (cooler's default; also assumed when the attribute is missing): the pixel table lists every contact
        # once, bin1 <= bin2.  'square': both triangles are stored - the lower one is left out (cooler.matrix hands the matrix
        # over as stored, and scripts/pyHICCUPS:147 takes its upper diagonals)
        mode = self.h.attr('.', 'storage-mode')
        mode = mode.decode() if isinstance(mode, bytes) else mode
        self.square = isinstance(mode, str) and mode.lower() == 'square'
        self.chromnames = self.h.read('chroms/name', kind='s')
        self.chrom_offset = self.h.read('indexes/chrom_offset', kind='i')
        self._cid = {c: i for i, c in enumerate(self.chromnames)}
        self.pool = None                # an ArrayPool: `pixels` takes its arrays from it, `release` hands them back
        self._colpool = None            # three threads, one per pixel column of a long read

    def close(self):
        if self._colpool is not None:
            self._colpool.shutdown(wait=True)
            self._colpool = None
        self.h.close()

    def extent(self, chrom):
        i = self._cid[chrom]
        return int(self.chrom_offset[i]), int(self.chrom_offset[i + 1])

    def weights(self, chrom, name='weight'):
        """(weight f64 [n] of the chromosome's bins, divisive) - the column as stored, NaN = masked bin"""
        lo, hi = self.extent(chrom)
        col = 'bins/' + name
        if not self.h.exists(col):
            raise KeyError('no bin column %r (balance the cooler first, or pick --clr-weight-name)' % name)
        w = self.h.read(col, lo, hi, kind='f')
        # by the column's NAME, as cooler.Cooler.matrix(balance=name) decides it - the call the reference makes
        # (scripts/pyHICCUPS:143) - and as the `cooler`-package backend of hicpeaks_amd.io decides it: one rule whatever is
        # installed.  (A `divisive_weights` attribute on the column is not consulted: cooler.matrix does not read it either.)
        return w, name in DIVISIVE_NAMES

    def pixels(self, chrom):
        """The intra-chromosomal pixels of `chrom`, bins relative to its first bin: (bin1 i8, bin2 i8, count) - what
        `clr.matrix(balance=False, as_pixels=True, join=False).fetch(chrom)` holds (scripts/pyHICCUPS:142)."""
        lo, hi = self.extent(chrom)
        off = self.h.read('indexes/bin1_offset', lo, hi + 1, kind='i')
        p0, p1 = int(off[0]), int(off[-1])
        # (bin ids relative to the chromosome's first bin as they are decoded: no second pass over 800 MB arrays)
        # (the counts as int32 where the file stores them so - what the band builders take; the arrays from the pool, if there is one)
        pool = self.pool
        cols = (('pixels/bin1_id', 'i', lo), ('pixels/bin2_id', 'i', lo), ('pixels/count', 'i4', 0))
        if isinstance(self.h, _H5C) and p1 - p0 >= _H5C.PARALLEL_MIN and not os.environ.get('HPK_READ_SERIAL'):
            # the three columns at once: a column is a few hundred chunks, i.e. a handful per decoding thread - more threads per
            # column buy nothing, three columns side by side do (the HDF5 look-ups of one run while the others are decoded)
            if self._colpool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._colpool = ThreadPoolExecutor(3, thread_name_prefix='hpk-column')
            futs = [self._colpool.submit(self.h.read_big, nm, p0, p1, kd, bias=bs, pool=pool) for nm, kd, bs in cols]
            b1, b2, cnt = [f.result() for f in futs]
        else:
            b1, b2, cnt = [self.h.read_big(nm, p0, p1, kd, bias=bs, pool=pool) for nm, kd, bs in cols]
        # pixels are sorted by bin1: a row's trans pixels have bin2 beyond the chromosome (and, both triangles stored, before it).
        # Long tables: counted and moved by libhpk's host threads (hpk_compact_pixels); short ones, or no library: numpy.
        full = (b1, b2, cnt)
        done = False
        if b1.size >= _H5C.PARALLEL_MIN and b1.dtype == np.int64 and b2.dtype == np.int64 and cnt.dtype.itemsize in (4, 8) \
                and all(a.flags.c_contiguous for a in full) and not os.environ.get('HPK_READ_PYTHON'):
            try:
                from . import _lib
                lib = _lib.load()
            except Exception:
                lib = None
            if lib is not None:
                nthr = int(os.environ.get('HPK_READ_THREADS', 0)) or min(48, os.cpu_count() or 1)
                take = (lambda dt, n: pool.take(dt, n)) if pool is not None else (lambda dt, n: np.empty(n, dtype=dt))
                kept = lib.hpk_compact_pixels(b1.ctypes.data, b2.ctypes.data, cnt.ctypes.data, cnt.dtype.itemsize, b1.size, hi - lo,
                                              1 if self.square else 0, None, None, None, nthr)      # (count only: nothing to write to)
                if kept == b1.size:
                    done = True
                elif kept >= 0:
                    o1, o2, oc = take(np.int64, kept), take(np.int64, kept), take(cnt.dtype, kept)
                    if lib.hpk_compact_pixels(b1.ctypes.data, b2.ctypes.data, cnt.ctypes.data, cnt.dtype.itemsize, b1.size, hi - lo,
                                              1 if self.square else 0, o1.ctypes.data, o2.ctypes.data, oc.ctypes.data, nthr) == kept:
                        b1, b2, cnt = o1, o2, oc
                        done = True
        if not done:
            keep = (b2 < hi - lo) & (b2 >= 0)
            if self.square:
                keep &= b2 >= b1
            if not keep.all():
                b1, b2, cnt = b1[keep], b2[keep], cnt[keep]
        if pool is not None and b1 is not full[0]:
            pool.give(*full)
        return b1, b2, cnt

    def release(self, *arrays):
        """Hands result arrays of `pixels` back for the next chromosome (ArrayPool); the caller no longer touches them."""
        if self.pool is not None:
            self.pool.give(*arrays)
