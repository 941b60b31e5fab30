"""ctypes binding of libhpk.so (include/hpk.h).

The library is the only compute path: if it cannot be loaded, or no gfx950 device is usable, the
callers raise `HpkError` - there is no CPU fallback in this package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('HPK_LIB') or os.path.join(HERE, 'libhpk.so')      # HPK_LIB: profiling builds (scripts/)
CSRC = os.path.join(HERE, 'csrc')

HPK_MAX_PAIRS = 8
HPK_MAX_W = 20
HPK_MAX_STEPS = 64
HPK_NB = 128
HPK_MAX_BATCH = 256

MODE_HICCUPS = 0
MODE_BHFDR = 1
FLAG_DENSE_E = 1
FLAG_DENSE_SUMS = 2
FLAG_NO_SCORE = 4
FLAG_PHASE_TIMING = 8
FLAG_NO_STENCIL_TIMING = 16

HPK_OK = 0
ERR_INVALID, ERR_HIP, ERR_NO_DEVICE, ERR_EMPTY_STEP, ERR_PLAN, ERR_NOMEM, ERR_BUSY = -1, -2, -3, -4, -5, -6, -7

# every symbol include/hpk.h declares (tests check the built library exports all of them)
ABI_SYMBOLS = ['hpk_create', 'hpk_destroy', 'hpk_last_error', 'hpk_abi_version', 'hpk_score_band',
               'hpk_pipeline_depth', 'hpk_submit_band', 'hpk_collect', 'hpk_submit_batch', 'hpk_collect_batch', 'hpk_set_option',
               'hpk_result_free', 'hpk_plan_rings', 'hpk_chunk_bounds', 'hpk_set_chunk_bounds',
               'hpk_device_info', 'hpk_poisson_sf', 'hpk_bruteforce_sums', 'hpk_probe_sums', 'hpk_band_from_coo',
               'hpk_devband_create', 'hpk_devband_free', 'hpk_decode_chunks', 'hpk_decode_chunks_fd', 'hpk_compact_pixels']


class HpkError(RuntimeError):
    def __init__(self, status, msg):
        RuntimeError.__init__(self, 'libhpk status %d: %s' % (status, msg))
        self.status = status
        self.msg = msg

    def __reduce__(self):       # picklable: a Pool worker that raises must not hang the parent's result handler
        return (type(self), (self.status, self.msg))


class EmptyStepError(HpkError, ValueError, ZeroDivisionError):
    """A widening step was entered with no unresolved candidate.  The reference raises at this point
    (ValueError from scipy 1.7 fancy indexing, ZeroDivisionError with newer scipy; hicpeaks/callers.py:203-208,
    487-492), so the drop-in raises too; the class derives from both."""


class Params(C.Structure):
    _fields_ = [('mode', C.c_int32), ('npairs', C.c_int32), ('pw', C.c_int32 * HPK_MAX_PAIRS),
                ('ww', C.c_int32 * HPK_MAX_PAIRS), ('maxww', C.c_int32), ('min_local_reads', C.c_int32),
                ('maxapart', C.c_int64), ('res', C.c_int64), ('sig', C.c_double), ('flags', C.c_int32),
                ('reserved', C.c_int32)]


class Band(C.Structure):
    _fields_ = [('n', C.c_int32), ('num', C.c_int32), ('ld', C.c_int64), ('raw', C.c_void_p),
                ('balanced', C.c_void_p), ('weight', C.c_void_p), ('IR', C.c_void_p), ('bias1', C.c_void_p),
                ('bias2', C.c_void_p), ('on_device', C.c_int32), ('reserved', C.c_int32)]


class Set(C.Structure):
    _fields_ = [('pair', C.c_int32), ('fl', C.c_int32), ('nvalid', C.c_int64), ('numbin', C.c_int32),
                ('reserved', C.c_int32), ('emax', C.c_double), ('begin', C.c_int64), ('end', C.c_int64),
                ('chunk_tests', C.POINTER(C.c_uint32)), ('chunk_below', C.POINTER(C.c_uint32))]


class Result(C.Structure):
    _fields_ = [('nsteps', C.c_int32), ('step_pi', C.c_int32 * HPK_MAX_STEPS), ('step_wi', C.c_int32 * HPK_MAX_STEPS),
                ('step_executed', C.c_int32 * HPK_MAX_STEPS), ('step_resolved', C.c_int64 * HPK_MAX_STEPS),
                ('frozen_w', C.c_int32), ('nslots', C.c_int32), ('slot_pi', C.c_int32 * HPK_MAX_PAIRS),
                ('ncand', C.c_int64), ('nsets', C.c_int32), ('sets', Set * (2 * HPK_MAX_PAIRS)),
                ('nsig', C.c_int64), ('x', C.POINTER(C.c_int32)), ('y', C.POINTER(C.c_int32)),
                ('O', C.POINTER(C.c_double)), ('bal', C.POINTER(C.c_double)), ('E', C.POINTER(C.c_double)),
                ('p', C.POINTER(C.c_double)), ('q', C.POINTER(C.c_double)), ('other_zero', C.POINTER(C.c_uint8)),
                ('gap', C.POINTER(C.c_uint8)), ('dense_ld', C.c_int64), ('dense_E', C.POINTER(C.c_double)),
                ('dense_w', C.POINTER(C.c_uint8)), ('dense_sums', C.POINTER(C.c_double)),
                ('ms_h2d', C.c_float), ('ms_stencil', C.c_float), ('ms_freeze', C.c_float), ('ms_score', C.c_float),
                ('ms_tighten', C.c_float), ('ms_gap', C.c_float), ('ms_d2h', C.c_float), ('ms_host_bh', C.c_float), ('ms_total', C.c_float),
                ('stencil_kernel', C.c_int32), ('record_bound', C.c_int32), ('redone', C.c_int32), ('nsurv_sig', C.c_int64), ('nsurv_cut', C.c_int64),
                ('stencil_tiles', C.c_int64), ('band_px', C.c_int64), ('batch_bands', C.c_int32), ('halo_w', C.c_int32),
                ('lean_tiles', C.c_int32), ('lean_redone', C.c_int32), ('lean_explicit', C.c_int64)]


def build(force=False, quiet=True):
    """Compile libhpk.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'hpk.h')]
    if not force and os.path.exists(LIB_PATH):
        newest = max(os.path.getmtime(s) for s in srcs)
        if os.path.getmtime(LIB_PATH) >= newest:
            return LIB_PATH
    # force: -B, make's own time stamps must not answer "does it build" either
    out = subprocess.run(['make'] + (['-B'] if force else []) + ['-C', CSRC], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if out.returncode != 0:
        raise RuntimeError('building libhpk.so failed:\n' + out.stdout)
    if not quiet:
        print(out.stdout)
    return LIB_PATH


_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch's ROCm wheels bundle their own libamdhip64.so (soname libamdhip64.so.7, found through
    an RPATH under its file name), libhpk.so asks for libamdhip64.so.7: whoever comes second gets a second runtime beside the
    first, and the second one no longer finds the device - `import hicpeaks_amd` before `import torch` used to break torch (round
    5: handled in tests/conftest.py only).  When torch is installed its bundled runtime is therefore loaded here, globally, before
    libhpk.so: libhpk's soname request and torch's later load by path both resolve to that one copy - the same arrangement as
    when torch is imported first.  No torch (the command lines need none): the system's ROCm as linked.  HPK_NO_TORCH_HIP=1: leave
    the loader alone."""
    import sys
    if os.environ.get('HPK_NO_TORCH_HIP') or 'torch' in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec('torch')
        if spec is None or not spec.origin:
            return
        libdir = os.path.join(os.path.dirname(spec.origin), 'lib')
        hip = os.path.join(libdir, 'libamdhip64.so')
        if os.path.exists(hip):
            C.CDLL(hip, mode=C.RTLD_GLOBAL)
    except Exception:       # (a torch that cannot be located changes nothing: the library loads as linked)
        pass


def load():
    """Load libhpk.so (never builds; `__graft_entry__.build()` / `make -C hicpeaks_amd/csrc` does)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HpkError(ERR_NO_DEVICE, 'libhpk.so is not built (%s); run `make -C hicpeaks_amd/csrc` - there is '
                       'no CPU fallback' % LIB_PATH)
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    lib.hpk_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.hpk_create.restype = C.c_int
    lib.hpk_destroy.argtypes = [C.c_void_p]
    lib.hpk_destroy.restype = None
    lib.hpk_last_error.argtypes = [C.c_void_p]
    lib.hpk_last_error.restype = C.c_char_p
    lib.hpk_abi_version.restype = C.c_int
    lib.hpk_score_band.argtypes = [C.c_void_p, C.POINTER(Band), C.POINTER(Params), C.POINTER(C.POINTER(Result))]
    lib.hpk_score_band.restype = C.c_int
    lib.hpk_result_free.argtypes = [C.POINTER(Result)]
    lib.hpk_result_free.restype = None
    lib.hpk_pipeline_depth.restype = C.c_int
    lib.hpk_submit_band.argtypes = [C.c_void_p, C.POINTER(Band), C.POINTER(Params), C.POINTER(C.c_void_p)]
    lib.hpk_submit_band.restype = C.c_int
    lib.hpk_collect.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.POINTER(Result))]
    lib.hpk_collect.restype = C.c_int
    lib.hpk_submit_batch.argtypes = [C.c_void_p, C.POINTER(Band), C.c_int32, C.POINTER(Params), C.POINTER(C.c_void_p)]
    lib.hpk_submit_batch.restype = C.c_int
    lib.hpk_collect_batch.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.POINTER(Result)), C.POINTER(C.c_int32), C.c_char_p, C.c_int32]
    lib.hpk_collect_batch.restype = C.c_int
    lib.hpk_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    lib.hpk_set_option.restype = C.c_int
    lib.hpk_plan_rings.argtypes = [C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hpk_plan_rings.restype = C.c_int
    lib.hpk_chunk_bounds.argtypes = [C.c_void_p, C.c_int32]
    lib.hpk_chunk_bounds.restype = C.c_int
    lib.hpk_set_chunk_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    lib.hpk_set_chunk_bounds.restype = C.c_int
    lib.hpk_device_info.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    lib.hpk_device_info.restype = C.c_int
    lib.hpk_poisson_sf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    lib.hpk_poisson_sf.restype = C.c_int
    lib.hpk_bruteforce_sums.argtypes = [C.c_void_p, C.POINTER(Band), C.POINTER(Params), C.c_int32, C.c_void_p,
                                        C.c_void_p, C.c_int64, C.c_void_p]
    lib.hpk_bruteforce_sums.restype = C.c_int
    lib.hpk_probe_sums.argtypes = [C.c_void_p, C.POINTER(Band), C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_int64,
                                   C.c_void_p]
    lib.hpk_probe_sums.restype = C.c_int
    lib.hpk_band_from_coo.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                                      C.c_int64, C.c_void_p]
    lib.hpk_band_from_coo.restype = C.c_int64
    lib.hpk_decode_chunks.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_int64, C.c_int32]
    lib.hpk_decode_chunks.restype = C.c_int
    lib.hpk_decode_chunks_fd.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_int64, C.c_int32]
    lib.hpk_decode_chunks_fd.restype = C.c_int
    lib.hpk_compact_pixels.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int32]
    lib.hpk_compact_pixels.restype = C.c_int64
    lib.hpk_devband_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(Band)]
    lib.hpk_devband_create.restype = C.c_int64
    lib.hpk_devband_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.hpk_devband_free.restype = None
    _lib = lib
    return lib


def make_params(mode, pw, ww, maxww, sig, maxapart, res, min_local_reads=16, flags=0):
    pw = [int(v) for v in (pw if np.ndim(pw) else [pw])]
    ww = [int(v) for v in (ww if np.ndim(ww) else [ww])]
    if len(pw) != len(ww):
        # The reference zips the lists for the pairs (callers.py:18, 239) but takes min(ww) / min(pw) over the full
        # lists (callers.py:58, 102, 197, 294); the two readings differ when the surplus entries hold the minimum.
        # Rather than guess, refuse what cannot be reproduced.
        npairs = min(len(pw), len(ww))
        if npairs and (min(ww) != min(ww[:npairs]) or min(pw) != min(pw[:npairs])):
            raise HpkError(ERR_INVALID, 'pw and ww have different lengths and the unpaired entries hold the minimum '
                           '(pw=%r, ww=%r): not supported' % (pw, ww))
        pw, ww = pw[:npairs], ww[:npairs]
    if not pw:
        raise ValueError('pw / ww are empty')
    if len(pw) > HPK_MAX_PAIRS:
        raise HpkError(ERR_INVALID, 'at most %d (pw, ww) pairs' % HPK_MAX_PAIRS)
    p = Params()
    p.mode = mode
    p.npairs = len(pw)
    for i, (a, b) in enumerate(zip(pw, ww)):
        p.pw[i] = a
        p.ww[i] = b
    p.maxww = int(maxww)
    p.min_local_reads = int(np.ceil(min_local_reads))
    p.maxapart = int(maxapart)
    p.res = int(res)
    p.sig = float(sig)
    p.flags = int(flags)
    return p


def plan_rings(params):
    """(steps [(pi, wi)], mult_K [nsteps, W+1], mult_reads) - host only, no device needed."""
    lib = load()
    sp = np.zeros(HPK_MAX_STEPS, np.int32)
    sw = np.zeros(HPK_MAX_STEPS, np.int32)
    mk = np.zeros((HPK_MAX_STEPS, HPK_MAX_W + 1), np.int32)
    mr = np.zeros((HPK_MAX_STEPS, HPK_MAX_W + 1), np.int32)
    rc = lib.hpk_plan_rings(C.byref(params), sp.ctypes.data, sw.ctypes.data, mk.ctypes.data, mr.ctypes.data)
    if rc < 0:
        raise HpkError(rc, lib.hpk_last_error(None).decode())
    return list(zip(sp[:rc].tolist(), sw[:rc].tolist())), mk[:rc].copy(), mr[:rc].copy()


def _arr(ptr, n, dtype):
    """Owned numpy copy of n elements behind a ctypes pointer (one memcpy; as_array costs twice as much per call)."""
    if n == 0 or not ptr:
        return np.zeros(0, dtype)
    ct = ptr._type_
    raw = C.string_at(C.addressof(ptr.contents), n * C.sizeof(ct))
    return np.frombuffer(raw, dtype=np.dtype(ct)).astype(dtype, copy=True)


class BandResult(object):
    """Host copy of one hpk_result.  The scalars are copied at once; the arrays (`sets`, `gap`, the dense outputs) are
    copied out of the library's result when they are first read - a caller that only wants counts and timings of a
    chromosome (bench.py does, for all but one chromosome of a batch) does not pay ~13 numpy conversions for it.  The
    library's result is released when both have been read, or with this object."""

    def __init__(self, r, n, owner=None):
        self.steps = [(r.step_pi[s], r.step_wi[s], int(r.step_resolved[s]), bool(r.step_executed[s]))
                      for s in range(r.nsteps)]
        self.frozen_w = r.frozen_w
        self.slot_pi = [r.slot_pi[i] for i in range(r.nslots)]
        self.ncand = int(r.ncand)
        self.band_px = int(r.band_px)
        self.tiles = int(r.stencil_tiles)
        self.stencil_kernel = int(r.stencil_kernel)
        self.record_bound = int(r.record_bound)      # records for candidates resolved up to this width (255: all)
        self.rescored = bool(r.redone & 2)           # the survivor bound was too narrow: scoring and cut ran twice
        self.redone = bool(r.redone & 1)                 # the bound from the previous chromosome was too narrow: computed twice
        self.halo_w = int(r.halo_w)                  # halo of the stencil tiles: maxww, or the record bound (spec_halo)
        self.lean_tiles = int(r.lean_tiles)          # tiles built without their f64 plane | of those computed once more | candidates summed cell by cell
        self.lean_redone = int(r.lean_redone)
        self.lean_explicit = int(r.lean_explicit)
        self.batch_bands = int(r.batch_bands)        # chromosomes that shared this one's launches (kernel times are its share)
        self.nsig = int(r.nsig)                      # pixels reported over all sets
        self.nsurv_sig, self.nsurv_cut = int(r.nsurv_sig), int(r.nsurv_cut)
        self.timing = dict(h2d=r.ms_h2d, stencil=r.ms_stencil, freeze=r.ms_freeze, score=r.ms_score, tighten=r.ms_tighten,
                           gap=r.ms_gap,
                           d2h=r.ms_d2h, host_bh=r.ms_host_bh, total=r.ms_total)
        self._n = n
        self._r = r                  # the library's result: valid until _release
        self._owner = owner          # (lib, pointer) to free it with, or None: the caller frees it and wants everything now
        self._sets = self._gap = None
        self._dense = None
        if owner is None:
            self._materialise()

    def _materialise(self):
        if self._r is None:
            return
        r, n = self._r, self._n
        ns = int(r.nsig)
        x, y = _arr(r.x, ns, np.int64), _arr(r.y, ns, np.int64)
        O, bal, E = _arr(r.O, ns, np.float64), _arr(r.bal, ns, np.float64), _arr(r.E, ns, np.float64)
        p, q, oz = _arr(r.p, ns, np.float64), _arr(r.q, ns, np.float64), _arr(r.other_zero, ns, np.uint8)
        sets = []
        for i in range(r.nsets):
            s = r.sets[i]
            sl = slice(int(s.begin), int(s.end))
            sets.append(dict(pair=s.pair, fl='KY'[s.fl], nvalid=int(s.nvalid), numbin=s.numbin, emax=s.emax,
                             x=x[sl], y=y[sl], O=O[sl], bal=bal[sl], E=E[sl], p=p[sl], q=q[sl],
                             other_zero=oz[sl].astype(bool),
                             chunk_tests=_arr(s.chunk_tests, HPK_NB + 1, np.int64),
                             chunk_below=_arr(s.chunk_below, HPK_NB + 1, np.int64)))
        self._sets = sets
        self._gap = _arr(r.gap, n, np.uint8).astype(bool)
        dE = dw = dS = None
        if r.dense_E:
            ld = int(r.dense_ld)
            k = r.nslots * n * ld
            dE = np.ctypeslib.as_array(r.dense_E, shape=(k * 2,)).reshape(r.nslots, n, ld, 2).copy()
            dw = np.ctypeslib.as_array(r.dense_w, shape=(k,)).reshape(r.nslots, n, ld).copy()
            if r.dense_sums:
                dS = np.ctypeslib.as_array(r.dense_sums, shape=(k * 4,)).reshape(r.nslots, n, ld, 4).copy()
        self._dense = (dE, dw, dS)
        self._release()

    def _release(self):
        r, owner = self._r, self._owner
        self._r = self._owner = None
        if r is not None and owner is not None:
            owner[0].hpk_result_free(owner[1])

    sets = property(lambda self: (self._materialise(), self._sets)[1])
    gap = property(lambda self: (self._materialise(), self._gap)[1])
    dense_E = property(lambda self: (self._materialise(), self._dense[0])[1])
    dense_w = property(lambda self: (self._materialise(), self._dense[1])[1])
    dense_sums = property(lambda self: (self._materialise(), self._dense[2])[1])

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass


class DeviceBand(object):
    """One chromosome's band built in device memory from its pixel table (hpk_devband_create): what travels over the bus is
    the pixels, not the dense band.  `band` goes into Context.submit / submit_batch; the memory is handed back when this
    object goes (keep it until the job that uses it has been collected - BatchJob / Job do, through `keep`)."""

    def __init__(self, ctx, handle, band, stored):
        self.ctx, self.handle, self.band, self.stored = ctx, handle, band, int(stored)
        self.shape = (int(band.n), int(band.num))
        self.nbytes = 4 * int(band.n) * int(band.ld)

    def close(self):
        h, self.handle = self.handle, None
        if h is not None and getattr(self.ctx, 'h', None):
            self.ctx.lib.hpk_devband_free(self.ctx.h, h)

    __del__ = close


class Job(object):
    """A chromosome in flight on one lane of a Context (hpk_submit_band / hpk_collect)."""

    def __init__(self, ctx, handle, n, keep=None):
        self.ctx, self.handle, self.n, self._keep = ctx, handle, n, keep

    def result(self):
        if self.handle is None:
            raise HpkError(ERR_INVALID, 'job already collected')
        res = C.POINTER(Result)()
        h, self.handle = self.handle, None
        rc = self.ctx.lib.hpk_collect(self.ctx.h, h, C.byref(res))
        self._keep = None
        self.ctx._check(rc)
        return BandResult(res.contents, self.n, owner=(self.ctx.lib, res))

    def __del__(self):          # a dropped job still has to give its lane back
        if getattr(self, 'handle', None) is not None and getattr(self.ctx, 'h', None):
            h, self.handle = self.handle, None
            self.ctx.lib.hpk_collect(self.ctx.h, h, None)


class BatchJob(object):
    """A batch of chromosomes in flight on one lane of a Context (hpk_submit_batch / hpk_collect_batch)."""

    def __init__(self, ctx, handle, ns, keep=None):
        self.ctx, self.handle, self.ns, self._keep = ctx, handle, list(ns), keep

    def results(self, raise_on_error=True):
        """-> [BandResult, ...] in submission order.  A chromosome on which the reference raises (an empty widening step)
        raises here too (the first one, after every result has been taken over), or - raise_on_error=False - has the
        exception object in its place."""
        if self.handle is None:
            raise HpkError(ERR_INVALID, 'job already collected')
        nb = len(self.ns)
        outs = (C.POINTER(Result) * nb)()
        status = (C.c_int32 * nb)()
        msg = C.create_string_buffer(nb * 512)
        h, self.handle = self.handle, None
        rc = self.ctx.lib.hpk_collect_batch(self.ctx.h, h, outs, status, msg, 512)
        self._keep = None
        self.ctx._check(rc)
        res = []
        for b in range(nb):
            if status[b] == HPK_OK:
                # (the pointer object is the result's own: `outs` is an array, whose elements are temporaries)
                res.append(BandResult(outs[b].contents, self.ns[b], owner=(self.ctx.lib, C.cast(outs[b], C.POINTER(Result)))))
            else:
                text = msg.raw[b * 512:(b + 1) * 512].split(b'\0', 1)[0].decode()
                res.append((EmptyStepError if status[b] == ERR_EMPTY_STEP else HpkError)(status[b], text))
        if raise_on_error:
            for r in res:
                if isinstance(r, Exception):
                    raise r
        return res

    def __del__(self):          # a dropped job still has to give its lane back
        if getattr(self, 'handle', None) is not None and getattr(self.ctx, 'h', None):
            h, self.handle = self.handle, None
            self.ctx.lib.hpk_collect_batch(self.ctx.h, h, None, None, None, 0)


class Context(object):
    """One hpk_ctx = one GPU.  Not shared between threads."""

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.hpk_create(int(device), C.byref(h))
        if rc != HPK_OK:
            raise HpkError(rc, self.lib.hpk_last_error(None).decode())
        self.h = h
        self.device = device
        # chunk boundaries exactly as the reference's numpy produces them (callers.py:36-37)
        b = np.array([np.power(2, ((i - 1) / 3.)) for i in range(1, HPK_NB + 1)], dtype=np.float64)
        self._check(self.lib.hpk_set_chunk_bounds(self.h, b.ctypes.data, HPK_NB))
        self.bounds = b

    def _check(self, rc):
        if rc != HPK_OK:
            msg = self.lib.hpk_last_error(self.h).decode()
            raise (EmptyStepError if rc == ERR_EMPTY_STEP else HpkError)(rc, msg)

    def close(self):
        if getattr(self, 'h', None):
            self.lib.hpk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value):
        """Tuning / test switch of the context (hpk_set_option, include/hpk.h): 'rounds', 'surv_cap', 'spec', 'spec_margin',
        'spec_force', 'spec_halo', 'spec_surv', 'spec_surv_margin', 'spec_surv_force', 'host_threads', 'risk_log2',
        'tile_order', 'gap_kernel', 'score_div', 'dbg_stop', 'grid_cap', 'lean', 'lean_max', 'lean_frac_pct', 'kcrit', 'cpu_threads',
        'reset_hints'."""
        self._check(self.lib.hpk_set_option(self.h, name.encode(), int(value)))

    def info(self):
        name = C.create_string_buffer(128)
        cus = C.c_int32()
        hbm = C.c_int64()
        self._check(self.lib.hpk_device_info(self.h, name, 128, C.byref(cus), C.byref(hbm)))
        return dict(name=name.value.decode(), cus=cus.value, hbm_bytes=hbm.value)

    # -- band helpers
    @staticmethod
    def _band(n, num, ld, raw, balanced, weight, IR, b1, b2, on_device):
        bd = Band()
        bd.n, bd.num, bd.ld = int(n), int(num), int(ld)
        bd.raw, bd.balanced, bd.weight, bd.IR, bd.bias1, bd.bias2 = raw, balanced, weight, IR, b1, b2
        bd.on_device = 1 if on_device else 0
        return bd

    def _host_band(self, raw, IR, bias1, bias2, balanced=None, weight=None, num_hint=None):
        raw = np.ascontiguousarray(raw, dtype=np.float32)
        n, ld = raw.shape
        keep = [raw]
        derive = IR is None         # IR (and, unless given, the biases) derived on the device from the weights
        if derive:
            num = num_hint if num_hint is not None else ld
            irp = b1p = b2p = None
            if bias1 is not None:
                b1 = np.ascontiguousarray(bias1, dtype=np.float64)
                b2 = b1 if (bias2 is bias1 or bias2 is None) else np.ascontiguousarray(bias2, dtype=np.float64)
                keep += [b1, b2]
                b1p, b2p = b1.ctypes.data, b2.ctypes.data
        else:
            IR = np.ascontiguousarray(IR, dtype=np.float64)
            b1 = np.ascontiguousarray(bias1, dtype=np.float64)
            b2 = b1 if bias2 is bias1 else np.ascontiguousarray(bias2, dtype=np.float64)
            keep += [IR, b1, b2]
            num, irp, b1p, b2p = IR.size, IR.ctypes.data, b1.ctypes.data, b2.ctypes.data
        balp = wp = None
        if balanced is not None:
            balanced = np.ascontiguousarray(balanced, dtype=np.float64)
            assert balanced.shape == raw.shape
            keep.append(balanced)
            balp = balanced.ctypes.data
        if weight is not None:
            weight = np.ascontiguousarray(weight, dtype=np.float64)
            keep.append(weight)
            wp = weight.ctypes.data
        bd = self._band(n, num, ld, raw.ctypes.data, balp, wp, irp, b1p, b2p, False)
        return bd, keep

    def devband(self, bin1, bin2, count, n, num, weight, bias=None):
        """Pixels (bin1, bin2, count) of one chromosome, bins relative to its first bin, each pixel once in either
        orientation -> DeviceBand (the GPU counterpart of band.band_from_coo; weights / biases as `hpk_band` takes them)."""
        i = np.ascontiguousarray(bin1, dtype=np.int64)
        j = np.ascontiguousarray(bin2, dtype=np.int64)
        v = np.asarray(count)
        f64 = v.dtype.kind == 'f' or (v.size > 0 and v.dtype.itemsize >= 4 and v.dtype != np.int32 and
                                      (int(v.max()) > 0x7fffffff or int(v.min()) < -0x80000000))
        if v.size and float(v.max()) >= float(1 << 24):
            raise ValueError('devband: a count of %g is beyond what the f32 band holds exactly (2^24)' % float(v.max()))
        v = np.ascontiguousarray(v, dtype=np.float64 if f64 else np.int32)
        w = np.ascontiguousarray(weight, dtype=np.float64)
        b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float64)
        assert w.size == n and (b is None or b.size == n) and i.size == j.size == v.size
        h, bd = C.c_void_p(), Band()
        rc = self.lib.hpk_devband_create(self.h, i.ctypes.data, j.ctypes.data, v.ctypes.data, 1 if f64 else 0, i.size, int(n), int(num),
                                         w.ctypes.data, None if b is None else b.ctypes.data, C.byref(h), C.byref(bd))
        if rc < 0:
            self._check(int(rc))
        return DeviceBand(self, h, bd, rc)

    def score(self, band, params, n):
        res = C.POINTER(Result)()
        rc = self.lib.hpk_score_band(self.h, C.byref(band), C.byref(params), C.byref(res))
        self._check(rc)
        try:
            return BandResult(res.contents, n)
        finally:
            self.lib.hpk_result_free(res)

    # ---- one chromosome ahead: submit() enqueues everything on a free lane, Job.result() waits and finishes on the host
    def submit(self, band, params, n, keep=None):
        job = C.c_void_p()
        self._check(self.lib.hpk_submit_band(self.h, C.byref(band), C.byref(params), C.byref(job)))
        return Job(self, job, n, keep)

    def submit_host(self, raw, IR, bias1, bias2, params, balanced=None, weight=None, num=None):
        """Like score_host, but returns a Job; the arrays are kept alive until Job.result()."""
        bd, keep = self._host_band(raw, IR, bias1, bias2, balanced, weight, num_hint=num)
        return self.submit(bd, params, raw.shape[0], keep)

    def submit_device(self, n, num, ld, raw_ptr, IR_ptr, b1_ptr, b2_ptr, params, balanced_ptr=None, weight_ptr=None):
        bd = self._band(n, num, ld, raw_ptr, balanced_ptr, weight_ptr, IR_ptr, b1_ptr, b2_ptr, True)
        return self.submit(bd, params, n)

    # ---- a batch of chromosomes through one set of launches
    def submit_batch(self, bands, params, ns, keep=None):
        """bands: [Band, ...] (from `_band` / `_host_band`), ns: their chromosome lengths -> BatchJob"""
        arr = (Band * len(bands))(*bands)
        job = C.c_void_p()
        self._check(self.lib.hpk_submit_batch(self.h, arr, len(bands), C.byref(params), C.byref(job)))
        return BatchJob(self, job, ns, keep)

    def submit_batch_host(self, items, params):
        """items: [dict(raw=, IR=, bias1=, bias2=, balanced=, weight=, num=), ...] of host arrays - or dict(raw=DeviceBand) -
        -> BatchJob (arrays / device bands kept alive until the results are in)"""
        bands, keep, ns = [], [], []
        for it in items:
            if isinstance(it.get('raw'), DeviceBand):           # built on the device from the pixel table: nothing to upload
                bands.append(it['raw'].band)
                keep.append(it['raw'])
                ns.append(it['raw'].band.n)
                continue
            bd, kp = self._host_band(it['raw'], it.get('IR'), it.get('bias1'), it.get('bias2'), it.get('balanced'), it.get('weight'),
                                     num_hint=it.get('num'))
            bands.append(bd)
            keep.append(kp)
            ns.append(bd.n)
        return self.submit_batch(bands, params, ns, keep)

    @property
    def pipeline_depth(self):
        return int(self.lib.hpk_pipeline_depth())

    def score_host(self, raw, IR, bias1, bias2, params, balanced=None, weight=None, num=None):
        """IR = bias1 = bias2 = None: derived on the device from `weight` (then `num` = stored diagonals, default
        raw.shape[1])."""
        bd, keep = self._host_band(raw, IR, bias1, bias2, balanced, weight, num_hint=num)
        return self.score(bd, params, raw.shape[0])

    def score_device(self, n, num, ld, raw_ptr, IR_ptr, b1_ptr, b2_ptr, params, balanced_ptr=None, weight_ptr=None):
        bd = self._band(n, num, ld, raw_ptr, balanced_ptr, weight_ptr, IR_ptr, b1_ptr, b2_ptr, True)
        return self.score(bd, params, n)

    def poisson_sf(self, k, lam):
        k = np.ascontiguousarray(k, dtype=np.float64)
        lam = np.ascontiguousarray(np.broadcast_to(lam, k.shape), dtype=np.float64)
        out = np.empty_like(k)
        self._check(self.lib.hpk_poisson_sf(self.h, k.ctypes.data, lam.ctypes.data, out.ctypes.data, k.size))
        return out

    def bruteforce_sums(self, raw, IR, bias1, bias2, params, step, rows, cols, balanced=None, weight=None):
        bd, keep = self._host_band(raw, IR, bias1, bias2, balanced, weight)
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        out = np.empty((rows.size, 5), dtype=np.float64)
        self._check(self.lib.hpk_bruteforce_sums(self.h, C.byref(bd), C.byref(params), int(step), rows.ctypes.data,
                                                 cols.ctypes.data, rows.size, out.ctypes.data))
        return out


    def probe_sums(self, band, params, rows, cols, nslots):
        """(bS_K, bE_K, bS_Y, bE_Y, width) per pixel and slot from the production kernels' records; `band` from
        `_band` / `_host_band`.  -> [count, nslots, 5]"""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        out = np.empty((rows.size, nslots, 5), dtype=np.float64)
        self._check(self.lib.hpk_probe_sums(self.h, C.byref(band), C.byref(params), rows.ctypes.data, cols.ctypes.data,
                                            rows.size, out.ctypes.data))
        return out

    def bruteforce_band(self, band, params, step, rows, cols):
        """`bruteforce_sums` on a prepared band (device pointers allowed)."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        out = np.empty((rows.size, 5), dtype=np.float64)
        self._check(self.lib.hpk_bruteforce_sums(self.h, C.byref(band), C.byref(params), int(step), rows.ctypes.data,
                                                 cols.ctypes.data, rows.size, out.ctypes.data))
        return out


_default_ctx = {}


def default_context(device=0):
    """Process-wide context per device (the callers are invoked once per chromosome) - what `callers.hiccups` / `bhfdr` and the
    command lines score on.  It runs under spec_halo = 2 (include/hpk.h): a chromosome's E / p / q are a function of the chromosome
    alone, whatever was scored on the context before it - like the reference's functions, whose results do not depend on the calls
    made before (HPK_SPEC_HALO in the environment overrides; a Context() of one's own starts with the library's default, 1)."""
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
        if 'HPK_SPEC_HALO' not in os.environ:
            _default_ctx[device].set_option('spec_halo', 2)
    return _default_ctx[device]
