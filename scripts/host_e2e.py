#!/usr/bin/env python
"""Host side of the path on a REAL cooler file, stage by stage (counterpart of worker(),
scripts/pyHICCUPS:139-166, and of the Pool.map loop at :192-198):

    open the .mcool  ->  read one chromosome's pixel table (HDF5 chunks, gzip)  ->  band builder (hpk_band_from_coo)
    ->  upload + kernels + host half (Benjamini-Hochberg, clustering)           per chromosome, cold and warm page cache

and the command line itself (scripts/pyHICCUPS) on the same file.  The file is written by scripts/make_cool.py (plain
h5py under /opt/conda: the `cooler` package is not in this image) from the synthetic genome at 5 kb, 10 Mb band
(num = 2011 stored diagonals) - by default chr1, chr21 and chrX (90 343 of the genome's 617 665 bins: the file for all
23 chromosomes takes ~40 s to write: --chroms 1 2 ... 22 X, profiles/r03_host_e2e_wg.txt), extrapolated to the genome by bins.

    python scripts/host_e2e.py [--chroms 1 21 X] [--res 5000] [--file /tmp/hpk_e2e.mcool] > profiles/r03_host_e2e.txt
"""
import argparse
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def drop_caches():
    try:
        subprocess.check_call(['sync'])
        with open('/proc/sys/vm/drop_caches', 'w') as f:
            f.write('3\n')
        return True
    except Exception:
        return False


def stages(uri, chroms, a, ctx):
    """one pass over the chromosomes -> {chrom: (read_s, band_s, score_s, pixels, peaks)}, open_s"""
    from hicpeaks_amd import io, band, callers
    t0 = time.perf_counter()
    src = io.CoolerSource(uri)
    t_open = time.perf_counter() - t0
    res = src.binsize
    num = a.maxapart // res + a.maxww + 1
    out = {}
    from hicpeaks_amd import cli
    args_dict = dict(maxapart=a.maxapart, maxww=a.maxww, clr_weight_name='weight')
    for c in chroms:
        key = 'chr' + c
        t0 = time.perf_counter()
        got = cli._read(args_dict, src, key, not a.host_bands)       # HDF5 chunks -> inflate -> the chromosome's pixel table
        t1 = time.perf_counter()
        if a.host_bands:
            raw, w = got[2], got[3]
            npx = int((raw != 0).sum())
        else:                   # sparse: the pixel table goes to the GPU and the band is built there; dense: threaded host scatter
            _, raw, w, _b = cli._to_item(key, got, ctx)
            npx = int(got[2].size)
        n = int(raw.shape[0])
        t2 = time.perf_counter()
        call = callers.hiccups_batch_submit([(c, raw, w, None)], pw=[a.pw], ww=[a.ww], maxww=a.maxww, sig=0.1, sumq=0.01,
                                            double_fold=1.75, single_fold=2, maxapart=a.maxapart, res=res, use_raw=False,
                                            min_marginal_peaks=2, onlyanchor=False, min_local_reads=16, ctx=ctx)
        table = call.results()[0]
        t3 = time.perf_counter()
        out[c] = (t1 - t0, t2 - t1, t3 - t2, npx, len(table), n)
    return out, t_open


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--chroms', nargs='*', default=['1', '21', 'X'])
    ap.add_argument('--res', type=int, default=5000)
    ap.add_argument('--maxapart', type=int, default=10000000)
    ap.add_argument('--maxww', type=int, default=10)
    ap.add_argument('--pw', type=int, default=4)
    ap.add_argument('--ww', type=int, default=7)
    ap.add_argument('--file', default='/tmp/hpk_e2e.mcool')
    ap.add_argument('--depth', type=float, default=25.0)
    ap.add_argument('--host-bands', action='store_true', help='build the dense bands on the host (band_from_coo) instead of on the GPU')
    ap.add_argument('--deep', action='store_true',
                    help='the file is a deep map written by scripts/make_cool_deep.py (--depth, pixels beyond the band included)')
    a = ap.parse_args()
    from hicpeaks_amd import _lib, synthetic
    num = a.maxapart // a.res + a.maxww + 1
    group = '/resolutions/%d' % a.res
    uri = '%s::%s' % (a.file, group)
    print('# scripts/host_e2e.py: chromosomes %s @%d bp, band %d diagonals, (p, w) = (%d, %d); bands built on the %s' % (
        ' '.join(a.chroms), a.res, num, a.pw, a.ww, 'host (band_s = hpk_band_from_coo)' if a.host_bands else 'GPU (band_s = hpk_devband_create: upload of the pixels + scatter)'))
    if not os.path.exists(a.file):
        t0 = time.perf_counter()
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
        if a.deep:
            subprocess.check_call(['/opt/conda/bin/python3.9', os.path.join(REPO, 'scripts', 'make_cool_deep.py'), a.file, '--res', str(a.res),
                                   '--num', str(num), '--depth', str(a.depth), '--far', '--chroms'] + a.chroms,
                                  env=env, stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
        else:
            subprocess.check_call(['/opt/conda/bin/python3.9', os.path.join(REPO, 'scripts', 'make_cool.py'), a.file, '--genome', 'hg38',
                                   '--res', str(a.res), '--num', str(num), '--group', group, '--depth', str(a.depth), '--chroms'] + a.chroms,
                                  env=env, stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
        print('# wrote %s (%.1f MB, gzip-6 chunks as cooler writes them) in %.0f s' % (a.file, os.path.getsize(a.file) / 1e6,
                                                                                     time.perf_counter() - t0))
    sizes = synthetic.hg38_bins(a.res)
    scale = sum(sizes.values()) / float(sum(sizes[c] for c in a.chroms))
    ctx = _lib.default_context(0)
    print('# backend of hicpeaks_amd.cool: %s; GPU: %s' % ('h5py' if _have('h5py') else 'libhdf5 through ctypes', ctx.info()['name']))
    for label in (('cold', 'warm') if a.deep else ('cold', 'warm', 'warm')):
        cold = label == 'cold'
        if cold and not drop_caches():
            print('# (page cache could not be dropped: no cold pass)')
            continue
        per, t_open = stages(uri, a.chroms, a, ctx)
        print('## %s page cache: open %.3f s' % (label, t_open))
        print('%-6s %8s %12s %9s %9s %9s %7s' % ('chrom', 'bins', 'pixels', 'read_s', 'band_s', 'score_s', 'peaks'))
        tot = np.zeros(3)
        for c in a.chroms:
            r, b, s, npx, npk, n = per[c]
            tot += (r, b, s)
            print('%-6s %8d %12d %9.3f %9.3f %9.3f %7d' % (c, n, npx, r, b, s, npk))
        print('%-6s %8s %12s %9.3f %9.3f %9.3f   -> whole genome by bins (x %.2f): read %.1f s, band %.1f s, score %.1f s' % (
            'sum', '', '', tot[0], tot[1], tot[2], scale, tot[0] * scale, tot[1] * scale, tot[2] * scale))
    # the command line end to end on the same file (interpreter start, HIP initialisation, reading, kernels, BEDPE lines)
    for label in ('cold', 'warm'):
        if label == 'cold' and not drop_caches():
            continue
        outp = '/tmp/hpk_e2e_%s.bedpe' % label
        cmd = [sys.executable, os.path.join(REPO, 'scripts', 'pyHICCUPS'), '-p', uri, '-O', outp, '--pw', str(a.pw), '--ww', str(a.ww),
               '--maxww', str(a.maxww), '--maxapart', str(a.maxapart), '-C'] + a.chroms + ['--logFile', '/tmp/hpk_e2e.log']
        t0 = time.perf_counter()
        rc = subprocess.call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                             env=dict(os.environ, **({'HPK_HOST_BANDS': '1'} if a.host_bands else {})))
        dt = time.perf_counter() - t0
        nl = sum(1 for _ in open(outp)) if rc == 0 and os.path.exists(outp) else -1
        print('## scripts/pyHICCUPS on the file, %s page cache: %.2f s wall, rc %d, %d BEDPE lines (x %.2f by bins: %.0f s for the genome on one GPU, '
              'reading on up to 16 decoding threads)' % (label, dt, rc, nl, scale, dt * scale))


def _have(mod):
    try:
        __import__(mod)
        return True
    except ImportError:
        return False


if __name__ == '__main__':
    main()
