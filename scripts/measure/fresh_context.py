#!/usr/bin/env python
"""The first genome of a fresh context (what one command-line invocation pays once): wall time of the first and of the second
submission, for the profile of HIP API calls around it (rocprofv3 --hip-trace --stats -- python scripts/measure/fresh_context.py wg_5kb)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from hicpeaks_amd import _lib, band, bandgen, synthetic
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'wg_5kb']
dev = torch.device('cuda', 0)
res, mw, D = cfg['res'], min(cfg['ww']), cfg['maxapart'] // cfg['res']
num = D + cfg['maxww'] + 1
ld = (num + 63) // 64 * 64
sizes = synthetic.hg38_bins(res)
bands = []
for i, c in enumerate(sorted(sizes, key=lambda k: -sizes[k])):
    n = sizes[c]
    raw_d, w_d, _, _ = bandgen.device_band(n, num, ld, mw, depth=cfg['depth'], nloops=max(1, cfg['nloops'] * n // cfg['n']), seed=i, device=dev, want_expected=False)
    bands.append((n, raw_d, w_d))
torch.cuda.synchronize()
prm = _lib.make_params(_lib.MODE_HICCUPS, cfg['pw'], cfg['ww'], cfg['maxww'], bench.SIG, cfg['maxapart'], cfg['res'], bench.MIN_READS, _lib.FLAG_NO_STENCIL_TIMING)
t0 = time.perf_counter()
c = _lib.Context(0)
c.set_option('spec_halo', 2)
t1 = time.perf_counter()
for k in range(3):
    bd = [c._band(n, num, ld, r.data_ptr(), None, w.data_ptr(), None, None, None, True) for (n, r, w) in bands]
    torch.cuda.synchronize()
    ta = time.perf_counter()
    job = c.submit_batch(bd, prm, [n for (n, _, _) in bands])
    tb = time.perf_counter()
    rs = job.results()
    tc = time.perf_counter()
    print('call %d: submit %.1f ms, collect %.1f ms (redone %d)' % (k, (tb - ta) * 1e3, (tc - tb) * 1e3, sum(int(r.redone) for r in rs)))
print('context %.1f ms' % ((t1 - t0) * 1e3))
