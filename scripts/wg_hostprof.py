import sys, os, time, collections, cProfile, pstats
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import bench
from hicpeaks_amd import _lib, band, bandgen, parallel, synthetic
cfg = bench.CONFIGS[sys.argv[1]]
ctx = _lib.Context(0)
dev = torch.device('cuda', 0)
res, mw, D = cfg['res'], min(cfg['ww']), cfg['maxapart'] // cfg['res']
num = D + cfg['maxww'] + 1
ld = (num + 63) // 64 * 64
sizes = synthetic.hg38_bins(res)
bands = []
for i, c in enumerate(parallel.lpt_partition(sizes, 1)[0]):
    n = sizes[c]
    raw_d, w_d, _, _ = bandgen.device_band(n, num, ld, mw, depth=cfg['depth'], nloops=max(1, cfg['nloops'] * n // cfg['n']), seed=i, device=dev, want_expected=False)
    bands.append((c, n, raw_d, w_d))
torch.cuda.synchronize()
prm = _lib.make_params(_lib.MODE_HICCUPS, cfg['pw'], cfg['ww'], cfg['maxww'], 0.05, cfg['maxapart'], cfg['res'], 16, 0)
tsub, tcol = [], []
def one_pass():
    pending = collections.deque()
    for c, n, raw_d, w_d in bands:
        t = time.perf_counter()
        pending.append(ctx.submit_device(n, num, ld, raw_d.data_ptr(), None, None, None, prm, weight_ptr=w_d.data_ptr()))
        tsub.append(time.perf_counter() - t)
        if len(pending) >= 2:
            t = time.perf_counter(); pending.popleft().result(); tcol.append(time.perf_counter() - t)
    while pending:
        t = time.perf_counter(); pending.popleft().result(); tcol.append(time.perf_counter() - t)
one_pass(); tsub.clear(); tcol.clear()
per = []
t0 = time.perf_counter()
for _ in range(int(os.environ.get('PASSES', '3'))):
    t1 = time.perf_counter(); one_pass(); per.append(round((time.perf_counter() - t1) * 1e3, 2))
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / len(per)
print('per pass ms', per)
print('pass %.2f ms; submit mean %.3f ms (sum %.2f); collect mean %.3f ms (sum %.2f)' % (el * 1e3, np.mean(tsub) * 1e3, np.sum(tsub) / len(per) * 1e3, np.mean(tcol) * 1e3, np.sum(tcol) / len(per) * 1e3))
cProfile.run('one_pass()', '/tmp/p.prof')
pstats.Stats('/tmp/p.prof').sort_stats('tottime').print_stats(8)
