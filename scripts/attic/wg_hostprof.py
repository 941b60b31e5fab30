"""Where a whole-genome pass spends its wall time: bench.py's wg_* loop with the library calls timed on the host.
usage: wg_hostprof.py <wg_10kb_union|wg_5kb>   (GPU box)"""
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
flags = _lib.FLAG_NO_STENCIL_TIMING
prm = _lib.make_params(_lib.MODE_HICCUPS, cfg['pw'], cfg['ww'], cfg['maxww'], 0.05, cfg['maxapart'], cfg['res'], 16, flags)
prm_ph = _lib.make_params(_lib.MODE_HICCUPS, cfg['pw'], cfg['ww'], cfg['maxww'], 0.05, cfg['maxapart'], cfg['res'], 16, _lib.FLAG_PHASE_TIMING)
tsub, tcol = [], []
def submit(p):
    bd = [ctx._band(n, num, ld, r.data_ptr(), None, w.data_ptr(), None, None, None, True) for (_, n, r, w) in bands]
    return ctx.submit_batch(bd, p, [n for (_, n, _, _) in bands])
def passes(k, p=prm):
    pending = collections.deque()
    for _ in range(k):
        t = time.perf_counter(); pending.append(submit(p)); tsub.append(time.perf_counter() - t)
        if len(pending) >= 2:
            t = time.perf_counter(); rs = pending.popleft().results(); tcol.append(time.perf_counter() - t)
    while pending:
        t = time.perf_counter(); rs = pending.popleft().results(); tcol.append(time.perf_counter() - t)
    return rs
passes(3); tsub.clear(); tcol.clear()
torch.cuda.synchronize()
K = int(os.environ.get('PASSES', '10'))
t0 = time.perf_counter(); passes(K); torch.cuda.synchronize(); el = (time.perf_counter() - t0) / K
print('pass %.2f ms; submit mean %.3f ms; collect mean %.3f ms' % (el * 1e3, np.mean(tsub) * 1e3, np.mean(tcol) * 1e3))
# synchronous passes: the GPU's own time for a genome (events around the phases) and the host half alone
for _ in range(2):
    t1 = time.perf_counter(); j = submit(prm_ph); t2 = time.perf_counter(); rs = j.results(); t3 = time.perf_counter()
ph = {k: float(sum(r.timing[k] for r in rs)) for k in rs[0].timing if k != 'total'}
print('synchronous pass: submit %.2f ms, results() %.2f ms; phases (sum over the batch) ms:' % ((t2 - t1) * 1e3, (t3 - t2) * 1e3), {k: round(v, 3) for k, v in ph.items()})
print('significant px', sum(int(sum(s['x'].size for s in r.sets)) for r in rs), 'records copied back', sum(r.nsurv_cut for r in rs))
cProfile.run('passes(4)', '/tmp/p.prof')
pstats.Stats('/tmp/p.prof').sort_stats('tottime').print_stats(10)
