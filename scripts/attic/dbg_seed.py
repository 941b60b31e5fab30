import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import gpu_fuzz as fz
from hicpeaks_amd import _lib, callers, synthetic
from oracle import hiccups_oracle as orc
seed = int(sys.argv[1])
ctx = _lib.Context(0)
status, desc, note = fz.one_case(seed, ctx)
print(status, desc, note)
n, D, maxww, pw, ww, depth, min_reads, sig = (desc[k] for k in ('n', 'D', 'maxww', 'pw', 'ww', 'depth', 'min_reads', 'sig'))
num = D + maxww + 1
rng = np.random.default_rng(seed)
raw, weight, _ = synthetic.synth_band(n, num, depth=depth, nloops=0, seed=seed)
print('raw rows 0..3, k 0..9:\n', raw[:4, :10])
print('weight[:6]', weight[:6])
IR, cband, biases = orc.prep_from_band(raw, weight, min(ww))
detail = dict(dense=True)
callers.hiccups_band(raw.astype(np.float32), IR, biases, biases, chrom='T', pw=pw, ww=ww, maxww=maxww, sig=sig, maxapart=D * 10000, res=10000,
                     min_local_reads=min_reads, min_marginal_peaks=2, onlyanchor=False, ctx=ctx, detail=detail, weight=weight)
R = detail['result']
print('dense_w rows 0..3, k 0..9:\n', R.dense_w[0][:4, :10])
loc = orc.hiccups_local_sums(raw, cband, IR, n, num, pw, ww, maxww, D * 10000, 10000, min_reads)
W = np.zeros((4, 10), int)
for x, y, w in zip(loc['vx'], loc['vy'], loc['wres'][0]):
    if x < 4 and y - x < 10: W[x, y - x] = w
print('oracle widths:\n', W)
print('halo', R.halo_w, 'record bound', R.record_bound, 'frozen', R.frozen_w)
