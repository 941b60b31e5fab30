"""debug: resolving widths / sums per candidate of the HIP path against the oracle on one fixture; prints where they differ"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np
from conftest import load_golden
from hicpeaks_amd import _lib
from oracle import hiccups_oracle as orc
import test_gpu_parity as T

name = sys.argv[1] if len(sys.argv) > 1 else 'hiccups_p2w5_shallow'
mode = sys.argv[2] if len(sys.argv) > 2 else 'weight'
g = load_golden(name)
p = g.params
raw, IR, cband, biases = T._inputs(g)
n, num = raw.shape
loc = orc.hiccups_local_sums(raw, cband, IR, n, num, p['pw'], p['ww'], p['maxww'], p['maxapart'], p['res'], p['min_local_reads'])
ctx = _lib.Context(0)
detail = dict(dense=True)
T._call(g, ctx, mode, detail)
R = detail['result']
print('n', n, 'num', num, 'params', {k: p[k] for k in ('pw', 'ww', 'maxww', 'maxapart', 'res', 'min_local_reads')}, 'halo', R.halo_w, 'frozen', R.frozen_w)
vx, vy = loc['vx'], loc['vy']
for slot, pi in enumerate(R.slot_pi):
    w = R.dense_w[slot][vx, vy - vx].astype(np.int64)
    w = np.where(w > R.frozen_w, 0, w)
    want = loc['wres'][pi]
    bad = np.nonzero(w != want)[0]
    print('slot', slot, 'candidates', vx.size, 'width mismatches', bad.size)
    for i in bad[:40]:
        print('   r %4d c %4d k %4d got %d want %d' % (vx[i], vy[i], vy[i] - vx[i], w[i], want[i]))
    if bad.size:
        print('   rows of mismatches:', np.unique(vx[bad])[:60])
        print('   diagonals of mismatches:', np.unique((vy - vx)[bad])[:80])
    sums = R.dense_sums[slot][vx, vy - vx]
    res_ = (w > 0) & (w == want)
    for col, (fl, arr) in enumerate([('K', 'bSV'), ('K', 'bEV'), ('Y', 'bSV'), ('Y', 'bEV')]):
        wv = loc[arr][pi][fl]
        err = np.abs(sums[res_, col] - wv[res_]) / np.maximum(np.abs(wv[res_]), 1e-300)
        nb = int((err > 1e-10).sum())
        print('   col', col, fl, arr, 'bad', nb, 'of', int(res_.sum()))
        if nb:
            idx = np.nonzero(res_)[0][err > 1e-10]
            for i in idx[:20]:
                print('      r %4d c %4d k %4d got %.6g want %.6g' % (vx[i], vy[i], vy[i] - vx[i], sums[i, col], wv[i]))
