import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
from hicpeaks_amd import _lib, bandgen
cfg = bench.CONFIGS['chr1_10kb']
mw, D = min(cfg['ww']), cfg['maxapart'] // cfg['res']
num = D + cfg['maxww'] + 1
ld = (num + 63) // 64 * 64
ctx = _lib.Context(0)
prm = _lib.make_params(_lib.MODE_HICCUPS, cfg['pw'], cfg['ww'], cfg['maxww'], bench.SIG, cfg['maxapart'], cfg['res'], bench.MIN_READS, 0)
dev = torch.device('cuda', 0)
ctx.set_option('spec_surv', 0)
for dp in (15.0, 40.0, 60.0, 150.0):
    raw_d, w_d, _, _ = bandgen.device_band(cfg['n'], num, ld, mw, depth=dp, nloops=cfg['nloops'], seed=5, device=dev, want_expected=False)
    torch.cuda.synchronize()
    for rep in range(3):
        R = ctx.score_device(cfg["n"], num, ld, raw_d.data_ptr(), 0, 0, 0, prm, weight_ptr=w_d.data_ptr())
    print(dp, 'ncand', R.ncand, 'tiles', R.tiles, 'lean', R.lean_tiles, 'halo', R.halo_w, 'bound', R.record_bound, 'frozen', R.frozen_w, 'cand/tile', R.ncand / max(R.tiles, 1), 'stencil ms', R.timing.get('stencil'), 'p<=sig records', R.nsurv_sig, 'of band px %.4f' % (R.nsurv_sig / R.band_px), 'sets', len(R.sets))
