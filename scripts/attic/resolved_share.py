import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch, bench
from hicpeaks_amd import _lib, bandgen
for name in sys.argv[1:]:
    cfg = bench.CONFIGS[name]
    n = cfg['n']; mw = min(cfg['ww']); D = cfg['maxapart'] // cfg['res']; num = D + cfg['maxww'] + 1; ld = (num + 63) // 64 * 64
    dev = torch.device('cuda', 0)
    raw_d, w_d, ir_d, b_d = bandgen.device_band(n, num, ld, mw, depth=cfg['depth'], nloops=cfg['nloops'], seed=0, device=dev)
    ctx = _lib.Context(0)
    prm = _lib.make_params(_lib.MODE_HICCUPS, cfg['pw'], cfg['ww'], cfg['maxww'], 0.05, cfg['maxapart'], cfg['res'], 16, 0)
    R = ctx.submit_device(n, num, ld, raw_d.data_ptr(), ir_d.data_ptr(), b_d.data_ptr(), b_d.data_ptr(), prm, weight_ptr=w_d.data_ptr()).result()
    print(name, 'ncand', R.ncand, 'steps', R.steps, 'resolved', sum(s[2] for s in R.steps))
