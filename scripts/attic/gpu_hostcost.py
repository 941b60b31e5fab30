"""Host cost of collecting a batch whose kernels have long finished: hpk_collect_batch (C: Benjamini-Hochberg, result
assembly) and the Python wrapping of the results.  usage: gpu_hostcost.py [config] [group]"""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import bench
from hicpeaks_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else 'chr1_10kb'
group = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = bench.CONFIGS[name]
ctx = _lib.Context(0)
dev = torch.device('cuda', 0)
n, mw, D = cfg['n'], min(cfg['ww']), cfg['maxapart'] // cfg['res']
num = D + cfg['maxww'] + 1
ld = (num + 63) // 64 * 64
raw, weight, IR, biases, num = bench.make_band_host(cfg, seed=0)
raw_d = torch.zeros((n, ld), dtype=torch.float32, device=dev); raw_d[:, :num] = torch.from_numpy(raw.astype(np.float32)).to(dev)
w_d, ir_d, b_d = torch.from_numpy(weight).to(dev), torch.from_numpy(IR).to(dev), torch.from_numpy(biases).to(dev)
prm = _lib.make_params(_lib.MODE_HICCUPS, cfg['pw'], cfg['ww'], cfg['maxww'], 0.05, cfg['maxapart'], cfg['res'], 16, _lib.FLAG_NO_STENCIL_TIMING)
def submit():
    bd = [ctx._band(n, num, ld, raw_d.data_ptr(), None, w_d.data_ptr(), ir_d.data_ptr(), b_d.data_ptr(), b_d.data_ptr(), True) for _ in range(group)]
    return ctx.submit_batch(bd, prm, [n] * group)
for _ in range(3):
    submit().results()
ts, tc = [], []
for _ in range(10):
    t0 = time.perf_counter(); j = submit(); t1 = time.perf_counter()
    torch.cuda.synchronize(); time.sleep(0.02)
    t2 = time.perf_counter(); rs = j.results(); t3 = time.perf_counter()
    ts.append(t1 - t0); tc.append(t3 - t2)
print('%s group %d: submit %.3f ms, results() with the GPU idle %.3f ms = %.4f ms per chromosome; C-side host_bh %.4f ms per chromosome' % (
    name, group, np.median(ts) * 1e3, np.median(tc) * 1e3, np.median(tc) * 1e3 / group, np.mean([r.timing['host_bh'] for r in rs])))
