"""debug: the GPU slice's fuzz seeds one by one, printing every case that does not match"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpu_fuzz as fz
from hicpeaks_amd import _lib
ctx = _lib.Context(0)
crash7 = [100955, 101171, 101482, 101572, 101689, 101801, 101993, 102243, 102246, 102399, 102415, 102446]
seeds = list(range(300, 360)) + [1410599, 1300317] + crash7
if len(sys.argv) > 1:
    seeds = [int(a) for a in sys.argv[1:]]
for seed in seeds:
    status, desc, note = fz.one_case(seed, ctx)
    if status.startswith('MISMATCH'):
        print(seed, status, desc, note)
print('done')
