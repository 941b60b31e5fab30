#!/usr/bin/env python
"""Diagnosis of a flagged bhfdr fuzz case: which pixels count as tests on either side (oracle: bE != 0, E > 0; HIP path:
the same from its dense sums), and their sums.  usage: gpu_fuzz_rows.py seed   (test infrastructure)"""
import os, sys
import numpy as np
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
src = open(os.path.join(ROOT, 'scripts', 'gpu_fuzz.py')).read()
body = src[:src.index('def main')]
hook = '''
    if mode == 'bhfdr':
        od = {}
        orc.bhfdr(raw, cband, biases, biases, IR, n, num, pw=pw[0], ww=ww[0], sig=sig, maxww=maxww, maxapart=maxapart, res=res,
                  min_marginal_peaks=2, onlyanchor=False, detail=od)
        R = detail['result']
        S = R.dense_sums[0]
        vx, vk = np.nonzero(raw); keep = (vk >= ww[0]) & (vk <= D); vx, vk = vx[keep], vk[keep]
        bS, bE = S[vx, vk, 0], S[vx, vk, 1]
        with np.errstate(all='ignore'):
            E = ((IR[vk] * (bS / bE)) * biases[vx]) * biases[vx + vk]
        hv = (bE != 0) & (E > 0)
        hip = set(zip(vx[hv].tolist(), (vx + vk)[hv].tolist()))
        orcs = set(zip(od['vx'].tolist(), od['vy'].tolist()))
        print('tests: hip', len(hip), 'oracle', len(orcs), 'steps hip', R.steps, 'frozen', R.frozen_w, 'oracle steps', od['steps'])
        ox = dict(zip(zip(od['x'].tolist(), od['y'].tolist()), od['ratio'].tolist()))
        for px in sorted(hip ^ orcs):
            i = np.nonzero((vx == px[0]) & (vx + vk == px[1]))[0][0]
            print(px, 'in', 'hip' if px in hip else 'oracle', 'hip bS', bS[i], 'bE', bE[i], 'w', R.dense_w[0][vx[i], vk[i]], 'oracle ratio', ox.get(px), 'raw', raw[vx[i], vk[i]], 'IR', IR[vk[i]], 'b', biases[px[0]], biases[px[1]])
'''
body = body.replace("    k, v = table_arrays(got)\n    kw, vw = table_arrays(want)\n", hook + "    k, v = table_arrays(got)\n    kw, vw = table_arrays(want)\n")
ns = {'__name__': 'fz', '__file__': os.path.join(ROOT, 'scripts', 'gpu_fuzz.py')}
exec(compile(body, 'fz', 'exec'), ns)
print(ns['one_case'](int(sys.argv[1]), ns['_lib'].Context(0))[0])
