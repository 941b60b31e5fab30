#!/usr/bin/env python
"""Widths of one fuzz case, HIP path next to the oracle (diagnosis of a MISMATCH-widths).  usage: gpu_fuzz_dump.py seed"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hicpeaks_amd import _lib, callers, synthetic
from oracle import hiccups_oracle as orc
seed = int(sys.argv[1])
# the case of seed 1300317 (scripts/gpu_fuzz.py draws): kept literal so that the dump does not depend on the generator
n, D, maxww, pw, ww, depth, min_reads, sig, res = 25, 15, 6, [0, 3], [6, 7], 25.0, 16, 0.1, 10000
rng = np.random.default_rng(seed)
num = D + maxww + 1
# replay the generator's draws up to the band (same order as one_case)
_ = rng.integers(3, 21); npairs = int(rng.integers(1, 4)); _ = rng.integers(2, maxww + 3, npairs)
raw = None
import importlib.util
spec = importlib.util.spec_from_file_location('fz', os.path.join(os.path.dirname(__file__), 'gpu_fuzz.py'))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
# simplest: monkeypatch one_case's comparison by re-running its body up to the inputs
src = open(os.path.join(os.path.dirname(__file__), 'gpu_fuzz.py')).read()
ctx = _lib.Context(0)
import types
ns = {}
body = src[src.index('def one_case'):src.index('def main')]
body = body.replace("            if not np.array_equal(w, loc['wres'][pi]):", "            print('slot', slot, 'pi', pi, 'steps', R.steps, 'frozen', R.frozen_w)\n            bad = np.nonzero(w != loc['wres'][pi])[0]\n            print('raw dense_w', R.dense_w[slot][vx, vy - vx][bad].tolist()); print('gpu', w[bad].tolist()); print('orc', loc['wres'][pi][bad].tolist()); print('x', vx[bad].tolist(), 'y', vy[bad].tolist())\n            if not np.array_equal(w, loc['wres'][pi]):")
exec(compile(body, 'one_case', 'exec'), fz.__dict__)
print(fz.one_case(seed, ctx))
