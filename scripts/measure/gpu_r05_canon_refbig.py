#!/usr/bin/env python
"""The fixtures above fixture size (tests/golden/ref_*.npz: the REAL reference's outputs) under option spec_halo = 2, in a context that scored
other chromosomes before - the same checks as tests/test_gpu_ref_big.py::test_reference_at_size (survivors, final table, printed text), which
runs them under the library's default."""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import refbig                                       # noqa: E402
import test_gpu_ref_big as T                        # noqa: E402
from hicpeaks_amd import _lib, callers, synthetic   # noqa: E402
from oracle import hiccups_oracle as orc            # noqa: E402

ok = 0
for name in refbig.names():
    g = refbig.load(name)
    raw, weight = refbig.band(g)
    n, num = raw.shape
    IR, cband, biases = orc.prep_from_band(raw, weight, g.mw)
    rawf = raw.astype(np.float32)
    kw = T._kw(g)
    call = callers.hiccups_band if g.mode == 'hiccups' else callers.bhfdr_band
    c = _lib.Context(0)
    c.set_option('spec_halo', 2)
    try:
        for depth, seed in ((150.0, 5), (8.0, 6)):          # history: a deep and a shallow chromosome
            other, ow, _ = synthetic.synth_band(max(num + 40, 900), num, depth=depth, nloops=10, seed=seed)
            oIR, _, ob = orc.prep_from_band(other, ow, g.mw)
            call(other.astype(np.float32), oIR, ob, ob, chrom='o', weight=ow, ctx=c, **kw)
        d = {}
        final = call(rawf, IR, biases, biases, chrom='T', weight=weight, ctx=c, detail=d, **kw)
        Rr = d['result']
        mw = min(g.params['ww']) if isinstance(g.params['ww'], (list, tuple)) else g.params['ww']
        assert Rr.halo_w == min(g.params['maxww'], max(Rr.frozen_w, mw, 4)), (Rr.halo_w, Rr.frozen_w)
        T._check_result(g, Rr, final)
        print('%-28s ok: frozen %d halo %d redone %d lean %d/%d' % (name, Rr.frozen_w, Rr.halo_w, int(Rr.redone), Rr.lean_tiles, Rr.tiles))
        ok += 1
    finally:
        c.close()
print('%d of %d fixtures equal the reference under spec_halo = 2' % (ok, len(refbig.names())))
