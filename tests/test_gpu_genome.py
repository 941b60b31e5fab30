"""BASELINE.json configs[2] and configs[3] as GENOMES (the batch machinery - depth classes, per-band tile geometry, lean
prediction, redo queue across 23 bands of 23 sizes at full size - had met the oracle only on batches of three).

The 23 hg38 chromosomes (chr1-22, X: the command lines' default --chroms, scripts/pyHICCUPS:184-190) at their real bin counts go
through ONE hpk_submit_batch under the mode the command lines ship (spec_halo = 2), in a context that has scored other
chromosomes before:

  * every chromosome's result is bit-identical to its own single call in a context without history (its E / p / q are a function
    of the chromosome alone: DESIGN 6),
  * the cheapest chromosomes against the oracle - hicpeaks/callers.py:44-362 restated, pinned to the reference by the fixtures -
    down to the final table: chr19-22 and X at 10 kb under the union plan (1,3)/(2,5)/(4,7), chr21 and chr22 at 5 kb under (4,7)
    with a 10 Mb band (2 011 diagonals).  The oracle runs on host cores of its own while the GPU works (a process pool: chr21 @5 kb
    alone is ~50 s of one core).
"""
import concurrent.futures as cf
import multiprocessing as mp

import numpy as np
import pytest

from hicpeaks_amd import _lib, callers, synthetic
from genome_oracle import GENOMES, MAXWW, MIN_READS, SIG, _oracle_job
from test_gpu_fullsize import _check_against_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def oracle_jobs():
    """every oracle chromosome of both genomes, started at once on cores of their own"""
    pool = cf.ProcessPoolExecutor(max_workers=8, mp_context=mp.get_context('spawn'))
    futs = {}
    for name, cfg in GENOMES.items():
        sizes = synthetic.hg38_bins(cfg['res'])
        order = sorted(sizes, key=lambda k: -sizes[k])
        for c in cfg['oracle']:
            futs[(name, c)] = pool.submit(_oracle_job, (name, c, sizes[c], 9000 + order.index(c)))
    yield futs
    pool.shutdown(wait=False, cancel_futures=True)


def _same(a, b):
    """two results of one chromosome: identical to the last bit"""
    assert a.ncand == b.ncand and a.frozen_w == b.frozen_w and list(a.steps) == list(b.steps)
    np.testing.assert_array_equal(a.gap, b.gap)
    assert len(a.sets) == len(b.sets)
    for s, t in zip(a.sets, b.sets):
        assert s['nvalid'] == t['nvalid']
        np.testing.assert_array_equal(s['chunk_tests'], t['chunk_tests'])
        oa, ob = np.lexsort((s['y'], s['x'])), np.lexsort((t['y'], t['x']))
        for k in ('x', 'y', 'O', 'E', 'p', 'q'):
            np.testing.assert_array_equal(s[k][oa], t[k][ob], err_msg=k)


@pytest.mark.parametrize('name', sorted(GENOMES))
def test_genome_in_one_batch(name, oracle_jobs):
    import torch
    from hicpeaks_amd import bandgen
    cfg = GENOMES[name]
    res, pw, ww = cfg['res'], cfg['pw'], cfg['ww']
    mw, D = min(ww), cfg['maxapart'] // res
    num = D + MAXWW + 1
    ld = (num + 63) // 64 * 64
    dev = torch.device('cuda', 0)
    sizes = synthetic.hg38_bins(res)
    order = sorted(sizes, key=lambda k: -sizes[k])
    assert len(order) == 23
    prm = _lib.make_params(_lib.MODE_HICCUPS, pw, ww, MAXWW, SIG, cfg['maxapart'], res, MIN_READS, 0)
    # the bands: the oracle's chromosomes are the host recipe's (uploaded as they are), the others are generated in HBM
    bands, host = {}, {}
    for i, c in enumerate(order):
        n = sizes[c]
        if c in cfg['oracle']:
            continue
        raw_d, w_d, _, _ = bandgen.device_band(n, num, ld, mw, depth=cfg['depth'], nloops=max(1, cfg['nloops'] * n // cfg['n_ref']),
                                               seed=9000 + i, device=dev, want_expected=False)
        bands[c] = (n, raw_d, w_d)
    for c in cfg['oracle']:
        _, _, rawf, weight, det, want = oracle_jobs[(name, c)].result(timeout=600)
        n = sizes[c]
        assert rawf.shape == (n, num)
        raw_d = torch.zeros((n, ld), dtype=torch.float32, device=dev)
        raw_d[:, :num] = torch.from_numpy(rawf).to(dev)
        bands[c] = (n, raw_d, torch.from_numpy(weight).to(dev))
        host[c] = (det, want)
    torch.cuda.synchronize()

    def desc(c_, c):
        n, r, w = bands[c]
        return c_._band(n, num, ld, r.data_ptr(), None, w.data_ptr(), None, None, None, True)

    ctx = _lib.Context(0)
    lone = _lib.Context(0)
    try:
        ctx.set_option('spec_halo', 2)
        lone.set_option('spec_halo', 2)
        # history: a deep and a shallow chromosome of another shape (other depth class, other frozen width)
        for dp, sd in ((4.0 * cfg['depth'], 1), (0.25 * cfg['depth'], 2)):
            r_, w_, _, _ = bandgen.device_band(6000, num, ld, mw, depth=dp, nloops=50, seed=sd, device=dev, want_expected=False)
            ctx.score(ctx._band(6000, num, ld, r_.data_ptr(), None, w_.data_ptr(), None, None, None, True), prm, 6000)
        Rs = ctx.submit_batch([desc(ctx, c) for c in order], prm, [sizes[c] for c in order]).results()
        assert len(Rs) == 23 and all(R.batch_bands == 23 for R in Rs)
        # (nothing is left under an inherited layout: every chromosome ran - or ran once more - under its own frozen width's)
        assert all(R.halo_w == min(MAXWW, max(R.frozen_w, mw, 4)) for R in Rs), [(R.halo_w, R.frozen_w) for R in Rs]
        # ---- each chromosome on its own, without history
        for k, (c, R) in enumerate(zip(order, Rs)):
            if k % 8 == 0:
                lone.close()
                lone = _lib.Context(0)                  # a truly fresh context now and then, forgotten hints otherwise
                lone.set_option('spec_halo', 2)
            else:
                lone.set_option('reset_hints', 1)
            _same(R, lone.score(desc(lone, c), prm, sizes[c]))
        # ---- the cheapest ones against the oracle, down to the final table
        nfinal = 0
        for c in cfg['oracle']:
            det, want = host[c]
            R = Rs[order.index(c)]
            final, _ = callers._finish_hiccups(R, sizes[c], c, pw, ww, SIG, 0.01, 1.75, 2, res, False, 2, False)
            _check_against_oracle(R, final, det, want, pw, ww, SIG, min_sig=cfg['min_sig'], min_final=0)
            nfinal += len(want)
        assert nfinal >= cfg['min_final']
    finally:
        ctx.close()
        lone.close()
