"""Back-end #0 of the C ABI - hpk_create(-1): the path on host threads (hicpeaks_amd/csrc/hpk_cpu.cpp; SURVEY.md §7 steps 2-3,
§8-B2 / D4(ii)) - against the same evidence as the GPU path, on a machine without a GPU:

  * the 24 fixtures of the real reference (tests/golden/*.npz: widening log, candidates, E / O / p / q of every set, family sizes,
    gap rows, final tables, the three crash edges), weights and f64 balanced input,
  * the reference above fixture size (tests/golden/ref_*.npz: configs[0]'s shape, chr21 @10 kb as one pair / union / bhfdr / with
    structure, chr1 @10 kb at full size - configs[1] -, the 2 011-diagonal band), down to the printed text,
  * IR and biases derived by the library (scripts/pyHICCUPS:149-166), a batch, the options that do not apply.

It is a back-end one asks for (device -1) - nothing selects it when a GPU is missing: `Context(0)` still raises there.
"""
import os

import numpy as np
import pytest

import refbig
from conftest import golden_names, load_golden
from hicpeaks_amd import _lib, callers
from oracle import hiccups_oracle as orc
from test_gpu_parity import _call, _check_golden
from test_gpu_ref_big import _check_result, _kw


@pytest.fixture(scope='module')
def cpu():
    c = _lib.Context(-1)
    yield c
    c.close()


def test_it_is_only_there_when_asked_for(cpu):
    info = cpu.info()
    assert 'back-end #0' in info['name'] and info['cus'] == 0
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(_lib.HpkError):              # no device: the default context does not turn into a CPU one
            _lib.Context(0)
    # what belongs to the device path is refused, not emulated
    for flag in (_lib.FLAG_DENSE_E, _lib.FLAG_DENSE_SUMS):
        g = load_golden(golden_names('hiccups')[0])
        prm = _lib.make_params(_lib.MODE_HICCUPS, [2], [5], 10, 0.05, 2000000, 10000, 16, flag)
        with pytest.raises(_lib.HpkError):
            cpu.score_host(g['raw'][:, :g.meta['num']].astype(np.float32), None, None, None, prm, weight=g['weight'])
    cpu.set_option('cpu_threads', 3)
    cpu.set_option('cpu_threads', max(1, os.cpu_count() or 1))


@pytest.mark.parametrize('mode', ['weight', 'balanced'])
@pytest.mark.parametrize('name', golden_names())
def test_golden_parity_on_host_threads(name, mode, cpu):
    g = load_golden(name)
    if 'prep_exception' in g.meta:
        pytest.skip('reference prep raised')
    detail = {}
    if 'exception' in g.meta:
        with pytest.raises((ValueError, ZeroDivisionError)):
            _call(g, cpu, mode, detail)
        return
    final = _call(g, cpu, mode, detail)
    R = detail['result']
    assert R.stencil_kernel == 0 and not R.redone and R.record_bound == 255
    _check_golden(g, R, final)


@pytest.mark.parametrize('name', refbig.names())
def test_reference_at_size_on_host_threads(name, cpu):
    g = refbig.load(name)
    raw, weight = refbig.band(g)
    IR, cband, biases = orc.prep_from_band(raw, weight, g.mw)
    rawf = raw.astype(np.float32)
    call = callers.hiccups_band if g.mode == 'hiccups' else callers.bhfdr_band
    d = {}
    final = call(rawf, IR, biases, biases, chrom='T', weight=weight, ctx=cpu, detail=d, **_kw(g))
    _check_result(g, d['result'], final)
    # IR and the biases left to the library: the same answer
    d2 = {}
    fin2 = call(rawf, None, None, None, chrom='T', weight=weight, ctx=cpu, detail=d2, **_kw(g))
    _check_result(g, d2['result'], fin2)


def test_batch_and_submit_collect_on_host_threads(cpu):
    names = [n for n in golden_names('hiccups') if 'exception' not in load_golden(n).meta and 'prep_exception' not in load_golden(n).meta]
    g = load_golden(names[0])
    p = g.params
    same = [n for n in names if load_golden(n).params == p][:3] or names[:1]
    prm = _lib.make_params(_lib.MODE_HICCUPS, p['pw'], p['ww'], p['maxww'], p['sig'], p['maxapart'], p['res'], p['min_local_reads'], 0)
    items = []
    for n in same:
        gg = load_golden(n)
        items.append(dict(raw=np.ascontiguousarray(gg['raw'][:, :gg.meta['num']].astype(np.float32)), weight=gg['weight'], num=gg.meta['num']))
    Rs = cpu.submit_batch_host(items, prm).results()
    assert len(Rs) == len(same) and all(R.batch_bands == len(same) for R in Rs)
    for n, R in zip(same, Rs):
        gg = load_golden(n)
        assert R.ncand == gg.meta['ncand']
        assert [(a, b, c) for a, b, c, ex in R.steps if ex] == [tuple(int(v) for v in s) for s in gg['steps']]
        np.testing.assert_array_equal(R.gap, gg['cband'].sum(axis=1) == 0)


def test_poisson_sf_on_host_threads_matches_scipy(cpu):
    from scipy import stats
    k = np.array([0, 1, 5, 30, 100, 1000, 3, 70000], dtype=np.float64)
    lam = np.array([0.5, 1.0, 2.5, 10.0, 120.0, 900.0, 40.0, 69000.0])
    got = cpu.poisson_sf(k, lam)
    want = 1.0 - stats.poisson(lam).cdf(k)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
