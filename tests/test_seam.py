"""The drop-in seam itself: `hiccups(M, cM, B1, B2, IR, chromLen, Diags, cDiags, num, chrom, ...)` and
`bhfdr(...)` called *positionally* with worker()-style inputs (scripts/pyHICCUPS:146-173, scripts/pyBHFDR:120-144):
scipy `sparse.diags` matrices, the list of raw diagonals, the list of balanced diagonals from min(ww) on and the IR
dict - rebuilt here from the fixtures exactly as the worker builds them from a cooler.

CPU part: `_bands_from_diags` turns (Diags, cDiags, IR) back into the dense bands the fixtures hold, the exception
classes travel through pickle (a Pool worker that raises must not hang its parent), --nproc to GPU mapping.
GPU part: the two entry points against the reference's final tables, and its crash edges."""
import pickle

import numpy as np
import pytest
from scipy import sparse

from conftest import load_golden
from hicpeaks_amd import _lib, callers, cli


def worker_inputs(g, extra_diags=0):
    """(M, cM, biases, IR, chromLen, Diags, cDiags, num) as scripts/pyHICCUPS:146-166 hands them over, from a fixture's
    band.  extra_diags > 0: a caller that keeps more diagonals than maxapart // res + maxww + 1."""
    num0 = g.meta['num']
    num = num0 + extra_diags
    raw = g['raw']
    n = raw.shape[0]
    assert raw.shape[1] >= num
    mw = g.mw
    Diags = [raw[:n - i, i].astype(np.int64) for i in range(num)]
    M = sparse.diags(Diags, np.arange(num), format='csr')
    w = g['weight']
    IR, cDiags = {}, []
    for i in range(mw, num):
        cnt = raw[:n - i, i].astype(np.float64)
        diag = (cnt * w[:n - i]) * w[i:]
        diag[cnt == 0] = 0.0                  # unstored pixels are 0 even in masked bins (COO .diagonal())
        mask = np.isnan(diag)
        IR[i] = diag[~mask].mean()
        diag[mask] = 0
        cDiags.append(diag)
    cM = sparse.diags(cDiags, np.arange(mw, num), format='csr')
    ok = ~((w == 0) | np.isnan(w))
    biases = np.zeros_like(w)
    biases[ok] = 1 / w[ok]
    return M, cM, biases, IR, n, Diags, cDiags, num


def _table(t):
    k = sorted(t)
    if not k:
        return np.zeros((0, 2), np.int64), np.zeros((0, 0))
    return np.array(k, dtype=np.int64), np.array([[float(v) for v in t[x]] for x in k])


# ----------------------------------------------------------------------------- CPU
@pytest.mark.parametrize('name', ['hiccups_union_shallow', 'hiccups_p2w5', 'bhfdr_p2w5', 'hiccups_nonan'])
def test_bands_from_diags_rebuilds_the_fixture_bands(name):
    g = load_golden(name)
    M, cM, biases, IR, n, Diags, cDiags, num = worker_inputs(g)
    raw, bal, IRa = callers._bands_from_diags(Diags, cDiags, IR, n, num, g.mw)
    assert raw.dtype == np.float32 and raw.shape == (n, num) and bal.shape == (n, num)
    np.testing.assert_array_equal(raw, g['raw'][:, :num])
    np.testing.assert_array_equal(bal, g['cband'])                  # the reference's own cDiags, NaN already zeroed
    np.testing.assert_array_equal(IRa[g.mw:], g['IR'])
    assert not IRa[:g.mw].any() and not bal[:, :g.mw].any()
    np.testing.assert_array_equal(biases, g['biases'])
    # the sparse matrices the seam also receives hold the same numbers
    np.testing.assert_array_equal(np.asarray(M[np.arange(n - 7), np.arange(7, n)]).ravel(), raw[:n - 7, 7])
    np.testing.assert_array_equal(np.asarray(cM[np.arange(n - g.mw), np.arange(g.mw, n)]).ravel(), bal[:n - g.mw, g.mw])


def test_exceptions_survive_pickling():
    for cls in (_lib.HpkError, _lib.EmptyStepError):
        e = pickle.loads(pickle.dumps(cls(-4, 'step (1,4) entered with no unresolved candidate')))
        assert type(e) is cls and e.status == -4 and 'step (1,4)' in str(e) and e.msg.startswith('step')
    assert isinstance(pickle.loads(pickle.dumps(_lib.EmptyStepError(-4, 'x'))), (ValueError, ZeroDivisionError))


def _raise_in_worker(status):
    raise _lib.EmptyStepError(status, 'raised in a pool worker')


def test_exception_from_a_pool_worker_reaches_the_parent():
    """Before the classes were picklable this hung the parent: run under a timeout."""
    import multiprocessing as mp
    with mp.get_context('spawn').Pool(2) as pool:
        res = pool.map_async(_raise_in_worker, [-4, -4])
        with pytest.raises(_lib.EmptyStepError) as ei:
            res.get(timeout=60)
    assert ei.value.status == -4


def test_nproc_maps_onto_the_gpus_present():
    # the reference's --nproc counts CPU processes (scripts/pyHICCUPS:192-198); a worker here needs a GPU
    assert cli.worker_devices(8, None, 2) == (2, [0, 1])
    assert cli.worker_devices(2, None, 8) == (2, [0, 1])
    assert cli.worker_devices(8, 3, 8) == (1, [3])              # --device: one GPU, one worker
    assert cli.worker_devices(1, None, 8) == (1, [0])
    assert cli.worker_devices(4, None, 0)[0] == 4                # no GPU visible: hpk_create reports it, loudly


def test_unequal_pw_ww_lengths():
    # zip() truncation is reproduced when the surplus entries do not hold the minimum (callers.py:18 vs 58, 102)
    p = _lib.make_params(_lib.MODE_HICCUPS, [1, 2, 4], [3, 5], 10, 0.05, 2000000, 10000)
    assert p.npairs == 2 and list(p.pw)[:2] == [1, 2]
    with pytest.raises(_lib.HpkError):
        _lib.make_params(_lib.MODE_HICCUPS, [1, 2], [5, 7, 3], 10, 0.05, 2000000, 10000)    # min(ww) sits in the surplus


# ----------------------------------------------------------------------------- GPU
HIC_KW = ('pw', 'ww', 'maxww', 'sig', 'sumq', 'double_fold', 'single_fold', 'maxapart', 'res', 'use_raw',
          'min_marginal_peaks', 'onlyanchor', 'min_local_reads')
BH_KW = ('pw', 'ww', 'sig', 'maxww', 'maxapart', 'res', 'min_marginal_peaks', 'onlyanchor')


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['hiccups_union_shallow', 'hiccups_p2w5', 'hiccups_p1w3_short', 'hiccups_useraw_anchor',
                                  'hiccups_defaults_kw', 'hiccups_swapped_pairs', 'hiccups_w8_pairdrop'])
def test_hiccups_positional_seam(name):
    g = load_golden(name)
    M, cM, biases, IR, n, Diags, cDiags, num = worker_inputs(g)
    p = g.params
    final = callers.hiccups(M, cM, biases, biases, IR, n, Diags, cDiags, num, 'T', **{k: p[k] for k in HIC_KW})
    k, v = _table(final)
    np.testing.assert_array_equal(k, g['final_keys'])
    if k.size:
        np.testing.assert_allclose(v, g['final_vals'], rtol=1e-9, atol=1e-9)
    # the text of scripts/pyHICCUPS:200-210 through the positional seam: the reference's own lines (a '%.3g' field whose
    # value sits on a rounding boundary of its third digit may differ by one unit of that digit)
    from test_gpu_cli import _numeric_equal
    _numeric_equal(sorted(cli.format_hiccups('T', final, p['res'], sort=True).splitlines()), sorted(g.meta['lines'].splitlines()))


@pytest.mark.gpu
def test_hiccups_seam_with_reference_keyword_defaults():
    """No keywords at all: the defaults of callers.py:44-46 (maxww=20, sig=0.1, maxapart=2000000, res=10000, ...)."""
    g = load_golden('hiccups_defaults_kw')
    p = g.params
    M, cM, biases, IR, n, Diags, cDiags, num = worker_inputs(g)
    # the fixture was produced with maxapart=300000: everything else is the reference's default
    final = callers.hiccups(M, cM, biases, biases, IR, n, Diags, cDiags, num, 'T', maxapart=p['maxapart'])
    k, v = _table(final)
    np.testing.assert_array_equal(k, g['final_keys'])
    if k.size:
        np.testing.assert_allclose(v, g['final_vals'], rtol=1e-9, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['bhfdr_p2w5', 'bhfdr_shallow', 'bhfdr_w20'])
def test_bhfdr_positional_seam(name):
    g = load_golden(name)
    M, cM, biases, IR, n, Diags, cDiags, num = worker_inputs(g)
    p = g.params
    final = callers.bhfdr(M, cM, biases, biases, IR, n, Diags, cDiags, num, 'T', **{k: p[k] for k in BH_KW})
    k, v = _table(final)
    np.testing.assert_array_equal(k, g['final_keys'])
    if k.size:
        np.testing.assert_allclose(v, g['final_vals'], rtol=1e-9, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['hiccups_empty', 'hiccups_exhausted_pi'])
def test_seam_raises_where_the_reference_raises(name):
    g = load_golden(name)
    assert 'exception' in g.meta
    M, cM, biases, IR, n, Diags, cDiags, num = worker_inputs(g)
    p = g.params
    with pytest.raises((ValueError, ZeroDivisionError)):
        callers.hiccups(M, cM, biases, biases, IR, n, Diags, cDiags, num, 'T', **{k: p[k] for k in HIC_KW})


@pytest.mark.gpu
def test_seam_with_more_diagonals_than_the_command_line_keeps():
    """A caller may hand over num > maxapart // res + maxww + 1 diagonals.  The reference then (i) lets windows near
    d = D reach the extra diagonals and (ii) sums every stored diagonal for the gap rows (callers.py:238).  Checked
    against the oracle on the same wide band: survivors, gap rows and the final table."""
    from oracle import hiccups_oracle as orc
    from hicpeaks_amd import synthetic
    n, res, maxww, D, extra = 500, 10000, 10, 40, 25
    num = D + maxww + 1 + extra
    raw, weight, _ = synthetic.synth_band(n, num, depth=40.0, nloops=12, seed=5)
    # rows whose only signal sits on the extra diagonals: gap rows for the command line, live rows here
    r0 = 200
    raw[r0 - 3:r0 + 4, :D + maxww + 1] = 0
    raw[r0 - 3:r0 + 4, D + maxww + 3] = 7
    pw, ww = [2], [5]
    IRo, cband, biases = orc.prep_from_band(raw, weight, min(ww))
    assert cband[r0, :D + maxww + 1].sum() == 0 and cband[r0].sum() > 0
    want = orc.hiccups(raw, cband, biases, biases, IRo, n, num, pw=pw, ww=ww, maxww=maxww, sig=0.1, maxapart=D * res,
                       res=res, min_local_reads=16, min_marginal_peaks=2, onlyanchor=False)
    Diags = [raw[:n - i, i] for i in range(num)]
    cDiags = [cband[:n - i, i] for i in range(min(ww), num)]
    IR = {i: IRo[i] for i in range(min(ww), num)}
    M = sparse.diags(Diags, np.arange(num), format='csr')
    cM = sparse.diags(cDiags, np.arange(min(ww), num), format='csr')
    detail = {}
    got = callers.hiccups(M, cM, biases, biases, IR, n, Diags, cDiags, num, 'T', pw=pw, ww=ww, maxww=maxww, sig=0.1,
                          maxapart=D * res, res=res, min_local_reads=16, min_marginal_peaks=2, onlyanchor=False,
                          detail=detail)
    np.testing.assert_array_equal(detail['result'].gap, cband.sum(axis=1) == 0)
    k, v = _table(got)
    kw, vw = _table(want)
    np.testing.assert_array_equal(k, kw)
    if k.size:
        np.testing.assert_allclose(v, vw, rtol=1e-9, atol=1e-9)


@pytest.mark.gpu
def test_hicpeaks_amd_before_torch_in_a_fresh_process():
    """`import hicpeaks_amd` and a scored chromosome first, torch's first touch of the GPU afterwards (round 5: torch then no longer
    found the device; _lib._share_torch_hip_runtime)."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import numpy as np\n'
            'from hicpeaks_amd import _lib, synthetic, band\n'
            'raw, w, _ = synthetic.synth_band(900, 211, depth=60.0, nloops=10, seed=1)\n'
            'IR, b = band.expected_and_biases(raw, w, 5)\n'
            'c = _lib.Context(0)\n'
            'prm = _lib.make_params(_lib.MODE_HICCUPS, [2], [5], 10, 0.05, 2000000, 10000, 16, 0)\n'
            'R = c.score_host(raw.astype(np.float32), IR, b, b, prm, weight=w)\n'
            'assert R.ncand > 1000\n'
            'import torch\n'
            'assert torch.cuda.is_available()\n'
            'x = torch.arange(10, device="cuda").sum().item()\n'
            'assert x == 45\n'
            'R2 = c.score_host(raw.astype(np.float32), IR, b, b, prm, weight=w)\n'
            'assert R2.ncand == R.ncand\n'
            'print("ok")\n')
    r = subprocess.run([sys.executable, '-c', code], cwd=repo, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0 and r.stdout.decode().split()[-1] == 'ok', r.stderr.decode()[-3000:]
