"""End to end through the command lines on the GPU: band archive in, 16-/13-column text out, compared with the
lines the real reference produced for the same input."""
import os

import numpy as np
import pytest

from conftest import load_golden
from hicpeaks_amd import cli, io

pytestmark = pytest.mark.gpu


def _lines(path):
    return sorted(open(path).read().splitlines())


def _numeric_equal(got, want):
    """The text itself: coordinates and counts identical, and every '%.3g' field the same string as the reference's -
    except where a value sits on a rounding boundary of its third digit (the device's sums agree with the reference's
    to ~1e-12, which can still tip such a value): those may differ by one unit of that digit, and there must be
    next to none of them."""
    assert len(got) == len(want), (len(got), len(want))
    fields = tipped = 0
    for g, w in zip(got, want):
        gf, wf = g.split('\t'), w.split('\t')
        assert gf[:7] == wf[:7] and gf[8:10] == wf[8:10]
        for a, b in zip(gf[7:8] + gf[10:], wf[7:8] + wf[10:]):
            fields += 1
            if a == b:
                continue
            tipped += 1
            fb = float(b)
            ulp3 = 10.0 ** (np.floor(np.log10(abs(fb))) - 2) if fb != 0.0 else 0.0
            assert abs(float(a) - fb) <= 1.01 * ulp3, (g, w)
    assert tipped <= max(1, fields // 500), (tipped, fields)


@pytest.mark.parametrize('name', ['hiccups_union_shallow', 'hiccups_p2w5'])
def test_pyhiccups_cli(name, tmp_path):
    g = load_golden(name)
    p = g.params
    arc = str(tmp_path / 'in.npz')
    io.save_band_archive(arc, p['res'], {'chrT': (g['raw'], g['weight'])})
    out = str(tmp_path / 'out.bedpe')
    argv = ['-O', out, '-p', arc, '-C', 'T', '--pw'] + [str(v) for v in p['pw']] + ['--ww'] + [str(v) for v in p['ww']] + [
        '--maxww', str(p['maxww']), '--siglevel', str(p['sig']), '--sumq', str(p['sumq']), '--maxapart', str(p['maxapart']),
        '--min-marginal-peaks', str(p['min_marginal_peaks']), '--min-local-reads', str(p['min_local_reads']),
        '--logFile', str(tmp_path / 'log.txt')]
    assert cli.main_hiccups(argv) == 0
    _numeric_equal(_lines(out), sorted(g.meta['lines'].splitlines()))


def test_pybhfdr_cli(tmp_path):
    g = load_golden('bhfdr_p2w5')
    p = g.params
    arc = str(tmp_path / 'in.npz')
    io.save_band_archive(arc, p['res'], {'chrT': (g['raw'], g['weight'])})
    out = str(tmp_path / 'out.txt')
    argv = ['-O', out, '-p', arc, '-C', 'T', '--pw', str(p['pw']), '--ww', str(p['ww']), '--maxww', str(p['maxww']),
            '--siglevel', str(p['sig']), '--maxapart', str(p['maxapart']), '--logFile', str(tmp_path / 'log.txt')]
    assert cli.main_bhfdr(argv) == 0
    # pyBHFDR calls bhfdr() with its keyword defaults for clustering (min_marginal_peaks=3, onlyanchor=False)
    _numeric_equal(_lines(out), sorted(g.meta['lines'].splitlines()))


def test_pyhiccups_cli_several_chromosomes_one_ahead(tmp_path):
    """Three chromosomes through the one-ahead loop of the command line (hpk_submit_band / hpk_collect): every
    chromosome's lines equal the reference's lines for that band, in the order of --chroms."""
    ga, gb = load_golden('hiccups_p2w5'), load_golden('hiccups_p2w5_shallow')
    pa, pb = ga.params, gb.params
    same = all(pa[k] == pb[k] for k in ('pw', 'ww', 'maxww', 'sig', 'sumq', 'maxapart', 'res', 'min_local_reads',
                                        'min_marginal_peaks'))
    if not same:
        gb = ga
    arc = str(tmp_path / 'in.npz')
    io.save_band_archive(arc, pa['res'], {'chr1': (ga['raw'], ga['weight']), 'chr2': (gb['raw'], gb['weight']),
                                          'chr3': (ga['raw'], ga['weight'])})
    out = str(tmp_path / 'out.bedpe')
    argv = ['-O', out, '-p', arc, '-C', '1', '2', '3', '--pw'] + [str(v) for v in pa['pw']] + ['--ww'] + [
        str(v) for v in pa['ww']] + ['--maxww', str(pa['maxww']), '--siglevel', str(pa['sig']), '--sumq', str(pa['sumq']),
        '--maxapart', str(pa['maxapart']), '--min-marginal-peaks', str(pa['min_marginal_peaks']),
        '--min-local-reads', str(pa['min_local_reads']), '--logFile', str(tmp_path / 'log.txt')]
    assert cli.main_hiccups(argv) == 0
    got = open(out).read().splitlines()
    order = [l.split('\t')[0] for l in got]
    assert order == sorted(order, key=lambda c: int(c[3:]))            # chromosomes stay in --chroms order
    for name, g in (('chr1', ga), ('chr2', gb), ('chr3', ga)):
        want = sorted(l.replace('chrT', name) for l in g.meta['lines'].splitlines())
        _numeric_equal(sorted(l for l in got if l.startswith(name + '\t')), want)


def test_pyhiccups_cli_reads_a_cooler_file(tmp_path):
    """`-p file.cool` (scripts/pyHICCUPS:178-179): the cooler-format fixture through the package's own reader, every
    chromosome (one of them shorter than the band) - the text equals what the same bands give as a band archive; with the
    divisive 'KR' column the balanced band is the same and the biases are the reference's 1 / column."""
    import os
    from conftest import GOLDEN_DIR
    from hicpeaks_amd import callers, band as hband
    cool = os.path.join(GOLDEN_DIR, 'tiny.cool')
    try:
        src = io.open_source(cool)
    except SystemExit:
        pytest.skip('neither h5py nor libhdf5 here')
    num = 500000 // 10000 + 10 + 1
    bands = {c: src.fetch(c, num)[:2] for c in src.chromnames}
    arc = str(tmp_path / 'same.npz')
    io.save_band_archive(arc, 10000, bands)
    outs = {}
    for tag, path in (('cool', cool), ('npz', arc)):
        out = str(tmp_path / (tag + '.bedpe'))
        argv = ['-O', out, '-p', path, '-C', 'A', 'B', 'C', '--pw', '2', '--ww', '5', '--maxapart', '500000', '--siglevel', '0.1',
                '--logFile', str(tmp_path / 'log.txt')]
        assert cli.main_hiccups(argv) == 0
        outs[tag] = open(out).read()
    assert outs['cool'] == outs['npz'] and len(outs['cool'].splitlines()) >= 3
    assert [l.split('\t')[0] for l in outs['cool'].splitlines()] == sorted(l.split('\t')[0] for l in outs['cool'].splitlines())
    # the divisive column: IR derived on the device from 1 / KR, biases given (1 / KR as stored) == everything given
    raw, wkr, bkr = src.fetch('chrA', num, weight_name='KR')
    assert bkr is not None
    from hicpeaks_amd import _lib
    ctx = _lib.default_context(0)
    prm = _lib.make_params(_lib.MODE_HICCUPS, [2], [5], 10, 0.1, 500000, 10000, 16, 0)
    IR, _ = hband.expected_and_biases(raw, wkr, 5)
    want = ctx.score_host(raw, IR, bkr, bkr, prm, weight=wkr)
    got = ctx.submit_batch_host([dict(raw=raw, weight=wkr, bias1=bkr, bias2=bkr)], prm).results()[0]
    assert got.ncand == want.ncand and got.steps == want.steps
    for sg, sw in zip(got.sets, want.sets):
        assert sg['nvalid'] == sw['nvalid'] > 1000
        np.testing.assert_allclose(sg['emax'], sw['emax'], rtol=1e-12)
        np.testing.assert_array_equal(sg['x'], sw['x'])
    # (with biases = 1 / KR = w the corrected expected is w^2-small: the reference's rule for a divisive column, as coded)
    assert want.sets[0]['emax'] < 1.0
