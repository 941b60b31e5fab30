import numpy as np
import pytest

from conftest import golden_names, load_golden
from hicpeaks_amd import band, synthetic


@pytest.mark.parametrize('name', golden_names()[:6])
def test_expected_and_biases_match_reference_prep(name):
    g = load_golden(name)
    if 'IR' not in g:
        pytest.skip('reference prep raised')
    num = g.meta['num']
    IR, biases = band.expected_and_biases(g['raw'][:, :num], g['weight'], g.mw)
    np.testing.assert_allclose(IR[g.mw:], g['IR'], rtol=1e-13, atol=0)
    np.testing.assert_array_equal(biases, g['biases'])


def test_band_from_coo_roundtrip():
    raw, w, _ = synthetic.synth_band(200, 31, depth=5.0, seed=3)
    i, j, v = synthetic.band_to_coo(raw)
    # feed both triangles in scrambled order
    rng = np.random.default_rng(0)
    o = rng.permutation(i.size)
    back = band.band_from_coo(j[o], i[o], v[o], 200, 31, dtype=np.int64)
    np.testing.assert_array_equal(back, raw)
    assert band.band_pixels(200, 31, 5, 20) == sum(200 - d for d in range(5, 21))


def test_bench_configurations_match_the_survey_table():
    """Band pixel counts of bench.py's configurations = SURVEY.md section 8 size table (d in [min(ww), D])."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location('bench', os.path.join(os.path.dirname(__file__), '..', 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from hicpeaks_amd import band, synthetic

    def px(cfg, n):
        D = cfg['maxapart'] // cfg['res']
        return band.band_pixels(n, D + cfg['maxww'] + 1, min(cfg['ww']), D)
    c = bench.CONFIGS
    assert px(c['chr1_10kb'], c['chr1_10kb']['n']) == 12223176
    assert px(c['deep_1kb'], c['deep_1kb']['n']) == 496999010
    assert px(c['chr1_5kb'], c['chr1_5kb']['n']) == 97284269
    g10 = sum(px(c['wg_10kb_union'], n) for n in synthetic.hg38_bins(10000).values())
    g5 = sum(px(c['wg_5kb'], n) for n in synthetic.hg38_bins(5000).values())
    assert g10 == 148070091 and g10 * 3 == 444210273
    assert g5 == 1162778169
    assert bench.BYTES_PER_PX == 20.0 and bench.HBM_PEAK_GBS == 8000.0
