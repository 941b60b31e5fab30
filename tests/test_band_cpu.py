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
