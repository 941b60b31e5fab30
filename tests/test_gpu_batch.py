"""hpk_submit_batch / hpk_collect_batch: many chromosomes through one set of kernel launches (the stencil walks the tiles
of all of them in one persistent launch; expected tables, scoring, cut and copy-back are one launch each over the batch).
The contract is bit-identity with the chromosome-by-chromosome path, which the other GPU tests pin to the reference's
fixtures and to the oracle: every test here compares a batch with single calls on a fresh footing (`_same`).  The loop
being replaced is scripts/pyHICCUPS:192-198 (`map(worker, Params)`).  Needs an MI355X."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from hicpeaks_amd import _lib, synthetic, band as hband
from oracle import hiccups_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    c = _lib.Context(0)
    c.set_option('spec', 0)             # single calls and batches on the same footing: no record bound from earlier calls
    yield c
    c.close()


def _same(a, b):
    assert a.steps == b.steps and a.frozen_w == b.frozen_w and a.ncand == b.ncand and a.band_px == b.band_px
    assert len(a.sets) == len(b.sets)
    for sa, sb in zip(a.sets, b.sets):
        assert sa['nvalid'] == sb['nvalid'] and sa['numbin'] == sb['numbin'] and sa['emax'] == sb['emax']
        for k in ('x', 'y', 'O', 'bal', 'E', 'p', 'q', 'other_zero', 'chunk_tests', 'chunk_below'):
            np.testing.assert_array_equal(sa[k], sb[k])
    np.testing.assert_array_equal(a.gap, b.gap)
    assert a.nsurv_sig == b.nsurv_sig


def _chroms(sizes, num, depth, seed0, nloops=30):
    out = []
    for i, n in enumerate(sizes):
        raw, weight, _ = synthetic.synth_band(n, num, depth=depth, nloops=max(1, nloops * n // 3000), seed=seed0 + i)
        out.append((raw.astype(np.float32), weight))
    return out


SIZES = [3000, 700, 59, 1201, 2500, 118, 40, 1900]       # several tiles per XCD down to less than one tile; 40 < the band width


@pytest.mark.parametrize('mode,pw,ww', [('hiccups', [2], [5]), ('hiccups', [1, 2, 4], [3, 5, 7]), ('bhfdr', [2], [5])])
@pytest.mark.parametrize('inp', ['weight', 'derived', 'balanced'])
def test_batch_equals_single_calls(mode, pw, ww, inp, ctx):
    """Eight chromosomes of very different sizes in one batch: every field of every result equals the single call's.
    Inputs: weights with IR / biases given, weights only (IR / biases derived on the device, also batched), f64 balanced
    band."""
    res, maxapart, maxww = 10000, 1500000, 10
    num = maxapart // res + maxww + 1
    mw = min(ww)
    prm = _lib.make_params(_lib.MODE_HICCUPS if mode == 'hiccups' else _lib.MODE_BHFDR, pw, ww, maxww, 0.1, maxapart, res, 16, 0)
    items = []
    for raw, weight in _chroms(SIZES, num, 40.0, 100):
        if inp == 'derived':
            items.append(dict(raw=raw, weight=weight, num=num))
            continue
        # (not the oracle's prep: like the reference's, it raises on chromosomes shorter than the band, scripts/pyHICCUPS:148)
        IR, biases = hband.expected_and_biases(raw, weight, mw)
        IR = np.nan_to_num(IR)
        cband = synthetic.balanced_band(raw, weight, mw)
        if inp == 'weight':
            items.append(dict(raw=raw, IR=IR, bias1=biases, bias2=biases, weight=weight))
        else:
            items.append(dict(raw=raw, IR=IR, bias1=biases, bias2=biases, balanced=cband))
    want = []
    for it in items:
        try:
            want.append(ctx.score_host(it['raw'], it.get('IR'), it.get('bias1'), it.get('bias2'), prm, balanced=it.get('balanced'),
                                       weight=it.get('weight'), num=it.get('num')))
        except _lib.EmptyStepError as e:
            want.append(e)
    assert sum(not isinstance(w, Exception) for w in want) >= 5
    got = ctx.submit_batch_host(items, prm).results(raise_on_error=False)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        if isinstance(w, Exception):          # a chromosome on which the reference raises fails alone
            assert isinstance(g, _lib.EmptyStepError) and g.status == w.status
            continue
        assert g.batch_bands == len(items)
        _same(g, w)
    # two batches in flight on the two lanes, then a single call: the lanes' pooled workspaces do not leak into each other
    j1 = ctx.submit_batch_host(items[:5], prm)
    j2 = ctx.submit_batch_host(items[3:], prm)
    for g, w in zip(j2.results(raise_on_error=False), want[3:]):
        if not isinstance(w, Exception):
            _same(g, w)
    for g, w in zip(j1.results(raise_on_error=False), want[:5]):
        if not isinstance(w, Exception):
            _same(g, w)


def test_batch_of_one_band_many_times(ctx):
    """The same chromosome 40 times in one batch (what bench.py submits): all forty results are the single call's; the
    kernel times reported per chromosome add up to the batch's launches."""
    res, maxapart, maxww = 10000, 2000000, 10
    num = maxapart // res + maxww + 1
    (raw, weight), = _chroms([2400], num, 60.0, 7)
    prm = _lib.make_params(_lib.MODE_HICCUPS, [2], [5], maxww, 0.05, maxapart, res, 16, 0)
    want = ctx.score_host(raw, None, None, None, prm, weight=weight, num=num)
    got = ctx.submit_batch_host([dict(raw=raw, weight=weight, num=num)] * 40, prm).results()
    for g in got:
        _same(g, want)
    assert all(g.timing['stencil'] > 0 for g in got)
    assert sum(g.timing['stencil'] for g in got) < 40 * want.timing['stencil']        # one launch, one ramp-up, one tail


def test_eight_workgroups_walk_forty_chromosomes(ctx):
    """The stencil's tile walk with few workgroups (option grid_cap: one per XCD): every workgroup crosses every band boundary of a
    batch of 40 chromosome-sized bands - its resolve counts folded into LDS and published by one wave a tile later, the next band's
    first tile requested from the last tile of the one before - and walks more than 8 192 tiles, the length at which a walk ends a
    segment and flushes so that its 16-bit per-lane counters cannot wrap (TileWalk::bword).  Two different chromosomes in turn:
    counts that ended up in a neighbour's totals would show in both."""
    res, maxapart, maxww = 10000, 5000000, 10
    num = maxapart // res + maxww + 1
    chroms = _chroms([24896, 20011], num, 60.0, 17)
    prm = _lib.make_params(_lib.MODE_HICCUPS, [2], [5], maxww, 0.05, maxapart, res, 16, 0)
    want = [ctx.score_host(raw, None, None, None, prm, weight=weight, num=num) for raw, weight in chroms]
    items = [dict(raw=chroms[i % 2][0], weight=chroms[i % 2][1], num=num) for i in range(40)]
    try:
        ctx.set_option('grid_cap', 8)
        got = ctx.submit_batch_host(items, prm).results()
        ctx.set_option('lean', 0)                   # ... and every tile through the one kernel
        got2 = ctx.submit_batch_host(items, prm).results()
        ctx.set_option('grid_cap', 0)
        want2 = [ctx.score_host(raw, None, None, None, prm, weight=weight, num=num) for raw, weight in chroms]
    finally:
        ctx.set_option('grid_cap', 0)
        ctx.set_option('lean', 1)
    assert sum(w.ncand for w in want) > 0 and all(len(w.sets[0]['x']) > 0 for w in want)
    assert sum(g.tiles for g in got2) > 8 * 8192, [g.tiles for g in got2[:2]]       # (every workgroup's walk ends a segment)
    for i, (g, g2) in enumerate(zip(got, got2)):
        _same(g, want[i % 2])
        _same(g2, want2[i % 2])


def test_batch_raises_like_the_single_call(ctx):
    """results() raises for a chromosome the reference raises on (empty widening step) - after the whole batch was
    collected, so the lane is free again; an oversized batch and mixed input kinds are refused."""
    res, maxapart, maxww = 10000, 1500000, 10
    num = maxapart // res + maxww + 1
    prm = _lib.make_params(_lib.MODE_HICCUPS, [2], [5], maxww, 0.1, maxapart, res, 16, 0)
    (raw, weight), = _chroms([900], num, 40.0, 3)
    empty = np.zeros_like(raw)
    with pytest.raises((ValueError, ZeroDivisionError)):
        ctx.submit_batch_host([dict(raw=raw, weight=weight, num=num), dict(raw=empty, weight=weight, num=num)], prm).results()
    ok = ctx.submit_batch_host([dict(raw=raw, weight=weight, num=num)], prm).results()[0]
    want = ctx.score_host(raw, None, None, None, prm, weight=weight, num=num)
    _same(ok, want)
    # a batch large enough for its host half to run on several threads (hpk_collect_batch: from 8 chromosomes on), two
    # of them empty: every chromosome keeps its own outcome and its own message
    items = [dict(raw=(empty if i in (3, 10) else raw), weight=weight, num=num) for i in range(12)]
    got = ctx.submit_batch_host(items, prm).results(raise_on_error=False)
    for i, g in enumerate(got):
        if i in (3, 10):
            assert isinstance(g, _lib.EmptyStepError) and 'no unresolved candidate' in str(g)
        else:
            _same(g, want)
    with pytest.raises(_lib.HpkError) as ei:
        ctx.submit_batch_host([dict(raw=raw, weight=weight, num=num)] * (_lib.HPK_MAX_BATCH + 1), prm)
    assert ei.value.status == _lib.ERR_INVALID
    IR, cband, biases = orc.prep_from_band(raw, weight, 5)
    with pytest.raises(_lib.HpkError) as ei:
        ctx.submit_batch_host([dict(raw=raw, weight=weight, num=num), dict(raw=raw, IR=IR, bias1=biases, bias2=biases, balanced=cband)], prm)
    assert ei.value.status == _lib.ERR_INVALID


@pytest.mark.parametrize('name', [n for n in golden_names() if n.startswith('hiccups')])
def test_fixture_in_the_middle_of_a_batch(name, ctx):
    """Every reference fixture as the middle chromosome of a three-chromosome batch (its neighbours share its
    parameters): survivors, family sizes and the widening log equal the single call's, which test_gpu_parity.py pins to
    the reference."""
    g = load_golden(name)
    if 'prep_exception' in g.meta or 'exception' in g.meta:
        pytest.skip('the reference raises on this one')
    p = g.params
    num = g.meta['num']
    raw = g['raw'][:, :num].astype(np.float32)
    prm = _lib.make_params(_lib.MODE_HICCUPS, p['pw'], p['ww'], p['maxww'], p['sig'], p['maxapart'], p['res'], p['min_local_reads'], 0)
    n = raw.shape[0]
    other, ow, _ = synthetic.synth_band(max(n // 2, 30), num, depth=50.0, nloops=5, seed=5)
    third, tw, _ = synthetic.synth_band(n + 77, num, depth=20.0, nloops=5, seed=6)
    items = [dict(raw=other.astype(np.float32), weight=ow, num=num), dict(raw=raw, weight=g['weight'], num=num),
             dict(raw=third.astype(np.float32), weight=tw, num=num)]
    want = ctx.score_host(raw, None, None, None, prm, weight=g['weight'], num=num)
    got = ctx.submit_batch_host(items, prm).results(raise_on_error=False)
    assert not isinstance(got[1], Exception)
    _same(got[1], want)


def test_device_band_builder_equals_host_band(ctx):
    """hpk_devband_create (the pixel table goes to the GPU, the band is scattered together there) against the host
    builder (band_from_coo + upload of the dense band): the same results bit for bit - pixels in either orientation,
    repeated pixels adding up, pixels beyond the band, NaN weights, given biases (a divisive weight column); a bin outside
    the chromosome is refused."""
    from hicpeaks_amd import band as hband
    res, maxapart, maxww = 10000, 600000, 10
    num = maxapart // res + maxww + 1
    prm = _lib.make_params(_lib.MODE_HICCUPS, [2], [5], maxww, 0.1, maxapart, res, 16, 0)
    (raw, weight), (raw2, weight2) = _chroms([1500, 90], num, 60.0, 31)
    weight = weight.copy(); weight[100:110] = np.nan
    items_h, items_d = [], []
    rng = np.random.default_rng(5)
    for rw, w, bias in ((raw, weight, None), (raw2, weight2, None), (raw, weight, np.where(np.isnan(weight), 0.0, 1.0 / np.where(np.isnan(weight), 1.0, weight)))):
        n = rw.shape[0]
        r, k = np.nonzero(rw)
        keep = r + k < n
        r, k = r[keep], k[keep]
        i, j, v = r.copy(), r + k, rw[r, k].astype(np.int32)
        flip = rng.random(i.size) < 0.3                          # either orientation
        i[flip], j[flip] = j[flip].copy(), i[flip].copy()
        # a repeated pixel (its count split in two) and a pixel beyond the band
        big = np.nonzero(v >= 2)[0][:50]
        v2 = v.copy(); v2[big] -= 1
        i2 = np.concatenate([i, i[big], [0]]); j2 = np.concatenate([j, j[big], [min(n - 1, num + 5)]]); v3 = np.concatenate([v2, np.ones(big.size, np.int32), [7]])
        host = hband.band_from_coo(i2, j2, v3, n, num)
        np.testing.assert_array_equal(host, np.where(np.arange(n)[:, None] + np.arange(num)[None, :] < n, rw, 0).astype(np.float32))
        db = ctx.devband(i2, j2, v3, n, num, w, bias)
        assert db.stored == int((np.abs(j2 - i2) < num).sum()) and db.shape == (n, num)
        items_d.append(dict(raw=db))
        items_h.append(dict(raw=host, weight=w, bias1=bias, bias2=bias, num=num))
    got = ctx.submit_batch_host(items_d, prm).results()
    want = ctx.submit_batch_host(items_h, prm).results()
    for g, w in zip(got, want):
        _same(g, w)
    with pytest.raises(_lib.HpkError) as ei:
        ctx.devband(np.array([0, 5]), np.array([3, 1500]), np.array([1, 1], dtype=np.int32), 1500, num, weight)
    assert ei.value.status == _lib.ERR_INVALID
