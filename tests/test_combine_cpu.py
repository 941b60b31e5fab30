"""Multi-resolution merge against vectors produced by the reference's own functions
(oracle/gen_golden_combine.py; utilities.py:442-552)."""
import json, os
import pytest
from hicpeaks_amd import combine

HERE = os.path.dirname(__file__)
CASES = json.load(open(os.path.join(HERE, 'golden', 'combine_cases.json')))['cases']


@pytest.mark.parametrize('case', CASES, ids=lambda c: 'seed%d' % c['spec']['seed'])
def test_parse_and_combine(case, tmp_path):
    sp = case['spec']
    byres = {}
    for r, text in case['files'].items():
        p = tmp_path / ('peaks_%s.bedpe' % r)
        p.write_text(text)
        byres[int(r)] = combine._parse_peakfile(str(p), 1)
        want = {c: [tuple(x) for x in v] for c, v in case['parsed'][r].items()}
        assert byres[int(r)] == want
    got = combine.combine_annotations(byres, good_res=sp['good_res'], mindis=sp['mindis'], max_res=sp['max_res'])
    assert [tuple(t) for t in case['expected']] == got


def test_cli_roundtrip(tmp_path):
    case = CASES[0]
    sp = case['spec']
    paths = []
    for r in sp['resolutions']:
        p = tmp_path / ('%d.bedpe' % r)
        p.write_text(case['files'][str(r)])
        paths.append(str(p))
    out = tmp_path / 'combined.bedpe'
    rc = combine.main_combine(['-O', str(out), '-p'] + paths + ['-R'] + [str(r) for r in sp['resolutions']]
                              + ['-S', '1', '-G', str(sp['good_res']), '-M', str(sp['mindis']),
                                 '--max-res', str(sp['max_res'])])
    assert rc == 0
    lines = out.read_text().splitlines()
    assert len(lines) == len(case['expected'])
    first = case['expected'][0]
    assert lines[0].split('\t') == ['chr' + first[0], str(first[1]), str(first[2]), 'chr' + first[3], str(first[4]), str(first[5])]


def test_parser_defaults_match_reference():
    a = combine._parser().parse_args(['-O', 'x'])
    assert (a.skip_rows, a.good_res, a.min_dis, a.max_res) == (0, 20000, 200000, 10000)
