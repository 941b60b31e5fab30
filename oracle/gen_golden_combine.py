"""Golden vectors for the multi-resolution merge.  TEST INFRASTRUCTURE ONLY.

hicpeaks/utilities.py cannot be imported in this image (it needs `cooler`), so the two pure functions
`_parse_peakfile` and `combine_annotations` (utilities.py:442-552) are compiled from the reference file's AST in
memory and run on seeded synthetic peak lists; only inputs and outputs are stored (tests/golden/combine_cases.json).
Run here only:  python oracle/gen_golden_combine.py
"""
import ast, json, os, sys, tempfile
import numpy as np

REF = '/root/reference/hicpeaks/utilities.py'
OUT = os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'combine_cases.json')


def load_reference():
    tree = ast.parse(open(REF).read())
    want = {'find_chrom_pre', '_parse_peakfile', 'combine_annotations'}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    ns = {'np': np}
    exec(compile(ast.Module(body=body, type_ignores=[]), REF, 'exec'), ns)
    return ns


def synth(seed, resolutions, nchrom=3, nbase=60, prefix='chr'):
    """Peaks seen at several resolutions: a shared set of loops jittered and snapped to each bin size, plus
    resolution-private calls at short and long range."""
    rng = np.random.default_rng(seed)
    byres = {r: {} for r in resolutions}
    for ci in range(nchrom):
        chrom = prefix + ('X' if ci == nchrom - 1 else str(ci + 1))
        a = rng.integers(100, 4000, nbase) * 5000
        d = rng.integers(4, 400, nbase) * 5000
        for r in resolutions:
            rows = []
            for x, dist in zip(a, d):
                if rng.random() < 0.7:
                    jx, jy = rng.integers(-2, 3, 2) * r
                    s1 = (x + jx) // r * r
                    s2 = (x + dist + jy) // r * r
                    rows.append((int(s1), int(s1 + r), int(s2), int(s2 + r)))
            for _ in range(nbase // 3):
                x = int(rng.integers(100, 4000)) * 5000 // r * r
                dist = int(rng.integers(3, 600)) * r
                rows.append((x, x + r, x + dist, x + dist + r))
            if rows and not (ci == 1 and r == resolutions[-1]):     # one chromosome missing at the coarsest level
                byres[r][chrom] = rows
    return byres


def main():
    ns = load_reference()
    cases = []
    specs = [
        dict(seed=1, resolutions=[5000, 10000, 20000], good_res=20000, mindis=200000, max_res=10000, prefix='chr'),
        dict(seed=2, resolutions=[5000, 10000, 25000], good_res=10000, mindis=100000, max_res=10000, prefix='chr'),
        dict(seed=3, resolutions=[10000, 5000], good_res=10000, mindis=100000, max_res=10000, prefix=''),
        dict(seed=4, resolutions=[10000], good_res=10000, mindis=100000, max_res=10000, prefix='chr'),
        dict(seed=5, resolutions=[5000, 10000, 20000, 40000], good_res=20000, mindis=150000, max_res=20000, prefix='chr'),
        dict(seed=6, resolutions=[2000, 5000], good_res=20000, mindis=200000, max_res=1000, prefix='chr'),
    ]
    for sp in specs:
        raw = synth(sp['seed'], sp['resolutions'], prefix=sp['prefix'])
        parsed, texts = {}, {}
        for r, table in raw.items():
            lines = ['#header\n']
            for chrom, rows in table.items():
                for s1, e1, s2, e2 in rows:
                    lines.append('\t'.join(map(str, (chrom, s1, e1, chrom, s2, e2, '.', 12.5))) + '\n')
            text = ''.join(lines)
            with tempfile.NamedTemporaryFile('w', suffix='.bedpe', delete=False) as f:
                f.write(text)
            parsed[r] = ns['_parse_peakfile'](f.name, 1)
            os.unlink(f.name)
            texts[str(r)] = text
        out = ns['combine_annotations'](parsed, good_res=sp['good_res'], mindis=sp['mindis'], max_res=sp['max_res'])
        cases.append(dict(spec=sp, files=texts,
                          parsed={str(r): {c: [list(p) for p in v] for c, v in t.items()} for r, t in parsed.items()},
                          expected=[list(t) for t in out]))
        print(sp['seed'], {r: sum(len(v) for v in t.values()) for r, t in parsed.items()}, '->', len(out))
    json.dump(dict(generator='oracle/gen_golden_combine.py', numpy=np.__version__, python=sys.version.split()[0],
                   cases=cases), open(OUT, 'w'))
    print('wrote', os.path.abspath(OUT), os.path.getsize(OUT))


if __name__ == '__main__':
    main()
