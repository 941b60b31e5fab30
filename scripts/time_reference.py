#!/opt/conda/bin/python3.9
"""Time the *real* reference (hicpeaks 0.3.9, /root/reference) on the synthetic stand-ins of BASELINE.md section 2, so that
those numbers can be regenerated from the repository.  BUILD CONTAINER ONLY: imports
/root/reference (never shipped, nothing of it is copied) under the interpreter pinned by SURVEY.md §8-C1:

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 scripts/time_reference.py [case ...] [--nproc N]

Cases: cfg1 (chr21 @25 kb stand-in, (1,3), 10 Mb, n = 1869 x 411 diagonals), chr21_10kb ((2,5), 2 Mb, 4671 x 211),
chr21_10kb_union ((1,3)/(2,5)/(4,7), 5 Mb, 4671 x 511).  With --nproc N the case is run N times in N processes at once
(the reference's own parallelism is one process per chromosome, scripts/pyHICCUPS:192-198) and the aggregate rate
is reported.  Output: one line per case with band pixels / s / core; append to profiles/ by hand."""
import importlib.util, os, sys, time, warnings
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
sys.path.insert(0, os.path.join(REPO, 'oracle'))
warnings.filterwarnings('ignore')

CASES = {
    'cfg1': dict(n=1869, res=25000, maxapart=10000000, pw=[1], ww=[3], maxww=10, depth=150.0, nloops=30),
    'chr21_10kb': dict(n=4671, res=10000, maxapart=2000000, pw=[2], ww=[5], maxww=10, depth=60.0, nloops=60),
    'chr21_10kb_union': dict(n=4671, res=10000, maxapart=5000000, pw=[1, 2, 4], ww=[3, 5, 7], maxww=10, depth=60.0, nloops=60),
}


def run_case(name):
    import gen_golden as gg                     # the fixture generator's prep (pyHICCUPS:142-166 restated) and the reference
    c = CASES[name]
    num = c['maxapart'] // c['res'] + c['maxww'] + 1
    raw, weight, _ = gg.synthetic.synth_band(c['n'], num, depth=c['depth'], nloops=c['nloops'], seed=0)
    mw = min(c['ww'])
    t0 = time.perf_counter()
    H, cH = gg.cooler_like(raw, weight)
    M, cM, biases, IR, chromLen, Diags, cDiags, num = gg.worker_prep(H, cH, weight, mw, c['maxapart'], c['res'], c['maxww'])
    t1 = time.perf_counter()
    gg.ref.hiccups(M, cM, biases, biases, IR, chromLen, Diags, cDiags, num, 'T', pw=c['pw'], ww=c['ww'], maxww=c['maxww'],
                   sig=0.05, maxapart=c['maxapart'], res=c['res'], min_marginal_peaks=2, onlyanchor=False, min_local_reads=16)
    t2 = time.perf_counter()
    D = c['maxapart'] // c['res']
    px = sum(max(c['n'] - d, 0) for d in range(mw, D + 1))
    return name, px, len(c['pw']), t1 - t0, t2 - t1


def main():
    args = sys.argv[1:]
    nproc = 1
    if '--nproc' in args:
        k = args.index('--nproc')
        nproc = int(args[k + 1])
        del args[k:k + 2]
    names = args or ['cfg1']
    for name in names:
        if nproc == 1:
            res = [run_case(name)]
            wall = res[0][4]
        else:
            import multiprocessing as mp
            t0 = time.perf_counter()
            with mp.get_context('fork').Pool(nproc) as pool:
                res = pool.map(run_case, [name] * nproc)
            wall = max(r[4] for r in res)
        _, px, npairs, tprep, _ = res[0]
        print('%-18s n=%d pairs=%d band_px=%d  prep %.1f s  hiccups() %.1f s  -> %.1f k band px/s/process; %d process(es): %.1f k band px/s '
              '(hicpeaks 0.3.9, python %s, %d-core container)' % (name, CASES[name]['n'], npairs, px, tprep, wall, px / res[0][4] / 1e3, nproc,
                                                                   nproc * px / wall / 1e3, sys.version.split()[0], os.cpu_count()))


if __name__ == '__main__':
    main()
