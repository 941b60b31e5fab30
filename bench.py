#!/usr/bin/env python
"""bench.py - band pixels scored / s (donut + lower-left) on synthetic banded Hi-C matrices.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config chr1_10kb]

One "step" = the whole hot path (expected tables -> stencil -> freeze -> lambda-chunk Poisson scoring -> survivor
compaction -> Benjamini-Hochberg) over `--batch` synthetic chromosomes whose bands are already resident in HBM,
handed to the library `--group` chromosomes at a time (hpk_submit_batch: one launch per stage for the whole group,
one group in flight ahead).  N > 1: one process per GPU (torch.distributed / RCCL only for the barrier and the max
over ranks); chromosomes are independent, every rank scores its own chromosomes of the same shape, no data-path
collective (weak scaling).  Rank 0 prints ONE JSON line.

`roofline`   : the stencil kernel against HBM: algorithmic bytes = 20 B per band pixel per (p, w) pair
               (4 B f32 count read + 2 x 8 B f64 local expected written) x the band
               pixels one launch processes (a group of chromosomes) divided by the launch's mean duration from HIP
               events on the library's stream.
`cpu_baseline`: the library's own back-end #0 (hpk_create(-1): C++ on host threads, parity-gated like the GPU path) on all
               cores and on one, and the numpy oracle (oracle/hiccups_oracle.py, the restatement the fixtures pin to the
               reference) on one core - the same chromosome, rank 0, N = 1 only.
"""
import argparse
import collections
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

CONFIGS = {
    # BASELINE.json configs[1]: hg38 chr1 at 10 kb, (p=2, w=5), 5 Mb band
    'chr1_10kb': dict(n=24896, res=10000, maxapart=5000000, pw=[2], ww=[5], maxww=10, depth=60.0, nloops=400,
                      workload='hg38 chr1 @10kb (n=24896), (p,w)=(2,5), 5Mb band, weight-balanced, synthetic Poisson band'),
    # configs[2] style, one chromosome: union of three pairs
    'chr1_10kb_union': dict(n=24896, res=10000, maxapart=5000000, pw=[1, 2, 4], ww=[3, 5, 7], maxww=10, depth=60.0,
                            nloops=400, workload='hg38 chr1 @10kb, union (1,3)/(2,5)/(4,7), 5Mb band'),
    # configs[3] largest item: chr1 at 5 kb, (4,7), 10 Mb band
    'chr1_5kb': dict(n=49792, res=5000, maxapart=10000000, pw=[4], ww=[7], maxww=10, depth=25.0, nloops=800,
                     workload='hg38 chr1 @5kb (n=49792), (p,w)=(4,7), 10Mb band'),
    # configs[4]: synthetic 1 kb deep Hi-C stress
    'deep_1kb': dict(n=250000, res=1000, maxapart=2000000, pw=[2], ww=[5], maxww=10, depth=8.0, nloops=2000,
                     workload='synthetic 1kb (n=250000), (p,w)=(2,5), 2Mb band'),
    # BASELINE.json configs[2] / configs[3]: the whole genome (hg38 chr1-22,X, the CLI's default --chroms) as 23 work
    # items through the one-ahead queue; with --gpus N the chromosomes are dealt largest-first to the ranks (strong
    # scaling: the genome is the fixed total) and nothing is exchanged.  IR and the biases are derived on the device.
    'wg_10kb_union': dict(genome=True, n=24896, res=10000, maxapart=5000000, pw=[1, 2, 4], ww=[3, 5, 7], maxww=10, depth=60.0,
                          nloops=400, workload='hg38 chr1-22,X @10kb, union (1,3)/(2,5)/(4,7), 5Mb band, 23 chromosomes'),
    'wg_5kb': dict(genome=True, n=49792, res=5000, maxapart=10000000, pw=[4], ww=[7], maxww=10, depth=25.0, nloops=800,
                   workload='hg38 chr1-22,X @5kb, (p,w)=(4,7), 10Mb band, 23 chromosomes'),
    # the sibling path (callers.py:364-590, scripts/pyBHFDR defaults): donut only, per-pixel Poisson(lambda = E), one
    # Benjamini-Hochberg family per chromosome; 2 Mb band
    'chr1_10kb_bhfdr': dict(mode='bhfdr', n=24896, res=10000, maxapart=2000000, pw=[2], ww=[5], maxww=10, depth=60.0, nloops=400,
                            workload='hg38 chr1 @10kb (n=24896), bhfdr (p,w)=(2,5), 2Mb band'),
    'tiny': dict(n=3000, res=10000, maxapart=2000000, pw=[2], ww=[5], maxww=10, depth=60.0, nloops=40,
                 workload='tiny self-test'),
}
SIG, MIN_READS = 0.05, 16
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
TIMED_EVERY = 4          # launches of the timed region whose stencil and scoring kernels are bracketed by HIP events: one in four
BYTES_PER_PX = 20.0      # SURVEY.md §8-D3's convention for the >= 70 % target: 4 B read + 2 x 8 B of dense local expecteds written
# the roof the stencil actually leans on: vector issue.  1 024 SIMDs (256 CUs x 4), a wave64 VALU instruction occupies its SIMD
# 4 cycles (f64, DPP, packed: ~5; profiles/r02_instruction_cost.txt) - priced at 4, so the fraction is a lower bound of the issue
# slots taken; clock: MI355X's 2.4 GHz peak engine clock (MI355X_MICROARCH.md; under load the chip runs at or below it)
N_SIMD, VALU_CYCLES, CLOCK_GHZ = 1024, 4.0, 2.4
N_CU, SALU_CYCLES = 256, 2.0         # one scalar unit per CU, one instruction per two cycles with all sixteen waves issuing (measured with its loop: 2.07, profiles/r06_instruction_cost.txt)
FRAC_LABEL = ('frac / frac_20B_equivalent: 20 B per band pixel per pair (SURVEY 8-D3: what a kernel writing dense E_K, E_Y would move) '
              '/ kernel time / 8 TB/s - how far the instruction stream is from the contract\'s memory floor, NOT HBM utilisation (it can '
              'exceed 1); frac_measured: the bytes rocprofv3 counted / kernel time / 8 TB/s; roofline_valu: vector issue slots taken')


def make_band_host(cfg, seed, n=None):
    from hicpeaks_amd import synthetic, band
    n = n or cfg['n']
    num = cfg['maxapart'] // cfg['res'] + cfg['maxww'] + 1
    raw, weight, _ = synthetic.synth_band(n, num, depth=cfg['depth'], nloops=max(1, cfg['nloops'] * n // cfg['n']),
                                          seed=seed)
    IR, biases = band.expected_and_biases(raw, weight, min(cfg['ww']))
    return raw, weight, IR, biases, num


def cpu_baseline(cfg, rows, allcores_rows=0):
    """The CPU path timed beside the GPU's, on the same workload: the library's own back-end #0 - hpk_create(-1),
    C++ on host threads, parity-gated on the GPU path's fixtures (tests/test_cpu_backend.py), kind "native" - on every core the
    process may use and on one; and the numpy oracle (kind "port": oracle/hiccups_oracle.py, the restatement the fixtures pin to
    the reference) on one core.  rows: the slice of the workload (default: the whole chromosome)."""
    from oracle import hiccups_oracle as orc
    from hicpeaks_amd import _lib, band
    raw, weight, IR, biases, num = make_band_host(cfg, seed=12345, n=rows)
    mw = min(cfg['ww'])
    px = band.band_pixels(rows, num, mw, cfg['maxapart'] // cfg['res']) * len(cfg['pw'])
    # ---- native: IR / biases given (as the GPU's timed region has them), band resident in host memory
    rawf = np.ascontiguousarray(raw.astype(np.float32))
    prm = _lib.make_params(_lib.MODE_BHFDR if cfg.get('mode') == 'bhfdr' else _lib.MODE_HICCUPS, cfg['pw'], cfg['ww'], cfg['maxww'], SIG,
                           cfg['maxapart'], cfg['res'], MIN_READS, 0)
    cpu = _lib.Context(-1)
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)

    def native(threads, reps):
        cpu.set_option('cpu_threads', threads)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            R = cpu.score_host(rawf, IR, biases, biases, prm, weight=weight)
            ts.append(time.perf_counter() - t0)
        return min(ts), float(np.median(ts)), R
    native(ncores, 1)                               # (Poisson tables, first touch)
    # (a call is tens of milliseconds on a few hundred threads - what the box's other tenants do shows: twelve calls, the best
    #  one is the value, the median is stated)
    t_all, t_all_med, R = native(ncores, 12)
    t_one, _, _ = native(1, 1)
    cpu.close()
    # ---- port: the numpy oracle, one core
    t0 = time.perf_counter()
    IRo, cband, b = orc.prep_from_band(raw, weight, mw)
    t1 = time.perf_counter()
    orc.hiccups(raw, cband, b, b, IRo, rows, num, pw=cfg['pw'], ww=cfg['ww'], maxww=cfg['maxww'], sig=SIG,
                maxapart=cfg['maxapart'], res=cfg['res'], min_local_reads=MIN_READS, min_marginal_peaks=2,
                onlyanchor=False) if cfg.get('mode') != 'bhfdr' else None
    t2 = time.perf_counter()
    out = dict(value=px / t_all, unit='band px/s', cores=ncores, cores_available=os.cpu_count(), kind='native',
               sample='%d-row slice of the workload (%d band px): libhpk back-end #0 (hpk_create(-1), C++ on %d host threads), best of 12 '
                      'calls %.3f s (median %.3f s); candidates %d, significant %d' % (rows, px, ncores, t_all, t_all_med, R.ncand, R.nsig),
               one_core=dict(value=px / t_one, unit='band px/s', cores=1, kind='native', sample='the same call on one thread: %.1f s' % t_one))
    if cfg.get('mode') != 'bhfdr':
        out['port'] = dict(value=px / (t2 - t1), unit='band px/s', cores=1, kind='port',
                           sample='numpy oracle hiccups() on the same slice: %.1f s (+%.1f s prep)' % (t2 - t1, t1 - t0))
    if allcores_rows > 0:
        out['port_all_cores'] = cpu_baseline_all_cores(cfg, allcores_rows)
    return out


def _cpu_worker(job):
    """One host core: the oracle on its own synthetic chromosome slice (like one worker of the reference's Pool.map over
    chromosomes, scripts/pyHICCUPS:192-198).  -> (band px, seconds)"""
    cfg, rows, seed = job
    os.environ['OMP_NUM_THREADS'] = '1'
    from oracle import hiccups_oracle as orc
    from hicpeaks_amd import band
    raw, weight, IR, biases, num = make_band_host(cfg, seed=seed, n=rows)
    mw = min(cfg['ww'])
    t0 = time.perf_counter()
    IRo, cband, b = orc.prep_from_band(raw, weight, mw)
    orc.hiccups(raw, cband, b, b, IRo, rows, num, pw=cfg['pw'], ww=cfg['ww'], maxww=cfg['maxww'], sig=SIG,
                maxapart=cfg['maxapart'], res=cfg['res'], min_local_reads=MIN_READS, min_marginal_peaks=2,
                onlyanchor=False)
    return band.band_pixels(rows, num, mw, cfg['maxapart'] // cfg['res']) * len(cfg['pw']), time.perf_counter() - t0


def cpu_baseline_all_cores(cfg, rows):
    """Every host core at once, one process per core, each on its own `rows`-row slice: aggregate px/s with the core count
    stated.  The numpy port is bound by memory bandwidth long before the box runs out of cores (MI355X host:
    256 logical cores; 64 processes scored twice what 256 do), so the leg runs at both counts and reports both."""
    import multiprocessing as mp
    avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    nall = max(1, avail)                # every core the process may run on (round 5 capped this at 64 of the box's 256)
    try:                                # ... as long as half the free memory holds them (a worker peaks at ~0.3 GB)
        free_kb = [int(l.split()[1]) for l in open('/proc/meminfo') if l.startswith('MemAvailable')][0]
        nall = max(1, min(nall, int(free_kb / 2 / (0.4 * 1024 * 1024))))
    except Exception:
        nall = min(nall, 64)
    legs = []
    for nproc in sorted(set([min(64, nall), nall])):
        t0 = time.perf_counter()
        with mp.get_context('spawn').Pool(nproc) as pool:
            res = pool.map(_cpu_worker, [(cfg, rows, 777 + i) for i in range(nproc)])
        wall = time.perf_counter() - t0
        px = sum(r[0] for r in res)
        busy = max(r[1] for r in res)
        legs.append(dict(value=px / busy, unit='band px/s', cores=nproc, cores_available=os.cpu_count(), kind='port',
                         sample='%d processes x %d-row slices (%d band px in all), slowest worker %.1f s, wall incl. start-up %.1f s' % (
                             nproc, rows, px, busy, wall)))
    out = dict(legs[-1])                # the all-cores leg; `best` = the faster of the two
    out['legs'] = legs
    out['best'] = max(legs, key=lambda l: l['value'])['value']
    out['best_cores'] = max(legs, key=lambda l: l['value'])['cores']
    return out


def run_genome(args, cfg, ctx, rank, world, local, dist, emit=True, steps=None, warmup=None):
    """Whole-genome configurations: one step = every chromosome of the genome scored once - this rank's share of them as
    ONE batch (hpk_submit_batch: one launch per stage for all of them), the next pass submitted before this one is
    collected; value = band pixels of the whole genome x pairs x steps / wall time."""
    import torch
    from hicpeaks_amd import _lib, band, bandgen, parallel, synthetic
    dev = torch.device('cuda', local)
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    res, mw, D = cfg['res'], min(cfg['ww']), cfg['maxapart'] // cfg['res']
    num = D + cfg['maxww'] + 1
    ld = (num + 63) // 64 * 64
    sizes = synthetic.hg38_bins(res)
    parts = parallel.lpt_partition(sizes, world)
    mine = parts[rank]
    rank_px = [sum(band.band_pixels(sizes[c], cfg['maxapart'] // cfg['res'] + cfg['maxww'] + 1, min(cfg['ww']), cfg['maxapart'] // cfg['res'])
                   for c in p_) * len(cfg['pw']) for p_ in parts]
    bands = []
    gorder = sorted(sizes, key=lambda k: (-sizes[k], str(k)))       # a chromosome's band is the same whichever rank scores it
    for i, c in enumerate(mine):
        n = sizes[c]
        raw_d, w_d, _, _ = bandgen.device_band(n, num, ld, mw, depth=cfg['depth'], nloops=max(1, cfg['nloops'] * n // cfg['n']),
                                               seed=1000 + gorder.index(c), device=dev, want_expected=False)
        if args.host_inputs:       # PCIe-inclusive variant: the bands live in host memory (pageable numpy), as after cooler I/O
            bands.append((c, n, np.ascontiguousarray(raw_d[:, :num].cpu().numpy()), w_d.cpu().numpy()))
            del raw_d, w_d
        else:
            bands.append((c, n, raw_d, w_d))
    torch.cuda.synchronize()
    prm = _lib.make_params(_lib.MODE_BHFDR if cfg.get('mode') == 'bhfdr' else _lib.MODE_HICCUPS, cfg['pw'], cfg['ww'], cfg['maxww'], SIG, cfg['maxapart'], cfg['res'],
                           MIN_READS, 0)
    prm_quiet = _lib.make_params(_lib.MODE_BHFDR if cfg.get('mode') == 'bhfdr' else _lib.MODE_HICCUPS, cfg['pw'], cfg['ww'], cfg['maxww'], SIG, cfg['maxapart'], cfg['res'],
                                 MIN_READS, _lib.FLAG_NO_STENCIL_TIMING)      # see run_chromosome: every TIMED_EVERY-th launch is timed
    npass = [0]
    px_genome = sum(band.band_pixels(n, num, mw, D) for n in sizes.values()) * len(cfg['pw'])
    depth = max(1, min(args.pipeline_depth, ctx.pipeline_depth))
    group = len(bands) if args.group <= 0 else max(1, min(args.group, len(bands)))

    def light(R):         # what the report needs; holding every BandResult makes Python's cyclic GC slower pass after pass
        return (R.timing['stencil'], R.band_px, R.ncand, R.nsig, int(R.redone), R.frozen_w, int(R.rescored))

    def submit(part, p):
        if args.host_inputs:
            return ctx.submit_batch_host([dict(raw=r, weight=w, num=num) for (_, _, r, w) in part], p)
        bd = [ctx._band(n, num, ld, r.data_ptr(), None, w.data_ptr(), None, None, None, True) for (_, n, r, w) in part]
        return ctx.submit_batch(bd, p, [n for (_, n, _, _) in part])

    def one_pass(pending, done):
        npass[0] += 1
        for g0 in range(0, len(bands), group):
            p = prm if (npass[0] + g0) % TIMED_EVERY == 0 else prm_quiet
            pending.append(submit(bands[g0:g0 + group], p))
            if len(pending) >= depth:
                done.append([light(R) for R in pending.popleft().results()])

    def drain(pending, done):
        while pending:
            done.append([light(R) for R in pending.popleft().results()])

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    pending, done = collections.deque(), []
    for _ in range(depth + warmup):             # set-up (every lane allocates its workspaces once) + warm-up passes
        one_pass(pending, done)
    drain(pending, done)
    barrier()
    t0 = time.perf_counter()
    results = []
    for _ in range(steps):                      # the passes follow each other without a gap: one batch ahead
        one_pass(pending, results)
    drain(pending, results)
    barrier()
    elapsed = time.perf_counter() - t0
    # what the last pass found on this rank, and - N ranks - on all of them: every chromosome must have been scored by exactly one
    last_all = [t for rs in results[-((len(bands) + group - 1) // group):] for t in rs]
    tot = [float(sum(t[2] for t in last_all)), float(sum(t[3] for t in last_all)), float(len(mine))]
    scored_by = [list(mine)]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=args.coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        t = torch.tensor(tot, dtype=torch.float64, device=args.coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        tot = [float(v) for v in t.tolist()]
        scored_by = [None] * world
        dist.all_gather_object(scored_by, list(mine))
    out = None
    if rank == 0:
        # dominant kernel: the stencil launches of this rank that were bracketed by events; achieved = algorithmic bytes
        # of the launch's band pixels / its duration
        timed = [rs for rs in results if rs[0][0] > 0]
        st_ms = sum(t[0] for rs in timed for t in rs)
        st_px = sum(t[1] for rs in timed for t in rs) * len(cfg['pw'])
        nlaunch = max(1, len(timed))
        achieved = BYTES_PER_PX * st_px / (st_ms * 1e-3) / 1e9 if timed else 0.0
        last = [t for rs in results[-((len(bands) + group - 1) // group):] for t in rs]
        out = {
            'metric': 'band pixels scored/sec (donut+LL)', 'value': px_genome * steps / elapsed, 'unit': 'band px/s',
            'n_gpus': world, 'steps': steps, 'warmup': warmup, 'ms_per_step': elapsed / steps * 1e3,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': cfg['workload'], 'name': args.config, 'band_px_per_step': px_genome,
                       'chromosomes': len(sizes), 'chromosomes_rank0': len(mine), 'chromosomes_per_launch': group,
                       'pipeline_depth': depth, 'ranks_seen': args.ranks_seen,
                       'candidates_rank0': int(sum(t[2] for t in last)),
                       'significant_px_rank0': int(sum(t[3] for t in last)),
                       # record bound = the widest freeze of the chromosomes collected before: chromosomes of
                       # the last pass that froze later and were computed once more, and the widths they froze at
                       'redone_in_full_rank0': int(sum(t[4] for t in last)), 'frozen_w_rank0': sorted(set(t[5] for t in last)),
                       # ... and chromosomes whose Benjamini-Hochberg cut lay above the bound of their survivor records
                       'rescored_rank0': int(sum(t[6] for t in last)),
                       'parallelism': 'chromosomes dealt largest-first to the GPUs, no collective',
                       # the largest-first deal: band px x pairs per rank, and the heaviest rank against the mean
                       'per_rank_px': rank_px, 'lpt_imbalance': max(rank_px) / (sum(rank_px) / float(len(rank_px))),
                       # over all ranks (all-reduce / all-gather over the ranks' own results): candidates and significant pixels of the
                       # genome, and which rank scored which chromosome
                       'candidates_all_ranks': int(tot[0]), 'significant_px_all_ranks': int(tot[1]), 'chromosomes_scored': int(tot[2]),
                       'chromosomes_by_rank': scored_by,
                       'whole_genome_wall_ms': elapsed / steps * 1e3, 'host_inputs': bool(args.host_inputs)},
            'roofline': {'bound': 'valu-issue', 'bound_by_contract': 'hbm', 'kernel': 'hpk_stencil_s (+ hpk_stencil_lean)', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'frac_20B_equivalent': achieved / HBM_PEAK_GBS, 'frac_measured': None,
                         'frac_label': FRAC_LABEL, 'traffic': None, 'kernel_ms': st_ms / nlaunch,
                         'launches_timed': nlaunch, 'launches': len(results),
                         'algorithmic_bytes_per_launch': BYTES_PER_PX * st_px / nlaunch},
        }
        if emit:
            print(json.dumps(out))
    del bands
    if emit and dist is not None:
        dist.destroy_process_group()
    return out


def run_genome_cold(args, cfg, ctx, local, reps=5):
    """What one invocation of the command line meets (scripts/pyHICCUPS:192-210: one genome per process): ONE genome of 23
    chromosomes scored from a context without history, under the mode the command lines ship (spec_halo = 2: every chromosome under
    the tile layout of its own frozen width - with nothing inherited, a chromosome whose layout turns out different is computed a
    second time).  Median of `reps` passes, the context's hints reset before each (workspaces stay allocated); then the same genome
    again without a reset (warm, also synchronous: one call, collected before the next), and once in a truly fresh context
    (allocations included)."""
    import torch
    from hicpeaks_amd import _lib, band, bandgen, synthetic
    dev = torch.device('cuda', local)
    res, mw, D = cfg['res'], min(cfg['ww']), cfg['maxapart'] // cfg['res']
    num = D + cfg['maxww'] + 1
    ld = (num + 63) // 64 * 64
    sizes = synthetic.hg38_bins(res)
    bands = []
    for i, c in enumerate(sorted(sizes, key=lambda k: -sizes[k])):
        n = sizes[c]
        raw_d, w_d, _, _ = bandgen.device_band(n, num, ld, mw, depth=cfg['depth'], nloops=max(1, cfg['nloops'] * n // cfg['n']),
                                               seed=i, device=dev, want_expected=False)
        bands.append((n, raw_d, w_d))
    torch.cuda.synchronize()
    prm = _lib.make_params(_lib.MODE_HICCUPS, cfg['pw'], cfg['ww'], cfg['maxww'], SIG, cfg['maxapart'], cfg['res'], MIN_READS,
                           _lib.FLAG_NO_STENCIL_TIMING)
    px = sum(band.band_pixels(n, num, mw, D) for n in sizes.values()) * len(cfg['pw'])

    def one(c):
        bd = [c._band(n, num, ld, r.data_ptr(), None, w.data_ptr(), None, None, None, True) for (n, r, w) in bands]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rs = c.submit_batch(bd, prm, [n for (n, _, _) in bands]).results()
        ms = (time.perf_counter() - t0) * 1e3
        return ms, sum(int(r.redone) for r in rs), sum(int(r.rescored) for r in rs), sorted(set(int(r.frozen_w) for r in rs))

    ctx.set_option('spec_halo', 2)
    one(ctx)                                    # workspaces
    cold = []
    for _ in range(reps):
        ctx.set_option('reset_hints', 1)
        cold.append(one(ctx))
    warm = [one(ctx) for _ in range(3)]
    ctx.set_option('spec_halo', 1)
    c2 = _lib.Context(local)
    c2.set_option('spec_halo', 2)
    fresh = one(c2)
    c2.close()
    del bands
    cm = float(np.median([c_[0] for c_ in cold]))
    wm = float(np.median([w_[0] for w_ in warm]))
    k = int(np.argsort([c_[0] for c_ in cold])[len(cold) // 2])
    return {'workload': cfg['workload'], 'spec_halo': 2, 'band_px_per_genome': px, 'passes': reps,
            'ms_per_genome': cm, 'value': px / (cm * 1e-3), 'unit': 'band px/s',
            'passes_redone_in_full': cold[k][1], 'second_passes': cold[k][1], 'rescored': cold[k][2], 'frozen_w': cold[k][3],
            'ms_per_genome_warm_sync': wm, 'warm_redone': warm[-1][1], 'cold_over_warm': cm / wm,
            'ms_first_genome_fresh_context': fresh[0], 'fresh_context_redone': fresh[1]}


def launch_plan(gpus, env, ngpus_visible):
    """What `--gpus N` means for this process (no side effects; unit-tested on CPU):
       ('run',)            go ahead as the rank the environment describes (or as the only one),
       ('spawn', port)     N > 1 and not under torchrun: re-launch as N ranks,
       ('error', message)  the request cannot be honoured - never run fewer GPUs than asked for."""
    world = int(env.get('WORLD_SIZE', '0') or 0)
    if ngpus_visible <= 0:
        return ('error', 'needs an MI355X: there is no CPU path')
    if world > 0:                                   # under torchrun (the driver's launch line for N > 1)
        if world != gpus:
            return ('error', '--gpus %d but WORLD_SIZE=%d' % (gpus, world))
        if ngpus_visible < (1 if env.get('HPK_BENCH_ONE_GPU') else min(world, 8)):
            return ('error', '%d rank(s) but only %d GPU(s) visible' % (world, ngpus_visible))
        return ('run',)
    if gpus <= 1:
        return ('run',)
    if ngpus_visible < gpus:
        return ('error', '--gpus %d but only %d GPU(s) visible' % (gpus, ngpus_visible))
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    return ('spawn', port)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=0,
                    help='chromosomes per step: a step scores this many chromosome-sized bands '
                         '(default: 1280 for the 10 kb configurations, so that 20 steps keep the GPU busy for ~2 s '
                         'and the clocks settle; the whole-genome configurations\' step is the 23-chromosome genome)')
    ap.add_argument('--group', type=int, default=0,
                    help='chromosomes per library call (hpk_submit_batch: one launch per stage for the whole group); 1 = chromosome '
                         'by chromosome as in round 2; default: sized so that a group\'s workspaces stay below ~60 GB (two groups are in flight: 288 GB of HBM), at most 64; '
                         'whole-genome configurations: the rank\'s whole share')
    ap.add_argument('--config', default='chr1_10kb', choices=sorted(CONFIGS))
    ap.add_argument('--cpu-rows', type=int, default=1 << 30,
                    help='rows of the CPU-baseline sample (default: the whole workload, ~8 s on one core; 0 = skip)')
    ap.add_argument('--pipeline-depth', type=int, default=2,
                    help='groups in flight per GPU (hpk_submit_batch / hpk_collect_batch); 1 = synchronous calls')
    ap.add_argument('--balanced-f64', action='store_true',
                    help='hand over the balanced band as f64 [n][ld] (what the drop-in hiccups() receives as cDiags) instead of '
                         'the weights: 12 B/px read instead of 4')
    ap.add_argument('--host-inputs', action='store_true',
                    help='hand the band over as host (numpy) arrays every step: the PCIe-inclusive rate of DESIGN.md, never `value`')
    ap.add_argument('--distinct', type=int, default=64,
                    help='distinct bands resident in HBM that a step\'s chromosomes rotate through (seeds 0..N-1; configurations up to 20 M cells per band)')
    ap.add_argument('--depths', default='',
                    help='comma-separated sequencing depths the distinct bands cycle through (default: the configuration\'s depth x 1/4, 2/3, 1, 5/2)')
    ap.add_argument('--no-extra', action='store_true',
                    help='skip the extra measurements of the default run (the same workload without any bound from earlier chromosomes, '
                         'the whole-genome configurations)')
    ap.add_argument('--stencil-only', action='store_true', help='time the stencil kernel alone (HPK_FLAG_NO_SCORE)')
    ap.add_argument('--no-probes', action='store_true',
                    help='skip the single-chromosome latency probes and the phase-timed launches after the timed region (profiling '
                         'runs: every stencil launch of the process then carries a whole group)')
    ap.add_argument('--structure', action='store_true',
                    help='bands with structure on top of the distance decay (synthetic.structure_fields: TAD blocks, a compartment '
                         'checkerboard, dense far-field patches) - what the record bounds, depth classes and lean column chunks are '
                         'not tuned on; the line reports what they did (passes_redone_in_full, passes_rescored, lean_redone)')
    ap.add_argument('--cpu-allcores-rows', type=int, default=0,
                    help='rows per process of the numpy port\'s all-cores leg (one process per core; 0 = skip: the native back-end is the all-cores baseline)')
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    import torch
    plan = launch_plan(args.gpus, os.environ, torch.cuda.device_count() if torch.cuda.is_available() else 0)
    if plan[0] == 'error':
        raise SystemExit('bench.py: ' + plan[1])
    if plan[0] == 'spawn':          # `python bench.py --gpus N` by hand: become N ranks, one per GPU
        import subprocess
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(plan[1]), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # HPK_BENCH_ONE_GPU (tests on a one-GPU box): the ranks share the GPUs there are; RCCL refuses two ranks on one device, so
    # the barrier / max / rank count then travel over gloo (CPU tensors) - the process plumbing and the arithmetic over
    # ranks are the ones of a real N-GPU run, the collective's transport is not
    shared = bool(os.environ.get('HPK_BENCH_ONE_GPU')) and world > torch.cuda.device_count()
    if shared:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dist = None
    ranks_seen = 1
    args.coll_device = torch.device('cpu') if shared else torch.device('cuda', local)
    if world > 1 or os.environ.get('HPK_BENCH_FORCE_DIST'):     # the env knob exercises the RCCL path on a 1-GPU box
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group('gloo' if shared else 'nccl', rank=rank, world_size=world)
        ones = torch.ones(1, dtype=torch.int32, device=args.coll_device)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        assert ranks_seen == world, 'all-reduce saw %d ranks of %d' % (ranks_seen, world)
    args.ranks_seen = ranks_seen
    args.backend = None if dist is None else ('gloo' if shared else 'nccl')

    from hicpeaks_amd import _lib, band, bandgen
    ctx = _lib.Context(local)
    if cfg.get('genome'):
        return run_genome(args, cfg, ctx, rank, world, local, dist)
    n = cfg['n']
    mw = min(cfg['ww'])
    D = cfg['maxapart'] // cfg['res']
    num = D + cfg['maxww'] + 1
    ld = (num + 63) // 64 * 64
    dev = torch.device('cuda', local)
    # The step's chromosomes are DISTINCT bands resident in HBM (--distinct, default 64: 64 x 51 MB = 3.3 GB at 10 kb, past the
    # 256 MiB Infinity Cache) whose depths cycle through --depths (default: the configuration's depth x 1/4, 2/3, 1, 5/2 -
    # 15 / 40 / 60 / 150 at 10 kb, which freeze at widths 5 / 6 / 6 / 8), so that the widening freezes at different widths from chromosome to chromosome: the record
    # bound, the halo and the survivors' bound are inherited from *other* chromosomes, and the chromosomes
    # that have to be computed or scored once more because of it are inside `value`.  SURVEY 8-D2's recipe and seeds, generated in HBM
    # (bandgen.device_band; the large configurations - 0.4-2 GB per band - keep one band).
    small = n * num <= 20_000_000
    ndist = max(1, args.distinct) if small else 1
    depths = [float(v) for v in args.depths.split(',')] if args.depths else \
        ([round(cfg['depth'] * f, 1) for f in (0.25, 2 / 3., 1.0, 2.5)] if small and ndist >= 4 else [cfg['depth']])
    bands, band_depth = [], []
    for sd in range(ndist):
        dp = depths[sd % len(depths)]
        raw_d, w_d, ir_d, b_d = bandgen.device_band(n, num, ld, mw, depth=dp, nloops=cfg['nloops'], seed=ndist * rank + sd, device=dev,
                                                    structure={} if args.structure else None)
        bands.append((raw_d, w_d, ir_d, b_d))
        band_depth.append(dp)
    nseeds = ndist
    order = np.random.default_rng(2024).permutation(ndist)       # which band the i-th chromosome of the run is: depths interleaved
    raw_d, w_d, ir_d, b_d = bands[0]
    if args.host_inputs or args.balanced_f64:
        nseeds = 1
    torch.cuda.synchronize()
    flags = _lib.FLAG_NO_SCORE if args.stencil_only else 0
    prm = _lib.make_params(_lib.MODE_BHFDR if cfg.get('mode') == 'bhfdr' else _lib.MODE_HICCUPS, cfg['pw'], cfg['ww'], cfg['maxww'], SIG, cfg['maxapart'], cfg['res'],
                           MIN_READS, flags)
    # The stencil's duration comes from two HIP events around its launch on the library's stream; a pair of events
    # idles the GPU ~6 us, so every TIMED_EVERY-th launch of the timed region is bracketed and the others run as a
    # production call does (HPK_FLAG_NO_STENCIL_TIMING).
    prm_quiet = _lib.make_params(_lib.MODE_BHFDR if cfg.get('mode') == 'bhfdr' else _lib.MODE_HICCUPS, cfg['pw'], cfg['ww'], cfg['maxww'], SIG, cfg['maxapart'], cfg['res'],
                                 MIN_READS, flags | _lib.FLAG_NO_STENCIL_TIMING)
    nsub = [0]
    px_per_step = band.band_pixels(n, num, mw, D) * len(cfg['pw'])

    # One step = one whole pass of the path over the chromosome (every kernel, the download and the host half).  As in
    # a run over many chromosomes, the next pass is submitted before the previous one is collected, so the host half
    # (Benjamini-Hochberg, result assembly) of pass i overlaps the kernels of pass i + 1; all K passes are submitted
    # and collected inside the timed region.
    depth = max(1, min(args.pipeline_depth, ctx.pipeline_depth))

    if args.host_inputs:
        raw_h = np.ascontiguousarray(raw_d[:, :num].cpu().numpy())
        w_h, ir_h, b_h = w_d.cpu().numpy(), ir_d.cpu().numpy(), b_d.cpu().numpy()

    bal_d = None
    if args.balanced_f64:       # balanced = (raw * w_r) * w_c on diagonals >= min(ww), NaN -> 0 (scripts/pyHICCUPS:150-158)
        kk = torch.arange(ld, device=dev)
        rr = torch.arange(n, device=dev)
        cc = (rr[:, None] + kk[None, :]).clamp(max=n - 1)
        bal_d = (raw_d.to(torch.float64) * w_d[:, None]) * w_d[cc]
        bal_d = torch.nan_to_num(bal_d, nan=0.0)
        bal_d[:, :mw] = 0
        bal_d[(rr[:, None] + kk[None, :]) >= n] = 0
        del kk, rr, cc
        torch.cuda.synchronize()

    # chromosomes per library call: the records and survivor regions of a group live side by side in HBM
    tiles = -(-n // 59) * -(-(59 + D - mw) // 107)
    per_band = tiles * 6313 * (4 + 17 * len(set(cfg['pw']))) + 2 * 40 * band.band_pixels(n, num, mw, D) * 2 * len(cfg['pw']) // 6
    group = args.group if args.group > 0 else max(1, min(64, int(60e9 // per_band)))
    batch = args.batch if args.batch > 0 else (1280 if n * num <= 60_000_000 else 16)
    batch = max(group, batch // group * group)          # whole groups

    by_depth = [[b for b in range(ndist) if band_depth[b] == dp] for dp in depths]
    sample_run = [0]            # > 0: runs of that many library calls per depth ("samples" scored one after the other)

    def band_index(i):
        if sample_run[0] > 0 and nseeds > 1:
            pool = by_depth[(i // (group * sample_run[0])) % len(depths)]
            return pool[i % len(pool)]
        return int(order[i % nseeds])

    def band_of(i):
        if bal_d is not None:
            return ctx._band(n, num, ld, raw_d.data_ptr(), bal_d.data_ptr(), None, ir_d.data_ptr(), b_d.data_ptr(), b_d.data_ptr(), True)
        r_, w_, i_, b_ = bands[band_index(i)]
        return ctx._band(n, num, ld, r_.data_ptr(), None, w_.data_ptr(), i_.data_ptr(), b_.data_ptr(), b_.data_ptr(), True)

    def submit(timed=None, k=None):
        """one group of `k` chromosomes (the passes rotate through the seeds' bands)"""
        k = group if k is None else k
        if timed is None:
            timed = nsub[0] % TIMED_EVERY == 0
        p = prm if timed else prm_quiet
        first = nsub[0] * group
        nsub[0] += 1
        if args.host_inputs:
            job = ctx.submit_batch_host([dict(raw=raw_h, IR=ir_h, bias1=b_h, bias2=b_h, weight=w_h)] * k, p)
        else:
            job = ctx.submit_batch([band_of(first + i) for i in range(k)], p, [n] * k)
        job._first = first
        return job

    fw_by_depth = {}

    call_log = []                       # HPK_BENCH_CALLS=1: (seconds since the first, wait inside results() in ms) of every collection -> stderr

    def take(job, done):
        t_c = time.perf_counter()
        rs = job.results()
        if os.environ.get('HPK_BENCH_CALLS'):
            call_log.append((t_c, (time.perf_counter() - t_c) * 1e3, max(r.timing['host_bh'] for r in rs), sum(r.timing['host_bh'] for r in rs),
                             sum(r.timing['d2h'] for r in rs), max(r.nsurv_cut for r in rs), sum(int(r.redone) for r in rs),
                             sum(int(r.rescored) for r in rs), sum(int(r.lean_redone) for r in rs), sum(r.timing['stencil'] for r in rs),
                             sum(r.timing['score'] for r in rs), max(r.timing['total'] for r in rs)))
        del done[:]
        done.append(rs[-1])              # the report needs the kernel times (below) and one result, not all of them
        for j, r in enumerate(rs):       # widths the widening froze at, by the depth of the band (what the bounds are inherited across)
            dp = band_depth[band_index(job._first + j)] if nseeds > 1 else band_depth[0]
            fw_by_depth.setdefault(dp, set()).add(int(r.frozen_w))
        nredone[0] += sum(int(r.redone) for r in rs)
        nrescored[0] += sum(int(r.rescored) for r in rs)
        for k_ in ('tiles', 'lean_tiles', 'lean_redone', 'lean_explicit'):
            lean_ct[k_] += sum(int(getattr(r, k_)) for r in rs)
        st = sum(r.timing['stencil'] for r in rs)        # the group's launch: its chromosomes' shares add up to it
        if st > 0:
            stencil_ms.append(st)
            sc = sum(r.timing['score'] for r in rs)      # ... and its scoring launch (one more event on the timed calls)
            if sc > 0:
                score_ms.append(sc)
                score_cand.append(sum(c_ for r in rs for _, _, c_, ex in r.steps if ex))

    def run(ngroups):
        pending, done = collections.deque(), []
        for _ in range(ngroups):
            pending.append(submit())
            if len(pending) >= depth:
                take(pending.popleft(), done)
        while pending:
            take(pending.popleft(), done)
        return done

    stencil_ms, score_ms, score_cand = [], [], []
    nredone, nrescored = [0], [0]
    lean_ct = collections.Counter()
    R = None
    run(depth)              # set-up, not a step: every lane allocates its workspaces (GBs on the large configurations) once
    for R in run(max(args.warmup, 1) * batch // group if args.warmup > 0 else 0):
        pass

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    del stencil_ms[:], score_ms[:], score_cand[:]
    fw_by_depth.clear()
    nredone[0] = nrescored[0] = 0
    lean_ct.clear()
    t0 = time.perf_counter()
    results = run(args.steps * batch // group)
    barrier()
    elapsed = time.perf_counter() - t0
    nredone_timed, nrescored_timed = nredone[0], nrescored[0]
    if call_log and rank == 0:
        for t_c, w, hmax, hsum, dsum, nmax, nred, nres, nlr, st_, sc_, tot_ in call_log:
            print('call at %8.1f ms: results() %7.2f ms; host_bh max %6.2f sum %7.2f, d2h sum %6.2f ms, most survivors %d; redone %d rescored %d lean_redone %d; '
                  'stencil %.2f score %.2f total %.2f ms' % ((t_c - call_log[0][0]) * 1e3, w, hmax, hsum, dsum, nmax, nred, nres, nlr, st_, sc_, tot_), file=sys.stderr)
    lean_timed = dict(lean_ct)
    fw_timed = {str(k): sorted(v) for k, v in sorted(fw_by_depth.items())}
    assert len(stencil_ms) >= args.steps * batch // group // TIMED_EVERY
    stencil_ms, score_ms_timed, score_cand_timed = list(stencil_ms), list(score_ms), list(score_cand)
    R = results[-1]
    # outside the timed region: latency of one synchronous single-chromosome call (submit + collect), then a group with
    # the per-phase events switched on (they cost ~6 us of idle GPU each, so the timed passes run without them)
    lat = []
    for _ in range(0 if args.no_probes else min(5, args.steps)):
        t1 = time.perf_counter()
        submit(False, 1).results()
        lat.append((time.perf_counter() - t1) * 1e3)
    prm = _lib.make_params(_lib.MODE_BHFDR if cfg.get('mode') == 'bhfdr' else _lib.MODE_HICCUPS, cfg['pw'], cfg['ww'], cfg['maxww'], SIG, cfg['maxapart'], cfg['res'],
                           MIN_READS, flags | _lib.FLAG_PHASE_TIMING)
    phases = {}
    if not args.no_probes:
        for _ in range(2):
            Rs = submit(True).results()
        phases = {k: float(sum(r.timing[k] for r in Rs)) / len(Rs) for k in Rs[0].timing}       # per chromosome of a group
        # (candidates resolved at the executed steps, per chromosome of the same group: what its scoring launch worked on)
        phase_scored = float(sum(c_ for r in Rs for _, _, c_, ex in r.steps if ex)) / len(Rs)
        phases['total'] = float(Rs[-1].timing['total']) / len(Rs)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=args.coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # ---- extra measurements of the default run (single GPU): the same workload with the memory of earlier chromosomes switched
    # off (option spec = 0: no record bound, the plan's own halo - every chromosome as if it were the first), and the
    # whole-genome configurations (BASELINE configs[2], configs[3]) as lines of their own under `extra`
    extra = {}
    no_bound = None
    if world == 1 and not args.no_extra and not args.stencil_only:
        ctx.set_option('spec', 0)
        nb_steps = max(1, args.steps // 5)
        run(max(2, depth) * 2)
        barrier()
        t1 = time.perf_counter()
        run(nb_steps * batch // group)
        barrier()
        no_bound = px_per_step * batch * nb_steps / (time.perf_counter() - t1)
        ctx.set_option('spec', 1)
        if nseeds > 1 and len(depths) > 1:
            # the same chromosomes sample by sample: runs of 5 library calls of one depth each - at every switch to a deeper
            # sample the first calls run under a bound that is too narrow and are computed once more (hpk_result::redone)
            sample_run[0] = 5
            run(2 * depth)
            barrier()
            nredone[0] = 0
            t1 = time.perf_counter()
            run(nb_steps * batch // group)
            barrier()
            extra['sample_runs'] = {'value': px_per_step * batch * nb_steps / (time.perf_counter() - t1), 'unit': 'band px/s',
                                    'calls_per_sample': 5, 'chromosomes': nb_steps * batch, 'redone_in_full': nredone[0]}
            sample_run[0] = 0
        # the mode the command lines and the drop-in functions ship (spec_halo = 2): the default workload once more under it
        ctx.set_option('spec_halo', 2)
        run(max(2, depth) * 2)
        barrier()
        nredone[0] = 0
        t1 = time.perf_counter()
        run(nb_steps * batch // group)
        barrier()
        extra['cli_mode'] = {'value': px_per_step * batch * nb_steps / (time.perf_counter() - t1), 'unit': 'band px/s', 'spec_halo': 2,
                             'chromosomes': nb_steps * batch, 'redone_in_full': nredone[0],
                             'note': 'the timed region above runs a Context of its own under the library default (spec_halo = 1)'}
        ctx.set_option('spec_halo', 1)
        if args.config == 'chr1_10kb':
            for name in ('wg_10kb_union', 'wg_5kb'):
                extra[name + '_cold'] = run_genome_cold(args, CONFIGS[name], ctx, local)
            for name in ('wg_10kb_union', 'wg_5kb'):
                o = run_genome(args, CONFIGS[name], ctx, rank, world, local, None, emit=False, steps=3, warmup=1)
                extra[name] = {'value': o['value'], 'unit': o['unit'], 'ms_per_step': o['ms_per_step'], 'steps': o['steps'],
                               'workload': o['config']['workload'], 'band_px_per_step': o['config']['band_px_per_step'],
                               'redone_in_full': o['config']['redone_in_full_rank0'], 'rescored': o['config']['rescored_rank0'],
                               'frozen_w': o['config']['frozen_w_rank0'], 'roofline_frac': o['roofline']['frac'],
                               'kernel_ms': o['roofline']['kernel_ms']}

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        st = float(np.mean(stencil_ms))                     # one launch = one group of chromosomes
        achieved = BYTES_PER_PX * px_per_step * group / (st * 1e-3) / 1e9
        out = {
            'metric': 'band pixels scored/sec (donut+LL)', 'value': world * px_per_step * batch * args.steps / elapsed,
            'unit': 'band px/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': cfg['workload'], 'name': args.config, 'band_px_per_step': px_per_step * batch,
                       'chromosomes_per_step': batch, 'chromosomes_per_launch': group, 'band_px_per_chromosome': px_per_step,
                       'distinct_bands': nseeds, 'depths': depths,
                       'frozen_w_by_depth': fw_timed,
                       'no_bound_value': no_bound,
                       'ms_per_chromosome': ms_step / batch, 'ranks_seen': args.ranks_seen, 'collective_backend': args.backend,
                       'candidates': R.ncand, 'significant_px': int(sum(s['x'].size for s in R.sets)),
                       'px_with_p_le_sig': R.nsurv_sig, 'records_copied_back': R.nsurv_cut,
                       # candidates resolved beyond the width the widening freezes at are dropped by the scoring kernel; the
                       # stencil leaves their records out, bounded by the previous pass's frozen width (HPK_SPEC=0: no bound)
                       'record_bound_w': R.record_bound, 'frozen_w': R.frozen_w, 'passes_redone_in_full': nredone_timed,
                       'passes_rescored': nrescored_timed,
                       # tiles of the timed region built without their f64 plane (hpk_stencil_s, lean tiles), of those computed once
                       # more in full, candidates of lean tiles whose sums were formed cell by cell
                       'tiles': lean_timed.get('tiles', 0), 'lean_tiles': lean_timed.get('lean_tiles', 0),
                       'lean_redone': lean_timed.get('lean_redone', 0), 'lean_explicit': lean_timed.get('lean_explicit', 0),
                       'parallelism': 'one chromosome per GPU, no collective', 'pipeline_depth': depth,
                       'per_rank_px': [px_per_step * batch] * world, 'lpt_imbalance': 1.0,      # (every rank scores its own chromosomes of one shape)
                       'sync_call_ms': float(np.median(lat)) if lat else None,
                       'stencil_only': bool(args.stencil_only), 'host_inputs': bool(args.host_inputs), 'balanced_f64': bool(args.balanced_f64),
                       'structure': bool(args.structure)},
            # frac = frac_20B_equivalent: SURVEY.md §8-D3's convention, 20 B per band pixel per pair (4 B read + 2 x 8 B local expected
            # written) - an "equivalent bytes" figure that measures the kernel against the contract's memory floor, NOT HBM
            # utilisation: the kernels write compact records for the candidates that can count, so the bytes they really move are
            # fewer (frac_measured: rocprofv3's counters / kernel time / peak; compact_4Bpx: 8-D3's figure for a compacting mode).
            # What bounds the kernel is vector issue: roofline_valu below; `bound` names the roof with the larger fraction.
            'roofline': {'bound': 'hbm', 'bound_by_contract': 'hbm', 'kernel': 'hpk_stencil_s (+ hpk_stencil_lean)', 'achieved': achieved,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'frac_20B_equivalent': achieved / HBM_PEAK_GBS, 'frac_measured': None, 'frac_label': FRAC_LABEL, 'traffic': None,
                         'kernel_ms': st, 'kernel_ms_per_chromosome': st / group,
                         'algorithmic_bytes_per_launch': BYTES_PER_PX * px_per_step * group,
                         'launches_timed': len(stencil_ms), 'launches': args.steps * batch // group,
                         'compact_4Bpx': {'achieved': 4.0 * px_per_step * group / (st * 1e-3) / 1e9,
                                          'frac': 4.0 * px_per_step * group / (st * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         'hbm_frac_measured': None},
            # per chromosome, from two launches AFTER the timed region with events between the kernels (HPK_FLAG_PHASE_TIMING:
            # five more event pairs, ~3 us of idle GPU each - the stages come out 15-30 % above what they take in the timed
            # region, whose launches run without them; roofline.kernel_ms is the timed region's own figure): the split, not the sum
            'phases_ms': phases,
            'phases_note': 'instrumented launches after the timed region (events between the kernels): +15-30 %, for the split only',
            'extra': extra,
        }
        if score_ms_timed or phases.get('score'):
            # The second kernel, hpk_score (a gather kernel: issue-bound), against the same roof.  Algorithmic
            # bytes per scored candidate and pair: its record (4 B entry + 1 B step + 16 B sums) and what its expected value is
            # formed from (IR[d], B1[r], B2[c], two local-expected table entries: 40 B); candidates = those resolved at the
            # executed steps.  Time and candidates: the timed region's own launches (every TIMED_EVERY-th call carries one more
            # event behind its scoring kernel); without them (stencil-only, no timed call) the instrumented launch after it.
            if score_ms_timed:
                sc_ms, nrec, src = float(np.mean(score_ms_timed)) / group, float(np.mean(score_cand_timed)) / group, \
                    'events on every %dth call of the timed region (%d launches)' % (TIMED_EVERY, len(score_ms_timed))
            else:
                sc_ms, nrec, src = phases['score'], phase_scored, 'instrumented launch (phases_ms.score)'
            sbytes = nrec * 61.0
            sach = sbytes / (sc_ms * 1e-3) / 1e9
            out['roofline_score'] = {'bound': 'valu-issue', 'bound_by_contract': 'hbm', 'kernel': 'hpk_score', 'achieved': sach, 'peak': HBM_PEAK_GBS,
                                     'unit': 'GB/s', 'frac': sach / HBM_PEAK_GBS, 'frac_61B_equivalent': sach / HBM_PEAK_GBS, 'frac_measured': None,
                                     'traffic': None, 'kernel_ms_per_chromosome': sc_ms,
                                     'candidates_scored_per_chromosome': nrec, 'algorithmic_bytes_per_candidate': 61.0,
                                     'time_source': src}
        try:        # HBM traffic and vector instructions per launch of the dominant kernel, measured off-line with rocprofv3 --pmc (profiles/)
            tr = json.load(open(os.path.join(REPO, 'profiles', 'traffic.json'))).get(args.config)
            if tr:
                # (counters are recorded per chromosome; a launch carries a group of them)
                out['roofline']['traffic'] = tr['traffic_bytes'] * group
                out['roofline']['traffic_source'] = tr['source']
                fm = tr['traffic_bytes'] * group / (st * 1e-3) / 1e9 / HBM_PEAK_GBS
                out['roofline']['hbm_frac_measured'] = out['roofline']['frac_measured'] = fm
                if 'roofline_score' in out and tr.get('score_traffic_bytes'):
                    out['roofline_score']['traffic'] = tr['score_traffic_bytes']      # per chromosome, like its time
                    out['roofline_score']['frac_measured'] = tr['score_traffic_bytes'] / (out['roofline_score']['kernel_ms_per_chromosome'] * 1e-3) / 1e9 / HBM_PEAK_GBS
                if tr.get('valu_insts'):
                    # vector issue: SQ_INSTS_VALU (wave-instructions, per chromosome) x 4 cycles / (1 024 SIMDs x clock x kernel time)
                    def valu(n_inst, ms):
                        return n_inst * VALU_CYCLES / (N_SIMD * CLOCK_GHZ * 1e9 * ms * 1e-3)
                    vf = valu(tr['valu_insts'], st / group)
                    out['roofline_valu'] = {'bound': 'valu-issue', 'kernel': out['roofline']['kernel'], 'achieved': vf, 'peak': 1.0,
                                            'unit': 'fraction of the vector issue slots', 'frac': vf,
                                            'valu_wave_insts_per_chromosome': tr['valu_insts'], 'cycles_per_inst': VALU_CYCLES,
                                            'simds': N_SIMD, 'clock_ghz': CLOCK_GHZ,
                                            'valu_lane_insts_per_band_px': tr['valu_insts'] * 64.0 / px_per_step * len(cfg['pw']),
                                            'floor_ms_per_chromosome': tr['valu_insts'] * VALU_CYCLES / (N_SIMD * CLOCK_GHZ * 1e9) * 1e3,
                                            'source': tr['source']}
                    if vf > max(fm, 0.0):
                        out['roofline']['bound'] = 'valu-issue'
                    if 'roofline_score' in out and tr.get('score_valu_insts'):
                        out['roofline_score']['valu_frac'] = valu(tr['score_valu_insts'], out['roofline_score']['kernel_ms_per_chromosome'])
                # scalar issue: SQ_INSTS_SALU x 2 cycles (measured 2.07 with the loop around it: profiles/r06_instruction_cost.txt) / (256 CUs x clock x kernel time) -
                # the CU's one scalar unit serves its four SIMDs
                def salu(n_inst, ms):
                    return n_inst * SALU_CYCLES / (N_CU * CLOCK_GHZ * 1e9 * ms * 1e-3)
                if tr.get('salu_insts') and 'roofline_valu' in out:
                    out['roofline_valu']['salu_frac'] = salu(tr['salu_insts'], st / group)
                if 'roofline_score' in out and tr.get('score_salu_insts'):
                    out['roofline_score']['salu_frac'] = salu(tr['score_salu_insts'], out['roofline_score']['kernel_ms_per_chromosome'])
                    if out['roofline_score']['salu_frac'] > out['roofline_score'].get('valu_frac', 0.0):
                        out['roofline_score']['bound'] = 'salu-issue'
        except Exception:
            pass
        if world == 1 and args.cpu_rows > 0:
            out['cpu_baseline'] = cpu_baseline(cfg, min(args.cpu_rows, n), args.cpu_allcores_rows)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
