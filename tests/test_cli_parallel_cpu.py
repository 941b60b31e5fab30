"""Host-side plumbing that needs no GPU: flag parity with the reference command lines, chromosome selection,
band archives, the largest-first work queue (one process, multiprocessing workers, world_size-2 over gloo), LPT
sharding and the gather on rank 0."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO
from hicpeaks_amd import cli, io, parallel, synthetic


def test_hiccups_flags_and_defaults_match_reference():
    # scripts/pyHICCUPS:12-81
    a = cli._hiccups_parser().parse_args(['-O', 'x', '-p', 'y', '--pw', '1', '2', '--ww', '3', '5'])
    assert (a.maxww, a.siglevel, a.sumq, a.double_fold, a.single_fold) == (10, 0.05, 0.01, 1.75, 2)
    assert (a.clr_weight_name, a.use_raw, a.min_marginal_peaks, a.min_local_reads) == ('weight', False, 2, 16)
    assert (a.only_anchors, a.maxapart, a.nproc, a.chroms, a.logFile) == (False, 10000000, 1, ['#', 'X'], 'pyHICCUPS.log')
    assert a.pw == [1, 2] and a.ww == [3, 5]


def test_bhfdr_flags_and_defaults_match_reference():
    # scripts/pyBHFDR:12-58
    a = cli._bhfdr_parser().parse_args(['-O', 'x', '-p', 'y'])
    assert (a.pw, a.ww, a.maxww, a.siglevel, a.maxapart, a.nproc, a.logFile) == (2, 5, 10, 0.05, 2000000, 1, 'pyBHFDR.log')


def test_select_chroms_rule():
    names = ['chr1', 'chr2', 'chrX', 'chrY', 'chrM', 'chr10_random']
    assert cli.select_chroms(names, ['#', 'X']) == ['chr1', 'chr2', 'chrX']
    assert cli.select_chroms(names, []) == names
    assert cli.select_chroms(names, ['Y']) == ['chrY']


def test_band_archive_roundtrip(tmp_path):
    raw, w, _ = synthetic.synth_band(120, 31, depth=5.0, seed=1)
    io.save_band_archive(str(tmp_path / 'b.npz'), 10000, {'chr1': (raw, w), 'chrX': (raw[:80], w[:80])})
    src = io.open_source(str(tmp_path / 'b.npz'))
    assert src.binsize == 10000 and src.chromnames == ['chr1', 'chrX'] and src.nbins('chrX') == 80
    r, ww, _ = src.fetch("chr1", 25)
    assert r.shape == (120, 25) and r.dtype == np.float32
    np.testing.assert_array_equal(r, raw[:, :25])
    np.testing.assert_array_equal(ww, w)
    r, _, _ = src.fetch("chr1", 40)        # wider than stored: zero padded
    assert r.shape == (120, 40) and not r[:, 31:].any()


@pytest.mark.parametrize('compressed', [True, False])
def test_band_archive_members_mapped_or_read(tmp_path, compressed):
    """Uncompressed archives are mapped member by member (no copy, no CRC pass), compressed ones read by numpy: same
    arrays either way, and a band stored as f32 [n, num] reaches the library without a second copy."""
    raw, w, _ = synthetic.synth_band(90, 27, depth=5.0, seed=3)
    raw = raw.astype(np.float32)
    path = str(tmp_path / 'a.npz')
    io.save_band_archive(path, 5000, {'chr2': (raw, w), 'chr3': (raw[:50], w[:50])}, compressed=compressed)
    src = io.open_source(path)
    assert src.binsize == 5000 and src.chromnames == ['chr2', 'chr3'] and src.nbins('chr3') == 50
    for c, rr, wwant in (('chr2', raw, w), ('chr3', raw[:50], w[:50])):
        r, ww, _ = src.fetch(c, 27)
        assert r.dtype == np.float32 and r.flags.c_contiguous
        np.testing.assert_array_equal(np.asarray(r), rr)
        np.testing.assert_array_equal(ww, wwant)
        assert isinstance(r, np.memmap) == (not compressed)
    r, _, _ = src.fetch("chr2", 20)
    np.testing.assert_array_equal(r, raw[:, :20])


def test_lpt_partition_balances_hg38():
    sizes = synthetic.hg38_bins(10000)
    parts = parallel.lpt_partition(sizes, 8)
    assert sorted(c for p in parts for c in p) == sorted(sizes)
    loads = [sum(sizes[c] for c in p) for p in parts]
    assert max(loads) / (sum(loads) / 8.0) < 1.06          # 1.045
    assert parallel.lpt_partition(sizes, 1) == [sorted(sizes, key=lambda k: (-sizes[k], k))]
    assert parallel.lpt_partition({'a': 1}, 4)[0] == ['a']


WORKER = r'''
import os, sys
sys.path.insert(0, %(repo)r)
import torch.distributed as dist
from hicpeaks_amd import parallel
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
sizes = {'chr%%d' %% i: 100 - i for i in range(1, 8)}
seen = []
def score(c):
    seen.append(c)
    return {(1, 2): (rank, c)}
out = parallel.run_sharded(sizes, score, rank, world)
mine = parallel.lpt_partition(sizes, world)[rank]
assert seen == mine, (seen, mine)
if rank == 0:
    assert sorted(out) == sorted(sizes)
    for r in range(world):
        for c in parallel.lpt_partition(sizes, world)[r]:
            assert out[c] == {(1, 2): (r, c)}
    print('GATHER_OK', len(out))
else:
    assert out is None
dist.destroy_process_group()
'''


def test_run_sharded_world2_gloo(tmp_path):
    script = tmp_path / 'w.py'
    script.write_text(WORKER % dict(repo=REPO))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29631', WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert 'GATHER_OK 7' in outs[0]


def test_run_sharded_batch_fn_single_rank():
    """batch_fn path of run_sharded (the one-ahead loop hands back tables in the order of the chromosomes it got)."""
    from hicpeaks_amd import parallel
    sizes = {'1': 100, '2': 80, 'X': 60}
    seen = []

    def batch(chroms):
        seen.append(list(chroms))
        return [{'chrom': c} for c in chroms]
    out = parallel.run_sharded(sizes, None, 0, 1, batch_fn=batch)
    assert seen == [parallel.lpt_partition(sizes, 1)[0]]
    assert out == {c: {'chrom': c} for c in sizes}


def test_bench_gpus_flag_is_honoured_or_refused():
    """bench.py --gpus N: never a silent single-GPU run.  launch_plan() is the whole decision."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench', os.path.join(REPO, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    lp = bench.launch_plan
    assert lp(1, {}, 1) == ('run',)
    assert lp(1, {}, 0)[0] == 'error'                                   # no GPU: there is no CPU path
    kind, port = lp(8, {}, 8)                                           # by hand on an 8-GPU node: become 8 ranks
    assert kind == 'spawn' and 1024 < port < 65536
    assert lp(2, {}, 1)[0] == 'error' and 'only 1 GPU' in lp(2, {}, 1)[1]   # 1-GPU box: refuse, do not print n_gpus: 1
    assert lp(8, {'WORLD_SIZE': '8', 'RANK': '3'}, 8) == ('run',)      # the driver's torchrun line
    assert lp(8, {'WORLD_SIZE': '4'}, 8)[0] == 'error'                  # flag and launcher disagree
    assert lp(4, {'WORLD_SIZE': '4'}, 2)[0] == 'error'                  # more ranks than GPUs
    assert lp(1, {'WORLD_SIZE': '1'}, 1) == ('run',)


# ----------------------------------------------------------------------------- the shared largest-first queue
def test_work_queue_hands_out_every_chromosome_once_largest_first():
    sizes = synthetic.hg38_bins(10000)
    q = parallel.WorkQueue(sizes, parallel.local_counter())
    got = list(q)
    assert got == parallel.largest_first(sizes) and got[0] == '1' and len(got) == 23
    assert q.take() is None and q.take() is None                        # empty stays empty


def test_queue_beats_static_shares_when_costs_are_mispredicted():
    """cost follows the candidates, not the bins (1.5x spread per tile class).  With the real
    costs off the estimate by up to 1.5x, one queue keeps eight workers busy where static LPT shares leave the unlucky
    worker behind; with exact estimates the two are within a few percent."""
    sizes = synthetic.hg38_bins(5000)
    rng = np.random.default_rng(0)
    worse = []
    for trial in range(50):
        costs = {c: s * rng.uniform(1.0, 1.5) for c, s in sizes.items()}
        mq = parallel.simulate(costs, 8, 'queue', estimate=sizes)
        ml = parallel.simulate(costs, 8, 'lpt', estimate=sizes)
        ideal = sum(costs.values()) / 8.0
        assert mq >= ideal and ml >= ideal
        worse.append(ml / mq)
    assert np.mean(worse) > 1.03 and min(worse) > 0.97, (np.mean(worse), min(worse))
    exact = parallel.simulate(sizes, 8, 'queue') / parallel.simulate(sizes, 8, 'lpt')
    assert 0.95 < exact < 1.08
    # a slow worker (say a GPU shared with another job) takes fewer chromosomes instead of holding the others up
    slow = [1.0] * 7 + [0.5]
    assert parallel.simulate(sizes, 8, 'queue', speeds=slow) < 0.85 * parallel.simulate(sizes, 8, 'lpt', speeds=slow)


def _mp_drain(sizes, value, out, delay):
    import time
    q = parallel.WorkQueue(sizes, parallel.mp_counter(value))
    mine = []
    for c in q:
        time.sleep(delay)
        mine.append(c)
    out.put(mine)


def test_work_queue_across_processes():
    """Two worker processes around one multiprocessing counter (--nproc 2): every chromosome exactly once, and the worker
    that is four times slower ends up with fewer of them."""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    sizes = {'chr%d' % i: 100 - i for i in range(1, 21)}
    value, out = ctx.Value('i', 0), ctx.Queue()
    procs = [ctx.Process(target=_mp_drain, args=(sizes, value, out, d)) for d in (0.02, 0.08)]
    for p in procs:
        p.start()
    parts = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(parts[0] + parts[1]) == sorted(sizes)
    assert {len(parts[0]), len(parts[1])} != {10} and min(len(parts[0]), len(parts[1])) >= 1


QUEUE_WORKER = r'''
import os, sys, time
sys.path.insert(0, %(repo)r)
import torch.distributed as dist
from hicpeaks_amd import parallel
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
sizes = {'chr%%d' %% i: 100 - i for i in range(1, 12)}
queue = parallel.WorkQueue(sizes, parallel.store_counter())

class Call(object):
    def __init__(self, c): self.c = c
    def result(self):
        time.sleep(0.01 * (1 + 3 * rank))          # rank 1 is slower
        return {(1, 2): (rank, self.c)}
local = parallel.run_queue(queue, Call, depth=2)
out = parallel.gather_tables(local, rank, world)
if rank == 0:
    assert sorted(out) == sorted(sizes), sorted(out)
    by_rank = [sum(1 for v in out.values() if v[(1, 2)][0] == r) for r in range(world)]
    assert sum(by_rank) == len(sizes) and all(b >= 1 for b in by_rank), by_rank
    print('QUEUE_OK', by_rank)
else:
    assert out is None
dist.destroy_process_group()
'''


def test_work_queue_world2_gloo(tmp_path):
    """torchrun-style launch, world_size 2 on CPU: the ranks share the queue through the process group's key-value store
    (an atomic add per chromosome), each keeps two chromosomes in flight, rank 0 gathers every table exactly once."""
    script = tmp_path / 'q.py'
    script.write_text(QUEUE_WORKER % dict(repo=REPO))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29637', WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert 'QUEUE_OK' in outs[0]
