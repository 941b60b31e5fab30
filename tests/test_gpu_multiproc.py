"""The N-process paths on real hardware (not even the process plumbing had met a GPU), and the command lines'
--deterministic mode.  A one-GPU box is enough: the workers / ranks share device 0 (HPK_CLI_SHARE_GPU, HPK_BENCH_ONE_GPU).

  * `--nproc 2`: two spawned worker processes around the shared largest-first queue (multiprocessing counter),
  * `torchrun --nproc-per-node 2 scripts/pyHICCUPS ... --device 0`: two ranks, the queue's counter in the c10d store, tables
    gathered over gloo,
  * `bench.py` with HPK_BENCH_FORCE_DIST=1 (RCCL init, all-reduce and barrier with one rank) and as two torchrun ranks on one GPU,
  * the default (spec_halo = 2; formerly --deterministic): byte-identical text whatever the chromosome order, the batching and the number of workers (the
    reference's output does not depend on its map order, scripts/pyHICCUPS:192-210); under --history-dependent coordinates and counts
    are identical and the '%.3g' fields flip next to never - the measured rate is asserted to stay below 1 in 500.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from hicpeaks_amd import cli, io, synthetic

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEPTHS = [25.0, 90.0, 40.0, 60.0, 150.0, 12.0, 60.0, 25.0, 90.0, 40.0]
N, NUM, RES, MAXAPART = 2100, 211, 10000, 2000000


@pytest.fixture(scope='module')
def bands():
    """ten chromosomes of one size and different depths: the widening freezes at different widths"""
    return [synthetic.synth_band(N, NUM, depth=dp, nloops=25, seed=500 + i)[:2] for i, dp in enumerate(DEPTHS)]


def _archive(path, bands, perm):
    """chromosome i of the file = band perm[i]: same sizes, so the largest-first queue runs them in the file's order"""
    io.save_band_archive(path, RES, {'chr%d' % (i + 1): bands[b] for i, b in enumerate(perm)})


def _argv(out, arc, log, extra=()):
    return ['-O', out, '-p', arc, '-C', '#', '--pw', '2', '--ww', '5', '--maxapart', str(MAXAPART),
            '--logFile', log] + list(extra)


def _by_band(path, perm):
    """{band index: its lines with the chromosome name taken out}"""
    out = {b: [] for b in perm}
    for l in open(path).read().splitlines():
        f = l.split('\t')
        out[perm[int(f[0][3:]) - 1]].append('\t'.join(f[1:3] + f[4:]))
    return out


def _free_port():
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def test_deterministic_mode_is_order_and_batch_independent(bands, tmp_path, monkeypatch):
    perm_a = list(range(10))
    perm_b = [7, 2, 9, 0, 5, 3, 8, 1, 6, 4]
    arcs = {}
    for tag, perm in (('a', perm_a), ('b', perm_b)):
        arcs[tag] = str(tmp_path / (tag + '.npz'))
        _archive(arcs[tag], bands, perm)
    log = str(tmp_path / 'log.txt')

    def run(tag, perm, group, det, nproc=1):
        monkeypatch.setattr(cli, 'GROUP_CHROMS', group)
        out = str(tmp_path / ('%s_%d_%d_%d.bedpe' % (tag, group, det, nproc)))
        extra = ([] if det else ['--history-dependent']) + (['--nproc', str(nproc)] if nproc > 1 else [])     # (deterministic: the default)
        assert cli.main_hiccups(_argv(out, arcs[tag], log, extra)) == 0
        return _by_band(out, perm)

    ref = run('a', perm_a, 8, True)
    assert sum(len(v) for v in ref.values()) >= 40
    widths = set()
    for l in open(log).read().splitlines():
        if 'Valid Contact Number from This Loop' in l:
            widths.add(l.split('(')[1].split(')')[0])
    assert len(widths) >= 3                                    # the chromosomes do not all freeze at the same width
    # other order, other batching: byte-identical per band
    assert run('b', perm_b, 8, True) == ref
    assert run('a', perm_a, 1, True) == ref
    assert run('b', perm_b, 3, True) == ref
    # two worker processes on the one GPU
    monkeypatch.setenv('HPK_CLI_SHARE_GPU', '1')
    assert run('b', perm_b, 2, True, nproc=2) == ref
    monkeypatch.delenv('HPK_CLI_SHARE_GPU')
    assert 'independent of chromosome order' in open(log).read()
    # --history-dependent (the layout a chromosome inherits, no second pass): same pixels and counts; statistics equal to rounding,
    # the printed '%.3g' fields flip next to never
    fields = flips = 0
    for tag, perm, group in (('a', perm_a, 8), ('b', perm_b, 8), ('b', perm_b, 1)):
        got = run(tag, perm, group, False)
        for b in ref:
            assert len(got[b]) == len(ref[b])
            for g, w in zip(sorted(got[b]), sorted(ref[b])):
                gf, wf = g.split('\t'), w.split('\t')
                assert gf[:8] == wf[:8]
                for u, v in zip(gf[8:], wf[8:]):
                    fields += 1
                    if u != v:
                        flips += 1
                        assert abs(float(u) - float(v)) <= 1.01e-2 * abs(float(v))
    print('adaptive halo vs deterministic: %d of %d printed statistics differ' % (flips, fields))
    assert flips <= max(1, fields // 500)


def test_nproc_two_workers_share_the_queue(bands, tmp_path, monkeypatch):
    arc = str(tmp_path / 'in.npz')
    _archive(arc, bands, list(range(10)))
    log = str(tmp_path / 'log.txt')
    one, two = str(tmp_path / 'one.bedpe'), str(tmp_path / 'two.bedpe')
    assert cli.main_hiccups(_argv(one, arc, log, ['--deterministic'])) == 0
    monkeypatch.setenv('HPK_CLI_SHARE_GPU', '1')
    assert cli.main_hiccups(_argv(two, arc, log, ['--deterministic', '--nproc', '2'])) == 0
    assert open(one).read() == open(two).read() and len(open(one).read().splitlines()) >= 40
    assert '--nproc 2: 2 worker(s)' in open(log).read()
    # pyBHFDR likewise
    one, two = str(tmp_path / 'one.txt'), str(tmp_path / 'two.txt')
    argv = ['-p', arc, '-C', '#', '--logFile', log, '--deterministic']
    assert cli.main_bhfdr(['-O', one] + argv) == 0
    assert cli.main_bhfdr(['-O', two, '--nproc', '2'] + argv) == 0
    assert open(one).read() == open(two).read() and len(open(one).read().splitlines()) >= 40


def test_torchrun_two_ranks_on_one_gpu(bands, tmp_path):
    arc = str(tmp_path / 'in.npz')
    _archive(arc, bands, list(range(10)))
    log = str(tmp_path / 'log.txt')
    one, two = str(tmp_path / 'one.bedpe'), str(tmp_path / 'two.bedpe')
    assert cli.main_hiccups(_argv(one, arc, log, ['--deterministic'])) == 0
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(REPO, 'scripts', 'pyHICCUPS')] + \
        _argv(two, arc, str(tmp_path / 'log2.txt'), ['--deterministic', '--device', '0'])
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    assert open(one).read() == open(two).read()


def _bench(env_extra, args, launcher=()):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', **env_extra)
    cmd = [sys.executable] + list(launcher) + [os.path.join(REPO, 'bench.py'), '--config', 'tiny', '--steps', '2', '--warmup', '1',
                                               '--cpu-rows', '0', '--no-extra', '--no-probes', '--batch', '8', '--group', '4'] + list(args)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=REPO)
    assert r.returncode == 0, (r.stdout.decode()[-2000:], r.stderr.decode()[-3000:])
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1                                     # rank 0 prints ONE line
    return json.loads(lines[0])


def test_bench_rccl_path_with_one_rank():
    out = _bench({'HPK_BENCH_FORCE_DIST': '1', 'MASTER_PORT': str(_free_port())}, [])
    assert out['n_gpus'] == 1 and out['config']['ranks_seen'] == 1 and out['config']['collective_backend'] == 'nccl'
    assert out['value'] > 0 and out['roofline']['frac'] > 0


def test_bench_two_ranks_on_one_gpu():
    launcher = ['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                '--master-port', str(_free_port())]
    out = _bench({'HPK_BENCH_ONE_GPU': '1'}, ['--gpus', '2'], launcher)
    assert out['n_gpus'] == 2 and out['config']['ranks_seen'] == 2 and out['scaling'] == 'weak'
    one = _bench({}, [])
    # whole-job value of two ranks time-slicing one GPU: about the single process's (weak scaling arithmetic: world x per-rank work)
    assert 0.4 * one['value'] < out['value'] < 1.6 * one['value']


# ---------------------------------------------------------------------------------------------------------------------
# The target machine's world size (8 GPUs per node) without a node: eight ranks / eight workers share the one GPU.  What is
# checked is everything but the collective's transport: the largest-first deal, that every chromosome is scored exactly once,
# the gathering of the ranks' results, and that the answer does not depend on the number of workers.
def _bench_genome(env_extra, args, launcher=()):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', **env_extra)
    cmd = [sys.executable] + list(launcher) + [os.path.join(REPO, 'bench.py'), '--config', 'wg_10kb_union', '--steps', '1', '--warmup', '0',
                                               '--cpu-rows', '0', '--no-extra', '--no-probes'] + list(args)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=REPO)
    assert r.returncode == 0, (r.stdout.decode()[-2000:], r.stderr.decode()[-3000:])
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_eight_ranks_deal_the_genome():
    launcher = ['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
                '--master-port', str(_free_port())]
    out = _bench_genome({'HPK_BENCH_ONE_GPU': '1'}, ['--gpus', '8'], launcher)
    c = out['config']
    assert out['n_gpus'] == 8 and c['ranks_seen'] == 8 and out['scaling'] == 'strong'
    # every chromosome of the genome scored by exactly one rank
    names = sorted(synthetic.hg38_bins(10000))
    assert c['chromosomes_scored'] == 23 and len(c['chromosomes_by_rank']) == 8
    assert sorted(x for part in c['chromosomes_by_rank'] for x in part) == names
    # the largest-first deal: the heaviest rank within 5 % of the mean (1.045 on 8 GPUs), chr1 alone on its rank's top
    px = c['per_rank_px']
    assert len(px) == 8 and sum(px) == c['band_px_per_step']
    assert c['lpt_imbalance'] == max(px) / (sum(px) / 8.0) and c['lpt_imbalance'] <= 1.05
    assert c['chromosomes_by_rank'][0][0] == '1'
    # ... and the ranks' results add up to the single process's: the same candidates, the same significant pixels (a pixel within
    # 1e-8 of a threshold may fall either way under another history of layouts: library default spec_halo = 1)
    one = _bench_genome({}, [])
    c1 = one['config']
    assert c1['chromosomes_by_rank'] == [sorted(names, key=lambda k: (-synthetic.hg38_bins(10000)[k], str(k)))]
    assert c['candidates_all_ranks'] == c1['candidates_all_ranks'] > 10 ** 7
    assert abs(c['significant_px_all_ranks'] - c1['significant_px_all_ranks']) <= 3 and c1['significant_px_all_ranks'] > 1000
    assert out['value'] > 0 and one['value'] > 0


def test_cli_eight_workers_on_a_genome(tmp_path, monkeypatch):
    """scripts/pyHICCUPS --nproc 8 on an archive of 23 chromosomes of 23 sizes (hg38's proportions) against --nproc 1"""
    sizes = synthetic.hg38_bins(10000)
    chroms = {}
    for i, c in enumerate(sorted(sizes, key=lambda k: -sizes[k])):
        n = max(NUM + 40, sizes[c] // 12)
        chroms['chr' + c] = synthetic.synth_band(n, NUM, depth=DEPTHS[i % len(DEPTHS)], nloops=max(4, n // 80), seed=900 + i)[:2]
    arc = str(tmp_path / 'genome.npz')
    io.save_band_archive(arc, RES, chroms)
    log = str(tmp_path / 'log.txt')
    one, eight = str(tmp_path / 'one.bedpe'), str(tmp_path / 'eight.bedpe')
    argv = ['-p', arc, '-C', '#', 'X', '--pw', '2', '--ww', '5', '--maxapart', str(MAXAPART), '--logFile', log]
    assert cli.main_hiccups(['-O', one] + argv) == 0
    monkeypatch.setenv('HPK_CLI_SHARE_GPU', '1')
    assert cli.main_hiccups(['-O', eight, '--nproc', '8'] + argv) == 0
    text = open(one).read()
    assert open(eight).read() == text
    seen = set(l.split('\t')[0] for l in text.splitlines())
    assert len(text.splitlines()) >= 60 and len(seen) >= 15             # pixels from most of the 23 chromosomes
    assert '--nproc 8: 8 worker(s)' in open(log).read()


def test_history_dependent_run_puts_the_shared_context_back(bands, tmp_path, monkeypatch):
    """An in-process --history-dependent run scores on the process-wide context the drop-in hiccups() / bhfdr() use too: it must leave
    that context under spec_halo = 2 again (advisor, round 5), and HPK_SPEC_HALO in the environment wins over either flag."""
    from hicpeaks_amd import _lib
    arc = str(tmp_path / 'in.npz')
    _archive(arc, bands[:3], [0, 1, 2])
    calls = []
    real = _lib.Context.set_option

    def spy(self, name, value):
        calls.append((name, int(value)))
        return real(self, name, value)
    monkeypatch.setattr(_lib.Context, 'set_option', spy)
    log = str(tmp_path / 'log.txt')
    assert cli.main_hiccups(_argv(str(tmp_path / 'a.bedpe'), arc, log, ['--history-dependent'])) == 0
    halo = [v for n, v in calls if n == 'spec_halo']
    assert halo and halo[0] == 1 and halo[-1] == 2
    del calls[:]
    assert cli.main_hiccups(_argv(str(tmp_path / 'b.bedpe'), arc, log)) == 0
    halo = [v for n, v in calls if n == 'spec_halo']
    assert halo and set(halo) == {2}
    del calls[:]
    monkeypatch.setenv('HPK_SPEC_HALO', '1')
    assert cli.main_hiccups(_argv(str(tmp_path / 'c.bedpe'), arc, log)) == 0
    assert not [v for n, v in calls if n == 'spec_halo']          # the environment's choice stands
