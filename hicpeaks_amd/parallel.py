"""Chromosome-level sharding.  Every statistic of the path is per chromosome (candidates, frozen_w, lambda chunks,
BH, gaps, clustering), so the GPUs of a node never exchange data: work items are whole chromosomes, taken largest
first from ONE queue by whichever GPU worker has room (`WorkQueue`; `lpt_partition` is the static hand-out bench.py
uses), and the only communication is that queue's counter and gathering the per-chromosome tables on rank 0.

Counterpart of `Pool(args.nproc).map(worker, Params)` in scripts/pyHICCUPS:192-198.
"""
import itertools
import os


def lpt_partition(sizes, nworkers):
    """Longest-processing-time-first assignment.  sizes: {item: cost}.  Returns a list of item lists per worker,
    items of one worker in decreasing cost."""
    loads = [0] * nworkers
    parts = [[] for _ in range(nworkers)]
    for item in sorted(sizes, key=lambda k: (-sizes[k], str(k))):
        w = min(range(nworkers), key=lambda t: (loads[t], t))
        parts[w].append(item)
        loads[w] += sizes[item]
    return parts


def dist_env():
    """(rank, world, local_rank) from the torchrun environment, (0, 1, 0) outside of it."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')),
            int(os.environ.get('LOCAL_RANK', '0')))


def run_sharded(sizes, score_fn, rank=0, world=1, group=None, batch_fn=None):
    """Score this rank's share of the chromosomes with `score_fn(chrom) -> table` (or, one chromosome ahead,
    `batch_fn(chroms) -> [table, ...]`) and gather {chrom: table} on rank 0 (None elsewhere).  `group` is a
    torch.distributed process group when world > 1 (backend nccl on the GPU box, gloo in the CPU tests); only Python
    objects travel."""
    mine = lpt_partition(sizes, world)[rank]
    if batch_fn is not None:
        local = dict(zip(mine, batch_fn(mine)))
    else:
        local = {c: score_fn(c) for c in mine}
    if world == 1:
        return local
    import torch.distributed as dist
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(local, gathered, dst=0, group=group)
    if rank != 0:
        return None
    out = {}
    for part in gathered:
        out.update(part)
    return out


# ----------------------------------------------------------------------------- one largest-first queue for all workers
def largest_first(sizes):
    """The queue's order: decreasing cost, ties by name (every worker derives the same list)."""
    return sorted(sizes, key=lambda k: (-sizes[k], str(k)))


class WorkQueue(object):
    """Chromosomes in largest-first order behind one shared counter: `take()` hands the next one to whichever worker asks,
    None when the queue is empty.  A worker whose chromosomes turn out slow (cost follows the candidates, not the bins)
    simply asks less often - nobody idles behind a static share.  `counter()` must return 0, 1, 2, ... atomically across
    the workers: `local_counter` (one process), `mp_counter` (multiprocessing workers of one node), `store_counter`
    (torch.distributed ranks)."""

    def __init__(self, sizes, counter):
        self.items = largest_first(sizes)
        self.counter = counter

    def take(self):
        i = self.counter()
        return self.items[i] if i < len(self.items) else None

    def __iter__(self):
        return iter(self.take, None)


def local_counter():
    c = itertools.count()
    return lambda: next(c)


def mp_counter(value):
    """value: multiprocessing.Value('i', 0) shared with the workers (hand it over through the Pool's initializer)."""
    def nxt():
        with value.get_lock():
            i = value.value
            value.value = i + 1
        return i
    return nxt


def store_counter(store=None, key='hpk_next_chromosome'):
    """A counter in the key-value store of the default torch.distributed process group (add is atomic on the store's
    server, rank 0's process): the only traffic the data path's queue needs, a few bytes per chromosome."""
    if store is None:
        import torch.distributed as dist
        store = dist.distributed_c10d._get_default_store()
    return lambda: int(store.add(key, 1)) - 1


def run_queue(queue, submit_fn, depth=2):
    """Drain `queue` from this worker: keep up to `depth` chromosomes in flight (`submit_fn(chrom)` returns an object with
    .result()), collect in submission order.  -> {chrom: table} of the chromosomes this worker took."""
    import collections
    pending, out = collections.deque(), {}
    for c in queue:
        pending.append((c, submit_fn(c)))
        if len(pending) >= depth:
            k, call = pending.popleft()
            out[k] = call.result()
    while pending:
        k, call = pending.popleft()
        out[k] = call.result()
    return out


def gather_tables(local, rank=0, world=1, group=None):
    """{chrom: table} of every rank on rank 0 (None elsewhere); only Python objects travel."""
    if world == 1:
        return local
    import torch.distributed as dist
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(local, gathered, dst=0, group=group)
    if rank != 0:
        return None
    out = {}
    for part in gathered:
        out.update(part)
    return out


def simulate(costs, nworkers, policy='queue', estimate=None, speeds=None):
    """Virtual-time makespan of a hand-out policy (tests): 'queue' = largest-first by `estimate`, next item to the
    first free worker; 'lpt' = static partition by `estimate`.  costs / estimate: {item: seconds}; speeds: per worker."""
    estimate = estimate or costs
    speeds = speeds or [1.0] * nworkers
    if policy == 'lpt':
        parts = lpt_partition(estimate, nworkers)
        return max(sum(costs[c] for c in p) / speeds[w] for w, p in enumerate(parts))
    free = [0.0] * nworkers
    for c in largest_first(estimate):
        w = min(range(nworkers), key=lambda t: (free[t], t))
        free[w] += costs[c] / speeds[w]
    return max(free)
