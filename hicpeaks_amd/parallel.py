"""Chromosome-level sharding.  Every statistic of the path is per chromosome (candidates, frozen_w, lambda chunks,
BH, gaps, clustering), so the GPUs of a node never exchange data: work items are whole chromosomes, handed out
largest first (LPT), and the only communication is gathering the per-chromosome tables on rank 0.

Counterpart of `Pool(args.nproc).map(worker, Params)` in scripts/pyHICCUPS:192-198.
"""
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
