"""Multi-GPU plumbing of the refine stage: reads shard embarrassingly.

The reference's only coarse partitioning is ``-start S -stride N``
(alignment/Blasr.cpp:4057-4058): process i takes reads S, S+N, S+2N, ...
Candidates of one read never interact with another read's, so the path has no
exchange step: every rank refines its own shard and the only collective is one
all-reduce (sum) of four int64 counters at the end -- RCCL over xGMI on the
GPU box (backend "nccl"), gloo in the CPU tests.
"""
import torch

COUNTERS = ("reads", "aligned_reads", "aligned_bases", "dp_cells")


def stride_shard(n_items, rank, world):
    """Indices rank, rank+world, ... < n_items  (the -start/-stride split)."""
    return range(rank, n_items, world)


def block_shard(n_items, rank, world):
    """Contiguous [begin, end) split, sizes differing by at most one."""
    base, extra = divmod(n_items, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def reduce_counters(reads, aligned_reads, aligned_bases, dp_cells, device="cpu", group=None):
    """Sum the per-rank counters over all ranks; returns a dict of python ints.
    Without an initialised process group this is the identity (single GPU)."""
    t = torch.tensor([int(reads), int(aligned_reads), int(aligned_bases), int(dp_cells)], dtype=torch.int64,
                     device=device)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=group)
    return dict(zip(COUNTERS, (int(v) for v in t.tolist())))


def max_over_ranks(seconds, device="cpu", group=None):
    """Wall time of the slowest rank (what bench.py divides by)."""
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=group)
    return float(t.item())


def per_rank(values, device="cpu", group=None):
    """Every rank's list of floats, gathered on all ranks: a list (one entry per rank) of lists.  bench.py prints the ranks'
    own step and stage times with it, so that a scaling line explains itself (a slow rank, a slow stage of one rank)."""
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        out = [torch.zeros_like(t) for _ in range(torch.distributed.get_world_size(group))]
        torch.distributed.all_gather(out, t, group=group)
        return [[float(x) for x in o.tolist()] for o in out]
    return [[float(x) for x in t.tolist()]]
