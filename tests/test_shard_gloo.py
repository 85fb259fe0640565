"""The N>1 path on CPU: two gloo ranks shard a read list the way bench.py does
and combine their counters with the single all-reduce of the path."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from blasr_amd import shard


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_reads, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lengths = torch.arange(n_reads, dtype=torch.int64) * 7 + 100  # read i has 100 + 7 i bases
    begin, end = shard.block_shard(n_reads, rank, world)
    mine = lengths[begin:end]
    strided = lengths[list(shard.stride_shard(n_reads, rank, world))]
    aligned = mine[mine % 3 != 0]  # pretend reads with length % 3 == 0 found no alignment
    total = shard.reduce_counters(len(mine), len(aligned), int(aligned.sum()), int(mine.sum()) * 34)
    t = shard.max_over_ranks(1.0 + rank)
    stride_total = shard.reduce_counters(len(strided), 0, int(strided.sum()), 0)
    if rank == 0:
        out.put((total, t, stride_total))
    dist.destroy_process_group()


def test_two_rank_shard_and_count_reduce():
    n_reads, world = 1001, 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_reads, out)) for r in range(world)]
    for p in procs:
        p.start()
    total, t, stride_total = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lengths = torch.arange(n_reads, dtype=torch.int64) * 7 + 100
    aligned = lengths[lengths % 3 != 0]
    assert total == dict(reads=n_reads, aligned_reads=len(aligned), aligned_bases=int(aligned.sum()),
                         dp_cells=int(lengths.sum()) * 34)
    assert t == 2.0  # the slowest rank
    assert stride_total["reads"] == n_reads and stride_total["aligned_bases"] == int(lengths.sum())


def test_shards_partition_the_input():
    for n in (0, 1, 7, 100, 1001):
        for world in (1, 2, 3, 8):
            blocks = [shard.block_shard(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in blocks]
            assert max(sizes) - min(sizes) <= 1
            strided = sorted(i for r in range(world) for i in shard.stride_shard(n, r, world))
            assert strided == list(range(n))
