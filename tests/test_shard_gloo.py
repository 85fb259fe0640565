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
    ranks = shard.per_rank([10.0 + rank, 2.5 * rank])
    assert ranks == [[10.0, 0.0], [11.0, 2.5]], ranks  # every rank sees every rank's figures
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


def _run_bench(args, env_extra=None, timeout=300):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout)
    lines = [x for x in p.stdout.decode().splitlines() if x.startswith("{")]
    return p.returncode, ([json.loads(x) for x in lines]), p.stderr.decode()


def test_bench_launches_its_own_ranks():
    """`bench.py --gpus 2` started WITHOUT a launcher (the way DESIGN.md documents the scaling run) spawns its two
    ranks itself, rendezvous on 127.0.0.1, and rank 0 reports n_gpus = 2 with the counters of both shards: the real
    argument / launch / reduce code of bench.py, the device step stubbed out (--stub-device, gloo)."""
    rc, out, err = _run_bench(["--gpus", "2", "--stub-device", "--reads", "123", "--steps", "2", "--warmup", "0"])
    assert rc == 0, err[-2000:]
    assert len(out) == 1, out  # exactly one JSON line, from rank 0
    d = out[0]
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["config"]["reads"] == 2 * 123 and d["seeds"] == [3, 4]  # every rank its own shard
    assert "123 synthetic 10 kb reads per GPU" in d["config"]["workload"]  # the label follows the arguments


def test_bench_refuses_a_mismatched_launch():
    """Under a launcher that started a different number of ranks than --gpus says, bench.py stops instead of
    reporting one run as another."""
    rc, out, err = _run_bench(["--gpus", "4", "--stub-device"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and not out and "WORLD_SIZE=1" in err


def test_bench_workload_label_follows_the_arguments():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for argv, must in ((["--config", "4"], ["configs[4]", "6250 synthetic 30 kb", "ins 10%/del 5%", "8.00 chance clusters", "long:"]),
                       (["--config", "1", "--reads", "777", "--band", "30"], ["configs[1]", "777 synthetic 10 kb", "band 30", "0 planted"]),
                       ([], ["configs[2]", "100000 synthetic 10 kb", "4000 planted repeat families", "0.25 chance clusters"])):
        args = bench.parse_args(argv)
        d = bench.config_defaults(args)
        label = bench.workload_label(args, d, 1234, 1)
        for m in must:
            assert m in label, (m, label)
