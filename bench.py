#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X refine stage on BASELINE.json's workload.

One "step" = one pass of the hot path over one batch of synthetic candidates
that are already resident in HBM: guide geometry -> banded affine fill
(AffineGuidedAlign) -> traceback + ComputeAlignmentStats -> op compaction,
through the C ABI (blasr_hip_refine_device).  Workload at N=1 = BASELINE.json
configs[1]: 100k synthetic 10 kb PacBio-like reads (ins 11 % / del 4 % / sub
1 %) against a 4.64 Mb i.i.d. genome (E. coli K12 length; the real FASTA is not
available offline), band 16, default scores.  With N>1 every rank processes its
own 100k-read shard (weak scaling, no data-path collective; one RCCL all-reduce
of the per-rank counters at the end, as SURVEY.md section 8e prescribes).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def cpu_baseline(w, n_sample, band):
    """The CPU oracle (a bit-exact port of the reference's AffineGuidedAlign +
    ComputeAlignmentStats) timed on one host core over the first `n_sample`
    candidates of the same workload.  Checker code: never on the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from blasr_testlib import load_oracle

    oracle = load_oracle()
    import numpy as np

    seq = w["seq"].cpu().numpy()
    q_off, q_len = w["q_off"][:n_sample].cpu().numpy(), w["q_len"][:n_sample].cpu().numpy()
    t_off, t_len = w["t_off"][:n_sample].cpu().numpy(), w["t_len"][:n_sample].cpu().numpy()
    blk_off, n_blk = w["blk_off"][:n_sample].cpu().numpy(), w["n_blocks"][:n_sample].cpu().numpy()
    hi = int(blk_off[-1] + n_blk[-1])
    blocks = w["blocks"][: 3 * hi].cpu().numpy().reshape(-1, 3)
    tasks = [(seq[q_off[i]: q_off[i] + q_len[i]], seq[t_off[i]: t_off[i] + t_len[i]],
              blocks[blk_off[i]: blk_off[i] + n_blk[i]]) for i in range(n_sample)]
    cells = 0
    t0 = time.perf_counter()
    for q, t, bl in tasks:
        cells += oracle.affine_guided_align(q, t, bl, band=band)["nCells"]
    dt = time.perf_counter() - t0
    bases = int(q_len.sum())
    return dict(value=bases / dt, unit="aligned bases/s", cores=1, kind="port",
                sample="first %d candidates of the same workload (%.1f s, %.1f Mcells/s on one core)"
                       % (n_sample, dt, cells / dt / 1e6))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--band", type=int, default=16)
    ap.add_argument("--cpu-sample", type=int, default=1200, help="candidates timed on the CPU oracle (0 = skip)")
    ap.add_argument("--seed", type=int, default=3)
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the refine stage has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    import blasr_amd
    from blasr_amd import synth

    ctx = blasr_amd.Context(local_rank)
    ctx.set_workspace_limit(160 << 30)
    w = synth.make_refine_workload(args.reads, args.read_len, args.seed + rank, dev)
    n = int(w["q_len"].numel())
    ptrs = {k: w[k].data_ptr() for k in ("seq", "q_off", "q_len", "t_off", "t_len", "blocks", "blk_off", "n_blocks")}
    results = torch.zeros(n * 56, dtype=torch.uint8, device=dev)
    ops_cap = int((w["q_len"].sum().item() + w["t_len"].sum().item()) // 2 + 1024)
    ops = torch.zeros(ops_cap, dtype=torch.int32, device=dev)
    score = blasr_amd.ScoreParams()
    torch.cuda.synchronize()

    def step():
        return ctx.refine_device(ptrs, n, args.band, score, results.data_ptr(), ops.data_ptr(), ops_cap)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    fill_ms = fill_bytes = cells = launches = 0
    stage_ms = [0.0, 0.0, 0.0, 0.0]
    for _ in range(args.steps):
        step()
        tm = ctx.last_timing()  # HIP events recorded on the library's own stream
        fill_ms += tm["fill_ms"]
        fill_bytes += tm["fill_bytes"]
        cells += tm["cells"]
        launches += tm["fill_launches"]
        for i, k in enumerate(("geometry_ms", "fill_ms", "traceback_ms", "compact_ms")):
            stage_ms[i] += tm[k]
    barrier()
    elapsed = time.perf_counter() - t0

    rec = results.view(torch.int32).view(n, 14)
    ok = rec[:, 0] == 0
    from blasr_amd import shard

    # the only collective of the path: one all-reduce of the per-rank counters (RCCL over xGMI)
    totals = shard.reduce_counters(n, int(ok.sum()), int((w["q_len"].to(torch.int64) * ok).sum()),
                                   int(rec[:, 4].to(torch.int64).sum()), device=dev)
    elapsed = shard.max_over_ranks(elapsed, device=dev)
    reads_total, aligned_reads = totals["reads"], totals["aligned_reads"]
    aligned_bases, cells_total = totals["aligned_bases"], totals["dp_cells"]

    if rank == 0:
        value = aligned_bases * args.steps / elapsed
        achieved = fill_bytes / (fill_ms / 1e3) / 1e9 if fill_ms > 0 else 0.0
        out = {
            "metric": "aligned bases/sec (refine stage: guided banded affine DP + traceback + stats)",
            "value": value,
            "unit": "bases/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic: i.i.d. %d-base genome, %d x %d-base reads per GPU (ins 11%%/del 4%%/sub 1%%), "
                    "guide blocks = ungapped runs of the simulated alignment" % (synth.ECOLI_K12_LENGTH, args.reads,
                                                                                  args.read_len),
            "config": {"workload": "configs[1]: 100k synthetic 10 kb reads vs E. coli-K12-length genome, "
                                   "refine stage, band 16, SMRT matrix, ins=del=5, open=50, ext=0",
                       "reads_per_gpu": args.reads, "read_len": args.read_len, "band": args.band,
                       "candidates": reads_total, "aligned_reads": aligned_reads},
            "gcups_per_gpu": cells_total / world * args.steps / elapsed / 1e9,
            "stage_ms_per_step": {k: v / args.steps for k, v in zip(("geometry", "fill", "traceback", "compact"),
                                                                     stage_ms)},
            "roofline": {"bound": "hbm", "kernel": "blasr_hip::fill_kernel<false, true>", "achieved": achieved,
                         "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                         "traffic_note": "PMC passes (profiles/r01_pmc_counters_20k_reads.json): FETCH_SIZE x2 + "
                                         "WRITE_SIZE = 1.11 x the algorithmic bytes of the same launch",
                         "algorithmic_bytes_per_launch": fill_bytes / max(launches, 1),
                         "launches": launches, "avg_launch_ms": fill_ms / max(launches, 1),
                         "fill_gcups": cells / (fill_ms / 1e3) / 1e9 if fill_ms > 0 else 0.0},
        }
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(w, min(args.cpu_sample, n), args.band)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
