#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X refine path on BASELINE.json's workload.

One "step" = one pass of the hot path over one batch of synthetic candidates already resident in
HBM, through the C ABI of include/blasr_hip.h:

  blasr_hip_sdp_device     SDPAlign (k-mer seeding, chaining, SWAlign / recursive gap fills) +
                           ComputeAlignmentStats                 (alignment/Blasr.cpp:1716-1724)
  (device glue)            trim query/target to the SDP extent   (alignment/Blasr.cpp:850-859)
  blasr_hip_refine_device  AffineGuidedAlign + traceback + ComputeAlignmentStats (:862-878)

Workload at N=1 = BASELINE.json configs[1]: 100k synthetic 10 kb PacBio-like reads (ins 11 % /
del 4 % / sub 1 %) against a 4.64 Mb i.i.d. genome (E. coli K12 length; the real FASTA is not
available offline), -sdpTupleSize 11, band 16, default scores; one candidate window per read, built
by the window rule of AlignIntervals.  With N>1 every rank processes its own shard of the same size (weak
scaling, no data-path collective; one RCCL all-reduce of the run counters at the end).

--contexts C maps the batch with C host threads at once, each through its own context (the C ABI's "one context
per host thread", the reference's -nproc).  Kernels of two independent streams fill the tails and the
latency-bound phases of each other's: 247 ms per pass with two threads against 271 ms with one.
The default is ONE, so that every kernel has the GPU to itself and the roofline block means what it says; with
two the same launches take 36-58 ms depending on what they happen to share the GPU with.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


# bytes per fill launch measured with rocprofv3 --pmc for (reads per GPU, read length, band): see profiles/
PMC_FILL4_TRAFFIC = {(100000, 10000, 16, 1): 56375466304.0}
# wavefront VALU instructions of the same launch (SQ_INSTS_VALU, same profile)
PMC_FILL4_VALU_INSTS = {(100000, 10000, 16, 1): 31494305558.0}


def reference_candidate(ref, q, t, band):
    """One candidate through the REFERENCE's own functions (oracle/_ref/libblasr_ref.so: SDPAlign,
    ComputeAlignmentStats, AffineGuidedAlign compiled from the reference sources), in the order of
    AlignIntervals (Blasr.cpp:1716-1724) and RefineAlignment (:850-914); the same fields as the oracle's
    refine_candidate."""
    sdp = ref.sdp_align(q, t)
    blocks = sdp["blocks"]
    st = ref.compute_alignment_stats(q, t, sdp["qPos"], sdp["tPos"], blocks)
    out = dict(sdpScore=sdp["score"], sdpStatsScore=st["score"], sdpQPos=sdp["qPos"], sdpTPos=sdp["tPos"],
               nSdpBlocks=len(blocks), dpScore=0, nCells=0, score=0, nMatch=0, nMismatch=0, nIns=0, nDel=0,
               qPos=sdp["qPos"], tPos=sdp["tPos"])
    if len(blocks):
        last = blocks[-1]
        qs = q[sdp["qPos"]: sdp["qPos"] + int(last[0] + last[2])]
        ts = t[sdp["tPos"]: sdp["tPos"] + int(last[1] + last[2])]
        r = ref.affine_guided_align(qs, ts, blocks, band=band)
        out.update({k: r[k] for k in ("dpScore", "nCells", "score", "nMatch", "nMismatch", "nIns", "nDel")})
        out["qPos"], out["tPos"] = sdp["qPos"] + r["qPos"], sdp["tPos"] + r["tPos"]
    return out


def cpu_baseline(w, n_sample, band, sdp_rec, ref_rec, cores=1):
    """The path on one host core over the first `n_sample` candidates of the same workload: through the
    reference's own code where its build is at hand (oracle/_ref/libblasr_ref.so, kind "reference"), else through
    the oracle (its bit-exact restatement, kind "port").  Checker code: never on the product path.  Since the CPU
    answers are at hand, the device's result records of the same candidates (from the last timed step) are
    compared with them field by field."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from blasr_testlib import load_oracle, load_ref

    ref = load_ref()
    oracle = None if ref is not None else load_oracle()
    seq = w["seq"].cpu().numpy()
    q_off, q_len = w["q_off"][:n_sample].cpu().numpy(), w["q_len"][:n_sample].cpu().numpy()
    t_off, t_len = w["t_off"][:n_sample].cpu().numpy(), w["t_len"][:n_sample].cpu().numpy()
    tasks = [(seq[q_off[i]: q_off[i] + q_len[i]], seq[t_off[i]: t_off[i] + t_len[i]]) for i in range(n_sample)]
    one = (lambda qt: reference_candidate(ref, qt[0], qt[1], band)) if ref is not None else \
          (lambda qt: oracle.refine_candidate(qt[0], qt[1], band=band))
    t0 = time.perf_counter()
    if cores <= 1:
        res = [one(qt) for qt in tasks]
    else:
        # the reference's -nproc: its functions are re-entrant (one MappingBuffers per call here), and ctypes
        # releases the interpreter lock for the duration of a call, so host threads do run side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=cores) as pool:
            res = list(pool.map(one, tasks, chunksize=8))
    dt = time.perf_counter() - t0
    bases = int(q_len.sum())
    mismatches = 0
    for i, o in enumerate(res):
        s, r = sdp_rec[i], ref_rec[i]
        dev = (int(s[1]), int(s[5]), int(s[2]), int(s[3]), int(s[4]),           # SDP: score, stats score, qPos, tPos, blocks
               int(r[3]), int(r[4]), int(r[5]), int(r[6]), int(r[7]), int(r[8]), int(r[9]),  # refine: DP score .. nDel
               int(s[2]) + int(r[1]), int(s[3]) + int(r[2]))                   # window-relative qPos, tPos
        cpu = (o["sdpScore"], o["sdpStatsScore"], o["sdpQPos"], o["sdpTPos"], o["nSdpBlocks"], o["dpScore"],
               o["nCells"], o["score"], o["nMatch"], o["nMismatch"], o["nIns"], o["nDel"], o["qPos"], o["tPos"])
        mismatches += dev != cpu or int(s[0]) != 0 or int(r[0]) != 0
    what = ("the reference's own SDPAlign + AffineGuidedAlign + ComputeAlignmentStats (oracle/_ref/libblasr_ref.so)"
            if ref is not None else "the oracle's restatement of SDP + refine")
    return dict(value=bases / dt, unit="bases/s", cores=max(cores, 1), kind="reference" if ref is not None else "port",
                sample="first %d candidates of the same workload through %s (%.1f s on %d host thread%s, %.1f ms/read per thread)"
                       % (n_sample, what, dt, max(cores, 1), "" if cores <= 1 else "s", dt * max(cores, 1) / n_sample * 1e3),
                parity_checked=len(res), parity_mismatches=mismatches)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--band", type=int, default=16)
    ap.add_argument("--contexts", type=int, default=1,
                    help="host threads (each with its own context) that map the batch at the same time")
    ap.add_argument("--cpu-sample", type=int, default=4800, help="candidates timed on the host cores (0 = skip)")
    ap.add_argument("--cpu-cores", type=int, default=0, help="host threads of the CPU baseline (0 = min(8, cores))")
    ap.add_argument("--seed", type=int, default=3)
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the refine path has no CPU implementation")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    import threading

    import blasr_amd
    from blasr_amd import shard, synth

    class Mapper:
        """One host thread's share of the batch: its own context (own HIP streams and workspace), as the
        reference's -nproc threads each own their MappingBuffers (Blasr.cpp:318-369, :4811-4842)."""

        def __init__(self, reads, seed):
            self.ctx = blasr_amd.Context(local_rank)
            self.ctx.set_workspace_limit((160 << 30) // args.contexts)
            self.w = w = synth.make_sdp_workload(reads, args.read_len, seed, dev)
            self.n = n = int(w["q_len"].numel())
            self.sdp_ptrs = {k: w[k].data_ptr() for k in ("seq", "q_off", "q_len", "t_off", "t_len")}
            self.sdp_res = torch.zeros(n * 48, dtype=torch.uint8, device=dev)
            self.blk_cap = int(torch.minimum(w["q_len"], w["t_len"]).sum().item() // 4 + n)
            self.blocks = torch.zeros(3 * self.blk_cap, dtype=torch.int32, device=dev)
            self.ref_res = torch.zeros(n * 56, dtype=torch.uint8, device=dev)
            self.ops_cap = int((w["q_len"].sum().item() + w["t_len"].sum().item()) // 2 + 1024)
            self.ops = torch.zeros(self.ops_cap, dtype=torch.int32, device=dev)
            # the glue between the two calls runs on torch's stream: its own one per thread when several threads map
            # at once (the default stream would serialise them)
            self.glue = torch.cuda.Stream(device=dev) if args.contexts > 1 else torch.cuda.current_stream()
            self.acc = {}

        def step(self):
            n, w = self.n, self.w
            # a-2..a-5 + a-9: SDPAlign and its statistics
            self.ctx.sdp_device(self.sdp_ptrs, n, sdp_par, score, self.sdp_res.data_ptr(), self.blocks.data_ptr(),
                                self.blk_cap)
            t_sdp = self.ctx.sdp_last_timing()
            with torch.cuda.stream(self.glue):
                # RefineAlignment's trimming (Blasr.cpp:850-859): pure index arithmetic on the result records
                r = self.sdp_res.view(torch.int32).view(n, 12)
                q_pos, t_pos, n_blk = r[:, 2].to(torch.int64), r[:, 3].to(torch.int64), r[:, 4].contiguous()
                blk_off = self.sdp_res.view(torch.int64).view(n, 6)[:, 5].contiguous()
                last = (blk_off + n_blk - 1).clamp_min(0)
                b3 = self.blocks.view(-1, 3)
                has = n_blk > 0
                q_len2 = torch.where(has, b3[last, 0] + b3[last, 2], torch.zeros_like(n_blk)).contiguous()
                t_len2 = torch.where(has, b3[last, 1] + b3[last, 2], torch.zeros_like(n_blk)).contiguous()
                q_off2 = (w["q_off"] + q_pos).contiguous()
                t_off2 = (w["t_off"] + t_pos).contiguous()
                ptrs = dict(seq=w["seq"].data_ptr(), q_off=q_off2.data_ptr(), q_len=q_len2.data_ptr(),
                            t_off=t_off2.data_ptr(), t_len=t_len2.data_ptr(), blocks=self.blocks.data_ptr(),
                            blk_off=blk_off.data_ptr(), n_blocks=n_blk.data_ptr())
                self.glue.synchronize()  # the glue ran on a torch stream, the library uses its own
            # a-6..a-9: guided banded affine DP, traceback, statistics
            self.ctx.refine_device(ptrs, n, args.band, score, self.ref_res.data_ptr(), self.ops.data_ptr(),
                                   self.ops_cap)
            t_ref = self.ctx.last_timing()
            f4 = self.ctx.last_fill4()
            t_ref.update(fill4_ms=f4["ms"], fill4_launches=f4["launches"], fill4_bytes=f4["bytes"],
                         fill4_cells=f4["cells"])
            for k, v in list(t_sdp.items()) + list(t_ref.items()):
                self.acc[k] = self.acc.get(k, 0) + v

    score, sdp_par = blasr_amd.ScoreParams(), blasr_amd.SdpParams()
    C = max(1, args.contexts)
    share = [args.reads * (i + 1) // C - args.reads * i // C for i in range(C)]
    mappers = [Mapper(share[i], args.seed + rank * C + i) for i in range(C)]
    torch.cuda.synchronize()

    def run_steps(k):
        # the batch is mapped by C host threads at once, each through its own context: kernels of independent
        # streams fill the tails and the latency-bound phases of each other's
        def body(m):
            for _ in range(k):
                m.step()
        if C == 1:
            body(mappers[0])
            return
        ths = [threading.Thread(target=body, args=(m,)) for m in mappers]
        for t in ths:
            t.start()
        for t in ths:
            t.join()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    run_steps(args.warmup)
    for m in mappers:
        m.acc = {}
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    acc = {}
    for m in mappers:
        for k, v in m.acc.items():
            acc[k] = acc.get(k, 0) + v

    n = sum(m.n for m in mappers)
    ref_res = torch.cat([m.ref_res for m in mappers])
    q_len_all = torch.cat([m.w["q_len"] for m in mappers])
    rec = ref_res.view(torch.int32).view(n, 14)
    ok = rec[:, 0] == 0
    # the only collective of the path: one all-reduce of the per-rank counters (RCCL over xGMI)
    totals = shard.reduce_counters(n, int(ok.sum()), int((q_len_all.to(torch.int64) * ok).sum()),
                                   int(rec[:, 4].to(torch.int64).sum()), device=dev)
    elapsed = shard.max_over_ranks(elapsed, device=dev)

    if rank == 0:
        K = args.steps
        value = totals["aligned_bases"] * K / elapsed
        fill_ms = acc["fill_ms"]
        # the dominant kernel: fillp_kernel<16, true> (four candidates per wavefront), timed on its own with HIP events on
        # the library's stream; its algorithmic bytes = 1 B per band cell + the bases + 8 B per row record of
        # the candidates it processed
        f4_ms, f4_launches = acc["fill4_ms"], max(acc["fill4_launches"], 1)
        achieved = acc["fill4_bytes"] / (f4_ms / 1e3) / 1e9 if f4_ms > 0 else 0.0
        out = {
            "metric": "aligned bases/sec (refine path: SDP chaining + guided banded affine DP), GCUPS per GPU",
            "value": value,
            "unit": "bases/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic: i.i.d. %d-base genome, %d x %d-base reads per GPU (ins 11%%/del 4%%/sub 1%%), one "
                    "candidate window per read (AlignIntervals window rule)" % (synth.ECOLI_K12_LENGTH, args.reads,
                                                                                args.read_len),
            "config": {"workload": "configs[1]: 100k synthetic 10 kb reads vs E. coli-K12-length genome, "
                                   "-sdpTupleSize 11, band 16, SMRT matrix, ins=del=5, open=50, ext=0; SDP + refine",
                       "reads_per_gpu": args.reads, "read_len": args.read_len, "band": args.band,
                       "host_threads_per_gpu": C,
                       "candidates": totals["reads"], "aligned_reads": totals["aligned_reads"]},
            "gcups_per_gpu": totals["dp_cells"] / world * K / elapsed / 1e9,
            # HIP-event times of the stages summed over the contexts; with more than one they overlap in time
            "stage_ms_per_step": {
                "sdp_keys_sort": acc["keys_ms"] / K, "sdp_match_order": acc["match_ms"] / K,
                "sdp_chain": acc["chain_ms"] / K, "sdp_blocks_fills_assembly": acc["blocks_ms"] / K,
                "refine_geometry": acc["geometry_ms"] / K, "refine_fill": fill_ms / K,
                "refine_traceback_stats": acc["traceback_ms"] / K, "refine_compact": acc["compact_ms"] / K},
            "sdp_counters_per_step": {"sdpalign_invocations": acc["invocations"] // K,
                                      "fragments": acc["fragments"] // K, "swalign_fills": acc["sw_fills"] // K},
            "roofline": {"bound": "hbm", "kernel": "blasr_hip::fillp_kernel<16, true>", "achieved": achieved,
                         "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         # HBM bytes of one launch from the PMC passes of this same workload (separate rocprofv3
                         # --pmc runs, FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md); not measurable in-process
                         "traffic": PMC_FILL4_TRAFFIC.get((args.reads, args.read_len, args.band, C)),
                         "traffic_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, "
                                         "profiles/r01_v24_pmc_counters_100k_reads.json: 2 x FETCH_SIZE + WRITE_SIZE "
                                         "of the fillp_kernel<16, true> launch",
                         "algorithmic_bytes_per_launch": acc["fill4_bytes"] / f4_launches,
                         "launches": f4_launches, "avg_launch_ms": f4_ms / f4_launches,
                         "kernel_gcups": acc["fill4_cells"] / (f4_ms / 1e3) / 1e9 if f4_ms > 0 else 0.0,
                         "candidates_note": "candidates whose rows all fit 112 cells (82 % here); the others take "
                                            "fill_kernel, one per wavefront",
                         "fill_stage_gcups": acc["cells"] / (fill_ms / 1e3) / 1e9 if fill_ms > 0 else 0.0},
        }
        valu = PMC_FILL4_VALU_INSTS.get((args.reads, args.read_len, args.band, C))
        if valu and f4_ms > 0:
            # what actually bounds the kernel (integer DP: no MFMA, far from the HBM roof): VALU issue slots.
            # A wave64 VALU instruction holds its SIMD for 4 cycles (tools/valu_rate.hip); 256 CUs x 4 SIMDs.
            out["roofline"]["valu_issue"] = {
                "wave_insts_per_launch": valu, "cycles_per_wave_inst": 4, "simds": 1024, "clock_ghz_max": 2.4,
                "frac_of_issue_slots_at_max_clock": valu * 4 / (1024 * 2.4e9 * (f4_ms / f4_launches) / 1e3),
                "note": "SQ_INSTS_VALU of the launch (profiles/r01_v24_pmc_counters_100k_reads.json) x 4 cycles over "
                        "the SIMD cycles of the measured launch time at the 2.4 GHz maximum clock"}
        if world == 1 and args.cpu_sample > 0:
            m0 = mappers[0]
            ns = min(args.cpu_sample, m0.n)
            cores = args.cpu_cores if args.cpu_cores > 0 else min(8, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(m0.w, ns, args.band,
                                               m0.sdp_res.view(torch.int32).view(m0.n, 12)[:ns].cpu().numpy(),
                                               rec[:ns].cpu().numpy(), cores)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
