#!/usr/bin/env python3
"""next-1 measured: blasr_hip_anchor_reads_device on reads sampled from a synthetic genome whose suffix array
blasr_hip_build_suffix_array builds on the device (--sa sawriter: the reference's own tool instead), with the
reference's MapReadToGenome (oracle/_ref/libblasr_refmapper.so, one host thread) timed beside it on a bounded sample of
the same reads and every anchor of that sample compared.  One JSON line.

    python tools/anchor_bench.py [--genome-mb 50] [--reads 2000] [--read-len 10000] [--steps 3]

The parameters are what Blasr.cpp ends up with for a default sawriter index: lookup prefix 13, -minMatch 14
(MappingParameters.h:244), minPrefixMatchLength = params.lookupTableLength = 8, stopMappingOnceUnique,
maxAnchorsPerPosition 1000, expand 0."""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mb", type=float, default=50.0)
    ap.add_argument("--reads", type=int, default=2000)
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--cpu-reads", type=int, default=40)
    ap.add_argument("--blt", type=int, default=0)
    ap.add_argument("--families", type=int, default=200)
    ap.add_argument("--sa", choices=("device", "sawriter"), default="device")
    args = ap.parse_args()
    import blasr_amd
    from blasr_amd import hip, synth
    import blasr_testlib as T

    rng = np.random.default_rng(5)
    glen = int(args.genome_mb * 1e6)
    genome = synth.random_genome(glen, 17)
    for f in range(args.families):  # repeat families: 2..6 copies of a 3-8 kb unit at 1-6 % divergence
        unit = int(rng.integers(3000, 8000))
        src = int(rng.integers(0, glen - unit))
        for _ in range(int(rng.integers(1, 6))):
            d = float(rng.uniform(0.01, 0.06))
            piece = np.ascontiguousarray(synth.mutate(genome[src:src + unit], rng, d / 3, d / 3, d / 3))
            at = int(rng.integers(0, glen - len(piece)))
            genome[at:at + len(piece)] = piece
    reads = np.zeros(args.reads, dtype=hip.READ_DTYPE)
    pool, off = [], 0
    for r in range(args.reads):
        pos = int(rng.integers(0, glen - args.read_len))
        tmpl = genome[pos:pos + args.read_len]
        if r % 2:
            tmpl = synth.revcomp(tmpl)
        rd = np.ascontiguousarray(synth.mutate(tmpl, rng, 0.11, 0.04, 0.01))
        reads[r] = (off, len(rd), 0, len(rd), 0)
        pool.append(rd)
        off += len(rd)
    pool = np.concatenate(pool)
    tmp = tempfile.mkdtemp(prefix="anchor_bench_")
    ctx = blasr_amd.Context(0)
    if args.sa == "sawriter":
        t0 = time.time()
        fasta = b">g\n" + bytes(genome) + b"\n"
        G, starts, sa, sa_path = T.write_suffix_array(tmp, fasta, args.blt or None)
        t_sa = time.time() - t0
        ctx.upload_genome(G)
        ctx.upload_suffix_array(sa)
    else:
        G = np.concatenate([genome, np.array([ord("N")], np.uint8)])  # one contig and the N after it
        starts = np.array([0, len(G)], np.uint32)
        ctx.upload_genome(G)
        t0 = time.time()
        sa = ctx.build_suffix_array(args.blt or 13)  # sawriter's default prefix
        t_sa = time.time() - t0
    del genome
    prefix = int(sa["lookup_prefix_length"])
    min_match = max(14, prefix)

    p = ctx.anchor_params(min_match_length=min_match, min_prefix_match_length=8)
    mem = T.DeviceMemory()
    d_pool = mem.upload(pool)
    cap = 8 * int(reads["length"].astype(np.int64).sum())
    d_out = mem.zeros(cap * hip.ANCHOR_DTYPE.itemsize)
    os.environ["BLASR_HIP_ANCHOR_COUNT_PROBES"] = "1"
    n, loff = ctx.anchor_reads_device(reads, d_pool, p, d_out, cap)  # warm-up, and the probe count
    probes = ctx.anchor_last_timing()["probes"]
    del os.environ["BLASR_HIP_ANCHOR_COUNT_PROBES"]
    ms = []
    for _ in range(args.steps):
        t0 = time.time()
        n, loff = ctx.anchor_reads_device(reads, d_pool, p, d_out, cap)
        wall = (time.time() - t0) * 1e3
        t = ctx.anchor_last_timing()
        ms.append((wall, t["search_ms"], t["emit_ms"], t["sort_ms"]))
    best = min(ms)
    anchors = mem.download(d_out, hip.ANCHOR_DTYPE, n)

    # the reference itself on the first reads, one thread
    cpu = None
    if os.path.exists(T.REFMAPPER_SO):
        ref = C.CDLL(T.REFMAPPER_SO)
        ref.ref_sa_adopt.restype = C.c_void_p
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        h = ref.ref_sa_adopt(vp(sa["index"]), len(sa["index"]), vp(sa["start_pos_table"]), vp(sa["end_pos_table"]),
                             len(sa["start_pos_table"]), prefix)
        aps = (min_match, 0, 1, 1000, 0, 0, 1)
        k = min(args.cpu_reads, args.reads)
        differ, t_cpu, n_cpu = 0, 0.0, 0
        for r in range(k):
            o, L = int(reads[r]["off"]), int(reads[r]["length"])
            rd = np.ascontiguousarray(pool[o:o + L])
            for strand in (0, 1):
                t0 = time.time()
                w = T.ref_anchors(ref, h, G, rd, 0, L, strand, 8, aps, True)
                t_cpu += time.time() - t0
                mine = anchors[int(loff[2 * r + strand]): int(loff[2 * r + strand + 1])]
                got = np.stack([mine["t"], mine["q"], mine["l"]], axis=1) if len(mine) else np.zeros((0, 3), np.uint32)
                differ += not np.array_equal(got, w)
                n_cpu += len(w)
        bases = int(reads["length"][:k].astype(np.int64).sum())
        cpu = dict(kind="reference", cores=1, sample="%d reads, both strands, MapReadToGenome + SortMatchPosList" % k,
                   seconds=round(t_cpu, 3), bases_per_s=round(bases / t_cpu), anchors=n_cpu, lists_that_differ=differ)
    # the host half: anchors -> candidate intervals (RemoveOverlappingAnchors, FindMaxIncreasingInterval, interval set)
    starts = np.asarray(starts, dtype=np.uint32)
    t0 = time.time()
    count_table = hip.count_tuples(G, 8)
    t_ct = time.time() - t0
    ip = hip.interval_params()
    t0 = time.time()
    ivs, ioff, pvs = hip.find_intervals(G, starts, count_table, reads, anchors, loff, ip)
    t_iv = time.time() - t0
    host = dict(intervals=int(len(ivs)), find_intervals_ms=round(t_iv * 1e3, 1), threads="up to 16",
                count_table_s=round(t_ct, 2), reads_per_s=round(args.reads / t_iv))
    if cpu is not None:
        ref.ref_ct_open.restype = C.c_void_p
        hct = ref.ref_ct_open(G.ctypes.data_as(C.c_void_p), len(G), 8)
        k = min(args.cpu_reads, args.reads)
        t_ref, differ = 0.0, 0
        zero = np.zeros((2, 3), np.uint32)
        for r in range(k):
            L = int(reads[r]["length"])
            lists = []
            for strand in (0, 1):
                mine = anchors[int(loff[2 * r + strand]): int(loff[2 * r + strand + 1])]
                lists.append(np.stack([mine["t"], mine["q"], mine["l"]], axis=1) if len(mine) else np.zeros((0, 3), np.uint32))
            t0 = time.time()
            want, wpv = T.ref_find_intervals(ref, hct, G, starts, L, 0, L, lists[0], lists[1], zero)
            t_ref += time.time() - t0
            m = ivs[int(ioff[r]): int(ioff[r + 1])]
            got = (np.stack([m["start"], m["end"], m["q_start"], m["q_end"], m["strand"].astype(np.uint32)], axis=1)
                   if len(m) else np.zeros((0, 5), np.uint32))
            differ += not (np.array_equal(got, want) and np.array_equal(pvs[int(ioff[r]): int(ioff[r + 1])], wpv))
        ref.ref_ct_close(C.c_void_p(hct))
        host["reference"] = dict(kind="reference", cores=1, sample="%d reads" % k, seconds=round(t_ref, 3),
                                 reads_per_s=round(k / t_ref), reads_that_differ=differ)
    t = ctx.anchor_last_timing()
    total_bases = int(reads["length"].astype(np.int64).sum())
    # algorithmic bytes of the search: every probe reads one index entry (4 B) and one genome byte, in separate sectors
    out = dict(stage="anchor (MapReadToGenome x 2 strands + SortMatchPosList)", genome_bases=len(G), lookup_prefix=prefix,
               min_match=min_match, reads=args.reads, read_len=args.read_len, positions=t["positions"], anchors=int(n),
               pieces=t["pieces"], probes=int(probes), probes_per_position=round(probes / max(t["positions"], 1), 1),
               wall_ms=round(best[0], 2), search_ms=round(best[1], 2), emit_ms=round(best[2], 2), sort_ms=round(best[3], 2),
               bases_per_s=round(total_bases / (best[0] / 1e3)),
               probes_per_s=round(probes / (best[1] / 1e3)) if best[1] > 0 else None,
               sector_GBps=round(probes * 64 / (best[1] / 1e3) / 1e9, 1) if best[1] > 0 else None,
               sa_build=args.sa, sa_build_s=round(t_sa, 1), cpu_baseline=cpu, host_half=host)
    if cpu:
        out["vs_cpu_thread"] = round(out["bases_per_s"] / cpu["bases_per_s"], 1)
    print(json.dumps(out))
    if cpu is not None:
        ref.ref_sa_close(C.c_void_p(h))
    mem.free()
    ctx.close()
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
