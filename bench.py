#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X refine path on BASELINE.json's workload.

One "step" = one pass of the WHOLE hot path over one block of synthetic reads whose inputs (genome, reads,
candidate intervals) are resident in HBM: ONE call of the library's C++ MapRead driver through the C ABI of
include/blasr_hip.h (blasr_hip_map_reads_device, blasr_amd/csrc/mapread_api.hip), in the reference's order:

  a-1   windows of AlignIntervals from the resident genome           alignment/Blasr.cpp:1208-1336
  a-2.. SDPAlign + ComputeAlignmentStats of every candidate interval  :1716-1724
  a-10  sort by score, RemoveLowQualityAlignments (host, C++)         :2547, :2594-2596
  trim  RefineAlignment's trimming, as a kernel                       :850-859
  a-6.. AffineGuidedAlign + traceback + ComputeAlignmentStats of the survivors  :862-878
  a-10  sort, RemoveLowQualityAlignments, RemoveOverlappingAlignments (host, C++)  :2178-2180, :2615-2616

No torch kernel runs inside a step (torch only generated the inputs and carries the counters' all-reduce).

Workloads (--config, BASELINE.json configs[]):
  2 (default)  N reads of 10 kb (ins 11 % / del 4 % / sub 1 %) against an hg19-SIZED synthetic genome (24 contigs of
               the hg19 chromosome lengths, 3.1 Gbases resident; the real FASTA is not available offline) with planted
               repeat families; per read 1-10 candidate intervals: its true locus, the matching stretch of every
               other copy of a repeat family it overlaps, and chance clusters against unrelated sequence.
  1            the same reads against a 4.64 Mb genome (E. coli K12 length), one candidate per read.
  4            30 kb reads, ins 10 % / del 5 % / sub 1 %, and next to every read's true interval eight long CHANCE
               clusters against unrelated genome, as the reference's anchoring reports them for such reads: the
               heavy-tailed SDP regime of BASELINE.md (455 ms of SDPAlign per read on one core).
With N>1 every rank maps its own shard of the same size against its own replica of the genome (weak scaling, no
data-path collective; one RCCL all-reduce of the run counters at the end).  `python bench.py --gpus N` started
without a launcher starts its N ranks itself (fresh child processes, before anything touches a GPU).

Output on rank 0: the LAST stdout line is ONE compact JSON object (< 4 KB: the driver keeps an 8 KB tail) with the
contract's fields, `roofline` (the step's dominant kernel), `cpu_baseline` and a five-field stub per other configuration.
Everything else -- every kernel's roofline entry, the SDP phases, the whole-mapper leg, the reference binary's baseline, the
other configurations' full records -- goes to EARLIER stdout lines ("[bench detail] <key>: <json>") and, whole, to
--detail-out (default gpurun_out/bench_detail.json).
"""
import argparse
import json
import os
import subprocess
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before torch starts the HIP runtime (blasr_hip_create says why)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# PMC figures (HBM bytes, VALU instructions per launch) come from rocprofv3 --pmc passes kept under profiles/
# (tools/pmc_config.sh -> tools/pmc_summary.py): a file is used only when it was taken on the SAME workload and with the
# SAME build of the library (blasr_hip_build_id(): a hash over the library's sources), else `traffic` is null.
def pmc_for(config, reads, read_len, band, build):
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc*.json")), reverse=True):
        try:
            d = json.load(open(path))
        except ValueError:
            continue
        w = d.get("workload") or {}
        if d.get("build_id") == build and (w.get("config"), w.get("reads"), w.get("read_len"), w.get("band")) == (
                config, reads, read_len, band):
            return os.path.relpath(path, ROOT), d.get("kernels", {})
    return None, {}


# Issue cost of one wave64 VALU instruction on the MI355X (tools/valu_rate.hip, profiles/r02_valu_rate.txt, 8 waves per
# SIMD): 4.1-4.4 cycles for min / max / min3 / compare / cndmask / shift / DPP forms, 2.3-2.9 for add / sub / and / or /
# xor / mov.  The fill kernels' hot loops are ~60 % of the first kind (ISA of fillp_kernel: 80 of 135 per 4-row step).
VALU_CYCLES = {"slow": 4.15, "fast": 2.4, "slow_share": 0.6}
SIMDS, CLOCK_HZ = 1024, 2.4e9


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=(1, 2, 3, 4),
                    help="BASELINE.json configs[] index (3 = configs[2]'s workload as configs[3] states it: 1 M reads in all, "
                         "sharded over the ranks -- strong scaling, 125 k reads per rank at 8 GPUs)")
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (0 = the config's: 100000; 6250 for config 4)")
    ap.add_argument("--read-len", type=int, default=0, help="template bases per read (0 = the config's)")
    ap.add_argument("--band", type=int, default=16)
    ap.add_argument("--no-guided", action="store_true",
                    help="the reference's -noUseGuidedAlign: the survivors go through RefineAlignment's k-band branch "
                         "(Blasr.cpp:917-1046: band from the SDP alignment's drift, AffineKBandAlign, ~1.3 k at 10 kb and "
                         "~2.7 k at 30 kb reads) instead of AffineGuidedAlign -- configs[4]'s 'wide k-band' form")
    ap.add_argument("--kband", type=int, default=0, help="with --no-guided: params.bandSize (0 = from the drift, the default)")
    ap.add_argument("--other-configs", type=int, default=-1, metavar="STEPS",
                    help="after the line's own workload, also run configs[1], configs[4] and configs[4] --no-guided for STEPS "
                         "steps each (child processes, each with its own parity sample and CPU leg) and report them as "
                         "`other_configs`.  -1 = 3 on the driver's default line (config 2, one GPU, default sizes), else 0")
    ap.add_argument("--families", type=int, default=-1, help="repeat families planted in the genome (-1 = the config's)")
    ap.add_argument("--spurious", type=float, default=-1.0, help="mean chance clusters per read (-1 = the config's)")
    ap.add_argument("--genome-len", type=int, default=0, help="override the genome length (single contig)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="reads timed on the host cores (0 = skip, -1 = auto)")
    ap.add_argument("--cpu-cores", type=int, default=0, help="host threads of the CPU baseline (0 = all the host's cores)")
    ap.add_argument("--blasrmc-reads", type=int, default=-1,
                    help="reads mapped by the reference binary (oracle/_ref/blasrmc) as a second baseline (0 = skip)")
    ap.add_argument("--workspace-gb", type=float, default=200.0,
                    help="limit of the context's grow-only device workspace (blasr_hip_set_workspace_limit)")
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--share-gpus", action="store_true",
                    help="let the ranks share the GPUs that exist (rank r on device r mod devices, gloo for the counters): "
                         "runs the N-rank path with the real device step on a box with fewer GPUs; the line says so")
    ap.add_argument("--from-reads", type=int, default=-1, metavar="STEPS",
                    help="after the timed steps, also run STEPS passes of the WHOLE mapper from the reads alone (index "
                         "built on the device before the timed steps, suffix-array anchoring, interval finding, then the "
                         "path) and report them as `end_to_end` (N=1 only; needs ~130 GB of HBM for an hg19-size genome).  "
                         "-1 = 1 pass at the default workload (config 2, one GPU, the config's own read count), else 0")
    ap.add_argument("--extend", action="store_true",
                    help="the reference's -extend (Blasr.cpp:1749-1940): both ends of every SDP alignment extended before "
                         "the first filter.  NOT the headline configuration; the CPU leg then runs the reference's "
                         "ExtendAlignment too (tools/fuzz_mapper_gpu.py pins -extend against the reference binary)")
    ap.add_argument("--save-workload", default="", metavar="PATH",
                    help="write the generated inputs (genome, reads, intervals) to PATH and go on")
    ap.add_argument("--load-workload", default="", metavar="PATH",
                    help="read the inputs a --save-workload run wrote instead of generating them (rocprofv3 --pmc dies "
                         "inside torch's generator on this image: tools/pmc_config.sh profiles a loaded workload)")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"), metavar="PATH",
                    help="where the FULL record goes (the last stdout line is the compact one); '' = nowhere")
    ap.add_argument("--stub-device", action="store_true",
                    help="CPU-only dry run of the launch / shard / reduce code (gloo, no device step): for the tests")
    return ap.parse_args(argv)


def config_defaults(args):
    """The workload of a BASELINE.json config as concrete numbers."""
    c = args.config
    d = dict(reads=100000, read_len=10000, ins=0.11, dele=0.04, sub=0.01, families=0, spurious=0.0, loose=0.0,
             max_unanchored=300, hg=False, spurious_span=None)
    if c in (2, 3):
        # the reference's anchoring reports exactly one interval for a 10 kb read of a repeat-free genome (blasrmc -V 1
        # on 40 such reads): what competes on hg19 are the other copies of repeats; a few short chance clusters on top
        d.update(families=4000, spurious=0.25, hg=True)
    if c == 4:
        # what blasrmc -V 1 reports for 30 kb reads at 15 % indels (12 reads, 4.6 Mb genome): the true interval, tight,
        # plus eight or nine CHANCE clusters per read spanning 5-29 kb of the read and 17-41 kb of unrelated genome
        d.update(reads=6250, read_len=30000, ins=0.10, dele=0.05, sub=0.01, families=0, spurious=8.0, loose=0.0,
                 max_unanchored=300, hg=False, spurious_span=(0.15, 0.9, 0.55, 1.3))
    if c == 3:  # configs[3]: 1 M reads in all, a contiguous 1/N shard per rank
        d["reads"] = 1000000 // max(int(os.environ.get("WORLD_SIZE", "1") or 1), args.gpus, 1)
        d["total_reads"] = 1000000
    if args.reads > 0:
        d["reads"] = args.reads
    if args.read_len > 0:
        d["read_len"] = args.read_len
    if args.families >= 0:
        d["families"] = args.families
    if args.spurious >= 0:
        d["spurious"] = args.spurious
    return d


def workload_label(args, d, genome_len, n_contigs):
    """config.workload, built from what actually ran."""
    names = {1: "configs[1]", 2: "configs[2]", 3: "configs[3] (1 M reads in all, this many per GPU: strong scaling)",
             4: "configs[4] (per-GPU share)"}
    return ("%s: %d synthetic %.0f kb reads per GPU (ins %.0f%%/del %.0f%%/sub %.0f%%) vs a %d-base synthetic genome "
            "(%d contig%s%s), %d planted repeat families, %.2f chance clusters per read%s%s, -sdpTupleSize 11, %s, "
            "SMRT matrix, ins=del=5, open=50, ext=0, nCandidates 10; timed step = windows + SDP + sort/filter + refine "
            "+ sort/filters" % (
                names[args.config], d["reads"], d["read_len"] / 1e3, d["ins"] * 100, d["dele"] * 100, d["sub"] * 100,
                genome_len, n_contigs, "" if n_contigs == 1 else "s",
                ", hg19 chromosome lengths" if d["hg"] and not args.genome_len else "", d["families"], d["spurious"],
                " (long: %.0f-%.0f%% of the read against %.0f-%.0f%% of its length of unrelated genome)"
                % tuple(100 * x for x in d["spurious_span"]) if d["spurious_span"] else "",
                ", intervals loosened by up to %.0f%% of the read per side" % (d["loose"] * 100) if d["loose"] else "",
                ("-noUseGuidedAlign (RefineAlignment's k-band branch: bandSize %d%s, AffineKBandAlign with indel 5)"
                 % (args.kband, " = from the SDP alignment's drift" if args.kband == 0 else "")) if args.no_guided
                else "band %d" % args.band)) + (" (run with -extend: ExtendAlignmentForward / Reverse before the first filter)"
                               if args.extend else "")


def launch_ranks(args):
    """`bench.py --gpus N` without a launcher: start the N ranks as fresh children (the contract's own command
    line) before this process has touched a GPU, and leave with their exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------------------
# CPU baselines (checker code: never on the product path)
# ---------------------------------------------------------------------------------------------------------------
def cpu_baseline(wl, genome_t, starts, n_reads, band, hits, hit_off, cores, extend=False, guided=True):
    """The path for the first `n_reads` reads of the same workload on the host cores: every candidate interval
    through the reference's OWN SDPAlign + ComputeAlignmentStats, the survivors through its AffineGuidedAlign +
    ComputeAlignmentStats (oracle/_ref/libblasr_ref.so, built from the reference's sources where they lie: kind
    "reference"; the oracle's bit-exact restatement where that build is absent: kind "port"), windows and the per-read
    sort / filters through the oracle's restatement of AlignIntervals' window rule and the post filters.  Since the
    CPU answers are at hand, the device's hits of the same reads (last timed step) are compared with them."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from blasr_testlib import HIT_FIELDS, MapperOracle, load_oracle, load_ref, oracle_map_reads

    ref = load_ref()
    engine = ref if ref is not None else load_oracle()
    mo = MapperOracle()
    reads, iv = wl["reads"][:n_reads], wl["intervals"]
    iv = iv[iv["read"] < n_reads]
    pool = wl["pool"][: int(reads["off"][-1] + reads["length"][-1])].cpu().numpy()
    genome = genome_t.cpu().numpy()  # (3 GB for configs[2]: the checker's window rule reads it by address)
    contigs = None if len(starts) <= 2 else starts.astype(np.uint32)

    def one(r):
        mine = iv[iv["read"] == r].copy()
        mine["read"] = 0
        rd = reads[r: r + 1].copy()
        p = pool[int(rd["off"][0]): int(rd["off"][0]) + int(rd["length"][0])]
        rd["off"] = 0
        return oracle_map_reads(engine, mo, genome, contigs, p, rd, mine, band=band, extend=extend, guided=guided, with_ops=False)[0]

    t0 = time.perf_counter()
    if cores <= 1:
        res = [one(r) for r in range(n_reads)]
    else:
        # the reference's -nproc: its functions are re-entrant, and ctypes releases the interpreter lock for the
        # duration of a call, so host threads do run side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=cores) as pool_ex:
            res = list(pool_ex.map(one, range(n_reads), chunksize=4))
    dt = time.perf_counter() - t0
    aligned = sum(int(reads["length"][r]) for r in range(n_reads) if res[r])
    mismatches = 0
    for r in range(n_reads):
        mine = hits[int(hit_off[r]): int(hit_off[r + 1])]
        local = np.nonzero(iv["read"] == r)[0]  # the checker numbered the read's intervals 0..
        glob = np.nonzero(wl["intervals"]["read"] == r)[0]
        ok = len(mine) == len(res[r])
        for h, x in zip(mine, res[r]):
            for f in HIT_FIELDS:
                want = x[f] if f != "interval" else int(glob[x[f]])
                ok = ok and h[f] == want
        mismatches += not ok
    refiner = "AffineGuidedAlign" if guided else "ComputeDrift / band rule + AffineKBandAlign (-noUseGuidedAlign)"
    what = ("the reference's own SDPAlign + %s%s + ComputeAlignmentStats (oracle/_ref/libblasr_ref.so)"
            % ("ExtendAlignmentForward / Reverse + " if extend else "", refiner)
            if ref is not None else "the oracle's restatement of SDP + %s%s" % ("extension + " if extend else "", refiner))
    return dict(value=aligned / dt, unit="bases/s", cores=max(cores, 1), kind="reference" if ref is not None else "port",
                sample="first %d reads of the same workload (%d candidate intervals) through %s, windows / sort / filters "
                       "through the oracle (%.1f s on %d host thread%s, %.1f ms/read per thread)"
                       % (n_reads, len(iv), what, dt, max(cores, 1), "" if cores <= 1 else "s",
                          dt * max(cores, 1) / max(n_reads, 1) * 1e3),
                parity_checked=n_reads, parity_mismatches=int(mismatches))


def blasrmc_baseline(wl, genome_t, n_reads, read_len, cores, extend=False, device_index=0):
    """The reference BINARY (oracle/_ref/blasrmc -nproc N -m 4 -bestn 1 -sa ...) on a FASTA dump of the first `n_reads`
    reads against a FASTA of their own loci (+- 5 kb, one contig each: the 3-Gbase genome's index is 12.4 GB on disk and
    the binary needs minutes to load it), wall time less a zero-read run: whole BLASR -- anchoring, intervals, the path,
    mapQV, printing -- per read.  The loci genome's suffix array is written beforehand (this library's device builder,
    byte-identical to sawriter's file: tests/test_sa_build_gpu.py) so that neither run builds it in-process."""
    import tempfile

    import numpy as np
    exe = os.path.join(ROOT, "oracle", "_ref", "blasrmc")
    if not os.path.exists(exe) or n_reads <= 0:
        return None
    import blasr_amd
    from blasr_amd import hip
    pos, strand = wl["origin"]
    pos = pos[:n_reads].cpu().numpy()
    G = int(genome_t.numel())
    reads = wl["reads"][:n_reads]
    pool = wl["pool"][: int(reads["off"][-1] + reads["length"][-1])].cpu().numpy()
    with tempfile.TemporaryDirectory() as tmp:
        parts, starts = [], [0]
        with open(os.path.join(tmp, "genome.fasta"), "w") as f:
            for r in range(n_reads):
                a, b = max(0, int(pos[r]) - 5000), min(G, int(pos[r]) + read_len + 5000)
                locus = genome_t[a:b].cpu().numpy()
                f.write(">locus%d\n%s\n" % (r, bytes(locus).decode()))
                parts += [locus, np.array([ord("N")], np.uint8)]
                starts.append(starts[-1] + len(locus) + 1)
        t0 = time.perf_counter()
        c2 = blasr_amd.Context(device_index)  # a context of its own: the bench's holds the big genome
        c2.upload_genome(np.concatenate(parts), np.array(starts, dtype=np.uint32))
        hip.write_suffix_array(os.path.join(tmp, "genome.fasta.sa"), c2.build_suffix_array(13))
        c2.close()
        t_index = time.perf_counter() - t0
        del parts
        with open(os.path.join(tmp, "reads.fasta"), "w") as f:
            for r in range(n_reads):
                s = pool[int(reads["off"][r]): int(reads["off"][r]) + int(reads["length"][r])]
                f.write(">read%d/%d/0_%d\n%s\n" % (r, r, len(s), bytes(s).decode()))
        open(os.path.join(tmp, "none.fasta"), "w").write(">none/0/0_4\nACGT\n")
        wall = {}
        for tag, fa in (("none", "none.fasta"), ("reads", "reads.fasta")):
            t0 = time.perf_counter()
            p = subprocess.run([exe, os.path.join(tmp, fa), os.path.join(tmp, "genome.fasta"), "-sa",
                                os.path.join(tmp, "genome.fasta.sa"), "-m", "4", "-bestn", "1",
                                "-nproc", str(cores), "-out", os.path.join(tmp, tag + ".m4")] + (["-extend"] if extend else []),
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1200)
            wall[tag] = time.perf_counter() - t0
            if p.returncode != 0:
                return dict(error="blasrmc exit %d" % p.returncode)
        lines = open(os.path.join(tmp, "reads.m4")).read().splitlines()
    dt = max(wall["reads"] - wall["none"], 1e-9)
    names = set(int(x.split("/", 1)[0][4:]) for x in lines)
    aligned = sum(int(reads["length"][r]) for r in names)
    return dict(value=aligned / dt, unit="bases/s", cores=cores, kind="reference binary", reads_per_s=n_reads / dt,
                signal_s=dt,
                sample="oracle/_ref/blasrmc -nproc %d -m 4 -bestn 1 -sa on the first %d reads against a FASTA of their own "
                       "loci (+-5 kb; %d bases, its suffix array written beforehand in %.1f s): %.1f s wall less %.1f s of "
                       "a zero-read run (index load), %d reads reported; whole BLASR incl. anchoring, not only the path"
                       % (cores, n_reads, starts[-1], t_index, wall["reads"], wall["none"], len(names)))


def blasrmc_true_genome(ctx, wl, genome_t, starts, n_reads, cores, extend=False):
    """The reference BINARY on the configuration as stated: oracle/_ref/blasrmc -nproc <cores> -m 4 -bestn 1 -sa on the
    first `n_reads` reads against a FASTA of the WHOLE genome of the run (the hg19-size one at configs[2]) and the .sa file
    of that genome -- the index resident on the device, written by blasr_hip_sa_write_resident (sawriter's file byte for
    byte: tests/test_sa_build_gpu.py).  Wall time less a zero-read run (which loads genome and index the same way).
    Needs ~5 x the genome in temporary disk space and ~5.5 x in host memory; returns None where the box has not got them."""
    import shutil
    import tempfile

    import numpy as np
    exe = os.path.join(ROOT, "oracle", "_ref", "blasrmc")
    G = int(genome_t.numel())
    if not os.path.exists(exe) or n_reads <= 0:
        return None
    tmp_root = tempfile.gettempdir()
    try:
        free_disk = shutil.disk_usage(tmp_root).free
        free_mem = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    except (OSError, ValueError):
        return None
    if free_disk < 6 * G + (1 << 30) or free_mem < 7 * G + (4 << 30):
        return dict(skipped="not enough room for the whole genome's FASTA and index (%.0f GB of disk, %.0f GB of memory free; "
                            "%.0f and %.0f wanted)" % (free_disk / 1e9, free_mem / 1e9, 6 * G / 1e9, 7 * G / 1e9))
    reads = wl["reads"][:n_reads]
    pool = wl["pool"][: int(reads["off"][-1] + reads["length"][-1])].cpu().numpy()
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        fa = os.path.join(tmp, "genome.fasta")
        with open(fa, "wb") as f:
            for c in range(len(starts) - 1):  # a contig's bases without the N the reader appends after every record
                f.write(b">contig%d\n" % c)
                a, b = int(starts[c]), int(starts[c + 1]) - 1
                for at in range(a, b, 1 << 28):
                    f.write(genome_t[at: min(b, at + (1 << 28))].cpu().numpy().tobytes())
                f.write(b"\n")
        ctx.write_resident_suffix_array(fa + ".sa")
        t_files = time.perf_counter() - t0
        with open(os.path.join(tmp, "reads.fasta"), "w") as f:
            for r in range(n_reads):
                s = pool[int(reads["off"][r]): int(reads["off"][r]) + int(reads["length"][r])]
                f.write(">read%d/%d/0_%d\n%s\n" % (r, r, len(s), bytes(s).decode()))
        open(os.path.join(tmp, "none.fasta"), "w").write(">none/0/0_4\nACGT\n")
        wall = {}
        for tag, rf in (("none", "none.fasta"), ("reads", "reads.fasta")):
            t0 = time.perf_counter()
            try:
                p = subprocess.run([exe, os.path.join(tmp, rf), fa, "-sa", fa + ".sa", "-m", "4", "-bestn", "1", "-nproc", str(cores),
                                    "-out", os.path.join(tmp, tag + ".m4")] + (["-extend"] if extend else []),
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1500)
            except subprocess.TimeoutExpired:
                return dict(error="blasrmc on the whole genome: no end after 1500 s")
            wall[tag] = time.perf_counter() - t0
            if p.returncode != 0:
                return dict(error="blasrmc exit %d" % p.returncode)
        lines = open(os.path.join(tmp, "reads.m4")).read().splitlines()
    dt = max(wall["reads"] - wall["none"], 1e-9)
    names = set(int(x.split("/", 1)[0][4:]) for x in lines)
    aligned = sum(int(reads["length"][r]) for r in names)
    return dict(value=aligned / dt, unit="bases/s", cores=cores, kind="reference binary", reads_per_s=n_reads / dt, signal_s=dt,
                genome_bases=G,
                sample="oracle/_ref/blasrmc -nproc %d -m 4 -bestn 1 -sa on the first %d reads against the run's WHOLE genome "
                       "(%d bases in %d contigs; FASTA and the device-built index written in %.0f s): %.1f s wall less %.1f s "
                       "of a zero-read run (genome + index load), %d reads reported; whole BLASR incl. anchoring, not only the path"
                       % (cores, n_reads, G, len(starts) - 1, t_files, wall["reads"], wall["none"], len(names)))


# ---------------------------------------------------------------------------------------------------------------

# ---------------------------------------------------------------------------------------------------------------
# the whole mapper from the reads alone (SURVEY next-1 + next-4 in front of the path): opt-in, reported beside the metric
# ---------------------------------------------------------------------------------------------------------------
def end_to_end(ctx, genome_t, starts, wl, params, steps, dev):
    import numpy as np
    import torch
    from blasr_amd import hip
    out = {"index_build_s": wl["index_build_s"]}
    t0 = time.perf_counter()
    genome_h = genome_t.cpu().numpy()  # the interval finder's p-values read the genome and its 8-mer counts on the host
    seq_starts = np.ascontiguousarray(starts, dtype=np.uint32)
    count_table = ctx.count_tuples(8)  # on the device
    out["host_genome_copy_and_count_table_s"] = round(time.perf_counter() - t0, 2)
    reads, pool_ptr = wl["reads"], wl["pool"].data_ptr()
    n_reads = len(reads)
    # ONE C++ call per pass (blasr_hip_map_from_reads_device, blasr_amd/csrc/pipeline_api.hip): the anchoring in sub-blocks
    # of reads, each sub-block's anchors copied to pinned host memory on a side stream and turned into intervals by host
    # threads while the device searches the next sub-block, then the path on all the intervals found
    ap = ctx.anchor_params(min_match_length=14, min_prefix_match_length=8)  # Blasr's values for a sawriter index
    ip = hip.interval_params(threads=min(64, os.cpu_count() or 1))
    # BLASR_HIP_PIPE="reads per sub-block,overlap": the default is the serial form (anchor everything, then ONE call of the
    # path); "25000,1" runs the path on sub-blocks of 25 000 reads beside the anchoring of the next one
    # (blasr_hip_pipeline_params.overlap) -- same hits, and measured slower on one MI355X: 1312 against 1154 ms, the search
    # and the path's kernels slow each other down by as much as they overlap (profiles/NOTES.md, round 5)
    per_block, overlap = (int(v) for v in os.environ.get("BLASR_HIP_PIPE", "100000,0").split(","))
    if os.environ.get("BLASR_HIP_INTERVALS_HOST"):
        per_block, overlap = 6250, 0
    pp = ctx.pipeline_params(reads_per_block=per_block, anchor=ap, interval=ip, map=params, overlap=overlap)
    hits = np.zeros(10 * n_reads, dtype=hip.HIT_DTYPE)
    hit_off = np.zeros(n_reads + 1, dtype=np.int64)
    stages = {}
    last = {}
    # one untimed pass first: the device and pinned anchor buffers are allocated on it (14 GB each at configs[2])
    ctx.map_from_reads_device(genome_h, seq_starts, count_table, reads, wl["reads_t"].data_ptr(), pool_ptr, pp, hits, hit_off)
    t_all = time.perf_counter()
    for _ in range(steps):
        n_hits, ivs = ctx.map_from_reads_device(genome_h, seq_starts, count_table, reads, wl["reads_t"].data_ptr(), pool_ptr,
                                                pp, hits, hit_off, intervals_capacity=10 * n_reads)
        pt = ctx.pipeline_last_timing()
        for k in ("anchoring_ms", "wait_for_intervals_ms", "path_ms", "host_intervals_ms"):  # (host_intervals_ms: the interval finding)
            stages[k] = stages.get(k, 0.0) + pt[k] / 1e3
        iv_off = np.concatenate([[0], np.cumsum(np.bincount(ivs["read"], minlength=n_reads))])
        last = dict(anchors=int(pt["anchors"]), intervals=int(len(ivs)), timing=ctx.map_last_timing(), ivs=ivs, iv_off=iv_off)
    wall = (time.perf_counter() - t_all) / steps
    n_blocks = -(-n_reads // per_block)
    same_as_serial = None
    if overlap and per_block < n_reads:
        # the overlapped form against the serial one (every sub-block anchored, then ONE call of the path), untimed: every
        # field of every hit but the three arena offsets, and the hits' places
        hits2, off2 = np.zeros_like(hits), np.zeros_like(hit_off)
        ps = ctx.pipeline_params(reads_per_block=n_reads, anchor=ap, interval=ip, map=params, overlap=0)
        n2, _ = ctx.map_from_reads_device(genome_h, seq_starts, count_table, reads, wl["reads_t"].data_ptr(), pool_ptr, ps,
                                          hits2, off2)
        fields = [f for f in hits.dtype.names if f not in ("ops_off", "q_off", "t_off")]
        same_as_serial = bool(n2 == n_hits and np.array_equal(off2, hit_off) and
                              all(np.array_equal(hits[:n_hits][f], hits2[:n2][f]) for f in fields))
        del hits2
    # how many reads have their true locus among the candidate intervals (the synthetic set-up knows it)
    true_iv = wl["intervals"][wl["kind"] == 0]
    found = 0
    ivs, iv_off = last["ivs"], last["iv_off"]
    for k in range(0, len(true_iv), max(1, len(true_iv) // 2000)):
        tv = true_iv[k]
        mine = ivs[int(iv_off[tv["read"]]): int(iv_off[tv["read"] + 1])]
        mid = (int(tv["start"]) + int(tv["end"])) // 2
        found += bool(((mine["start"] <= mid) & (mine["end"] >= mid) & (mine["strand"] == tv["strand"])).any())
    sampled = len(range(0, len(true_iv), max(1, len(true_iv) // 2000)))
    tm = last["timing"]
    out.update({
        "what": ("reads -> hits in one C++ call (blasr_hip_map_from_reads_device): MapReadToGenome x 2 strands + "
                 "SortMatchPosList on the device; the anchors never leave it: RemoveOverlappingAnchors, the window clustering and "
                 "GlobalChain run as kernels right behind the search (intervals_device.hip) and one 32-byte record per cluster "
                 "comes back; the host (%d threads) ends with the p-values and the interval sets; then the timed path of the "
                 "metric on the intervals found (instead of the synthetic ones)" % ip.threads
                 if not os.environ.get("BLASR_HIP_INTERVALS_HOST") else
                 "reads -> hits in one C++ call, round 2's route (BLASR_HIP_INTERVALS_HOST): anchors to pinned host memory, "
                 "the interval finder on %d host threads beside the device's search of the next sub-block" % ip.threads),
        "steps": steps, "ms_per_step": wall * 1e3, "reads_per_s": n_reads / wall,
        "aligned_bases_per_s": tm["aligned_bases"] / wall,
        "stage_ms_per_step": {k: v / steps * 1e3 for k, v in stages.items()},
        # the anchor stage on its own (blasr_hip_anchor_last_timing, last sub-block of the last pass): the suffix-array
        # search is what anchoring_ms is made of; `interval_finding_ms` (kernels + the host's end) is inside anchoring_ms
        "anchor_stage_ms": {k: round(float(v), 2) for k, v in ctx.anchor_last_timing().items() if k.endswith("_ms")},
        "reads_per_sub_block": per_block, "overlap": bool(overlap and per_block < n_reads),
        "hits_equal_the_serial_form": same_as_serial,
        "anchors_per_step": last["anchors"], "candidate_intervals": last["intervals"], "hits": int(tm["hits"]),
        "aligned_reads": int(tm["aligned_reads"]),
        "anchor_blocks": n_blocks,
        "true_locus_among_intervals": "%d of %d sampled reads" % (found, sampled),
    })
    return out


def other_configs(steps):
    """configs[1], configs[4] and configs[4] under -noUseGuidedAlign, each through this same script as a child process
    (fresh context, own inputs, own parity sample against the CPU checker), cut down to the figures of a bench line."""
    runs = (("configs[1]", ["--config", "1"]), ("configs[4]", ["--config", "4"]),
            ("configs[4] -noUseGuidedAlign (wide k-band)", ["--config", "4", "--no-guided"]))
    import tempfile
    res = {}
    for name, flags in runs:
        with tempfile.TemporaryDirectory() as tmp:
            detail = os.path.join(tmp, "detail.json")
            cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", "1",
                   "--from-reads", "0", "--blasrmc-reads", "0", "--other-configs", "0", "--detail-out", detail] + flags
            t0 = time.perf_counter()
            try:
                p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
            except subprocess.TimeoutExpired:
                res[name] = {"error": "timed out after 900 s"}
                continue
            if p.returncode != 0 or not os.path.exists(detail):
                res[name] = {"error": "exit %d: %s" % (p.returncode, p.stderr.decode()[-400:])}
                continue
            o = json.load(open(detail))  # the child's FULL record (its last stdout line is the compact one)
        keep = ("metric", "value", "unit", "steps", "ms_per_step", "dtype", "gcups_per_gpu", "stage_ms_per_step", "roofline",
                "roofline_of_the_largest_refine_kernel", "kernel_rooflines", "sdp_rooflines", "cpu_baseline",
                "device_memory_held_gb", "build_id")
        res[name] = {k: o[k] for k in keep if k in o}
        res[name]["config"] = {k: o["config"][k] for k in ("workload", "reads_per_gpu", "read_len", "refiner",
                                                           "candidate_intervals_per_gpu", "candidates_refined_per_gpu",
                                                           "aligned_reads") if k in o["config"]}
        res[name]["wall_s_of_the_child"] = round(time.perf_counter() - t0, 1)
    return res

# ---------------------------------------------------------------------------------------------------------------
# the driver's line: compact, < 4 KB (the driver parses the last stdout line out of an 8 KB tail)
# ---------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 4096


def _sig(x, digits=5):
    """Floats to `digits` significant figures (the full-precision numbers are in the detail record)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (digits, x))


def _pick(d, keys):
    return {k: _sig(d[k]) for k in keys if d is not None and k in d}


ROOFLINE_KEYS = ("kernel", "stage", "bound", "limiter", "achieved", "peak", "unit", "frac", "traffic", "valu_issue_frac", "avg_launch_ms",
                 "launches_per_step", "ms_per_step", "concurrent_launches", "share_of_step", "stage_frac", "stage_gcups",
                 "salu_over_valu")
CPU_KEYS = ("value", "unit", "cores", "kind", "parity_checked", "parity_mismatches")


def other_config_stub(o):
    """The five fields an other configuration leaves on the driver's line."""
    if "error" in o:
        return {"error": str(o["error"])[:120]}
    rl, cb = o.get("roofline") or {}, o.get("cpu_baseline") or {}
    return {"ms_per_step": _sig(o.get("ms_per_step")), "value": _sig(o.get("value")), "roofline_frac": _sig(rl.get("frac")),
            "roofline_kernel": rl.get("kernel"), "parity_mismatches": cb.get("parity_mismatches"),
            "parity_checked": cb.get("parity_checked")}


def minimal_line(out, detail_path=None, note=None):
    """The last resort of emit(): the contract's fields alone, every free-text field cut short -- a run of several minutes
    must not end without the driver's line because an optional part of the record grew."""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype"))
    line["metric"] = str(line.get("metric", ""))[:200]
    line["data"] = "synthetic"
    line["config"] = {"workload": str((out.get("config") or {}).get("workload_short") or (out.get("config") or {}).get("workload", ""))[:160]}
    rl = out.get("roofline")
    line["roofline"] = _pick(rl, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic")) if rl else None
    if line["roofline"] and "kernel" in line["roofline"]:
        line["roofline"]["kernel"] = str(line["roofline"]["kernel"])[:80]
    cb = out.get("cpu_baseline")
    line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "parity_checked", "parity_mismatches")) if cb else None
    if "build_id" in out:
        line["build_id"] = str(out["build_id"])[:16]
    if detail_path:
        line["detail"] = str(detail_path)[:200]
    line["truncated"] = str(note or "the full line did not fit")[:120]
    return json.dumps(line, separators=(",", ":"))


def compact_line(out, detail_path=None, strict=True):
    """The full record `out` cut down to the one line the driver reads: the contract's fields, `roofline` of the step's
    dominant kernel, `cpu_baseline`, the wall time of the stages, a stub per other configuration.  When the result would
    not fit LINE_LIMIT bytes even without its optional parts: raises (strict; tests/test_bench_line_cpu.py feeds it full
    records) or falls back to minimal_line (emit(): a finished run always prints a line the driver can read)."""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype"))
    line["data"] = "synthetic (generated on the device from seeds; see the detail record)"
    cfg = out.get("config") or {}
    line["config"] = _pick(cfg, ("reads_per_gpu", "read_len", "band", "refiner", "candidate_intervals_per_gpu",
                                 "candidates_refined_per_gpu", "reads", "aligned_reads", "hits_per_gpu"))
    line["config"]["workload"] = cfg.get("workload_short") or str(cfg.get("workload", ""))[:240]
    if "gcups_per_gpu" in out:
        line["gcups_per_gpu"] = _sig(out["gcups_per_gpu"])
    if out.get("stage_ms_per_step"):
        line["stage_ms_per_step"] = {k: _sig(v, 4) for k, v in out["stage_ms_per_step"].items()}
    rl = out.get("roofline")
    line["roofline"] = _pick(rl, ROOFLINE_KEYS) if rl else None
    if out.get("roofline_of_the_largest_refine_kernel"):
        line["roofline_of_the_largest_refine_kernel"] = _pick(out["roofline_of_the_largest_refine_kernel"],
                                                              ("kernel", "bound", "frac", "valu_issue_frac", "avg_launch_ms",
                                                               "launches_per_step", "traffic"))
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = _pick(cb, CPU_KEYS)
        line["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:160]
    wb = out.get("cpu_baseline_whole_blasr")
    if wb:
        line["cpu_baseline_whole_blasr"] = _pick(wb, ("value", "unit", "cores", "kind", "reads_per_s", "genome_bases", "error"))
    ee = out.get("end_to_end")
    if ee:
        line["end_to_end"] = _pick(ee, ("ms_per_step", "reads_per_s", "aligned_bases_per_s", "vs_whole_blasr"))
    if out.get("other_configs"):
        line["other_configs"] = {name: other_config_stub(o) for name, o in out["other_configs"].items()}
    for k in ("device_memory_held_gb", "build_id", "physical_gpus", "rccl_ranks", "counters_backend"):
        if k in out:
            line[k] = _sig(out[k])
    if out.get("per_rank_ms_per_step") and out.get("n_gpus", 1) > 1:
        line["per_rank_ms_per_step"] = [_sig(r.get("step"), 4) for r in out["per_rank_ms_per_step"]]
    if out.get("note"):
        line["note"] = str(out["note"])[:200]
    if detail_path:
        line["detail"] = detail_path
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:  # drop the optional parts, largest first, rather than print a line the driver cannot read
        for k in ("stage_ms_per_step", "end_to_end", "cpu_baseline_whole_blasr", "roofline_of_the_largest_refine_kernel",
                  "per_rank_ms_per_step"):
            line.pop(k, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) < LINE_LIMIT:
                break
    if len(text) >= LINE_LIMIT:
        if strict:
            raise ValueError("bench.py: the driver's line would be %d bytes (limit %d)" % (len(text), LINE_LIMIT))
        text = minimal_line(out, detail_path, "the compact line would be %d bytes (limit %d)" % (len(text), LINE_LIMIT))
    return text


def emit(out, detail_path):
    """Detail first (earlier stdout lines + the detail file), the compact line LAST."""
    rel = None
    if detail_path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
            with open(detail_path, "w") as f:
                json.dump(out, f, indent=1)
            rel = os.path.relpath(detail_path, ROOT) if os.path.abspath(detail_path).startswith(ROOT + os.sep) else detail_path
        except OSError as e:
            print("[bench detail] could not write %s: %s" % (detail_path, e))
    for k, v in out.items():
        if isinstance(v, (dict, list)):
            print("[bench detail] %s: %s" % (k, json.dumps(v)))
    sys.stdout.flush()
    try:
        text = compact_line(out, rel, strict=False)
    except Exception as e:  # (a malformed optional part must not cost the run its line either)
        text = minimal_line(out, rel, "compact_line failed: %s" % e)
    print(text)
    sys.stdout.flush()


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        raise SystemExit(launch_ranks(args))  # nothing below has run: no GPU call in this process
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a %d-rank run as %d GPUs"
                         % (args.gpus, world, world, args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:  # one process per GPU: the library's host-side pools (filters, interval finder) share the node's cores
        os.environ.setdefault("BLASR_HIP_HOST_THREADS", str(max(1, (os.cpu_count() or 1) // world)))
    d = config_defaults(args)
    default_line = (args.config == 2 and args.gpus == 1 and args.reads <= 0 and args.read_len <= 0 and args.genome_len <= 0 and
                    not args.stub_device and not args.load_workload and not args.no_guided and not args.extend)
    if args.other_configs < 0:
        args.other_configs = 3 if default_line else 0
    if args.from_reads < 0:  # the whole-mapper leg rides along on the driver's default line only
        args.from_reads = 1 if (args.config == 2 and args.gpus == 1 and args.reads <= 0 and args.read_len <= 0 and
                                args.genome_len <= 0 and not args.stub_device and not args.load_workload) else 0

    import numpy as np
    import torch

    from blasr_amd import shard

    if args.stub_device:
        # dry run of everything around the device step: launch, rendezvous, shard seeds, counters, reduce, report
        dist = None
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
        n = d["reads"]
        t0 = time.perf_counter()
        elapsed = shard.max_over_ranks(time.perf_counter() - t0 + 1e-3)
        totals = shard.reduce_counters(n, n, n * d["read_len"], 0)
        if rank == 0:
            print(json.dumps({"metric": "stub", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "scaling": "weak", "config": {"workload": workload_label(args, d, 0, 1),
                                                            "reads": totals["reads"], "aligned_reads": totals["aligned_reads"]},
                              "value": totals["aligned_bases"] * args.steps / elapsed, "stub": True,
                              "seeds": [args.seed + r for r in range(world)]}))
        if dist is not None:
            dist.destroy_process_group()
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the refine path has no CPU implementation")
    n_dev = torch.cuda.device_count()
    if world > n_dev and not args.share_gpus:
        raise SystemExit("bench.py: %d ranks but %d GPUs (--share-gpus runs them on the GPUs there are)" % (world, n_dev))
    shared = args.share_gpus and world > n_dev
    device_index = local_rank % n_dev
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if shared:  # RCCL refuses two ranks on one device: the four counters go over gloo instead
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    red_dev = "cpu" if shared else dev
    if world > 1 and not shared:
        # one process per GPU: every rank must sit on a device of its own
        mine = torch.tensor([torch.cuda.current_device()], device=dev, dtype=torch.int64)
        seen = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(seen, mine)
        ids = sorted(int(t.item()) for t in seen)
        if len(set(ids)) != world:
            raise SystemExit("bench.py: ranks share devices %s without --share-gpus" % ids)

    import blasr_amd
    from blasr_amd import hip, synth

    # ---- inputs, generated on the device ------------------------------------------------------------------------
    t_gen = time.perf_counter()
    if args.genome_len > 0:
        contig_lengths = [args.genome_len]
    elif d["hg"]:
        contig_lengths = list(synth.HG19_CHROM_LENGTHS)
    else:
        contig_lengths = [synth.ECOLI_K12_LENGTH]
    if args.load_workload:
        blob = torch.load(args.load_workload, weights_only=False)
        genome, starts, n_copies = blob["genome"].to(dev), blob["starts"], blob["n_copies"]
        wl = dict(blob["wl"])
        wl["pool"] = wl["pool"].to(dev)
        wl["origin"] = tuple(t.to(dev) for t in wl["origin"])
    else:
        genome, starts, copies = synth.make_repeat_genome(contig_lengths, 1, dev, families=d["families"])
        n_copies = 0 if copies is None else int(copies["start"].numel())
        wl = synth.make_map_workload(genome, copies if d["families"] > 0 else None, d["reads"], d["read_len"],
                                     args.seed + rank, dev, ins=d["ins"], dele=d["dele"], sub=d["sub"],
                                     max_unanchored=d["max_unanchored"], spurious=d["spurious"], loose=d["loose"],
                                     spurious_span=d["spurious_span"])
        del copies
    if args.save_workload:
        host = dict(wl)
        host["pool"] = wl["pool"].cpu()
        host["origin"] = tuple(t.cpu() for t in wl["origin"])
        torch.save(dict(genome=genome.cpu(), starts=starts, n_copies=n_copies, wl=host), args.save_workload)
    G = int(genome.numel())
    n_reads, n_iv = len(wl["reads"]), len(wl["intervals"])
    reads_t = torch.from_numpy(wl["reads"].view(np.uint8)).to(dev)
    wl["reads_t"] = reads_t
    iv_t = torch.from_numpy(wl["intervals"].view(np.uint8)).to(dev)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen

    ctx = blasr_amd.Context(device_index)
    ctx.set_workspace_limit(int(args.workspace_gb * (1 << 30)) // (1 if not shared else -(-world // n_dev)))
    ctx.adopt_genome_device(genome.data_ptr(), G, starts.astype(np.uint32) if len(starts) > 2 else None)
    if args.from_reads > 0 and world == 1:
        # the index of the genome, built on the device while the 32 B per base it takes to build are still free
        # (sawriter's default lookup prefix; the array stays resident for the end-to-end leg)
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        ctx.build_suffix_array(13, download=False)
        wl["index_build_s"] = round(time.perf_counter() - t0, 2)
    if args.no_guided:
        params = ctx.map_params(band=args.kband, use_guided_align=0, extend_alignments=1 if args.extend else 0)
    else:
        params = ctx.map_params(band=args.band, extend_alignments=1 if args.extend else 0)
    hits = np.zeros(n_iv, dtype=hip.HIT_DTYPE)
    hit_off = np.zeros(n_reads + 1, dtype=np.int64)
    acc = {}
    kacc = {}  # the refine stage's heavy kernels, each on its own (blasr_hip_refine_last_kernels), summed over the steps
    sacc = {}  # the SDP stage's (blasr_hip_sdp_last_kernels)

    def step():
        w0 = time.perf_counter()
        ctx.map_reads_device(n_reads, reads_t.data_ptr(), wl["pool"].data_ptr(), n_iv, iv_t.data_ptr(), params, hits, hit_off)
        w1 = time.perf_counter()
        t = dict(ctx.map_last_timing())
        if os.environ.get("BLASR_HIP_STEP_TRACE"):  # the call's wall time against the sum of its stages' timers
            print("[step] call %.1f ms, stages %.1f ms" % ((w1 - w0) * 1e3, sum(t[k] for k in (
                "windows_ms", "sdp_ms", "filter1_ms", "refine_ms", "filter2_ms", "fetch_ms"))), file=sys.stderr)
        t.update(ctx.sdp_last_timing())
        t.update(ctx.last_timing())
        f4 = ctx.last_fill4()
        t.update(fill4_ms=f4["ms"], fill4_launches=f4["launches"], fill4_bytes=f4["bytes"], fill4_cells=f4["cells"])
        for k, v in t.items():
            acc[k] = acc.get(k, 0) + v
        for kr in ctx.last_kernels():
            a = kacc.setdefault(kr["name"], dict(ms=0.0, launches=0, bytes=0, cells=0, candidates=0))
            for f in a:
                a[f] += kr[f]
        for kr in ctx.sdp_last_kernels():  # the SDP stage's heavy kernels, timed the same way (blasr_hip_sdp_last_kernels)
            a = sacc.setdefault(kr["name"], dict(ms=0.0, launches=0, bytes=0, fragments=0, invocations=0, concurrent=1))
            for f in ("ms", "launches", "bytes", "fragments", "invocations"):
                a[f] += kr[f]
            a["concurrent"] = max(a["concurrent"], kr["concurrent"])
        return t

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    acc.clear()
    kacc.clear()
    sacc.clear()
    barrier()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    barrier()
    elapsed = time.perf_counter() - t0

    # the only collective of the path: one all-reduce of the per-rank counters (RCCL over xGMI)
    totals = shard.reduce_counters(n_reads, last["aligned_reads"], last["aligned_bases"], last["dp_cells"], device=red_dev)
    own_elapsed = elapsed
    elapsed = shard.max_over_ranks(elapsed, device=red_dev)
    K_ = max(args.steps, 1)
    rank_ms = shard.per_rank([own_elapsed / K_ * 1e3, acc["windows_ms"] / K_, acc["sdp_ms"] / K_, acc["filter1_ms"] / K_,
                              acc["refine_ms"] / K_, acc["filter2_ms"] / K_, float(last["refined"])], device=red_dev)

    if rank == 0:
        K = args.steps
        value = totals["aligned_bases"] * K / elapsed

        def gbps(nbytes, ms):
            return nbytes / (ms / 1e3) / 1e9 if ms > 0 else 0.0

        kinds = np.bincount(wl["kind"], minlength=3)
        out = {
            "metric": "aligned bases/sec (refine path: windows + SDP chaining + %s + filters), GCUPS per GPU"
                      % ("k-band affine DP (-noUseGuidedAlign)" if args.no_guided else "guided banded affine DP"),
            "value": value,
            "unit": "bases/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.config == 3 and args.reads <= 0 else "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic, generated on the device in %.0f s: %d-base i.i.d. genome with %d planted repeat families "
                    "(%d copies), %d reads, %d candidate intervals (%d true loci, %d repeat copies, %d chance clusters)"
                    % (t_gen, G, d["families"], n_copies, n_reads, n_iv,
                       kinds[0], kinds[1], kinds[2]),
            "config": {"workload": workload_label(args, d, G, len(starts) - 1),
                       "reads_per_gpu": n_reads, "read_len": d["read_len"], "band": args.kband if args.no_guided else args.band,
                       "refiner": "AffineKBandAlign" if args.no_guided else "AffineGuidedAlign",
                       "candidate_intervals_per_gpu": n_iv, "candidates_refined_per_gpu": int(last["refined"]),
                       "reads": totals["reads"], "aligned_reads": totals["aligned_reads"], "hits_per_gpu": int(last["hits"])},
            "gcups_per_gpu": totals["dp_cells"] / world * K / elapsed / 1e9,
            # wall time of the mapper's stages (host clock around each stage of the C++ driver) ...
            "stage_ms_per_step": {
                "windows": acc["windows_ms"] / K, "sdp": acc["sdp_ms"] / K, "filters_after_sdp": acc["filter1_ms"] / K,
                "refine": acc["refine_ms"] / K, "filters": acc["filter2_ms"] / K,
                # ... and HIP-event times of the device phases inside the two big ones
                "sdp_keys_sort": acc["keys_ms"] / K, "sdp_match_order": acc["match_ms"] / K,
                "sdp_chain": acc["chain_ms"] / K, "sdp_blocks_fills_assembly": acc["blocks_ms"] / K,
                "refine_geometry": acc["geometry_ms"] / K, "refine_fill": acc["fill_ms"] / K,
                "refine_traceback_stats": acc["traceback_ms"] / K, "refine_compact": acc["compact_ms"] / K},
            "sdp_counters_per_step": {"sdpalign_invocations": acc["invocations"] // K,
                                      "fragments": acc["fragments"] // K, "swalign_fills": acc["sw_fills"] // K},
            # the dominant kernel of the step = the kernel with the largest HIP-event time, whatever stage it is in (filled in
            # below from blasr_hip_refine_last_kernels and blasr_hip_sdp_last_kernels: every heavy kernel timed on its own, on
            # the stream it ran on, with the algorithmic bytes of exactly what it processed); all of them in kernel_rooflines
            "roofline": None, "kernel_rooflines": None,
            # the SDP phases against the same roof: algorithmic bytes by SURVEY.md 8(d)'s model over HIP-event time
            "sdp_rooflines": {
                name: {"bound": "hbm", "achieved": gbps(acc[b], acc[m]), "peak": 8000.0, "unit": "GB/s",
                       "frac": gbps(acc[b], acc[m]) / 8000.0, "algorithmic_bytes_per_step": acc[b] / K,
                       "ms_per_step": acc[m] / K}
                for name, b, m in (("keys_sort", "keys_bytes", "keys_ms"), ("match_order", "match_bytes", "match_ms"),
                                   ("chain", "chain_bytes", "chain_ms"), ("blocks_fills_assembly", "blocks_bytes", "blocks_ms"))},
            "device_memory_held_gb": ctx.held_bytes() / 2**30,
            # every rank's own figures (the step's wall time is the slowest rank's)
            "per_rank_ms_per_step": [dict(zip(("step", "windows", "sdp", "filters_after_sdp", "refine", "filters",
                                               "candidates_refined"), r)) for r in rank_ms],
            "host_threads_per_rank": int(os.environ.get("BLASR_HIP_HOST_THREADS", "0")) or None,
        }
        # how the ranks' counters were reduced: RCCL (one process per GPU, `nccl` backend) or gloo (--share-gpus)
        out["rccl_ranks"] = world if (world > 1 and not shared) else 0
        out["counters_backend"] = None if world == 1 else ("gloo" if shared else "nccl")
        if shared:
            out["physical_gpus"] = n_dev
            out["note"] = ("%d ranks SHARE %d GPU%s (--share-gpus, counters over gloo): a run of the multi-rank path, "
                           "not a scaling measurement" % (world, n_dev, "" if n_dev == 1 else "s"))
        build = hip.build_id()
        out["build_id"] = build
        pmc_path, pmc = pmc_for(args.config, n_reads, d["read_len"], args.kband if args.no_guided else args.band, build)
        cyc = VALU_CYCLES["slow"] * VALU_CYCLES["slow_share"] + VALU_CYCLES["fast"] * (1 - VALU_CYCLES["slow_share"])
        lines = []
        for name, a in kacc.items():
            if a["ms"] <= 0 or a["launches"] <= 0:
                continue
            per_launch_ms = a["ms"] / a["launches"]
            ach = gbps(a["bytes"], a["ms"])
            rl = {"bound": "hbm", "kernel": name, "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                  "traffic": None, "algorithmic_bytes_per_launch": a["bytes"] / a["launches"],
                  "launches_per_step": a["launches"] / K, "avg_launch_ms": per_launch_ms,
                  "candidates_per_launch": a["candidates"] / a["launches"],
                  "kernel_gcups": a["cells"] / (a["ms"] / 1e3) / 1e9}
            p = pmc.get(name)
            if p:  # same workload, same build: HBM bytes and VALU instructions per launch from the counters
                if p.get("hbm_bytes_per_dispatch") is not None:
                    rl["traffic"] = p["hbm_bytes_per_dispatch"]
                    rl["traffic_over_algorithmic"] = p["hbm_bytes_per_dispatch"] / max(rl["algorithmic_bytes_per_launch"], 1)
                if p.get("SQ_INSTS_VALU"):
                    insts = p["SQ_INSTS_VALU"] / max(p.get("dispatches", 1), 1)
                    # the roof that binds the fills: wave64 VALU instructions x their measured issue cost over the SIMD
                    # cycles of the launch (1024 SIMDs x 2.4 GHz x the HIP-event time of this run)
                    rl["valu_issue_frac"] = insts * cyc / (SIMDS * CLOCK_HZ * per_launch_ms / 1e3)
                    rl["valu_issue_frac_range"] = [insts * VALU_CYCLES["fast"] / (SIMDS * CLOCK_HZ * per_launch_ms / 1e3),
                                                   insts * VALU_CYCLES["slow"] / (SIMDS * CLOCK_HZ * per_launch_ms / 1e3)]
                    rl["valu_wave_instructions_per_launch"] = insts
                rl["pmc_profile"] = pmc_path
            if name == "blasr_hip::kband_wide_kernel" and "valu_issue_frac" not in rl:
                # no counters of this build at hand: the instruction count from the kernel's ISA instead (a
                # step of four cells per lane is 86 wave64 instructions in the unmasked loop, 106 in the masked one, which
                # ~15 % of the steps of a k = 1.3-2.7 k band take), each lane-step covering 4 x 64 cells
                insts = a["cells"] / a["launches"] / 256.0 * (0.85 * 86 + 0.15 * 106)
                rl["valu_issue_frac"] = insts * cyc / (SIMDS * CLOCK_HZ * per_launch_ms / 1e3)
                rl["valu_issue_frac_source"] = "instruction count of the ISA x cells (no PMC pass of this build)"
                rl["valu_wave_instructions_per_launch"] = insts
            # `bound` stays the contract's ("hbm" | "mfma": this integer path has no MFMA work, its roof of the two is HBM) and the
            # figures stay the HBM ones; what actually limits the kernel goes beside it: vector issue where that fraction (PMC
            # of this build, or -- said so -- a count of the ISA) exceeds the HBM fraction
            if rl.get("valu_issue_frac", 0.0) > rl["frac"]:
                rl["limiter"] = "valu issue (modelled)" if "valu_issue_frac_source" in rl else "valu issue"
            rl["stage"] = "refine"
            if name == "blasr_hip::kband_wide_kernel" and a["launches"] > K:
                # the chunks of a k-band batch run as two lanes on two streams: the launches overlap, so a
                # launch's own time is that of a shared device; the stage-level figure is the one to read
                rl["concurrent_launches"] = 2
                if acc["fill_ms"] > 0:
                    rl["stage_achieved"] = gbps(a["bytes"], acc["fill_ms"])
                    rl["stage_frac"] = rl["stage_achieved"] / 8000.0
                    rl["stage_gcups"] = a["cells"] / (acc["fill_ms"] / 1e3) / 1e9
            lines.append(rl)
        for name, a in sacc.items():  # the SDP stage's kernels: 8(d)'s byte model over their own HIP-event time
            if a["ms"] <= 0 or a["launches"] <= 0:
                continue
            ach = gbps(a["bytes"], a["ms"])
            rl = {"bound": "hbm", "kernel": name, "stage": "sdp", "achieved": ach, "peak": 8000.0, "unit": "GB/s",
                  "frac": ach / 8000.0, "traffic": None, "algorithmic_bytes_per_launch": a["bytes"] / a["launches"],
                  "launches_per_step": a["launches"] / K, "avg_launch_ms": a["ms"] / a["launches"],
                  "fragments_per_step": a["fragments"] / K, "invocations_per_step": a["invocations"] / K}
            if a["concurrent"] > 1:
                rl["concurrent_launches"] = a["concurrent"]
            # the counters of the kernels the record covers (a PMC profile is one step: tools/pmc_config.sh), per bracket
            parts = [x if x.startswith("blasr_hip::") else "blasr_hip::" + x for x in name.split(" + ")]
            mine = [v for kn, v in pmc.items() if any(kn == x or kn.startswith(x + "<") or kn.startswith(x.replace("_kernel", "_copy_kernel"))
                                                      for x in parts)]
            if mine:
                if all(v.get("hbm_bytes") is not None for v in mine):
                    rl["traffic"] = sum(v["hbm_bytes"] for v in mine) / max(rl["launches_per_step"], 1)
                    rl["traffic_over_algorithmic"] = rl["traffic"] / max(rl["algorithmic_bytes_per_launch"], 1)
                valu, salu = sum(v.get("SQ_INSTS_VALU", 0) for v in mine), sum(v.get("SQ_INSTS_SALU", 0) for v in mine)
                if valu > 0:
                    rl["salu_over_valu"] = salu / valu
                    if a["fragments"] > 0:
                        rl["wave_instructions_per_fragment"] = (valu + salu) / (a["fragments"] / K)
                rl["pmc_profile"] = pmc_path
            # a serial chain per invocation: neither an HBM nor a VALU-issue roof binds it
            if "chain" in name or "small" in name:
                rl["bound_note"] = "latency of a serial dependency chain per invocation; achieved / peak / frac are the HBM figures"
            lines.append(rl)
        step_ms = elapsed / K * 1e3
        for rl in lines:
            rl["ms_per_step"] = rl["avg_launch_ms"] * rl["launches_per_step"]
            rl["share_of_step"] = rl["ms_per_step"] / step_ms
        lines.sort(key=lambda r: -r["ms_per_step"])
        if lines:
            out["roofline"] = dict(lines[0])
            if out["roofline"].get("limiter", "").startswith("valu issue"):
                out["roofline"]["limiter_note"] = ("integer min / compare / DPP forms issue at 4.1-4.4 cycles per wave64 "
                                                   "instruction (profiles/r02_valu_rate.txt): valu_issue_frac is what binds; "
                                                   "bound / achieved / peak / frac are the HBM figures the contract asks for")
            out["kernel_rooflines"] = lines
            refine_lines = [r for r in lines if r["stage"] == "refine"]
            if refine_lines and refine_lines[0]["kernel"] != lines[0]["kernel"]:
                out["roofline_of_the_largest_refine_kernel"] = dict(refine_lines[0])
            if refine_lines:
                out["fill_stage_gcups"] = acc["cells"] / (acc["fill_ms"] / 1e3) / 1e9 if acc["fill_ms"] > 0 else 0.0
        if world == 1:
            cores = args.cpu_cores if args.cpu_cores > 0 else (os.cpu_count() or 1)
            # ~10-30 s of host work on `cores` threads: 4.95 ms per 10 kb candidate through SDP + guided refine on one
            # core, 455 ms per 30 kb one; the k-band refiner takes 0.46 s (10 kb) to 4.4 s (30 kb) per read
            if args.no_guided and args.cpu_cores <= 0:
                # the reference's AffineKBandAlign keeps six 4-byte arrays of (q + 1)(2k + 1) cells per thread (3.9 GB for
                # a 30 kb read at k = 2.7 k): no more threads than fit 64 GB of host memory
                k_est = 0.09 * d["read_len"] if args.kband == 0 else 1.2 * args.kband
                cores = max(1, min(cores, int(64e9 / (24.0 * d["read_len"] * (2 * k_est + 1)))))
            per_read_s = (0.005 * (d["read_len"] / 1e4) ** 2.5 * max(1.0, n_iv / n_reads) +
                          (0.46 * (d["read_len"] / 1e4) ** 2 if args.no_guided else 0.0))
            ns = args.cpu_sample if args.cpu_sample >= 0 else max(8, min(2400, int(20.0 * cores / per_read_s)))
            ns = min(ns, n_reads)
            if ns > 0:
                out["cpu_baseline"] = cpu_baseline(wl, genome, starts, ns, args.kband if args.no_guided else args.band, hits,
                                                   hit_off, cores, args.extend, guided=not args.no_guided)
            # >= 10 s of signal: 32 threads of the reference binary map ~7 M bases/s of these reads
            nb = args.blasrmc_reads if args.blasrmc_reads >= 0 else (0 if args.no_guided else
                                                                     min(n_reads, max(8, 80000000 // d["read_len"])))
            if nb > 0:
                ncores = os.cpu_count() or 1
                b = None
                if args.from_reads > 0:  # the genome's index is resident: the reference binary on the genome as stated
                    b = blasrmc_true_genome(ctx, wl, genome, starts, nb, ncores, args.extend)
                    if b is not None and "value" not in b:
                        out["cpu_baseline_whole_blasr_on_the_whole_genome"] = b  # why not
                        b = None
                if b is None:  # a FASTA of the reads' own loci instead (the index of 3 Gbases takes 18 GB of disk and memory)
                    b = blasrmc_baseline(wl, genome, nb, d["read_len"], ncores, args.extend, device_index)
                if b is not None:
                    out["cpu_baseline_whole_blasr"] = b
            if args.from_reads > 0:
                torch.cuda.empty_cache()
                out["end_to_end"] = end_to_end(ctx, genome, starts, wl, params, args.from_reads, dev)
                if "cpu_baseline_whole_blasr" in out and out["cpu_baseline_whole_blasr"].get("reads_per_s"):
                    out["end_to_end"]["vs_whole_blasr"] = (out["end_to_end"]["reads_per_s"] /
                                                           out["cpu_baseline_whole_blasr"]["reads_per_s"])
        if args.other_configs > 0 and world == 1:
            # the other single-GPU configurations on the same line: each as a child process of its own (this process's
            # context and tensors are released first: configs[4] wants the workspace)
            ctx.close()
            ctx = None
            del genome, wl, reads_t, iv_t
            torch.cuda.empty_cache()
            out["other_configs"] = other_configs(args.other_configs)
        emit(out, args.detail_out)
    if dist is not None:
        dist.destroy_process_group()
    if ctx is not None:
        ctx.close()


if __name__ == "__main__":
    main()
