"""End to end on NOISY reads with competing candidates: PacBio-like reads against a two-contig genome with planted
repeats.  tests/golden/make_m4_noisy_golden.py ran the reference BINARY (oracle/_ref/blasrmc -m 4 -bestn 1 -V 1) and
kept, besides the -m 4 lines, the candidate intervals it derived from its anchors.  Given those intervals -- anchoring
(SURVEY next-1) is the one stage that is not built -- everything else of MapRead is done by blasr_amd.map_reads
(blasr_amd/mapread.py), in the reference's
order (Blasr.cpp:2530-2644, 3509-3516, 3086): windows -> SDPAlign + statistics per interval -> sort by score ->
RemoveLowQualityAlignments -> AffineGuidedAlign + statistics -> sort -> RemoveLowQualityAlignments ->
RemoveOverlappingAlignments -> AssignRefContigLocation -> StoreMapQVs -> the -m 4 line of the best one.
The lines must equal the reference's byte for byte, and so must the SAM records of the same alignments (-sam:
CreateCIGARString + SAMOutput::PrintAlignment)."""
import os
import subprocess

import numpy as np
import pytest

import blasr_amd
from blasr_amd import hip

pytestmark = pytest.mark.gpu

SAWRITER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "sawritermc")

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "m4_noisy_golden.npz")


def test_m4_lines_of_noisy_reads_with_the_reference_intervals():
    run_end_to_end(own_intervals=False)


@pytest.mark.skipif(not os.path.exists(SAWRITER), reason="oracle/_ref/sawritermc not built (the suffix array's writer)")
def test_m4_lines_of_noisy_reads_from_the_reads_alone(tmp_path):
    """No reference intervals: the library anchors the reads itself (suffix-array search on the device, interval finding
    on the host) and must arrive at the reference binary's 125 intervals, -m 4 lines and SAM records."""
    run_end_to_end(own_intervals=True, tmp_path=tmp_path)


def test_m4_lines_of_noisy_reads_under_extend():
    """-extend (Blasr.cpp:1749-1940): ExtendAlignmentForward / Reverse on both ends of every SDP alignment before the
    filters and the refinement, inside blasr_hip_map_reads (extend_alignments = 1).  The reference binary's lines under
    -extend (tests/golden/make_m4_noisy_extend_golden.py; 54 of the 80 differ from the default run's), byte for byte."""
    g = np.load(GOLDEN)
    want = bytes(np.load(os.path.join(os.path.dirname(GOLDEN), "m4_noisy_extend_golden.npz"))["lines"]).decode().splitlines()
    plain = bytes(g["lines"]).decode().splitlines()
    contigs = [g["contig0"], g["contig1"]]
    read_lens, pool, intervals, owner = g["read_lens"], g["reads"], g["intervals"], g["owner"]
    genome = np.concatenate([contigs[0], [ord("N")], contigs[1], [ord("N")]]).astype(np.uint8)
    starts = np.array([0, len(contigs[0]) + 1, len(contigs[0]) + 1 + len(contigs[1]) + 1], dtype=np.uint32)
    ctx = blasr_amd.Context(0)
    ctx.upload_genome(genome, starts)
    reads = np.zeros(len(read_lens), dtype=blasr_amd.READ_DTYPE)
    off = 0
    for i in range(len(read_lens)):
        reads[i] = (off, int(read_lens[i]), 0, int(read_lens[i]), 0)
        off += int(read_lens[i])
    ivs = np.zeros(len(intervals), dtype=blasr_amd.INTERVAL_DTYPE)
    for k in range(len(intervals)):
        s, e, qs, qe, strand = (int(v) for v in intervals[k])
        ivs[k] = (s, e, qs, qe, int(owner[k]), strand)
    got = []
    params = ctx.map_params(extend_alignments=1)
    for r, hits in enumerate(blasr_amd.map_reads(ctx, len(genome), starts, pool, reads, ivs, bestn=1, params=params)):
        name = "read%d/%d/0_%d/0_%d" % (r, r, int(read_lens[r]), int(read_lens[r]))
        got += [blasr_amd.m4_record(h, name, "contig%d" % h["contig"]).rstrip("\n") for h in hits]
    bad = [(a, b) for a, b in zip(got, want) if a != b]
    assert len(got) == len(want) and not bad, (len(got), len(want), len(bad), bad[:3])
    assert sum(a != b for a, b in zip(got, plain)) > 40  # and the extension did change the alignments
    ctx.close()


def run_end_to_end(own_intervals, tmp_path=None):
    g = np.load(GOLDEN)
    contigs = [g["contig0"], g["contig1"]]
    want = bytes(g["lines"]).decode().splitlines()
    want_sam = bytes(g["sam_lines"]).decode().splitlines()
    read_group = [x for x in bytes(g["sam_header"]).decode().splitlines() if x.startswith("@RG")][0].split("ID:")[1].split("\t")[0]
    read_lens, pool, intervals, owner = g["read_lens"], g["reads"], g["intervals"], g["owner"]
    genome = np.concatenate([contigs[0], [ord("N")], contigs[1], [ord("N")]]).astype(np.uint8)
    starts = np.array([0, len(contigs[0]) + 1, len(contigs[0]) + 1 + len(contigs[1]) + 1], dtype=np.uint32)
    ctx = blasr_amd.Context(0)
    ctx.upload_genome(genome, starts)
    n_reads, n_iv = len(read_lens), len(intervals)
    reads = np.zeros(n_reads, dtype=blasr_amd.READ_DTYPE)
    off = 0
    for i in range(n_reads):
        reads[i] = (off, int(read_lens[i]), 0, int(read_lens[i]), 0)
        off += int(read_lens[i])
    ivs = np.zeros(n_iv, dtype=blasr_amd.INTERVAL_DTYPE)
    for k in range(n_iv):
        s, e, qs, qe, strand = (int(v) for v in intervals[k])
        ivs[k] = (s, e, qs, qe, int(owner[k]), strand)
    if own_intervals:
        # the suffix array blasrmc builds on the fly (Blasr.cpp:4407-4420: lookup prefix 8) = sawriter -blt 8
        fa = os.path.join(str(tmp_path), "genome.fasta")
        with open(fa, "wb") as f:
            f.write(b">contig0\n" + bytes(contigs[0]) + b"\n>contig1\n" + bytes(contigs[1]) + b"\n")
        subprocess.run([SAWRITER, fa + ".sa", fa, "-blt", "8"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        ctx.upload_suffix_array(hip.read_suffix_array(fa + ".sa"))
        mine, _, _ = blasr_amd.candidate_intervals(ctx, genome, starts, hip.count_tuples(genome, 8), pool, reads)
        assert np.array_equal(mine, ivs)
        ivs = mine
    # -m 4: bestn 1, gap lists as the aligner left them
    got = []
    for r, hits in enumerate(blasr_amd.map_reads(ctx, len(genome), starts, pool, reads, ivs, bestn=1)):
        name = "read%d/%d/0_%d/0_%d" % (r, r, int(read_lens[r]), int(read_lens[r]))
        got += [blasr_amd.m4_record(h, name, "contig%d" % h["contig"]).rstrip("\n") for h in hits]
    # -sam switches forPicard on (MappingParameters.h:488-491): OrderGapsByType runs before StoreMapQVs and the CIGAR
    got_sam = []
    for r, hits in enumerate(blasr_amd.map_reads(ctx, len(genome), starts, pool, reads, ivs, for_picard=True, bestn=1)):
        L = int(read_lens[r])
        name = "read%d/%d/0_%d/0_%d" % (r, r, L, L)
        read_seq = pool[int(reads[r]["off"]): int(reads[r]["off"]) + L]
        got_sam += [blasr_amd.sam_record(h, name, "contig%d" % h["contig"], read_group, read_seq).rstrip("\n") for h in hits]
    bad = [(a, b) for a, b in zip(got, want) if a != b]
    assert len(got) == len(want) and not bad, (len(got), len(want), bad[:3])
    bad = [(a[:300], b[:300]) for a, b in zip(got_sam, want_sam) if a != b]
    assert len(got_sam) == len(want_sam) and not bad, (len(bad), bad[:2])


@pytest.mark.skipif(not os.path.exists(SAWRITER), reason="oracle/_ref/sawritermc not built (the suffix array's writer)")
@pytest.mark.parametrize("reads_per_block,extend,overlap", [(7, 0, 0), (1000, 0, 0), (13, 1, 0), (7, 0, 1), (13, 1, 1), (79, 0, 1)])
def test_the_one_call_mapper_finds_the_same_intervals_and_hits(tmp_path, reads_per_block, extend, overlap):
    """blasr_hip_map_from_reads_device (anchoring in sub-blocks, then the path, in C++): the reference binary's 125
    candidate intervals and the hits of the step-by-step route, for sub-blocks of 7 reads and for one sub-block; with
    `overlap` the path runs sub-block by sub-block beside the anchoring of the next one on the helper context -- same
    hits, same order, and the fetched sequences and op lists of every sub-block where the shifted offsets say."""
    from blasr_testlib import DeviceMemory
    g = np.load(GOLDEN)
    contigs = [g["contig0"], g["contig1"]]
    genome = np.concatenate([contigs[0], [ord("N")], contigs[1], [ord("N")]]).astype(np.uint8)
    starts = np.array([0, len(contigs[0]) + 1, len(genome)], dtype=np.uint32)
    read_lens, pool = g["read_lens"], np.ascontiguousarray(g["reads"])
    reads = np.zeros(len(read_lens), dtype=blasr_amd.READ_DTYPE)
    off = 0
    for i in range(len(read_lens)):
        reads[i] = (off, int(read_lens[i]), 0, int(read_lens[i]), 0)
        off += int(read_lens[i])
    want_iv = np.zeros(len(g["intervals"]), dtype=blasr_amd.INTERVAL_DTYPE)
    for k, (s, e, qs, qe, strand) in enumerate(g["intervals"]):
        want_iv[k] = (int(s), int(e), int(qs), int(qe), int(g["owner"][k]), int(strand))
    ctx = blasr_amd.Context(0)
    ctx.upload_genome(genome, starts)
    ctx.build_suffix_array(8, download=False)  # what blasrmc builds on the fly (lookup prefix 8)
    mem = DeviceMemory()
    d_reads, d_pool = mem.upload(reads.view(np.uint8)), mem.upload(pool)
    hits = np.zeros(len(want_iv) + 8, dtype=hip.HIT_DTYPE)
    hit_off = np.zeros(len(reads) + 1, dtype=np.int64)
    p = ctx.pipeline_params(reads_per_block=reads_per_block, overlap=overlap)
    p.map.extend_alignments = extend  # -extend travels with the path's parameters
    n, ivs, seq, ops = ctx.map_from_reads_device(genome, starts, ctx.count_tuples(8), reads, d_reads, d_pool, p, hits, hit_off,
                                                 intervals_capacity=len(want_iv) + 8, fetch_capacity=4 << 20)
    assert np.array_equal(ivs, want_iv)
    t = ctx.pipeline_last_timing()
    assert t["intervals"] == len(want_iv) and t["anchors"] > 10000
    want_hits, want_off, want_seq, want_ops = ctx.map_reads(reads, pool, want_iv, params=ctx.map_params(extend_alignments=extend))
    assert n == len(want_hits) and np.array_equal(hit_off, want_off)
    for f in ("read", "interval", "t_strand", "t_aligned_seq_pos", "q_pos", "t_pos", "score", "n_match", "n_ops"):
        assert np.array_equal(hits[:n][f], want_hits[f]), f
    assert len(seq) == len(want_seq) and len(ops) == len(want_ops)
    for a, b in zip(hits[:n], want_hits):  # (the arenas may be laid out differently: compare what the offsets point at)
        assert np.array_equal(ops[a["ops_off"]: a["ops_off"] + a["n_ops"]], want_ops[b["ops_off"]: b["ops_off"] + b["n_ops"]])
        for f, ln in (("q_off", "q_aligned_seq_length"), ("t_off", "t_aligned_seq_length")):
            assert np.array_equal(seq[a[f]: a[f] + a[ln]], want_seq[b[f]: b[f] + b[ln]]), f
    if overlap:  # too little room is reported with what the WHOLE call needs
        small = np.zeros(3, dtype=hip.HIT_DTYPE)
        with pytest.raises(hip.BlasrHipError) as e:
            ctx.map_from_reads_device(genome, starts, ctx.count_tuples(8), reads, d_reads, d_pool, p, small, hit_off)
        assert e.value.code == hip.ERR_CAPACITY and "%d hits" % n in str(e.value)
    mem.free()
    ctx.close()
