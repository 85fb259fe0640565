"""BASELINE.json configs[0] end to end: the -m 4 lines the reference BINARY printed for error-free 1 kb reads on a
two-contig genome (tests/golden/make_m4_golden.py runs oracle/_ref/blasrmc) must come out of the device path + the
epilogue functions byte for byte: windows (a-1) -> SDP + statistics (a-2..a-5, a-9) -> refine (a-6..a-9) -> mapQV,
contig location, -m 4 line (next-2).  The candidate interval of a read is where the read was cut from (anchoring,
SURVEY next-1, is the one stage of the mapper that is not built)."""
import os

import numpy as np
import pytest

import blasr_amd
from blasr_amd import synth

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "m4_golden.npz")


def test_m4_lines_of_the_reference_binary():
    g = np.load(GOLDEN)
    contigs = [g["contig0"], g["contig1"]]
    read_len = int(g["read_len"])
    want = bytes(g["lines"]).decode().splitlines()
    where = g["where"]
    # the reference concatenates the contigs with an N behind each (SequenceIndexDatabase::seqStartPos)
    genome = np.concatenate([contigs[0], [ord("N")], contigs[1], [ord("N")]]).astype(np.uint8)
    starts = np.array([0, len(contigs[0]) + 1, len(contigs[0]) + 1 + len(contigs[1]) + 1], dtype=np.uint32)
    ctx = blasr_amd.Context(0)
    ctx.upload_genome(genome, starts)
    n = len(where)
    reads = np.zeros(n, dtype=blasr_amd.READ_DTYPE)
    ivs = np.zeros(n, dtype=blasr_amd.INTERVAL_DTYPE)
    pool = []
    for i, (ci, pos, strand) in enumerate(where):
        s = contigs[ci][pos:pos + read_len]
        pool.append(synth.revcomp(s) if strand else s)
        reads[i] = (i * read_len, read_len, 0, read_len, 0)
        ivs[i] = (int(starts[ci]) + pos, int(starts[ci]) + pos + read_len, 0, read_len, i, strand)
    seq, q_off, q_len, t_off, t_len, win = ctx.windows(ivs, reads, np.concatenate(pool))
    pairs = [(seq[q_off[i]: q_off[i] + q_len[i]], seq[t_off[i]: t_off[i] + t_len[i]]) for i in range(n)]
    sdp = ctx.sdp(pairs)
    batch = blasr_amd.RefineBatch()
    for (q, t), s in zip(pairs, sdp):
        last = s["blocks"][-1]
        batch.add(q[s["qPos"]: s["qPos"] + last[0] + last[2]], t[s["tPos"]: s["tPos"] + last[1] + last[2]], s["blocks"])
    refined = ctx.refine(batch)
    for i, ((q, t), s, r) in enumerate(zip(pairs, sdp, refined)):
        ci, pos, strand = (int(v) for v in where[i])
        w = win[i]
        q_span = int(sum(int(x >> 2) for x in r.ops if x & 3 != 2))
        t_span = int(sum(int(x >> 2) for x in r.ops if x & 3 != 1))
        q_pos, t_pos = s["qPos"] + r.qPos, s["tPos"] + r.tPos  # RefineAlignment's copy-back (Blasr.cpp:905-914)
        last = s["blocks"][-1]
        qs = q[s["qPos"]: s["qPos"] + last[0] + last[2]]
        ts = t[s["tPos"]: s["tPos"] + last[1] + last[2]]
        q_start = int(w["q_aligned_seq_pos"]) + q_pos
        aln = dict(q=qs, t=ts, ops=r.ops, q_pos=r.qPos, t_pos=r.tPos, q_align_start=q_start, q_align_end=q_start + q_span,
                   q_strand=0, q_length=read_len, n_match=r.nMatch)
        map_qv = blasr_amd.map_qvs([aln])[0][0]  # -bestn 1: one alignment per read
        contig, contig_pos = blasr_amd.ref_contig_location(int(w["t_aligned_seq_pos"]), int(w["t_aligned_seq_length"]),
                                                           int(w["t_strand"]), starts, len(genome))
        assert contig == ci, i
        line = blasr_amd.m4_line("read%d/%d/0_%d/0_%d" % (i, i, read_len, read_len), "contig%d" % contig, r.score,
                                 float(r.pctSimilarity), 0, q_start, q_start + q_span, read_len, int(w["t_strand"]),
                                 contig_pos + t_pos, contig_pos + t_pos + t_span, int(w["t_length"]), map_qv)
        assert line == want[i] + "\n", (i, line, want[i])
