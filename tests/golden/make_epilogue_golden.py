#!/usr/bin/env python3
"""Golden vectors for the first pieces of the output epilogue, generated HERE with
the reference's own code (oracle/_ref/libblasr_refmapper.so = alignment/Blasr.cpp compiled where it lies):
AssignRefContigLocation, SAMOutput::CreateCIGARString (op-list and block form, all four clipping modes, both
strands), IntervalAlignmentPrinter::Print, StoreMapQVs, SAMOutput::PrintAlignment.  Run from the repo root:  python tests/golden/make_epilogue_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import blasr_testlib as T  # noqa: E402


def contig_cases(rng, n):
    out = []
    for it in range(n):
        ncontig = int(rng.integers(1, 6))
        lens = rng.integers(200, 2000, ncontig)
        starts = [0]
        for L in lens:
            starts.append(starts[-1] + int(L) + 1)  # an N separates contigs
        glen = starts[-1]
        strand = int(rng.integers(0, 2))
        ci = int(rng.integers(0, ncontig))
        wl = int(rng.integers(1, lens[ci]))
        fstart = int(starts[ci]) + int(rng.integers(0, lens[ci] - wl + 1))
        pos = fstart if strand == 0 else glen - (fstart + wl - 1) - 1
        out.append((pos, wl, strand, np.array(starts, dtype=np.uint32), glen))
    return out


def m4_cases(rng, n):
    out = []
    for it in range(n):
        nb = int(rng.integers(1, 6))
        blocks, q, t = [], 0, 0
        for b in range(nb):
            ln = int(rng.integers(1, 500))
            blocks.append((q, t, ln))
            q += ln + int(rng.integers(0, 9))
            t += ln + int(rng.integers(0, 9))
        pct = float(np.float32(rng.random() * 100)) if it % 7 else float(np.float32([100.0, 0.0, 99.5, 1e-5, 12345.678][it % 5]))
        out.append(("read/%d/0_%d" % (it, it * 7), "chr%d" % (it % 3), int(rng.integers(-60000, 100)), pct, it % 2,
                    int(rng.integers(0, 1000)), int(rng.integers(0, 100)), int(rng.integers(3000, 60000)),
                    int(rng.integers(0, 2)), int(rng.integers(0, 4000000)), int(rng.integers(0, 100)), 4641652,
                    int(rng.integers(0, 255)), np.array(blocks, dtype=np.int32)))
    return out


def main():
    ref = T.EpilogueChecker("ref")
    rng = np.random.default_rng(20261003)
    g = {}
    cc = contig_cases(rng, 60)
    g["contig_n"] = np.array(len(cc))
    for i, (pos, wl, strand, starts, glen) in enumerate(cc):
        g["contig%d_in" % i] = np.array([pos, wl, strand, glen], dtype=np.int64)
        g["contig%d_starts" % i] = starts
        g["contig%d_out" % i] = np.array(ref.ref_contig_location(pos, wl, strand, starts, glen), dtype=np.int64)
    cg = T.random_epilogue_cases(rng, 96)
    g["cigar_n"] = np.array(len(cg))
    for i, c in enumerate(cg):
        kw = dict(clipping=c["clipping"], t_strand=c["t_strand"], read_length=c["read_length"], lqp=c["lqp"], lqs=c["lqs"],
                  subread_start=c["subread_start"], subread_end=c["subread_end"], q_align_start=c["q_align_start"],
                  q_align_end=c["q_align_end"])
        g["cigar%d_q" % i], g["cigar%d_t" % i] = c["q"], c["t"]
        g["cigar%d_ops" % i], g["cigar%d_blocks" % i] = c["ops"], c["blocks"]
        g["cigar%d_par" % i] = np.array([c["q_pos"], c["t_pos"], c["clipping"], c["t_strand"], c["read_length"], c["lqp"],
                                         c["lqs"], c["subread_start"], c["subread_end"], c["q_align_start"],
                                         c["q_align_end"]], dtype=np.int64)
        s1, k1 = ref.cigar(c["q"], c["t"], c["q_pos"], c["t_pos"], ops=c["ops"], **kw)
        s2, k2 = ref.cigar(c["q"], c["t"], c["q_pos"], c["t_pos"], blocks=c["blocks"], **kw)
        g["cigar%d_from_ops" % i] = np.frombuffer(s1.encode(), dtype=np.uint8)
        g["cigar%d_from_blocks" % i] = np.frombuffer(s2.encode(), dtype=np.uint8)
        g["cigar%d_clips" % i] = np.array(list(k1) + list(k2), dtype=np.int64)
    mc = m4_cases(rng, 60)
    g["m4_n"] = np.array(len(mc))
    for i, c in enumerate(mc):
        line, derived = ref.m4_line(*c[:13], c[13])
        g["m4_%d_names" % i] = np.frombuffer((c[0] + "\n" + c[1]).encode(), dtype=np.uint8)
        g["m4_%d_num" % i] = np.array([c[2], c[4], c[5], c[6], c[7], c[8], c[9], c[10], c[11], c[12]], dtype=np.int64)
        g["m4_%d_pct" % i] = np.array([c[3]], dtype=np.float32)
        g["m4_%d_blocks" % i] = c[13]
        g["m4_%d_derived" % i] = np.array(derived, dtype=np.int64)
        g["m4_%d_line" % i] = np.frombuffer(line.encode(), dtype=np.uint8)
    # StoreMapQVs: the alignments of 80 reads
    n_reads = 80
    g["mapqv_n"] = np.array(n_reads)
    for i in range(n_reads):
        alns, qv = T.random_mapqv_case(rng, i)
        par = dict(affine_align=i % 2, affine_open=[50, 30, 10][i % 3], affine_extend=[0, 1][i % 2], ins=[5, 4, 8][i % 3],
                   dele=[5, 5, 8][i % 3])
        map_qv, prob = ref.store_map_qvs(alns, qv, **par)
        g["mapqv%d_par" % i] = np.array([par["ins"], par["dele"], par["affine_open"], par["affine_extend"],
                                         par["affine_align"], len(alns)], dtype=np.int64)
        g["mapqv%d_qv" % i] = qv if qv is not None else np.zeros(0, dtype=np.uint8)
        for j, a in enumerate(alns):
            g["mapqv%d_%d_q" % (i, j)], g["mapqv%d_%d_t" % (i, j)], g["mapqv%d_%d_ops" % (i, j)] = a["q"], a["t"], a["ops"]
            g["mapqv%d_%d_num" % (i, j)] = np.array([a["q_pos"], a["t_pos"], a["q_align_start"], a["q_align_end"],
                                                     a["q_strand"], a["q_length"], a["n_match"]], dtype=np.int64)
        g["mapqv%d_out" % i] = np.array(map_qv, dtype=np.int64)
        g["mapqv%d_prob" % i] = np.array(prob, dtype=np.float32)
    # Alignment::OrderGapsByType on the op lists of the CIGAR cases
    for i, c in enumerate(cg):
        g["cigar%d_ordered" % i] = ref.order_gaps_by_type(c["ops"])
    # SAMOutput::PrintAlignment
    n_sam = 64
    g["sam_n"] = np.array(n_sam)
    for i in range(n_sam):
        c = T.random_sam_case(rng, i)
        line, derived = ref.sam_line(c)
        g["sam%d_names" % i] = np.frombuffer("\n".join((c["q_name"], c["t_name"], c["read_group_id"])).encode(), dtype=np.uint8)
        g["sam%d_seq" % i] = c["read_seq"]
        g["sam%d_qual" % i] = c["read_qual"] if c["read_qual"] is not None else np.zeros(0, dtype=np.uint8)
        g["sam%d_blocks" % i] = c["blocks"]
        g["sam%d_int" % i] = np.array([c[k] for k in T.SAM_INT_FIELDS], dtype=np.int64)
        g["sam%d_float" % i] = np.array([c["read_quality"]] + list(c["snr"]), dtype=np.float32)
        g["sam%d_derived" % i] = np.array(derived, dtype=np.int64)
        g["sam%d_line" % i] = np.frombuffer(line.encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "epilogue_golden.npz"), **g)
    print("wrote epilogue_golden.npz:", len(cc), "contig,", len(cg), "cigar,", len(mc), "m4 cases,", n_reads,
          "reads for mapQV,", n_sam, "SAM records")


if __name__ == "__main__":
    main()
