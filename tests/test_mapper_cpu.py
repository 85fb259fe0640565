"""a-1 (window set-up of AlignIntervals) and a-10 (candidate ordering / pruning), CPU side.

* the oracle restatements against the golden vectors the reference mapper produced
  (tests/golden/mapper_golden.npz <- tests/golden/make_mapper_golden.py) and, where the reference
  build is present, against the reference itself on fresh random cases;
* the product's host filters (blasr_hip_sort_candidates / remove_low_quality / remove_overlapping,
  plain C++ inside libblasr_hip.so, no device) against the same vectors.
"""
import os

import numpy as np
import pytest

import blasr_amd
from blasr_testlib import (WINDOW_FIELDS, MapperOracle, load_refmapper, random_candidates, random_interval_case)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mg():
    return np.load(os.path.join(ROOT, "tests", "golden", "mapper_golden.npz"))


@pytest.fixture(scope="module")
def mo():
    return MapperOracle()


def window_case(mg, i):
    inp = mg["w%d_in" % i]
    contigs = mg["w%d_contigs" % i]
    return mg["w%d_read" % i], (int(inp[0]), int(inp[1])), tuple(int(v) for v in inp[2:7]), (contigs if len(contigs) else None)


def test_oracle_windows_match_golden(mg, mo):
    genome = mg["genome"]
    strands = set()
    for i in range(int(mg["n_windows"])):
        read, subread, iv, contigs = window_case(mg, i)
        win, tseq = mo.window(genome, len(read), subread, iv, contigs)
        assert [win[f] for f in WINDOW_FIELDS] == [int(v) for v in mg["w%d_win" % i]], i
        assert np.array_equal(tseq, mg["w%d_tseq" % i]), i
        strands.add(iv[4])
    assert strands == {0, 1}


def test_oracle_sdp_on_golden_windows(mg, mo, oracle):
    """The SDP alignment the reference's AlignIntervals produced for its own window (the real call site,
    Blasr.cpp:1716-1724) equals the oracle's SDPAlign + ComputeAlignmentStats on the oracle's window."""
    genome = mg["genome"]
    for i in range(0, int(mg["n_windows"]), 3):
        read, subread, iv, contigs = window_case(mg, i)
        win, tseq = mo.window(genome, len(read), subread, iv, contigs)
        q = read[win["qAlignedSeqPos"]: win["qAlignedSeqPos"] + win["qAlignedSeqLength"]]
        r = oracle.sdp_align(q, tseq)
        want = mg["w%d_sdp" % i]
        assert (r["qPos"], r["tPos"]) == (int(want[0]), int(want[1])), i
        assert np.array_equal(r["blocks"], mg["w%d_blocks" % i]), i


def filter_case(mg, i):
    c = {k: mg["f%d_%s" % (i, k)] for k in ("score", "t_pos", "t_aligned_seq_pos", "t_index", "t_strand", "pct",
                                             "blocks", "blk_off", "n_blocks", "t_end")}
    p = mg["f%d_params" % i]
    return c, dict(n_candidates=int(p[0]), max_gap=int(p[1]), max_score=int(p[2]), min_pct=float(p[3]))


def run_oracle_filter(mo, op, c, p):
    order = np.arange(len(c["score"]), dtype=np.int32)
    if op == 0:
        order = mo.sort_by_score(c["score"], c["t_pos"], c["t_aligned_seq_pos"], order)
        return order, c["score"][order]
    if op == 1:
        order, score = mo.remove_low_quality(order, c["score"], c["blocks"], c["blk_off"], c["n_blocks"],
                                             p["n_candidates"], p["max_gap"], p["max_score"])
        return order, score[order]
    order = mo.remove_overlapping(order, c["score"], c["t_pos"], c["t_aligned_seq_pos"], c["t_end"], c["t_index"],
                                  c["t_strand"], c["pct"], p["min_pct"])
    return order, c["score"][order]


def as_candidates(c):
    cands = np.zeros(len(c["score"]), dtype=blasr_amd.CANDIDATE_DTYPE)
    for k_src, k_dst in (("score", "score"), ("t_pos", "t_pos"), ("t_aligned_seq_pos", "t_aligned_seq_pos"),
                         ("t_end", "t_end"), ("t_index", "t_index"), ("t_strand", "t_strand"),
                         ("pct", "pct_similarity"), ("n_blocks", "n_blocks"), ("blk_off", "blk_off")):
        cands[k_dst] = c[k_src]
    return cands


def run_product_filter(op, c, p):
    cands = as_candidates(c)
    order = np.arange(len(cands), dtype=np.int32)
    if op == 0:
        order = blasr_amd.sort_candidates(cands, order)
        return order, cands["score"][order]
    if op == 1:
        order, cands = blasr_amd.remove_low_quality(cands, c["blocks"].reshape(-1), order, p["n_candidates"],
                                                    p["max_gap"], p["max_score"])
        return order, cands["score"][order]
    order = blasr_amd.remove_overlapping(cands, order, p["min_pct"])
    return order, cands["score"][order]


@pytest.mark.parametrize("runner", ["oracle", "product"])
def test_filters_match_golden(mg, mo, runner):
    seen_removed = 0
    for i in range(int(mg["n_filter"])):
        c, p = filter_case(mg, i)
        for op in range(3):
            order, score = run_oracle_filter(mo, op, c, p) if runner == "oracle" else run_product_filter(op, c, p)
            assert np.array_equal(order, mg["f%d_op%d_kept" % (i, op)]), (i, op)
            assert np.array_equal(score, mg["f%d_op%d_score" % (i, op)]), (i, op)
            seen_removed += len(order) < len(c["score"])
    assert seen_removed > 10


def test_filter_chain_as_mapread_runs_it(mg, mo):
    """sort -> RemoveLowQuality -> (refine) -> sort -> RemoveLowQuality -> RemoveOverlapping
    (Blasr.cpp:2547, :2594-2596, :2615-2619): the product against the oracle on the chained state."""
    for i in range(int(mg["n_filter"])):
        c, p = filter_case(mg, i)
        cands = as_candidates(c)
        po = blasr_amd.sort_candidates(cands)
        oo = mo.sort_by_score(c["score"], c["t_pos"], c["t_aligned_seq_pos"])
        assert np.array_equal(po, oo)
        po, cands = blasr_amd.remove_low_quality(cands, c["blocks"].reshape(-1), po, p["n_candidates"], p["max_gap"],
                                                 p["max_score"])
        oo, oscore = mo.remove_low_quality(oo, c["score"], c["blocks"], c["blk_off"], c["n_blocks"], p["n_candidates"],
                                           p["max_gap"], p["max_score"])
        assert np.array_equal(po, oo) and np.array_equal(cands["score"], oscore)
        po = blasr_amd.remove_overlapping(cands, po, p["min_pct"])
        oo = mo.remove_overlapping(oo, oscore, c["t_pos"], c["t_aligned_seq_pos"], c["t_end"], c["t_index"],
                                   c["t_strand"], c["pct"], p["min_pct"])
        assert np.array_equal(po, oo)


def test_oracle_vs_reference_mapper_live(mo):
    ref = load_refmapper()
    if ref is None:
        pytest.skip("oracle/_ref/libblasr_refmapper.so not built (needs /root/reference)")
    from blasr_amd import synth
    rng = np.random.default_rng(77)
    genome = synth.random_genome(30000, rng)
    for it in range(24):
        read, subread, iv, contigs = random_interval_case(rng, genome, it)
        rwin, rt, _ = ref.align_interval(genome, read, subread, iv, contigs)
        owin, ot = mo.window(genome, len(read), subread, iv, contigs)
        assert rwin == owin and np.array_equal(rt, ot), it
    for it in range(60):
        c = random_candidates(rng, int(rng.choice([2, 7, 10, 18, 33, 64])))
        p = dict(n_candidates=10, max_gap=int(rng.choice([0, 25])), max_score=-200, min_pct=float(rng.choice([0, 70])))
        for op in range(3):
            kept, score = ref.post_filter(op, c["score"], c["t_pos"], c["t_aligned_seq_pos"], c["t_index"],
                                          c["t_strand"], c["pct"], c["blocks"], c["blk_off"], c["n_blocks"],
                                          p["n_candidates"], p["max_gap"], p["max_score"], p["min_pct"])
            order, oscore = run_oracle_filter(mo, op, c, p)
            assert np.array_equal(order, kept) and np.array_equal(oscore, score), (it, op)


def test_extend_step_composition_against_the_reference_align_intervals(mo):
    """-extend: the checker's composition of the step (blasr_testlib.oracle_extend_candidate: the oracle's ExtendAlignment
    + index arithmetic) against the reference's OWN AlignIntervals with params.extendAlignments set, one interval at a
    time: window position and strings, qPos / tPos, every block, the statistics.  Both strands, contigs, windows at the
    genome / contig ends, several band / drop-off settings."""
    ref = load_refmapper()
    if ref is None:
        pytest.skip("oracle/_ref/libblasr_refmapper.so not built (needs /root/reference)")
    from blasr_amd import synth
    from blasr_testlib import ScoreParams, load_oracle, oracle_extend_candidate
    oracle = load_oracle()
    rng = np.random.default_rng(91)
    genome = synth.random_genome(40000, rng)
    sc = ScoreParams()
    changed = strands = 0
    for it in range(60):
        read, subread, iv, contigs = random_interval_case(rng, genome, it)
        band, drops = [(10, 10), (4, 3), (15, 25)][it % 3]
        rwin, rt, rs = ref.align_interval(genome, read, subread, iv, contigs, extend=(band, drops))
        w, tseq = mo.window(genome, len(read), subread, iv, contigs)
        q = read[w["qAlignedSeqPos"]: w["qAlignedSeqPos"] + w["qAlignedSeqLength"]]
        s0 = oracle.sdp_align(q, tseq, sc)
        w2, q2, t2, s2 = oracle_extend_candidate(oracle, np.asarray(genome), contigs, np.asarray(read), w, q, tseq, s0, sc, band, drops)
        st = oracle.compute_alignment_stats(q2, t2, s2["qPos"], s2["tPos"], s2["blocks"], sc=sc)
        for key in ("tAlignedSeqPos", "tStrand", "tIndex", "tLength", "qAlignedSeqPos"):
            assert w2[key] == rwin[key], (it, key, w2[key], rwin[key])
        assert (s2["qPos"], s2["tPos"]) == (rs["qPos"], rs["tPos"]), it
        assert np.array_equal(np.asarray(s2["blocks"]).reshape(-1, 3), rs["blocks"]), it
        if len(rs["blocks"]):
            assert np.array_equal(t2, rt), it  # the aligned target string, prefix and suffix pieces included
            assert (st["score"], st["nMatch"], st["nMismatch"], st["nIns"], st["nDel"]) == (
                rs["score"], rs["nMatch"], rs["nMismatch"], rs["nIns"], rs["nDel"]), it
        changed += len(s2["blocks"]) != len(s0["blocks"])
        strands += int(iv[4])
    assert changed > 20 and 10 < strands < 50
