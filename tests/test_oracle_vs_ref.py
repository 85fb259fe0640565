"""Randomised sweep: oracle/blasr_oracle.c against the reference build
(oracle/_ref/libblasr_ref.so).  Skipped where the reference build is absent."""
import numpy as np
import pytest

from blasr_amd import synth
from blasr_testlib import SMRT_MATRIX, ScoreParams, SdpParams
from golden_cases import REFINE_KEYS, first_difference


def test_default_parameters_sweep(oracle, ref):
    rng = np.random.default_rng(11)
    for it in range(40):
        kind = it % 6
        big = {1: ("ins", int(rng.integers(100, 500))), 2: ("del", int(rng.integers(100, 500)))}.get(kind)
        nrun = int(rng.integers(5, 60)) if kind == 3 else None
        flank = (0, 0) if kind == 4 else (0, 200)
        er = dict(ins=0.0, dele=0.0, sub=0.0) if kind == 5 else dict(ins=0.11, dele=0.04, sub=0.01)
        q, t = synth.make_pair(rng, int(rng.integers(100, 4000)), flank=flank, big_indel=big, n_run=nrun, **er)
        a, b = oracle.sdp_align(q, t), ref.sdp_align(q, t)
        assert first_difference(a, b, ["score", "qPos", "tPos", "blocks"]) is None, it
        if len(b["blocks"]) == 0:
            continue
        bl = b["blocks"]
        qs = q[b["qPos"]: b["qPos"] + bl[-1, 0] + bl[-1, 2]]
        ts = t[b["tPos"]: b["tPos"] + bl[-1, 1] + bl[-1, 2]]
        a2, b2 = oracle.affine_guided_align(qs, ts, bl), ref.affine_guided_align(qs, ts, bl)
        assert first_difference(a2, b2, REFINE_KEYS) is None, it
        s1 = oracle.compute_alignment_stats(q, t, b["qPos"], b["tPos"], bl)
        s2 = ref.compute_alignment_stats(q, t, b["qPos"], b["tPos"], bl)
        assert s1 == s2, it


def test_parameter_variety_sweep(oracle, ref):
    rng = np.random.default_rng(12)
    for it in range(120):
        q, t = synth.make_pair(rng, int(rng.integers(20, 1500)), flank=(0, int(rng.integers(0, 100))),
                               ins=float(rng.uniform(0, 0.15)), dele=float(rng.uniform(0, 0.1)),
                               sub=float(rng.uniform(0, 0.05)))
        m = SMRT_MATRIX.copy()
        if it % 3 == 0:
            m = rng.integers(-6, 8, size=(5, 5)).astype(np.int32)
            m[np.arange(4), np.arange(4)] = -rng.integers(1, 8, size=4)
        sc = ScoreParams(matrix=m, ins=int(rng.integers(1, 9)), dele=int(rng.integers(1, 9)),
                         affine_open=int(rng.integers(0, 60)), affine_extend=int(rng.integers(0, 3)))
        sp = SdpParams(k=int(rng.integers(4, 14)), sdp_ins=int(rng.integers(1, 12)), sdp_del=int(rng.integers(1, 12)),
                       indel_rate=float(rng.uniform(0.1, 1.0)), align_type=int(rng.integers(0, 2)),
                       detailed=int(rng.integers(0, 2)), extend_front=int(rng.integers(0, 2)),
                       prefix=int(rng.choice([0, 20, 50])), recurse=int(rng.integers(0, 4)),
                       recurse_over=int(rng.choice([100, 1000, 10000])), max_per_pos=int(rng.choice([0, 0, 3])))
        a, b = oracle.sdp_align(q, t, sc, sp), ref.sdp_align(q, t, sc, sp)
        assert first_difference(a, b, ["score", "qPos", "tPos", "blocks"]) is None, (it, vars(sp))
        if len(b["blocks"]) == 0:
            continue
        bl = b["blocks"].copy()
        bl[:, 0] += b["qPos"]
        bl[:, 1] += b["tPos"]  # absolute guide: exercises tStart != 0
        band = int(rng.choice([4, 16, 30]))
        a2, b2 = oracle.affine_guided_align(q, t, bl, sc, band), ref.affine_guided_align(q, t, bl, sc, band)
        assert first_difference(a2, b2, REFINE_KEYS) is None, it


def test_words_of_up_to_32_bases(oracle, ref):
    """-sdpTupleSize beyond 16 (a tuple is 64 bits, tuples/DNATuple.h:55-94): the oracle's tuples against the reference's, on
    the cases the device's tests use (blasr_testlib.long_word_cases)."""
    from blasr_testlib import long_word_cases
    some = 0
    for it, (q, t, sp) in enumerate(long_word_cases(np.random.default_rng(31), 90)):
        a, b = oracle.sdp_align(q, t, None, sp), ref.sdp_align(q, t, None, sp)
        assert first_difference(a, b, ["score", "qPos", "tPos", "blocks"]) is None, (it, vars(sp))
        some += len(b["blocks"]) > 0
    assert some >= 60


def test_long_noisy_reads(oracle, ref):
    rng = np.random.default_rng(13)
    for qlen, er in ((10000, dict(ins=0.11, dele=0.04, sub=0.01)), (20000, dict(ins=0.10, dele=0.05, sub=0.01))):
        q, t = synth.make_pair(rng, qlen, flank=(0, 500), **er)
        a, b = oracle.sdp_align(q, t), ref.sdp_align(q, t)
        assert first_difference(a, b, ["score", "qPos", "tPos", "blocks"]) is None
        bl = b["blocks"]
        qs = q[b["qPos"]: b["qPos"] + bl[-1, 0] + bl[-1, 2]]
        ts = t[b["tPos"]: b["tPos"] + bl[-1, 1] + bl[-1, 2]]
        assert first_difference(oracle.affine_guided_align(qs, ts, bl), ref.affine_guided_align(qs, ts, bl),
                                REFINE_KEYS) is None


def test_introsort_clone_matches_std_sort_on_ties(oracle, ref):
    """SDPAlign.h:256: duplicate (x,y) keys with different lengths; the
    surviving duplicate is decided by std::sort's schedule."""
    rng = np.random.default_rng(14)
    for it in range(30):
        # identical prefix -> every prefix 5-mer fragment duplicates a middle one
        q, t = synth.make_pair(rng, int(rng.integers(60, 3000)), ins=float(rng.choice([0, 0.05])),
                               dele=0.0, sub=0.0, flank=(0, 0))
        a, b = oracle.sdp_align(q, t), ref.sdp_align(q, t)
        assert first_difference(a, b, ["score", "qPos", "tPos", "blocks"]) is None, it


def test_extend_alignment_sweep(oracle, ref):
    """next-3: ExtendAlignmentForward / Reverse, the oracle against the reference's templates over the random shapes
    (bands 0..31, drop limits 0..25, sequences that end early, N, unrelated sequence).  The harness hands the
    reference dirty, oversized matrices, as the mapper's reused buffers are."""
    from blasr_testlib import extend_align, random_extend_case
    rng = np.random.default_rng(77)
    extended = 0
    for it in range(1500):
        q, qp, t, tp, k, drops, rev = random_extend_case(rng, it)
        a, b = extend_align(oracle, q, qp, t, tp, k, drops, rev), extend_align(ref, q, qp, t, tp, k, drops, rev)
        assert (a["score"], a["qPos"], a["tPos"]) == (b["score"], b["qPos"], b["tPos"]), (it, len(q), qp, len(t), tp, k, drops, rev)
        for key in ("blocks", "gapCounts", "gapItems"):
            assert np.array_equal(a[key], b[key]), (it, key)
        extended += len(a["blocks"]) > 0
    assert extended > 800


def test_epilogue_pieces_against_the_reference_printers():
    """next-2: the oracle's restatement of AssignRefContigLocation / CreateCIGARString / IntervalAlignmentPrinter
    against the reference's own code on fresh random cases (the committed goldens cover a fixed set)."""
    import os

    import blasr_testlib as T
    if not os.path.exists(T.REFMAPPER_SO):
        pytest.skip("reference mapper harness not built (oracle/_ref)")
    orc, rf = T.EpilogueChecker("oracle"), T.EpilogueChecker("ref")
    rng = np.random.default_rng(424242)
    for i, c in enumerate(T.random_epilogue_cases(rng, 400)):
        kw = dict(clipping=c["clipping"], t_strand=c["t_strand"], read_length=c["read_length"], lqp=c["lqp"],
                  lqs=c["lqs"], subread_start=c["subread_start"], subread_end=c["subread_end"],
                  q_align_start=c["q_align_start"], q_align_end=c["q_align_end"])
        assert orc.cigar(c["q"], c["t"], c["q_pos"], c["t_pos"], ops=c["ops"], **kw) == \
            rf.cigar(c["q"], c["t"], c["q_pos"], c["t_pos"], ops=c["ops"], **kw), i
        assert orc.cigar(c["q"], c["t"], c["q_pos"], c["t_pos"], blocks=c["blocks"], **kw) == \
            rf.cigar(c["q"], c["t"], c["q_pos"], c["t_pos"], blocks=c["blocks"], **kw), i
        assert np.array_equal(orc.order_gaps_by_type(c["ops"]), rf.order_gaps_by_type(c["ops"])), i
    for it in range(300):
        ncontig = int(rng.integers(1, 7))
        lens = rng.integers(100, 3000, ncontig)
        starts = np.concatenate([[0], np.cumsum(lens + 1)]).astype(np.uint32)
        glen, strand, ci = int(starts[-1]), int(rng.integers(0, 2)), int(rng.integers(0, ncontig))
        wl = int(rng.integers(1, lens[ci]))
        fstart = int(starts[ci]) + int(rng.integers(0, lens[ci] - wl + 1))
        pos = fstart if strand == 0 else glen - (fstart + wl - 1) - 1
        assert orc.ref_contig_location(pos, wl, strand, starts, glen) == rf.ref_contig_location(pos, wl, strand, starts, glen)
        blocks = [(0, 0, int(rng.integers(1, 900)))]
        args = ("m/%d" % it, "ref%d" % ci, int(rng.integers(-90000, 10)), float(np.float32(rng.random() * 100)), it % 2,
                int(rng.integers(0, 500)), int(rng.integers(0, 50)), int(rng.integers(1000, 40000)), strand,
                pos, int(rng.integers(0, 50)), glen, int(rng.integers(0, 255)))
        assert orc.m4_line(*args, blocks) == rf.m4_line(*args, blocks), it
    for it in range(400):
        alns, qv = T.random_mapqv_case(rng, it)
        par = dict(ins=[5, 4, 8][it % 3], dele=[5, 5, 8][it % 3], affine_open=[50, 30, 10][it % 3],
                   affine_extend=[0, 1][it % 2], affine_align=it % 2)
        assert orc.store_map_qvs(alns, qv, **par) == rf.store_map_qvs(alns, qv, **par), it
    for it in range(300):  # the SAM record: the reference harness against the oracle (CIGAR from the library)
        c = T.random_sam_case(rng, it)
        line, derived = rf.sam_line(c)
        n = max(c["q_pos"], c["t_pos"]) + int(sum(c["blocks"][-1])) + 8
        a = np.full(n, ord("A"), dtype=np.uint8)
        cig, clips = orc.cigar(a, a, c["q_pos"], c["t_pos"], blocks=c["blocks"], clipping=c["clipping"],
                               t_strand=c["t_strand"], read_length=c["read_length"], lqp=c["low_quality_prefix"],
                               lqs=c["low_quality_suffix"], subread_start=c["subread_start"],
                               subread_end=c["subread_end"], q_align_start=derived[0], q_align_end=derived[1])
        assert orc.sam_line(c, cig, clips, derived) == line, it


def test_nonguided_refine_branch_against_the_reference_RefineAlignment(oracle):
    """a-11, the whole branch: the reference's own RefineAlignment with useGuidedAlign == false (Blasr.cpp:917-1046:
    ComputeDrift, the trimming, k = |drift| * 1.5 or bandSize, PairwiseLocalAlign -> AffineKBandAlign(k * 1.2)) on the
    candidate its AlignIntervals built, against the oracle's restatement: the window's position and length fields, qPos /
    tPos, blocks, gap lists, the four counts and the identity -- everything but the score, which the reference computes
    with a matrix it never initialised (SURVEY a-11)."""
    from blasr_testlib import MapperOracle, load_refmapper, random_interval_case
    refm = load_refmapper()
    if refm is None:
        pytest.skip("oracle/_ref/libblasr_refmapper.so not built")
    mo = MapperOracle()
    rng = np.random.default_rng(31)
    genome = synth.random_genome(60000, 5)
    n_wide = 0
    for it in range(36):
        read, subread, iv, contigs = random_interval_case(rng, genome, it)
        band = [0, 0, 20][it % 3]
        win, tseq, sdp, want = refm.align_interval(genome, read, subread, iv, contigs, nonguided_band=band)
        q = read[win["qAlignedSeqPos"]: win["qAlignedSeqPos"] + win["qAlignedSeqLength"]]
        mine_sdp = oracle.sdp_align(q, tseq)
        assert np.array_equal(mine_sdp["blocks"], sdp["blocks"])
        if len(sdp["blocks"]) == 0:
            continue
        got = oracle.refine_nonguided(q, tseq, sdp, band=band)
        assert got["k"] == (band if band else max(10, int(abs(oracle.compute_drift(sdp["blocks"])) * 1.5)))
        n_wide += got["kband"] > 40
        assert want["tAlignedSeqPos"] == (win["tAlignedSeqPos"] + sdp["tPos"]) & 0xffffffff
        assert want["qAlignedSeqPos"] == win["qAlignedSeqPos"] + sdp["qPos"]
        assert (want["tAlignedSeqLength"], want["qAlignedSeqLength"]) == (len(got["ts"]), len(got["qs"]))
        assert (want["qPos"], want["tPos"]) == (0, 0)
        for f in ("nMatch", "nMismatch", "nIns", "nDel"):
            assert want[f] == got[f], (it, f)
        assert np.float32(want["pct"]) == np.float32(got["pct"])
        assert np.array_equal(want["blocks"], got["blocks"]), it
        assert np.array_equal(want["gapCounts"], got["gapCounts"]) and np.array_equal(want["gapItems"], got["gapItems"]), it
    assert n_wide >= 3


def test_nonguided_branch_composed_from_the_reference_templates(oracle, ref):
    """bench.py's CPU leg for --no-guided runs the branch with the reference's OWN AffineKBandAlign and
    ComputeAlignmentStats templates (oracle/_ref/libblasr_ref.so; the drift and band rule in between are three lines):
    that composition equals the oracle's single function, score included (both carry the matrix)."""
    rng = np.random.default_rng(41)
    keys = ("drift", "k", "kband", "dpScore", "score", "nMatch", "nMismatch", "nIns", "nDel", "nCells", "blocks", "gapCounts",
            "gapItems", "qs", "ts")
    for it in range(12):
        q, t = synth.make_pair(rng, int(rng.integers(200, 3000)), ins=0.10, dele=0.05, flank=(0, int(rng.integers(0, 150))))
        sdp = ref.sdp_align(q, t)
        if len(sdp["blocks"]) == 0:
            continue
        band = [0, 0, 30][it % 3]
        a, b = oracle.refine_nonguided(q, t, sdp, band=band), ref.refine_nonguided(q, t, sdp, band=band)
        assert first_difference(a, b, keys) is None, (it, first_difference(a, b, keys))
        assert np.float32(a["pct"]) == np.float32(b["pct"])
