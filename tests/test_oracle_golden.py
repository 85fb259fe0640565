"""The CPU oracle (oracle/blasr_oracle.c) against golden vectors produced by
the reference itself (tests/golden/make_golden.py).  Bit-exact: integer work."""
import numpy as np

from blasr_testlib import GLOBAL, LOCAL
from golden_cases import REFINE_KEYS, first_difference, golden_cases, guide_digest


def test_sdp_align_matches_reference(oracle, golden):
    for c in golden_cases(golden):
        r = oracle.sdp_align(c.q, c.t, c.sc, c.sp)
        assert (r["score"], r["qPos"], r["tPos"]) == (c.sdp_score, c.sdp_qpos, c.sdp_tpos), c.name
        assert np.array_equal(r["blocks"], c.sdp_blocks), c.name


def test_sdp_stats_match_reference(oracle, golden):
    for c in golden_cases(golden):
        s = oracle.compute_alignment_stats(c.q, c.t, c.sdp_qpos, c.sdp_tpos, c.sdp_blocks, sc=c.sc)
        got = [s["score"], s["nMatch"], s["nMismatch"], s["nIns"], s["nDel"]]
        assert got == list(c.sdp_stats), c.name
        assert s["pct"] == np.float32(c.sdp_pct), c.name


def test_guide_matches_reference(oracle, golden):
    for c in golden_cases(golden):
        if not c.has_refine:
            continue
        rows, ncells = oracle.alignment_to_guide(c.sdp_blocks, c.band)
        assert np.array_equal(guide_digest(rows, ncells), c.guide_digest), c.name


def test_affine_guided_align_matches_reference(oracle, golden):
    n = 0
    for c in golden_cases(golden):
        if not c.has_refine:
            continue
        got = oracle.affine_guided_align(c.qs, c.ts, c.sdp_blocks, c.sc, c.band)
        assert first_difference(got, c.expected_refine(), REFINE_KEYS) is None, c.name
        n += 1
    assert n >= 18


def test_sw_align_matches_reference(oracle, golden):
    for i in range(int(golden["n_sw"][0])):
        r = oracle.sw_align(golden["sw%02d_q" % i], golden["sw%02d_t" % i])
        assert r["score"] == int(golden["sw%02d_score" % i][0])
        assert np.array_equal(r["blocks"], golden["sw%02d_blocks" % i].reshape(-1, 3))


def test_chain_matches_reference(oracle, golden):
    for i in range(int(golden["n_chain"][0])):
        xyl = golden["ch%02d_xyl" % i]
        qlen, k = (int(v) for v in golden["ch%02d_par" % i])
        assert np.array_equal(oracle.sdp_chain(qlen, xyl, k, align_type=LOCAL), golden["ch%02d_local" % i])
        assert np.array_equal(oracle.sdp_chain(qlen, xyl, k, align_type=GLOBAL), golden["ch%02d_global" % i])


def test_oracle_affine_kband_matches_golden(oracle):
    """a-11: the oracle's AffineKBandAlign restatement against the vectors of the reference's template."""
    import os
    from blasr_testlib import affine_kband_align
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kband_golden.npz"))
    for i in range(int(g["n"])):
        r = affine_kband_align(oracle, g["c%d_q" % i], g["c%d_t" % i], int(g["c%d_k" % i]))
        assert r["score"] == int(g["c%d_score" % i]), i
        assert np.array_equal(r["blocks"], g["c%d_blocks" % i]), i
        assert np.array_equal(r["gapCounts"], g["c%d_gapCounts" % i]), i
        assert np.array_equal(r["gapItems"], g["c%d_gapItems" % i]), i


def test_oracle_kband_linear_matches_golden(oracle):
    """a-11: the oracle's KBandAlign restatement (Global / QueryFit / TargetFit / Fit) against the vectors of the
    reference's template."""
    import os
    from blasr_testlib import kband_align
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kband_golden.npz"))
    types = set()
    for i in range(int(g["n_linear"])):
        k, at, ins, dele, fi, fd = (int(v) for v in g["l%d_par" % i])
        r = kband_align(oracle, g["l%d_q" % i], g["l%d_t" % i], k, at, ins=ins, dele=dele, fn_ins=fi, fn_del=fd)
        assert [r["score"], r["qPos"], r["tPos"]] == [int(v) for v in g["l%d_res" % i]], i
        assert np.array_equal(r["blocks"], g["l%d_blocks" % i]), i
        assert np.array_equal(r["gapCounts"], g["l%d_gapCounts" % i]), i
        assert np.array_equal(r["gapItems"], g["l%d_gapItems" % i]), i
        types.add(at)
    assert types == {1, 2, 3, 7}


def test_oracle_extend_matches_golden(oracle):
    """next-3: the oracle's ExtendAlignment restatement (Forward and Reverse) against the vectors of the reference's
    templates."""
    import os
    from blasr_testlib import extend_align
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "extend_golden.npz"))
    seen = set()
    for i in range(int(g["n"])):
        qp, tp, k, drops, rev = (int(v) for v in g["e%d_par" % i])
        r = extend_align(oracle, g["e%d_q" % i], qp, g["e%d_t" % i], tp, k, drops, bool(rev))
        assert [r["score"], r["qPos"], r["tPos"]] == [int(v) for v in g["e%d_res" % i]], i
        assert np.array_equal(r["blocks"], g["e%d_blocks" % i]), i
        assert np.array_equal(r["gapCounts"], g["e%d_gapCounts" % i]), i
        assert np.array_equal(r["gapItems"], g["e%d_gapItems" % i]), i
        seen.add((rev, len(r["blocks"]) > 0))
    assert seen == {(0, False), (0, True), (1, False), (1, True)}


def test_oracle_nonguided_refine_matches_golden(oracle):
    """a-11, the whole -noUseGuidedAlign branch of RefineAlignment (Blasr.cpp:917-1046): the oracle's restatement
    against what the reference's own function left in 40 candidates (tests/golden/make_nonguided_golden.py)."""
    from blasr_testlib import MapperOracle, nonguided_golden_cases
    genome, cases = nonguided_golden_cases()
    mo = MapperOracle()
    for i, c in enumerate(cases):
        win, tseq = mo.window(genome, len(c["read"]), c["subread"], c["iv"], c["contigs"])
        q = c["read"][win["qAlignedSeqPos"]: win["qAlignedSeqPos"] + win["qAlignedSeqLength"]]
        sdp = oracle.sdp_align(q, tseq)
        got, w = oracle.refine_nonguided(q, tseq, sdp, band=c["band"]), c["want"]
        assert w["tAlignedSeqPos"] == (win["tAlignedSeqPos"] + sdp["tPos"]) & 0xffffffff, i
        assert w["qAlignedSeqPos"] == win["qAlignedSeqPos"] + sdp["qPos"], i
        assert (w["tAlignedSeqLength"], w["qAlignedSeqLength"], w["qPos"], w["tPos"]) == (len(got["ts"]), len(got["qs"]), 0, 0), i
        assert all(w[f] == got[f] for f in ("nMatch", "nMismatch", "nIns", "nDel")), i
        assert w["pct"] == np.float32(got["pct"]), i
        assert np.array_equal(w["blocks"], got["blocks"]) and np.array_equal(w["gapCounts"], got["gapCounts"]) and \
            np.array_equal(w["gapItems"], got["gapItems"]), i


def test_words_of_up_to_32_bases_golden(oracle):
    """tests/golden/long_words_golden.npz: the reference's SDPAlign with -sdpTupleSize 14 .. 32."""
    from blasr_testlib import long_word_golden_cases
    from golden_cases import first_difference
    cases = long_word_golden_cases()
    assert len(cases) >= 40 and max(sp.k for _, _, sp, _ in cases) == 32
    for i, (q, t, sp, want) in enumerate(cases):
        got = oracle.sdp_align(q, t, None, sp)
        assert first_difference(got, want, ["score", "qPos", "tPos", "blocks"]) is None, (i, vars(sp))
