"""Randomised sweep: oracle/blasr_oracle.c against the reference build
(oracle/_ref/libblasr_ref.so).  Skipped where the reference build is absent."""
import numpy as np

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
