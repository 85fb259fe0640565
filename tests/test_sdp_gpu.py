"""Parity of the HIP SDP stage (SDPAlign + ComputeAlignmentStats through the C ABI) with the
reference: golden vectors made by the reference itself, and randomised comparison with the CPU
oracle.  Bit-exact: score, qPos, tPos, every block, every statistic."""
import numpy as np
import pytest

import blasr_amd
from blasr_amd import synth
from blasr_testlib import GLOBAL, SMRT_MATRIX
from golden_cases import golden_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = blasr_amd.Context(0)
    yield c
    c.close()


def _sdp_params(sp):
    return blasr_amd.SdpParams(k=sp.k, sdp_ins=sp.sdp_ins, sdp_del=sp.sdp_del, indel_rate=sp.indel_rate,
                               align_type=sp.align_type, detailed=sp.detailed, extend_front=sp.extend_front,
                               prefix=sp.prefix, recurse=sp.recurse, recurse_over=sp.recurse_over,
                               max_per_pos=sp.max_per_pos)


def _score(sc):
    return blasr_amd.ScoreParams(sc.matrix, sc.ins, sc.dele, sc.affine_open, sc.affine_extend)


def _check(got, want, stats, tag):
    assert got["status"] == 0, tag
    assert (got["score"], got["qPos"], got["tPos"]) == (want["score"], want["qPos"], want["tPos"]), tag
    assert np.array_equal(got["blocks"], want["blocks"]), tag
    s = got["stats"]
    assert [s["score"], s["nMatch"], s["nMismatch"], s["nIns"], s["nDel"]] == list(stats), tag


def test_golden_vectors(ctx, golden):
    n = 0
    for c in golden_cases(golden):
        if c.sp.extend_front:  # outside the device domain, rejected below
            continue
        got = ctx.sdp([(c.q, c.t)], _sdp_params(c.sp), _score(c.sc))[0]
        want = dict(score=c.sdp_score, qPos=c.sdp_qpos, tPos=c.sdp_tpos, blocks=c.sdp_blocks)
        _check(got, want, c.sdp_stats, c.name)
        n += 1
    assert n >= 20


def test_golden_vectors_as_one_batch(ctx, golden):
    cases = [c for c in golden_cases(golden) if c.sp.k == 11 and c.sp.align_type == 0
             and np.array_equal(c.sc.matrix, SMRT_MATRIX) and c.sc.affine_open == 50 and not c.sp.extend_front]
    got = ctx.sdp([(c.q, c.t) for c in cases])
    for c, g in zip(cases, got):
        _check(g, dict(score=c.sdp_score, qPos=c.sdp_qpos, tPos=c.sdp_tpos, blocks=c.sdp_blocks), c.sdp_stats, c.name)


def test_random_tasks_against_oracle(ctx, oracle):
    rng = np.random.default_rng(201)
    genome = synth.random_genome(300000, 7)
    pairs = []
    for it in range(48):
        kind = it % 6
        if kind == 0:
            q, t, _ = synth.make_task(genome, rng, int(rng.integers(500, 6000)))
        else:
            big = {1: ("ins", int(rng.integers(100, 700))), 2: ("del", int(rng.integers(100, 1400)))}.get(kind)
            nrun = int(rng.integers(5, 60)) if kind == 3 else None
            flank = (0, 0) if kind == 4 else (0, 200)
            er = dict(ins=0.0, dele=0.0, sub=0.0) if kind == 5 else dict(ins=0.11, dele=0.04, sub=0.01)
            q, t = synth.make_pair(rng, int(rng.integers(100, 5000)), flank=flank, big_indel=big, n_run=nrun, **er)
        pairs.append((q, t))
    got = ctx.sdp(pairs)
    for i, ((q, t), g) in enumerate(zip(pairs, got)):
        w = oracle.sdp_align(q, t)
        st = oracle.compute_alignment_stats(q, t, w["qPos"], w["tPos"], w["blocks"])
        _check(g, w, [st["score"], st["nMatch"], st["nMismatch"], st["nIns"], st["nDel"]], i)


def test_parameter_variety_against_oracle(ctx, oracle):
    from blasr_testlib import ScoreParams as OSc, SdpParams as OSp
    rng = np.random.default_rng(202)
    for it in range(60):
        q, t = synth.make_pair(rng, int(rng.integers(20, 1500)), flank=(0, int(rng.integers(0, 100))),
                               ins=float(rng.uniform(0, 0.15)), dele=float(rng.uniform(0, 0.1)),
                               sub=float(rng.uniform(0, 0.05)))
        m = SMRT_MATRIX.copy()
        if it % 3 == 0:
            m = rng.integers(-6, 8, size=(5, 5)).astype(np.int32)
            m[np.arange(4), np.arange(4)] = -rng.integers(1, 8, size=4)
        osc = OSc(m, int(rng.integers(1, 9)), int(rng.integers(1, 9)), int(rng.integers(0, 60)), int(rng.integers(0, 3)))
        osp = OSp(k=int(rng.integers(4, 14)), sdp_ins=int(rng.integers(1, 12)), sdp_del=int(rng.integers(1, 12)),
                  indel_rate=float(rng.uniform(0.1, 1.0)), align_type=int(rng.integers(0, 2)), detailed=1,
                  extend_front=0, prefix=int(rng.choice([0, 20, 50])), recurse=int(rng.integers(0, 4)),
                  recurse_over=int(rng.choice([100, 1000, 10000])), max_per_pos=int(rng.choice([0, 0, 3])))
        got = ctx.sdp([(q, t)], _sdp_params(osp), _score(osc))[0]
        w = oracle.sdp_align(q, t, osc, osp)
        st = oracle.compute_alignment_stats(q, t, w["qPos"], w["tPos"], w["blocks"], sc=osc)
        _check(got, w, [st["score"], st["nMatch"], st["nMismatch"], st["nIns"], st["nDel"]], (it, vars(osp)))


def test_long_noisy_reads_against_oracle(ctx, oracle):
    rng = np.random.default_rng(203)
    pairs = [synth.make_pair(rng, 10000, flank=(0, 500)),
             synth.make_pair(rng, 30000, ins=0.10, dele=0.05, sub=0.01, flank=(0, 900))]
    for (q, t), g in zip(pairs, ctx.sdp(pairs)):
        w = oracle.sdp_align(q, t)
        st = oracle.compute_alignment_stats(q, t, w["qPos"], w["tPos"], w["blocks"])
        _check(g, w, [st["score"], st["nMatch"], st["nMismatch"], st["nIns"], st["nDel"]], len(q))


def test_unsupported_parameters_are_rejected(ctx):
    q = np.frombuffer(b"ACGTACGTACGTACGTACGT", np.uint8)
    with pytest.raises(blasr_amd.BlasrHipError) as e:
        ctx.sdp([(q, q)], blasr_amd.SdpParams(extend_front=1))
    assert e.value.code == blasr_amd.hip.ERR_UNSUPPORTED
