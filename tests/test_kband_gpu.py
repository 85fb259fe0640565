"""a-11 on the device: blasr_hip_kband_* (AffineKBandAlign, Global) against golden vectors produced by the
reference's own template and against the oracle, bit-exact: score, blocks, gap lists."""
import os

import numpy as np
import pytest

import blasr_amd
from blasr_testlib import affine_kband_align

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    c = blasr_amd.Context(0)
    yield c
    c.close()


def same(r, want):
    return (r["status"] == 0 and r["score"] == want["score"] and np.array_equal(r["blocks"], want["blocks"])
            and np.array_equal(r["gapCounts"], want["gapCounts"]) and np.array_equal(r["gapItems"], want["gapItems"]))


def test_kband_matches_golden(ctx):
    g = np.load(os.path.join(ROOT, "tests", "golden", "kband_golden.npz"))
    n = int(g["n"])
    pairs = [(g["c%d_q" % i], g["c%d_t" % i]) for i in range(n)]
    ks = [int(g["c%d_k" % i]) for i in range(n)]
    res = ctx.kband(pairs, ks)
    for i, r in enumerate(res):
        want = dict(score=int(g["c%d_score" % i]), blocks=g["c%d_blocks" % i], gapCounts=g["c%d_gapCounts" % i],
                    gapItems=g["c%d_gapItems" % i])
        assert same(r, want), (i, len(pairs[i][0]), len(pairs[i][1]), ks[i], r["score"], want["score"])


def test_kband_matches_oracle_random(ctx, oracle):
    from blasr_amd import synth
    rng = np.random.default_rng(12)
    pairs, ks = [], []
    for it in range(96):
        ql = int(rng.choice([3, 17, 64, 65, 200, 900, 2500]))
        q, t = synth.make_pair(rng, ql, flank=(0, 0))
        if it % 5 == 0:
            t = t[: max(1, len(t) - int(rng.integers(0, 40)))]
        if it % 7 == 0:
            q = q[: max(1, len(q) - int(rng.integers(0, 40)))]
        if it % 9 == 0:
            q = q.copy()
            q[len(q) // 3: len(q) // 3 + 9] = ord("T")  # a homopolymer run: the HP matrix gets used
        pairs.append((q, t))
        ks.append(int(rng.choice([10, 12, 31, 32, 33, 64, 100, 300, 543])))
    res = ctx.kband(pairs, ks)
    for (q, t), k, r in zip(pairs, ks, res):
        assert same(r, affine_kband_align(oracle, q, t, k)), (len(q), len(t), k)


def test_kband_chunked_workspace_and_errors(ctx, oracle):
    from blasr_amd import synth
    rng = np.random.default_rng(4)
    pairs = [synth.make_pair(rng, 700, flank=(0, 0)) for _ in range(12)]
    ks = [40] * 12
    small = blasr_amd.Context(0)
    small.set_workspace_limit(1 << 20)  # one or two candidates per chunk
    res = small.kband(pairs, ks)
    for (q, t), r in zip(pairs, res):
        assert same(r, affine_kband_align(oracle, q, t, 40))
    # a band too wide for the LDS rows takes the strip kernel, in the same call as a narrow one
    res = small.kband(pairs[:2], [40, 600])
    assert same(res[0], affine_kband_align(oracle, pairs[0][0], pairs[0][1], 40))
    assert same(res[1], affine_kband_align(oracle, pairs[1][0], pairs[1][1], 600))
    # k < 1: per-candidate status, the others still run
    res = small.kband(pairs[:2], [40, 0])
    assert res[0]["status"] == 0 and res[1]["status"] == blasr_amd.hip.TASK_BAD_GUIDE
    with pytest.raises(blasr_amd.BlasrHipError) as e:
        small.kband(pairs[:1], [40], ops_capacity=1)
    assert e.value.code == blasr_amd.hip.ERR_CAPACITY
    small.close()


@pytest.fixture
def strip_kernel_for_every_band(monkeypatch):
    """kband_api.hip reads the switch per call: every band, however narrow, goes through kband_wide_kernel."""
    monkeypatch.setenv("BLASR_HIP_KBAND_WIDE_COLS", "3")


def test_kband_strip_kernel_matches_golden(ctx, strip_kernel_for_every_band):
    test_kband_matches_golden(ctx)


@pytest.mark.parametrize("waves", [2, 3, 16])
def test_kband_strip_kernel_with_several_wavefronts_per_task(ctx, oracle, strip_kernel_for_every_band, monkeypatch, waves):
    """The strips of one task as a pipeline over `waves` wavefronts (normally chosen from the task's size): every task of
    the small vectors gets that many, whether or not it has as many strips."""
    monkeypatch.setenv("BLASR_HIP_KBAND_WAVES", str(waves))
    test_kband_matches_golden(ctx)
    from blasr_amd import synth
    rng = np.random.default_rng(waves)
    pairs, ks = [], []
    for ql, k in [(700, 40), (1500, 100), (2600, 30), (4000, 300), (900, 700)]:
        pairs.append(synth.make_pair(rng, ql, flank=(0, 0)))
        ks.append(k)
    for (q, t), k, r in zip(pairs, ks, ctx.kband(pairs, ks)):
        assert same(r, affine_kband_align(oracle, q, t, k)), (len(q), len(t), k, waves)


def test_kband_strip_kernel_matches_oracle_random(ctx, oracle, strip_kernel_for_every_band):
    test_kband_matches_oracle_random(ctx, oracle)
    # shapes around the strip geometry: 256 columns per strip, 64 rows per boundary block, eight steps per store
    from blasr_amd import synth
    rng = np.random.default_rng(21)
    pairs, ks = [], []
    for ql, k in [(255, 10), (256, 10), (257, 10), (511, 70), (513, 64), (1024, 128), (1300, 400), (700, 1000), (64, 5),
                  (1, 3), (2, 1), (5, 300), (320, 31), (3000, 150)]:
        q, t = synth.make_pair(rng, ql, flank=(0, 0))
        pairs.append((q, t))
        ks.append(k)
        pairs.append((q, t[: max(1, len(t) - min(k, len(t) // 2))]))  # SetKBoundedLengths cuts the query
        ks.append(k)
        pairs.append((q[: max(1, len(q) - min(k, len(q) // 2))], t))  # ... or the target
        ks.append(k)
    pairs.append((np.frombuffer(b"ACGTNNNNACGTTTTTTTTTTTTACGNACGT" * 20, dtype=np.uint8),
                  np.frombuffer(b"ACGTNNNACGTTTTTTTTTTACGNNACGT" * 20, dtype=np.uint8)))
    ks.append(90)
    res = ctx.kband(pairs, ks)
    for (q, t), k, r in zip(pairs, ks, res):
        assert same(r, affine_kband_align(oracle, q, t, k)), (len(q), len(t), k)
    # other penalties and an asymmetric matrix
    m = np.array([[-7, 3, 4, 5, 1], [6, -6, 2, 7, 0], [5, 4, -5, 3, 2], [2, 6, 7, -8, 0], [1, 0, 2, 0, 1]], dtype=np.int32)
    for indel in (3, 8):
        res = ctx.kband(pairs[:12], ks[:12], matrix=m, indel=indel)
        for (q, t), k, r in zip(pairs[:12], ks[:12], res):
            assert same(r, affine_kband_align(oracle, q, t, k, matrix=m, indel=indel)), (len(q), len(t), k, indel)


@pytest.mark.parametrize("read_len,k", [(10000, 1300), (30000, 2700)])
def test_kband_at_the_widths_noUseGuidedAlign_gives(ctx, oracle, read_len, k):
    """-noUseGuidedAlign: k = max(10, |drift| * 1.5) * 1.2 (Blasr.cpp:991-1001, :699) is ~1.3 k for 10 kb reads and
    ~2.7 k for 30 kb reads at 15 % indels (configs[2] / configs[4]): 2k+1 columns, far beyond the LDS rows."""
    from blasr_amd import synth
    rng = np.random.default_rng(read_len)
    q, t = synth.make_pair(rng, read_len, ins=0.10, dele=0.05, sub=0.01, flank=(0, 0))
    r = ctx.kband([(q, t)], [k])[0]
    want = affine_kband_align(oracle, q, t, k)
    assert same(r, want), (len(q), len(t), k, r["status"], r["score"], want["score"])


def same_linear(r, want):
    return (r["status"] == 0 and (r["score"], r["qPos"], r["tPos"]) == (want["score"], want["qPos"], want["tPos"])
            and np.array_equal(r["blocks"], want["blocks"]) and np.array_equal(r["gapCounts"], want["gapCounts"])
            and np.array_equal(r["gapItems"], want["gapItems"]))


def test_kband_linear_matches_golden(ctx):
    g = np.load(os.path.join(ROOT, "tests", "golden", "kband_golden.npz"))
    for i in range(int(g["n_linear"])):
        k, at, ins, dele, fi, fd = (int(v) for v in g["l%d_par" % i])
        r = ctx.kband_linear([(g["l%d_q" % i], g["l%d_t" % i])], [k], at, ins=ins, dele=dele, fn_ins=fi, fn_del=fd)[0]
        res = g["l%d_res" % i]
        want = dict(score=int(res[0]), qPos=int(res[1]), tPos=int(res[2]), blocks=g["l%d_blocks" % i],
                    gapCounts=g["l%d_gapCounts" % i], gapItems=g["l%d_gapItems" % i])
        assert same_linear(r, want), (i, k, at, r["score"], want["score"], r["qPos"], want["qPos"], r["tPos"], want["tPos"])


def test_kband_linear_matches_oracle_random(ctx, oracle):
    from blasr_amd import synth
    from blasr_testlib import kband_align
    rng = np.random.default_rng(13)
    for at in (1, 2, 3, 7):
        pairs, ks = [], []
        for it in range(40):
            ql = int(rng.choice([12, 40, 65, 200, 900, 2500]))
            q, t = synth.make_pair(rng, ql, flank=(0, int(rng.choice([0, 0, 20]))))
            if it % 5 == 0:
                t = t[: max(12, len(t) - int(rng.integers(0, 30)))]
            if it % 7 == 0:
                q = q[: max(12, len(q) - int(rng.integers(0, 30)))]
            pairs.append((q, t))
            ks.append(int(rng.choice([10, 12, 31, 32, 33, 64, 150, 400])))
        res = ctx.kband_linear(pairs, ks, at, ins=7, dele=6, fn_ins=5, fn_del=4)
        for (q, t), k, r in zip(pairs, ks, res):
            assert same_linear(r, kband_align(oracle, q, t, k, at, ins=7, dele=6, fn_ins=5, fn_del=4)), (at, len(q), len(t), k)
