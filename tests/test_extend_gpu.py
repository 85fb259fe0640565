"""next-3 on the device: blasr_hip_extend_* (ExtendAlignmentForward / ExtendAlignmentReverse, ExtendAlign.h:115-445)
against golden vectors produced by the reference's own templates and against the oracle, bit-exact: return value,
qPos, tPos, blocks, gap lists."""
import os

import numpy as np
import pytest

import blasr_amd
from blasr_testlib import extend_align, random_extend_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    c = blasr_amd.Context(0)
    yield c
    c.close()


def same(r, want):
    return (r["status"] == 0 and (r["score"], r["qPos"], r["tPos"]) == (want["score"], want["qPos"], want["tPos"])
            and np.array_equal(r["blocks"], want["blocks"]) and np.array_equal(r["gapCounts"], want["gapCounts"])
            and np.array_equal(r["gapItems"], want["gapItems"]))


def test_extend_matches_golden(ctx):
    g = np.load(os.path.join(ROOT, "tests", "golden", "extend_golden.npz"))
    by_params = {}
    for i in range(int(g["n"])):
        qp, tp, k, drops, rev = (int(v) for v in g["e%d_par" % i])
        by_params.setdefault((k, drops), []).append((i, (g["e%d_q" % i], qp, g["e%d_t" % i], tp, bool(rev))))
    checked = 0
    for (k, drops), items in by_params.items():
        res = ctx.extend([task for _, task in items], band=k, max_n_drops=drops)
        for (i, task), r in zip(items, res):
            score, q_pos, t_pos = (int(v) for v in g["e%d_res" % i])
            want = dict(score=score, qPos=q_pos, tPos=t_pos, blocks=g["e%d_blocks" % i],
                        gapCounts=g["e%d_gapCounts" % i], gapItems=g["e%d_gapItems" % i])
            assert same(r, want), (i, k, drops, task[1], task[3], task[4], r["score"], score, r["qPos"], q_pos)
            checked += 1
    assert checked == int(g["n"])


def test_extend_matches_oracle_random(ctx, oracle):
    rng = np.random.default_rng(31)
    groups = {}
    for it in range(1200):
        q, qp, t, tp, k, drops, rev = random_extend_case(rng, it)
        groups.setdefault((k, drops), []).append((q, qp, t, tp, rev))
    extended = 0
    for (k, drops), tasks in groups.items():
        res = ctx.extend(tasks, band=k, max_n_drops=drops, ins=4, dele=6)
        for (q, qp, t, tp, rev), r in zip(tasks, res):
            want = extend_align(oracle, q, qp, t, tp, k, drops, rev, ins=4, dele=6)
            assert same(r, want), (len(q), qp, len(t), tp, k, drops, rev, r["score"], want["score"])
            extended += len(r["blocks"]) > 0
    assert extended > 600


def test_extend_mapper_shapes_one_batch(ctx, oracle):
    """What AlignIntervals hands the two functions (Blasr.cpp:1805-1897): up to 500 bases of read end against up to
    500 of genome, Forward from (0, 0) and Reverse from (length - 1, length - 1); one batch, many wavefronts."""
    from blasr_amd import synth
    rng = np.random.default_rng(8)
    tasks = []
    for it in range(700):
        q, t = synth.make_pair(rng, int(rng.integers(1, 520)), flank=(0, int(rng.choice([0, 40]))))
        if len(q) == 0 or len(t) == 0:
            continue
        q, t = q[:500], t[:500]
        if it % 2:
            tasks.append((q[::-1].copy(), len(q) - 1, t[::-1].copy(), len(t) - 1, True))
        else:
            tasks.append((q, 0, t, 0, False))
    res = ctx.extend(tasks)
    for (q, qp, t, tp, rev), r in zip(tasks, res):
        assert same(r, extend_align(oracle, q, qp, t, tp, 10, 10, rev)), (len(q), len(t), rev)


def test_extend_chunked_workspace_and_errors(ctx, oracle):
    from blasr_amd import synth
    rng = np.random.default_rng(4)
    tasks = []
    for _ in range(300):
        q, t = synth.make_pair(rng, 450, ins=0.01, dele=0.01, sub=0.01, flank=(0, 0))  # clean: long extensions
        tasks.append((q, 0, t, 0, False))
    small = blasr_amd.Context(0)
    small.set_workspace_limit(1 << 20)
    res = small.extend(tasks, max_n_drops=25)
    assert max(int(r["blocks"][:, 2].sum()) for r in res) > 300
    for (q, qp, t, tp, rev), r in zip(tasks, res):
        assert same(r, extend_align(oracle, q, qp, t, tp, 10, 25, rev))
    # Reverse from beyond the sequence: the reference would read outside it
    res = small.extend([tasks[0], (tasks[1][0], len(tasks[1][0]) + 1, tasks[1][2], 5, True)])
    assert res[0]["status"] == 0 and res[1]["status"] == blasr_amd.hip.TASK_BAD_GUIDE
    with pytest.raises(blasr_amd.BlasrHipError) as e:
        small.extend(tasks[:1], band=32)
    assert e.value.code == blasr_amd.hip.ERR_UNSUPPORTED
    with pytest.raises(blasr_amd.BlasrHipError) as e:
        small.extend(tasks[:4], max_n_drops=25, ops_capacity=1)
    assert e.value.code == blasr_amd.hip.ERR_CAPACITY
    small.close()
