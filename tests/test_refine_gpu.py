"""Parity of the HIP refine path (through the C ABI) with the reference:
golden vectors made by the reference itself and randomised comparison with the
CPU oracle.  Bit-exact: every field, every block, every gap."""
import numpy as np
import pytest

import blasr_amd
from blasr_amd import hip, synth
from blasr_testlib import SMRT_MATRIX
from golden_cases import REFINE_KEYS, first_difference, golden_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = blasr_amd.Context(0)
    yield c
    c.close()


def _score(sc):
    return blasr_amd.ScoreParams(sc.matrix, sc.ins, sc.dele, sc.affine_open, sc.affine_extend)


def test_golden_vectors_one_by_one(ctx, golden):
    n = 0
    for c in golden_cases(golden):
        if not c.has_refine:
            continue
        b = blasr_amd.RefineBatch()
        b.add(c.qs, c.ts, c.sdp_blocks)
        got = ctx.refine(b, band=c.band, score=_score(c.sc))[0]
        assert got.status == 0, c.name
        assert first_difference(got.as_dict(), c.expected_refine(), REFINE_KEYS) is None, c.name
        n += 1
    assert n >= 18


def test_golden_vectors_as_one_batch(ctx, golden):
    cases = [c for c in golden_cases(golden) if c.has_refine and c.band == 16
             and np.array_equal(c.sc.matrix, SMRT_MATRIX) and c.sc.affine_open == 50]
    b = blasr_amd.RefineBatch()
    for c in cases:
        b.add(c.qs, c.ts, c.sdp_blocks)
    for c, got in zip(cases, ctx.refine(b)):
        assert first_difference(got.as_dict(), c.expected_refine(), REFINE_KEYS) is None, c.name


@pytest.mark.parametrize("plan_threads", [None, "4"])
def test_chunked_workspace_gives_identical_results(golden, monkeypatch, plan_threads):
    if plan_threads:  # (the candidates' sizes by four host threads, the pieces cut by one: what a batch of 16 k candidates and more gets)
        monkeypatch.setenv("BLASR_HIP_PLAN_THREADS", plan_threads)
    cases = [c for c in golden_cases(golden) if c.has_refine and c.band == 16
             and np.array_equal(c.sc.matrix, SMRT_MATRIX) and c.sc.affine_open == 50]
    b = blasr_amd.RefineBatch()
    for c in cases:
        b.add(c.qs, c.ts, c.sdp_blocks)
    small = blasr_amd.Context(0)
    small.set_workspace_limit(3 << 20)  # forces several kernel waves
    got = small.refine(b)
    assert small.last_timing()["fill_launches"] > 1
    for c, g in zip(cases, got):
        assert first_difference(g.as_dict(), c.expected_refine(), REFINE_KEYS) is None, c.name
    small.close()


def test_random_pairs_against_oracle(ctx, oracle):
    rng = np.random.default_rng(101)
    b, want = blasr_amd.RefineBatch(), []
    for it in range(48):
        kind = it % 6
        big = {1: ("ins", int(rng.integers(100, 700))), 2: ("del", int(rng.integers(100, 1400)))}.get(kind)
        nrun = int(rng.integers(5, 60)) if kind == 3 else None
        flank = (0, 0) if kind == 4 else (0, 200)
        er = dict(ins=0.0, dele=0.0, sub=0.0) if kind == 5 else dict(ins=0.11, dele=0.04, sub=0.01)
        q, t = synth.make_pair(rng, int(rng.integers(100, 5000)), flank=flank, big_indel=big, n_run=nrun, **er)
        s = oracle.sdp_align(q, t)
        if len(s["blocks"]) == 0:
            continue
        bl = s["blocks"]
        qs = q[s["qPos"]: s["qPos"] + bl[-1, 0] + bl[-1, 2]]
        ts = t[s["tPos"]: s["tPos"] + bl[-1, 1] + bl[-1, 2]]
        b.add(qs, ts, bl)
        want.append(oracle.affine_guided_align(qs, ts, bl))
    got = ctx.refine(b)
    assert len(got) == len(want) >= 40
    for i, (g, w) in enumerate(zip(got, want)):
        assert g.status == 0, i
        assert first_difference(g.as_dict(), w, REFINE_KEYS) is None, i


def test_parameter_variety_against_oracle(ctx, oracle):
    """Absolute guides (tStart != 0), several bands, asymmetric matrices, gap costs."""
    rng = np.random.default_rng(102)
    for it in range(40):
        q, t = synth.make_pair(rng, int(rng.integers(30, 1500)), flank=(0, int(rng.integers(0, 100))),
                               ins=float(rng.uniform(0, 0.15)), dele=float(rng.uniform(0, 0.1)),
                               sub=float(rng.uniform(0, 0.05)))
        m = SMRT_MATRIX.copy()
        if it % 3 == 0:
            m = rng.integers(-6, 8, size=(5, 5)).astype(np.int32)
            m[np.arange(4), np.arange(4)] = -rng.integers(1, 8, size=4)
        ext = int(rng.integers(0, 3))
        sc = blasr_amd.ScoreParams(m, ins=int(rng.integers(1, 9)), dele=int(rng.integers(ext, 9)),
                                   affine_open=int(rng.integers(0, 60)), affine_extend=ext)
        from blasr_testlib import ScoreParams as OSc
        osc = OSc(m, sc.ins, sc.dele, sc.affine_open, sc.affine_extend)
        s = oracle.sdp_align(q, t, osc)
        if len(s["blocks"]) == 0:
            continue
        bl = s["blocks"].copy()
        bl[:, 0] += s["qPos"]
        bl[:, 1] += s["tPos"]
        band = int(rng.choice([4, 16, 30]))
        b = blasr_amd.RefineBatch()
        b.add(q, t, bl)
        got = ctx.refine(b, band=band, score=sc)[0]
        want = oracle.affine_guided_align(q, t, bl, osc, band)
        assert got.status == 0
        assert first_difference(got.as_dict(), want, REFINE_KEYS) is None, it


def test_long_reads_against_oracle(ctx, oracle):
    rng = np.random.default_rng(103)
    b, want = blasr_amd.RefineBatch(), []
    for qlen, er in ((10000, dict(ins=0.11, dele=0.04, sub=0.01)), (30000, dict(ins=0.10, dele=0.05, sub=0.01))):
        q, t = synth.make_pair(rng, qlen, flank=(0, 500), **er)
        s = oracle.sdp_align(q, t)
        bl = s["blocks"]
        qs = q[s["qPos"]: s["qPos"] + bl[-1, 0] + bl[-1, 2]]
        ts = t[s["tPos"]: s["tPos"] + bl[-1, 1] + bl[-1, 2]]
        b.add(qs, ts, bl)
        want.append(oracle.affine_guided_align(qs, ts, bl))
    for g, w in zip(ctx.refine(b), want):
        assert first_difference(g.as_dict(), w, REFINE_KEYS) is None


def test_degenerate_candidates(ctx):
    b = blasr_amd.RefineBatch()
    acgt = np.frombuffer(b"ACGTACGTACGTACGTAAAA", np.uint8)
    b.add(acgt, acgt, np.zeros((0, 3), np.int32))         # empty guide (AffineGuidedAlign.h:151-155)
    b.add(acgt, acgt, [[0, 0, 8], [4, 9, 4]])             # overlapping blocks in q
    b.add(acgt, acgt, [[0, 0, 30]])                       # runs past both sequences
    b.add(acgt, acgt, [[0, 0, 20]])                       # fine: one block
    got = ctx.refine(b)
    assert got[0].status == hip.TASK_EMPTY_GUIDE and len(got[0].ops) == 0
    assert got[1].status == hip.TASK_BAD_GUIDE
    assert got[2].status == hip.TASK_BAD_GUIDE
    assert got[3].status == 0 and got[3].nMatch == 20 and got[3].score == -100 and got[3].dpScore == -100
    assert got[3].ops.tolist() == [(20 << 2) | hip.OP_M]


def test_unsupported_gap_costs_are_rejected(ctx):
    b = blasr_amd.RefineBatch()
    acgt = np.frombuffer(b"ACGTACGTACGT", np.uint8)
    b.add(acgt, acgt, [[0, 0, 12]])
    with pytest.raises(blasr_amd.BlasrHipError) as e:
        ctx.refine(b, score=blasr_amd.ScoreParams(dele=1, affine_extend=3))
    assert e.value.code == hip.ERR_UNSUPPORTED
    with pytest.raises(blasr_amd.BlasrHipError) as e:
        ctx.refine(b, ops_capacity=0)
    assert e.value.code == hip.ERR_CAPACITY


@pytest.mark.parametrize("switch", ["BLASR_HIP_NO_FILL4", "BLASR_HIP_FILL_LANES16", "BLASR_HIP_H8_SPLIT", "BLASR_HIP_HYBRID4"])
def test_alternative_fill_kernels_agree(ctx, golden, monkeypatch, switch):
    """The default (extension penalty 0, 6-bit matrix): ONE queue of all candidates through fillh8_kernel (eight per wavefront,
    each group of eight lanes a candidate at a time, each at its own row).  BLASR_HIP_H8_SPLIT: the narrow candidates through
    fill8_kernel (eight fixed candidates per wavefront) beside it; BLASR_HIP_HYBRID4: the candidates with wide rows through
    fillp_kernel<16, ., true> (four in lockstep); BLASR_HIP_FILL_LANES16: fillp_kernel<16> for both lists;
    BLASR_HIP_NO_FILL4: every candidate through fill_kernel (one per wavefront).  The same goldens must come out of all of them."""
    cases = [c for c in golden_cases(golden) if c.has_refine and c.band == 16
             and np.array_equal(c.sc.matrix, SMRT_MATRIX) and c.sc.affine_open == 50]
    b = blasr_amd.RefineBatch()
    for c in cases:
        b.add(c.qs, c.ts, c.sdp_blocks)
    monkeypatch.setenv(switch, "1")
    for c, got in zip(cases, ctx.refine(b)):
        assert first_difference(got.as_dict(), c.expected_refine(), REFINE_KEYS) is None, c.name
    monkeypatch.delenv(switch)


@pytest.mark.parametrize("lanes8", [False, True, "split"])
def test_many_candidates_of_ragged_lengths(ctx, oracle, monkeypatch, lanes8):
    """fillh8_kernel (a queue; "split": fill8_kernel, eight fixed candidates) / fillp_kernel pack eight / four candidates of
    different lengths and widths into one wavefront; rows of up to 112 cells take up to three passes (of 40 / 48 columns)."""
    if not lanes8:
        monkeypatch.setenv("BLASR_HIP_FILL_LANES16", "1")
    if lanes8 == "split":
        monkeypatch.setenv("BLASR_HIP_H8_SPLIT", "1")
    rng = np.random.default_rng(77)
    b = blasr_amd.RefineBatch()
    want = []
    for it in range(37):  # not a multiple of four
        q, t = synth.make_pair(rng, int(rng.integers(60, 1500)), flank=(0, int(rng.integers(0, 80))),
                               big_indel=(("del", int(rng.integers(20, 90))) if it % 5 == 0 else None))
        sdp = oracle.sdp_align(q, t)
        if len(sdp["blocks"]) == 0:
            continue
        last = sdp["blocks"][-1]
        qs = q[sdp["qPos"]: sdp["qPos"] + last[0] + last[2]]
        ts = t[sdp["tPos"]: sdp["tPos"] + last[1] + last[2]]
        b.add(qs, ts, sdp["blocks"])
        want.append(oracle.affine_guided_align(qs, ts, sdp["blocks"]))
    got = ctx.refine(b)
    assert len(got) == len(want) >= 30
    for i, (g, w) in enumerate(zip(got, want)):
        assert g.status == 0, i
        assert first_difference(g.as_dict(), w, REFINE_KEYS) is None, i


def _guide_with_gaps(rng, gaps, block=40, n_base=False):
    """A query / target pair whose guide is exact blocks of `block` bases separated by the given gaps ((q, t) unaligned
    bases each): the rows behind a target gap of g columns are 33 + g cells wide."""
    q, t, blocks = [], [], []
    qp = tp = 0
    for k, (gq, gt) in enumerate(list(gaps) + [(0, 0)]):
        seg = synth.random_genome(block, int(rng.integers(1, 1 << 30)))
        q.append(seg)
        t.append(seg.copy())
        blocks.append([qp, tp, block])
        qp += block
        tp += block
        if k < len(gaps):
            q.append(synth.random_genome(gq, int(rng.integers(1, 1 << 30))))
            t.append(synth.random_genome(gt, int(rng.integers(1, 1 << 30))))
            qp += gq
            tp += gt
    q, t = np.concatenate(q), np.concatenate(t)
    if n_base:
        q[rng.integers(0, len(q), size=3)] = ord("N")
        t[rng.integers(0, len(t), size=3)] = ord("N")
    return np.ascontiguousarray(q), np.ascontiguousarray(t), np.array(blocks, dtype=np.int32)


def test_block_geometry_equals_the_row_walk(ctx, monkeypatch):
    """geometry_blocks_kernel sizes every candidate's arrow slab from closed forms per guide block (cells, widest row, rows,
    validation) where geometry_kernel walked all rows.  BLASR_HIP_GEOMETRY_CHECK runs the walk as well and fails the call
    on the first candidate whose answers differ: guides with gaps of 0 .. 7000 rows / columns in either direction and both
    at once, blocks of one base, gaps back to back, a first block far from the origin, bands 0 .. 300, and guides that are not
    guides (overlapping, unordered, outside the sequences: the same TASK_BAD_GUIDE either way)."""
    monkeypatch.setenv("BLASR_HIP_GEOMETRY_CHECK", "1")
    rng = np.random.default_rng(20261005)
    for band in (16, 0, 1, 5, 40, 250, 300):
        b = blasr_amd.RefineBatch()
        for it in range(60):
            nb = int(rng.integers(1, 12))
            gaps = []
            for _ in range(nb - 1):
                kind = int(rng.integers(0, 5))
                g = int(rng.choice([0, 1, 2, 17, 33, 100, 249, 250, 251, 600, 3000, 7000]))
                h = int(rng.choice([0, 1, 5, 40, 250, 700]))
                gaps.append([(g, 0), (0, g), (g, g), (g, h), (h, g)][kind])
            q, t, blocks = _guide_with_gaps(rng, gaps, block=int(rng.choice([1, 1, 2, 7, 30, 200])))
            if it % 7 == 3:  # the guide starts away from the origin of both sequences
                dq, dt = int(rng.integers(0, 400)), int(rng.integers(0, 400))
                q = np.concatenate([synth.random_genome(dq, it + 1), q])
                t = np.concatenate([synth.random_genome(dt, it + 2), t])
                blocks = blocks + np.array([dq, dt, 0], dtype=np.int32)
            if it % 11 == 5 and len(blocks) > 1:  # not a guide
                blocks = blocks.copy()
                how = it % 3
                if how == 0:
                    blocks[1, 0] = blocks[0, 0]           # overlapping in q
                elif how == 1:
                    blocks[[0, 1]] = blocks[[1, 0]]       # unordered
                else:
                    blocks[-1, 2] += len(t)               # beyond the target
            b.add(np.ascontiguousarray(q), np.ascontiguousarray(t), blocks)
        got = ctx.refine(b, band=band)  # (raises when the two geometries differ)
        assert len(got) == 60
        assert any(g.status != 0 for g in got) and any(g.status == 0 for g in got)


@pytest.mark.parametrize("lanes16", [False, True, "split"])
def test_fill8_pass_boundaries_and_domain_edges(ctx, oracle, monkeypatch, lanes16):
    """The edges of the eight-lane narrow pass (fillh8_kernel's by default, fill8_kernel's under BLASR_HIP_H8_SPLIT) one by one: rows of exactly 40 / 41 / 80 / 81 / 112 cells (one, two, three passes of 40 columns:
    target gaps of 7 / 8 / 47 / 48 / 79 columns behind a 33-cell band), a row that grows by more than the eight cleaned
    ring entries, query gaps (the band stands still), N in both sequences (the fifth row / field of the 6-bit matrix
    registers), more candidates than one wavefront's eight and fewer, a band of 8 and of 20 -- and matrix entries at the
    6-bit limits (-32, 31), then beyond them (32: the launch falls back to fillp_kernel<16>).  Everything against the
    oracle; with BLASR_HIP_FILL_LANES16 the same through the four-per-wavefront kernel."""
    narrow_kernel = "fillh8_kernel"
    if lanes16 == "split":
        monkeypatch.setenv("BLASR_HIP_H8_SPLIT", "1")
        lanes16, narrow_kernel = False, "fill8_kernel"
    if lanes16:
        monkeypatch.setenv("BLASR_HIP_FILL_LANES16", "1")
    rng = np.random.default_rng(20261004)
    shapes = [[(0, 7)], [(0, 8)], [(0, 47)], [(0, 48)], [(0, 79)], [(0, 7), (0, 79), (0, 8)], [(30, 0)], [(5, 60), (60, 5)],
              [(0, 20), (0, 40), (0, 60), (0, 79)], [(3, 3)] * 12, [(0, 79)] * 4, [(70, 79)], [(0, 1), (0, 9), (0, 10)]]
    for band in (16, 8, 20):
        b = blasr_amd.RefineBatch()
        want = []
        for k, gaps in enumerate(shapes):
            q, t, blocks = _guide_with_gaps(rng, gaps, block=int(rng.integers(12, 60)), n_base=(k % 3 == 0))
            b.add(q, t, blocks)
            want.append(oracle.affine_guided_align(q, t, blocks, band=band))
        got = ctx.refine(b, band=band)
        for i, (g, w) in enumerate(zip(got, want)):
            assert g.status == 0, (band, i)
            assert first_difference(g.as_dict(), w, REFINE_KEYS) is None, (band, i)
        if not lanes16 and band == 16:
            names = [k["name"] for k in ctx.last_kernels()]
            assert any(narrow_kernel in n for n in names), names
    # the matrix at the limits of the 6-bit fields, and one entry beyond them
    for lim, kernel in ((31, narrow_kernel), (32, "fillp_kernel<16")):
        m = np.array(SMRT_MATRIX, dtype=np.int32).copy()
        m[0, 1] = lim
        m[2, 2] = -32
        m[4, 3] = lim - 1
        sc = blasr_amd.ScoreParams(m, 5, 5, 50, 0)
        b = blasr_amd.RefineBatch()
        want = []
        for gaps in shapes[:6]:
            q, t, blocks = _guide_with_gaps(rng, gaps, n_base=True)
            b.add(q, t, blocks)
            want.append(oracle.affine_guided_align(q, t, blocks, sc=sc))
        got = ctx.refine(b, score=sc)
        for i, (g, w) in enumerate(zip(got, want)):
            assert g.status == 0 and first_difference(g.as_dict(), w, REFINE_KEYS) is None, (lim, i)
        if not lanes16:
            assert any(kernel in k["name"] for k in ctx.last_kernels()), (lim, [k["name"] for k in ctx.last_kernels()])


@pytest.mark.parametrize("switch", [None, "BLASR_HIP_H8_SPLIT", "BLASR_HIP_HYBRID4", "BLASR_HIP_NO_HYBRID", "BLASR_HIP_NO_FILL4", "BLASR_HIP_PLAN_THREADS"])
def test_guides_with_large_drift_against_oracle(ctx, oracle, monkeypatch, switch):
    """Guides whose consecutive blocks drift by hundreds to thousands of columns (what SDPAlign(Local) makes of a repeat
    copy the read overlaps only in part: it chains on through chance fragments): rows of band + |drift| cells, on
    both sides of what the LDS ring holds (192 cells; the global-memory ring beyond), against the oracle.
    BLASR_HIP_PLAN_THREADS: the planner's offsets by three host threads, as it takes batches of 16 k candidates and more."""
    if switch:
        monkeypatch.setenv(switch, "3" if switch == "BLASR_HIP_PLAN_THREADS" else "1")
    rng = np.random.default_rng(909)
    b, want = blasr_amd.RefineBatch(), []
    for drift in (150, 400, 900, 1500, 3000, 4500, 6000):
        # two homologous stretches, the target carrying `drift` extra bases between them (deletion-like) or the read
        # (insertion-like): the guide's second block starts `drift` columns off the first one's diagonal
        for kind in ("del", "ins"):
            n1, n2 = int(rng.integers(300, 900)), int(rng.integers(300, 900))
            core = synth.random_genome(n1 + n2, int(rng.integers(1, 1 << 30)))
            extra = synth.random_genome(drift, int(rng.integers(1, 1 << 30)))
            read = np.ascontiguousarray(synth.mutate(core, rng, 0.08, 0.03, 0.01))
            cut = int(n1 * len(read) / len(core))
            if kind == "del":
                q, t = read, np.concatenate([core[:n1], extra, core[n1:]])
            else:
                q, t = np.concatenate([read[:cut], extra, read[cut:]]), core
            # the guide: a short block in each stretch (as SDP would leave them), the second `drift` off the first
            q2 = cut + (drift if kind == "ins" else 0) + 40
            t2 = n1 + (drift if kind == "del" else 0) + 40
            bl = np.array([[20, 20, 12], [q2, t2, 12], [q2 + 100, t2 + 100, 12]], dtype=np.int32)
            qs, ts = np.ascontiguousarray(q[: q2 + 112]), np.ascontiguousarray(t[: t2 + 112])
            b.add(qs, ts, bl)
            want.append(oracle.affine_guided_align(qs, ts, bl))
    got = ctx.refine(b)
    widest = 0
    for i, (g, w) in enumerate(zip(got, want)):
        assert g.status == 0, i
        assert first_difference(g.as_dict(), w, REFINE_KEYS) is None, i
    if switch:
        monkeypatch.delenv(switch)


@pytest.mark.parametrize("grid", [None, ("1", "8"), ("1", "1"), ("2", "3")])
def test_fillh8_wide_pieces_and_the_queue(ctx, oracle, monkeypatch, grid):
    """fillh8_kernel's own edges (round 6).  The wide mode takes a PIECE per entry: runs of wide rows longer than the 64
    records a wavefront holds (query and target both unaligned for 150 rows: three pieces), rows wider than a run's 576
    columns (one row through the ring in global memory, then back to narrow rows from there), consecutive wide rows that
    start more than a column apart (blocks of one and two bases behind target gaps), a wide row 0 (a first block far off the
    origin's diagonal), a wide LAST row, N in a wide row's query base and target.  The list is a QUEUE: with a grid of one
    workgroup (32 groups of eight lanes, or 4 with one group per wavefront, or 24) the 70 candidates go round several times
    and a group takes wide and narrow candidates one after the other; by default a short list uses one group per wavefront.
    Everything against the oracle."""
    if grid:
        monkeypatch.setenv("BLASR_HIP_H8_GRID", grid[0])
        monkeypatch.setenv("BLASR_HIP_H8_GROUPS", grid[1])
    rng = np.random.default_rng(20261007)
    shapes = [[(150, 200)], [(70, 400)], [(150, 200), (3, 3), (120, 330)], [(0, 900)], [(0, 640), (0, 20)], [(40, 1300), (0, 0)],
              [(0, 200)] * 5, [(5, 300), (0, 130), (200, 0)], [(64, 180)], [(65, 180)], [(129, 260)], [(0, 113)], [(0, 112)],
              [(300, 300)], [(10, 576 - 33)], [(10, 577 - 33)], [(0, 7)], [(3, 3)] * 6]
    b, want = blasr_amd.RefineBatch(), []

    def add(q, t, blocks):
        b.add(q, t, blocks)
        want.append(oracle.affine_guided_align(q, t, blocks))
    for rep in range(3):
        for k, gaps in enumerate(shapes):
            block = int(rng.choice([1, 2, 12, 40, 90])) if rep else 40
            q, t, blocks = _guide_with_gaps(rng, gaps, block=block, n_base=(k % 4 == 1))
            add(q, t, blocks)
    # a wide row 0: the guide starts 300 columns into the target (tStart - qStart = 300 > the band)
    for off in (150, 300, 700):
        q, t, blocks = _guide_with_gaps(rng, [(0, 40)], block=60)
        t = np.ascontiguousarray(np.concatenate([synth.random_genome(off, int(rng.integers(1, 1 << 30))), t]))
        blocks = blocks.copy()
        blocks[:, 1] += off
        add(q, t, blocks)
    # a wide last row: the last block is one base behind a target gap
    for gt in (200, 600, 1000):
        q, t, blocks = _guide_with_gaps(rng, [(0, 30), (2, gt)], block=50)
        blocks = blocks.copy()
        blocks[-1, 2] = 1
        add(np.ascontiguousarray(q[: blocks[-1, 0] + 1]), np.ascontiguousarray(t[: blocks[-1, 1] + 1]), blocks)
    # plain noisy pairs in between (narrow candidates in the same queue)
    for it in range(10):
        q, t = synth.make_pair(rng, int(rng.integers(100, 2500)))
        s = oracle.sdp_align(q, t)
        bl = s["blocks"]
        if len(bl):
            add(np.ascontiguousarray(q[s["qPos"]: s["qPos"] + bl[-1, 0] + bl[-1, 2]]),
                np.ascontiguousarray(t[s["tPos"]: s["tPos"] + bl[-1, 1] + bl[-1, 2]]), bl)
    got = ctx.refine(b)
    assert len(got) >= 70
    for i, (g, w) in enumerate(zip(got, want)):
        assert g.status == 0, i
        assert first_difference(g.as_dict(), w, REFINE_KEYS) is None, (i, grid)
    names = [k["name"] for k in ctx.last_kernels()]
    assert any("fillh8_kernel" in n for n in names) and not any("fill8_kernel" in n or "fillp_kernel" in n for n in names), names


@pytest.mark.parametrize("band", [0, 1, 39, 40, 55, 56, 60, 120, 300])
def test_fillh8_bands_from_none_to_every_row_wide(ctx, oracle, band):
    """The band decides what `fillh8_kernel` sees: with 39 / 40 columns either side a plain row has 79 / 81 cells (two and three
    passes of 40), with 55 / 56 it has 111 / 113 -- the last narrow width and the first WIDE one: from there on every row of every
    candidate is a wide row (runs as long as the candidate, row 0 included), and with 300 a row is beyond what a run holds (one row
    at a time through the ring in global memory).  Batches of noisy pairs and guides with gaps, against the oracle."""
    rng = np.random.default_rng(3100 + band)
    b, want = blasr_amd.RefineBatch(), []
    for it in range(14):
        if it % 3 == 2:
            q, t, blocks = _guide_with_gaps(rng, [(int(rng.integers(0, 90)), int(rng.integers(0, 400)))] * int(rng.integers(1, 4)),
                                            block=int(rng.integers(5, 70)), n_base=(it % 2 == 0))
        else:
            q, t = synth.make_pair(rng, int(rng.integers(60, 1200)), flank=(0, int(rng.integers(0, 60))))
            s = oracle.sdp_align(q, t)
            blocks = s["blocks"]
            if len(blocks) == 0:
                continue
            q = np.ascontiguousarray(q[s["qPos"]: s["qPos"] + blocks[-1, 0] + blocks[-1, 2]])
            t = np.ascontiguousarray(t[s["tPos"]: s["tPos"] + blocks[-1, 1] + blocks[-1, 2]])
        b.add(q, t, blocks)
        want.append(oracle.affine_guided_align(q, t, blocks, band=band))
    got = ctx.refine(b, band=band)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g.status == 0, (band, i)
        assert first_difference(g.as_dict(), w, REFINE_KEYS) is None, (band, i)
    assert any("fillh8_kernel" in k["name"] for k in ctx.last_kernels())


@pytest.mark.parametrize("switch", [None, "BLASR_HIP_NO_HYBRID"])
def test_guides_with_several_gaps_share_wavefronts(ctx, oracle, monkeypatch, switch):
    """What the hybrid fill kernel is for (candidates whose rows are narrow except where the guide jumps): guides of
    3-7 short blocks with gaps of tens to thousands of columns and rows between them, in either direction and back to
    back, unequal lengths, so that the four candidates of a wavefront meet their wide rows at different times, some in
    consecutive rows, some in the first rows; also a batch of fewer candidates than a wavefront holds."""
    if switch:
        monkeypatch.setenv(switch, "1")
    rng = np.random.default_rng(4242)
    for n_cand in (3, 41):
        b, want = blasr_amd.RefineBatch(), []
        for it in range(n_cand):
            nb = int(rng.integers(3, 8))
            q = t = int(rng.integers(0, 30))
            blocks = []
            for k in range(nb):
                length = int(rng.integers(8, 40))
                blocks.append([q, t, length])
                step = int(rng.integers(20, 400))
                gap = int(rng.choice([0, 30, 120, 200, 700, 2500, 5000])) if k + 1 < nb else 0
                if it % 7 == 0 and k == 0:
                    gap = 1800  # a jump right after the first block: the first rows are the wide ones
                kind = int(rng.integers(0, 3))
                q += length + step + (gap if kind == 1 else 0)
                t += length + step + (gap if kind == 0 else 0) + (gap // 3 if kind == 2 else 0)
            qs = synth.random_genome(q + 30, int(rng.integers(1, 1 << 30)))
            ts = synth.random_genome(t + 30, int(rng.integers(1, 1 << 30)))
            for (bq, bt, bl) in blocks:  # the blocks are exact matches, as SDP's are
                ts[bt: bt + bl] = qs[bq: bq + bl]
            bl = np.array(blocks, dtype=np.int32)
            b.add(qs, ts, bl)
            want.append(oracle.affine_guided_align(qs, ts, bl))
        got = ctx.refine(b)
        for i, (g, w) in enumerate(zip(got, want)):
            assert g.status == 0, (n_cand, i)
            assert first_difference(g.as_dict(), w, REFINE_KEYS) is None, (n_cand, i)
    if switch:
        monkeypatch.delenv(switch)
