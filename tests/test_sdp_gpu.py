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
    for it in range(120):
        q, t = synth.make_pair(rng, int(rng.integers(20, 1500)), flank=(0, int(rng.integers(0, 100))),
                               ins=float(rng.uniform(0, 0.15)), dele=float(rng.uniform(0, 0.1)),
                               sub=float(rng.uniform(0, 0.05)))
        m = SMRT_MATRIX.copy()
        if it % 3 == 0:
            m = rng.integers(-6, 8, size=(5, 5)).astype(np.int32)
            m[np.arange(4), np.arange(4)] = -rng.integers(1, 8, size=4)
        osc = OSc(m, int(rng.integers(1, 9)), int(rng.integers(1, 9)), int(rng.integers(0, 60)), int(rng.integers(0, 3)))
        osp = OSp(k=int(rng.integers(4, 14)), sdp_ins=int(rng.integers(1, 12)), sdp_del=int(rng.integers(1, 12)),
                  indel_rate=float(rng.uniform(0.1, 1.0)), align_type=int(rng.integers(0, 2)),
                  # the second half of the sweep: the two non-default branches of SDPAlign (:437-474 / :546-598 front
                  # and tail fill under extendFrontByLocalAlignment, :497-499 no fills without detailedAlignment)
                  detailed=1 if it < 60 else int(rng.integers(0, 2)), extend_front=0 if it < 60 else int(rng.integers(0, 2)),
                  prefix=int(rng.choice([0, 20, 50])), recurse=int(rng.integers(0, 4)),
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


def test_extend_front_and_coarse_alignment_on_noisy_reads(ctx, oracle):
    """-extendFrontAlignment (front and tail fills of the Local top-level call, the tail with word size k / 2) and
    detailedSDPAlignment = 0 (no fills between blocks) on PacBio-like reads with unaligned flanks."""
    from blasr_testlib import SdpParams as OSp
    rng = np.random.default_rng(204)
    pairs = [synth.make_pair(rng, int(rng.integers(1500, 9000)), flank=(50, 600)) for _ in range(8)]
    for ef, det in ((1, 1), (0, 0), (1, 0)):
        osp = OSp(extend_front=ef, detailed=det)
        got = ctx.sdp(pairs, _sdp_params(osp))
        for i, ((q, t), g) in enumerate(zip(pairs, got)):
            w = oracle.sdp_align(q, t, None, osp)
            st = oracle.compute_alignment_stats(q, t, w["qPos"], w["tPos"], w["blocks"])
            _check(g, w, [st["score"], st["nMatch"], st["nMismatch"], st["nIns"], st["nDel"]], (ef, det, i))


def test_unsupported_parameters_are_rejected(ctx):
    q = np.frombuffer(b"ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT", np.uint8)
    for k in (0, 33):  # (a tuple is 64 bits: the reference's own shifts are undefined beyond 32 bases)
        with pytest.raises(blasr_amd.BlasrHipError) as e:
            ctx.sdp([(q, q)], blasr_amd.SdpParams(k=k))
        assert e.value.code == blasr_amd.hip.ERR_UNSUPPORTED


def test_words_of_up_to_32_bases(ctx, oracle):
    """-sdpTupleSize 14 .. 32 (tuples/DNATuple.h:55-94): the sort key holds a word's first sixteen bases, the rest is
    compared on the sequences.  The reference's own results (tests/golden/long_words_golden.npz), one by one and as a batch
    of mixed word sizes is not possible -- the word size is the call's --, then random cases against the oracle."""
    from blasr_testlib import long_word_cases, long_word_golden_cases
    n = 0
    for i, (q, t, sp, want) in enumerate(long_word_golden_cases()):
        got = ctx.sdp([(q, t)], _sdp_params(sp))[0]
        assert got["status"] == 0, i
        assert (got["score"], got["qPos"], got["tPos"]) == (want["score"], want["qPos"], want["tPos"]), (i, vars(sp))
        assert np.array_equal(got["blocks"], want["blocks"]), (i, vars(sp))
        n += len(want["blocks"]) > 0
    assert n >= 30
    cases = long_word_cases(np.random.default_rng(207), 120)
    for k in sorted({sp.k for _, _, sp in cases}):  # batches by word size, the other parameters those of the batch's first case
        group = [(q, t) for q, t, sp in cases if sp.k == k]
        sp0 = next(sp for _, _, sp in cases if sp.k == k)
        for (q, t), g in zip(group, ctx.sdp(group, _sdp_params(sp0))):
            w = oracle.sdp_align(q, t, None, sp0)
            st = oracle.compute_alignment_stats(q, t, w["qPos"], w["tPos"], w["blocks"])
            _check(g, w, [st["score"], st["nMatch"], st["nMismatch"], st["nIns"], st["nDel"]], (k, len(q), len(t)))


def _repetitive_pair(rng, qlen, unit_len, copies):
    """A pair whose target holds a tandem repeat (many identical k-mers: long bucket chains, positions with
    more than three matches, ties between the prefix / middle / suffix lists)."""
    q, t = synth.make_pair(rng, qlen, flank=(0, 50))
    unit = synth.random_genome(unit_len, int(rng.integers(1, 1 << 30)))
    rep = np.tile(unit, copies)
    at_t, at_q = int(rng.integers(0, max(1, len(t) - len(rep)))), int(rng.integers(0, max(1, len(q) - len(rep))))
    t, q = t.copy(), q.copy()
    t[at_t: at_t + len(rep)] = rep[: len(t) - at_t]
    q[at_q: at_q + len(rep)] = rep[: len(q) - at_q]
    return q, t


def test_match_paths_against_oracle(ctx, oracle):
    """Every way the device finds k-mer matches -- targets <= 128 bases (registers), mid-size (sorted keys), long
    (LDS index, both size classes), word sizes below and above the index's limit, repeats with and without a
    cap on the matches per position -- against the oracle."""
    from blasr_testlib import ScoreParams as OSc, SdpParams as OSp
    rng = np.random.default_rng(204)
    osc = OSc()
    cases = []
    for qlen in (40, 100, 128, 200, 700, 1400, 1600, 3000, 9000):
        for k in (5, 8, 11, 13, 14):
            if qlen >= 3000 and k in (8, 14) and qlen != 3000:
                continue
            q, t = synth.make_pair(rng, qlen, flank=(0, int(rng.integers(0, 120))))
            cases.append((q, t, OSp(k=k, max_per_pos=int(rng.choice([0, 0, 2])), align_type=int(rng.integers(0, 2)),
                                    prefix=int(rng.choice([0, 50])))))
    for qlen, unit, copies, mpp in ((300, 3, 40, 0), (900, 7, 30, 0), (2500, 11, 40, 0), (2500, 2, 150, 5),
                                    (5000, 31, 20, 0), (5000, 1, 400, 10), (60, 2, 12, 0), (120, 5, 8, 3)):
        q, t = _repetitive_pair(rng, qlen, unit, copies)
        cases.append((q, t, OSp(k=int(rng.choice([5, 7, 11])), max_per_pos=mpp)))
    q, t = synth.make_pair(rng, 14000, flank=(0, 300))  # second LDS size class
    cases.append((q, t, OSp()))
    q, t = synth.make_pair(rng, 2000, flank=(0, 0), n_run=30)
    cases.append((q, t, OSp()))
    for it, (q, t, osp) in enumerate(cases):
        got = ctx.sdp([(q, t)], _sdp_params(osp), _score(osc))[0]
        w = oracle.sdp_align(q, t, osc, osp)
        st = oracle.compute_alignment_stats(q, t, w["qPos"], w["tPos"], w["blocks"], sc=osc)
        _check(got, w, [st["score"], st["nMatch"], st["nMismatch"], st["nIns"], st["nDel"]],
               (it, len(q), len(t), vars(osp)))


def test_batch_of_mixed_sizes_against_oracle(ctx, oracle):
    """One call, candidates of very different sizes side by side (all size classes of every kernel at once)."""
    rng = np.random.default_rng(205)
    pairs = [synth.make_pair(rng, int(n), flank=(0, int(rng.integers(0, 200))))
             for n in rng.choice([30, 90, 150, 400, 1200, 2000, 4500, 7000], size=40)]
    pairs += [_repetitive_pair(rng, 1800, 4, 60), _repetitive_pair(rng, 90, 3, 10)]
    got = ctx.sdp(pairs)
    for i, ((q, t), g) in enumerate(zip(pairs, got)):
        w = oracle.sdp_align(q, t)
        st = oracle.compute_alignment_stats(q, t, w["qPos"], w["tPos"], w["blocks"])
        _check(g, w, [st["score"], st["nMatch"], st["nMismatch"], st["nIns"], st["nDel"]], (i, len(q), len(t)))


def test_alternative_kernels_agree(ctx, oracle, monkeypatch):
    """The library picks its kernels by size class; the switches BLASR_HIP_NO_INDEX (long targets through the
    sorted-key kernels instead of the LDS index), BLASR_HIP_NO_CHAIN4 (the sweep with one invocation per
    wavefront instead of four) and BLASR_HIP_NO_SMALL4 (invocations of at most 16 fragments one per wavefront
    instead of four) force the general paths, which must give the same answers.  BLASR_HIP_MATCH_SERIAL and
    BLASR_HIP_ORDER_SERIAL put the kernels that otherwise run side by side (the matchers; the general order .. above
    kernels and the small ones) back on one stream, one after the other."""
    rng = np.random.default_rng(206)
    pairs = [synth.make_pair(rng, n, flank=(0, 150)) for n in (1700, 2600, 5200)]
    want = [oracle.sdp_align(q, t) for q, t in pairs]
    for env in ({}, {"BLASR_HIP_NO_INDEX": "1"}, {"BLASR_HIP_NO_CHAIN4": "1"}, {"BLASR_HIP_NO_SMALL4": "1"},
                {"BLASR_HIP_MATCH_SERIAL": "1"}, {"BLASR_HIP_ORDER_SERIAL": "1"},
                {"BLASR_HIP_MATCH_SERIAL": "1", "BLASR_HIP_ORDER_SERIAL": "1", "BLASR_HIP_NO_INDEX": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = ctx.sdp(pairs)
        for (q, t), g, w in zip(pairs, got, want):
            st = oracle.compute_alignment_stats(q, t, w["qPos"], w["tPos"], w["blocks"])
            _check(g, w, [st["score"], st["nMatch"], st["nMismatch"], st["nIns"], st["nDel"]], (env, len(q)))
        for k in env:
            monkeypatch.delenv(k)


def test_batch_taken_in_chunks(ctx, oracle, monkeypatch):
    """A batch too large for the memory the device can give the context is taken in pieces
    (BLASR_HIP_SDP_CHUNK_BASES forces small ones here): same records, block offsets counted through the pieces,
    and both equal the oracle's."""
    rng = np.random.default_rng(808)
    pairs = [synth.make_pair(rng, int(rng.integers(200, 3000)), flank=(0, int(rng.integers(0, 100)))) for _ in range(40)]
    whole = ctx.sdp(pairs)
    monkeypatch.setenv("BLASR_HIP_SDP_CHUNK_BASES", "9000")  # a handful of pairs per piece
    pieces = ctx.sdp(pairs)
    monkeypatch.delenv("BLASR_HIP_SDP_CHUNK_BASES")
    assert len(whole) == len(pieces) == len(pairs)
    for i, (a, b) in enumerate(zip(whole, pieces)):
        assert (a["status"], a["score"], a["qPos"], a["tPos"]) == (b["status"], b["score"], b["qPos"], b["tPos"]), i
        assert np.array_equal(a["blocks"], b["blocks"]), i
        assert a["stats"] == b["stats"], i
        w = oracle.sdp_align(*pairs[i])
        assert (b["score"], b["qPos"], b["tPos"]) == (w["score"], w["qPos"], w["tPos"]), i
        assert np.array_equal(b["blocks"], w["blocks"]), i


def test_workspace_stays_bounded_over_growing_batches():
    """Batches of growing size (what real reads are): the device memory the context holds follows the largest
    call instead of piling up an allocation per size (one region, grown to the call's high-water mark), and a
    repeated call allocates nothing."""
    rng = np.random.default_rng(31)
    ctx = blasr_amd.Context(0)
    held, bases = [], []
    for n in (4, 9, 17, 30, 30, 12):
        pairs = [synth.make_pair(rng, 3000) for _ in range(n)]
        ctx.sdp(pairs)
        held.append(ctx.held_bytes())
        bases.append(sum(len(q) + len(t) for q, t in pairs))
    assert held[4] == held[3] and held[5] == held[4]        # the same size again, and a smaller one: no growth
    assert all(b >= a for a, b in zip(held, held[1:]))
    assert held[3] < 2 * held[2] * bases[3] / bases[2] + (64 << 20)  # grows with the batch, not with the history
    # the refine stage shares the region
    before = ctx.held_bytes()
    q, t = synth.make_pair(rng, 2000)
    s = ctx.sdp([(q, t)])[0]
    batch = blasr_amd.RefineBatch()
    bl = s["blocks"]
    batch.add(q[s["qPos"]: s["qPos"] + bl[-1, 0] + bl[-1, 2]], t[s["tPos"]: s["tPos"] + bl[-1, 1] + bl[-1, 2]], bl)
    assert ctx.refine(batch)[0].status == 0
    assert ctx.held_bytes() == before
    ctx.close()


def test_unrelated_sequences_against_oracle(ctx, oracle, monkeypatch):
    """Candidate intervals on unrelated sequence (chance anchor clusters): a few random k-mer hits are chained and the
    recursion runs SDPAlign(Global) with shorter words on the large rectangles between them -- fragments scattered all
    over the rectangle, a column set of hundreds of entries touched anywhere (sdp_chain_bits_kernel: bitmap column
    set).  Local top level as AlignIntervals calls it, and Global calls on random rectangles directly; both chain
    kernels (BLASR_HIP_NO_CHAIN_BITS: the register-tail sweep) against the oracle."""
    rng = np.random.default_rng(4242)
    pairs = [(np.ascontiguousarray(synth.mutate(synth.random_genome(int(rng.integers(3000, 9000)), 500 + i), rng)),
              synth.random_genome(int(rng.integers(4000, 12000)), 600 + i)) for i in range(10)]
    want = [oracle.sdp_align(q, t) for q, t in pairs]
    def as_list(st):
        return [st["score"], st["nMatch"], st["nMismatch"], st["nIns"], st["nDel"]]

    stats = [as_list(oracle.compute_alignment_stats(q, t, w["qPos"], w["tPos"], w["blocks"])) for (q, t), w in zip(pairs, want)]
    # the scattered invocations through the bitmap sweep four per wavefront (default) and one per wavefront (2), the
    # 512-register sweep (1), and without any of them (the register-tail sweep of chain4)
    # ... and a row per step (3: sdp_chain_rows_kernel for every scattered invocation)
    for env, val in ((None, None), ("BLASR_HIP_CHAIN_SCATTER", "1"), ("BLASR_HIP_CHAIN_SCATTER", "2"),
                     ("BLASR_HIP_CHAIN_SCATTER", "3"), ("BLASR_HIP_NO_CHAIN_BITS", "1")):
        if env:
            monkeypatch.setenv(env, val)
        got = ctx.sdp(pairs)
        if env:
            monkeypatch.delenv(env)
        for i, (g, w, st) in enumerate(zip(got, want, stats)):
            _check(g, w, st, "unrelated %d %s" % (i, env))
    # Global calls with short words on random rectangles, as the recursion issues them
    from blasr_testlib import SdpParams as OSdp
    for k in (7, 5):
        rect = [(synth.random_genome(int(rng.integers(400, 2500)), 700 + 10 * k + i),
                 synth.random_genome(int(rng.integers(400, 3500)), 800 + 10 * k + i)) for i in range(8)]
        # one rectangle large enough for a column set beyond the 256 register entries (spill path of the wide sweep)
        rect.append((synth.random_genome(9000, 900 + k), synth.random_genome(12000, 950 + k)))
        if k == 7:  # and targets beyond 16384 and 65536 columns (the larger bitmap classes)
            rect.append((synth.random_genome(2500, 990), synth.random_genome(30000, 991)))
            rect.append((synth.random_genome(2000, 992), synth.random_genome(80000, 993)))
        osp = OSdp(k=k, align_type=GLOBAL, prefix=0, recurse=1)
        sp = blasr_amd.SdpParams(k=k, align_type=1, prefix=0, recurse=1)
        want = [oracle.sdp_align(q, t, None, osp) for q, t in rect]
        for mode in ("0", "1", "2", "3"):
            monkeypatch.setenv("BLASR_HIP_CHAIN_SCATTER", mode)
            got = ctx.sdp(rect, sdp=sp)
            monkeypatch.delenv("BLASR_HIP_CHAIN_SCATTER")
            for i, ((q, t), g, w) in enumerate(zip(rect, got, want)):
                st = as_list(oracle.compute_alignment_stats(q, t, w["qPos"], w["tPos"], w["blocks"]))
                _check(g, w, st, "rectangle k=%d %d mode %s" % (k, i, mode))


def test_dense_rows_through_the_row_sweep(ctx, oracle, monkeypatch):
    """sdp_chain_rows_kernel where its lanes interact: low-complexity sequence (runs of a base, two- and three-letter
    alphabets) gives rows of dozens of fragments, retired rows whose fragments delete each other's successors (the
    commuting-prefix rule has to cut), equal costs, and fragments whose `above` is the fragment just before them in the same
    row.  Global calls with short words, as the recursion issues them; every scattered invocation through the row sweep
    (BLASR_HIP_CHAIN_SCATTER=3) and through the default kernels, against the oracle."""
    from blasr_testlib import SdpParams as OSdp
    rng = np.random.default_rng(777)

    def lowcomplex(n, letters, run):
        out = np.empty(n, dtype=np.uint8)
        at = 0
        while at < n:
            c = ord("ACGT"[int(rng.integers(0, letters))])
            r = int(rng.integers(1, run + 1))
            out[at: at + r] = c
            at += r
        return out

    def as_list(st):
        return [st["score"], st["nMatch"], st["nMismatch"], st["nIns"], st["nDel"]]

    for k in (5, 7):
        rect = []
        for i in range(10):
            letters, run = (2, 3) if i % 3 == 0 else ((3, 4) if i % 3 == 1 else (4, 6))
            rect.append((lowcomplex(int(rng.integers(150, 900)), letters, run), lowcomplex(int(rng.integers(200, 1500)), letters, run)))
        rect.append((synth.random_genome(3000, 1234 + k), synth.random_genome(5000, 4321 + k)))  # ~ 3 k rows of 5 (k = 5)
        osp = OSdp(k=k, align_type=GLOBAL, prefix=0, recurse=1)
        sp = blasr_amd.SdpParams(k=k, align_type=1, prefix=0, recurse=1)
        want = [oracle.sdp_align(q, t, None, osp) for q, t in rect]
        for mode in ("3", "0"):
            monkeypatch.setenv("BLASR_HIP_CHAIN_SCATTER", mode)
            got = ctx.sdp(rect, sdp=sp)
            monkeypatch.delenv("BLASR_HIP_CHAIN_SCATTER")
            for i, ((q, t), g, w) in enumerate(zip(rect, got, want)):
                st = as_list(oracle.compute_alignment_stats(q, t, w["qPos"], w["tPos"], w["blocks"]))
                _check(g, w, st, "dense rows k=%d %d mode %s" % (k, i, mode))


def test_scattered_invocations_on_targets_of_4097_to_16384_columns(ctx, oracle, monkeypatch):
    """The bitmap sweep four per wavefront with the larger map (sdp_chain_bits4_kernel<true>: category 7 of the sweep's
    list -- few fragments on a target of 4097 .. 16384 columns): Global calls with short words on random rectangles of
    that shape, with the category (default), with half its fragment bound and without it (BLASR_HIP_BITS4_WIDE=0: a
    wavefront each), against the oracle; the kernel's record says whether it ran."""
    from blasr_testlib import SdpParams as OSdp
    rng = np.random.default_rng(20260)

    def as_list(st):
        return [st["score"], st["nMatch"], st["nMismatch"], st["nIns"], st["nDel"]]

    for k, shapes in ((7, ((2500, 12000), (900, 16384), (3000, 4097), (1200, 9000), (5000, 12000), (40, 5000))),
                      (5, ((150, 6000), (100, 9000), (60, 16000)))):
        rect = [(synth.random_genome(q, 3000 + 17 * i + k), synth.random_genome(t, 4000 + 19 * i + k)) for i, (q, t) in enumerate(shapes)]
        rect += [(synth.random_genome(int(rng.integers(300, 1500)), 5000 + i), synth.random_genome(int(rng.integers(4100, 16000)), 6000 + i))
                 for i in range(6)]
        osp = OSdp(k=k, align_type=GLOBAL, prefix=0, recurse=1)
        sp = blasr_amd.SdpParams(k=k, align_type=1, prefix=0, recurse=1)
        want = [oracle.sdp_align(q, t, None, osp) for q, t in rect]
        for bound in (None, "2000", "0"):
            if bound is not None:
                monkeypatch.setenv("BLASR_HIP_BITS4_WIDE", bound)
            got = ctx.sdp(rect, sdp=sp)
            if bound is not None:
                monkeypatch.delenv("BLASR_HIP_BITS4_WIDE")
            ran = [r for r in ctx.sdp_last_kernels() if r["name"].endswith("sdp_chain_bits4_kernel<true>")]
            assert (len(ran) == 1 and ran[0]["invocations"] > 0) == (bound != "0"), (k, bound, ran)
            for i, ((q, t), g, w) in enumerate(zip(rect, got, want)):
                st = as_list(oracle.compute_alignment_stats(q, t, w["qPos"], w["tPos"], w["blocks"]))
                _check(g, w, st, "wide bits4 k=%d %d bound %s" % (k, i, bound))
