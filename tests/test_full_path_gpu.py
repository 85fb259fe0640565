"""The whole path on the device, stage feeding stage as MapRead runs them -- candidate windows from the resident
genome (a-1), SDPAlign + statistics (a-2..a-5, a-9), ordering / pruning on the host (a-10), AffineGuidedAlign +
traceback + statistics (a-6..a-9) -- against the oracle, for a few hundred reads of mixed length and error rate."""
import numpy as np
import pytest

import blasr_amd
from blasr_amd import synth
from blasr_testlib import MapperOracle
from golden_cases import REFINE_KEYS, first_difference

pytestmark = pytest.mark.gpu


def test_windows_sdp_filters_refine_against_oracle(oracle):
    rng = np.random.default_rng(2026)
    genome = synth.random_genome(400000, 17)
    mo = MapperOracle()
    ctx = blasr_amd.Context(0)
    ctx.upload_genome(genome)
    n = 240
    reads = np.zeros(n, dtype=blasr_amd.READ_DTYPE)
    ivs = np.zeros(n, dtype=blasr_amd.INTERVAL_DTYPE)
    pool, at, meta = [], 0, []
    for i in range(n):
        rl = int(rng.choice([150, 400, 900, 2000, 4000, 8000, 12000]))
        err = [(0.11, 0.04, 0.01), (0.05, 0.02, 0.01), (0.10, 0.08, 0.02)][i % 3]
        pos = int(rng.integers(0, len(genome) - rl))
        strand = int(rng.integers(0, 2))
        tmpl = genome[pos:pos + rl]
        if strand:
            tmpl = synth.revcomp(tmpl)
        read = np.ascontiguousarray(synth.mutate(tmpl, rng, *err))
        L = len(read)
        ua0, ua1 = int(rng.integers(0, min(60, L // 4))), int(rng.integers(0, min(60, L // 4)))
        g0, g1 = pos + int(rng.integers(0, min(80, rl // 4))), pos + rl - int(rng.integers(0, min(80, rl // 4)))
        reads[i] = (at, L, 0, L, 0)
        ivs[i] = (g0, g1, ua0, L - ua1, i, strand)
        pool.append(read)
        at += L
        meta.append((read, (0, L), (g0, g1, ua0, L - ua1, strand)))
    seq, q_off, q_len, t_off, t_len, win = ctx.windows(ivs, reads, np.concatenate(pool))
    pairs = []
    for i, (read, subread, iv) in enumerate(meta):
        w, tseq = mo.window(genome, len(read), subread, iv)
        q, t = seq[q_off[i]: q_off[i] + q_len[i]], seq[t_off[i]: t_off[i] + t_len[i]]
        assert np.array_equal(t, tseq) and np.array_equal(q, read), i
        pairs.append((q, t))
    sdp = ctx.sdp(pairs)
    batch, want = blasr_amd.RefineBatch(), []
    cands = np.zeros(n, dtype=blasr_amd.CANDIDATE_DTYPE)
    for i, ((q, t), s) in enumerate(zip(pairs, sdp)):
        o = oracle.sdp_align(q, t)
        assert (s["status"], s["score"], s["qPos"], s["tPos"]) == (0, o["score"], o["qPos"], o["tPos"]), i
        assert np.array_equal(s["blocks"], o["blocks"]), i
        st = oracle.compute_alignment_stats(q, t, o["qPos"], o["tPos"], o["blocks"])
        assert [s["stats"][k] for k in ("score", "nMatch", "nMismatch", "nIns", "nDel")] == \
               [st[k] for k in ("score", "nMatch", "nMismatch", "nIns", "nDel")], i
        bl = s["blocks"]
        cands[i] = (s["stats"]["score"], s["tPos"], int(win[i]["t_aligned_seq_pos"]),
                    (bl[-1, 1] + bl[-1, 2]) if len(bl) else 0, int(win[i]["t_index"]), int(win[i]["t_strand"]),
                    st["pct"], len(bl), 0)
        if len(bl) == 0:
            continue
        qs = q[s["qPos"]: s["qPos"] + bl[-1, 0] + bl[-1, 2]]
        ts = t[s["tPos"]: s["tPos"] + bl[-1, 1] + bl[-1, 2]]
        batch.add(qs, ts, bl)
        want.append(oracle.affine_guided_align(qs, ts, bl))
    # a-10 on the whole set as if it were one read's candidates: the product's filters against the oracle's
    order = blasr_amd.sort_candidates(cands)
    assert np.array_equal(order, mo.sort_by_score(cands["score"], cands["t_pos"], cands["t_aligned_seq_pos"]))
    got = ctx.refine(batch)
    assert len(got) == len(want) >= n - 5
    for i, (g, w) in enumerate(zip(got, want)):
        assert g.status == 0, i
        assert first_difference(g.as_dict(), w, REFINE_KEYS) is None, i
    ctx.close()


def test_two_host_threads_with_their_own_contexts(oracle):
    """The C ABI's threading rule (include/blasr_hip.h): one context per host thread, any number of threads at
    once -- the reference's -nproc.  Two threads run SDP + refine on different pairs at the same time, several
    rounds each; every result must be what the oracle says, i.e. nothing is shared between contexts."""
    import threading

    rng = np.random.default_rng(515)
    sets = []
    for t in range(2):
        pairs = [synth.make_pair(rng, int(rng.integers(300, 5000)), flank=(0, int(rng.integers(0, 120))))
                 for _ in range(24)]
        sets.append(pairs)
    out, errors = [None, None], []

    def work(t):
        try:
            ctx = blasr_amd.Context(0)
            for _ in range(3):
                sdp = ctx.sdp(sets[t])
                b = blasr_amd.RefineBatch()
                keep = []
                for (q, tt), s in zip(sets[t], sdp):
                    if len(s["blocks"]) == 0:
                        continue
                    last = s["blocks"][-1]
                    qs = q[s["qPos"]: s["qPos"] + last[0] + last[2]]
                    ts = tt[s["tPos"]: s["tPos"] + last[1] + last[2]]
                    b.add(qs, ts, s["blocks"])
                    keep.append((qs, ts, s["blocks"]))
                out[t] = (sdp, keep, ctx.refine(b))
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append(e)

    ths = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
    for t in range(2):
        sdp, keep, refined = out[t]
        for i, ((q, tt), s) in enumerate(zip(sets[t], sdp)):
            o = oracle.sdp_align(q, tt)
            assert (s["status"], s["score"], s["qPos"], s["tPos"]) == (0, o["score"], o["qPos"], o["tPos"]), (t, i)
            assert np.array_equal(s["blocks"], o["blocks"]), (t, i)
        assert len(refined) == len(keep) > 0
        for i, ((qs, ts, blocks), g) in enumerate(zip(keep, refined)):
            w = oracle.affine_guided_align(qs, ts, blocks)
            assert g.status == 0 and first_difference(g.as_dict(), w, REFINE_KEYS) is None, (t, i)


def test_device_results_through_the_output_epilogue(oracle):
    """The refine stage's op lists go straight into the epilogue functions (include/blasr_hip.h: map_qvs, cigar,
    sam_line, m4_line): the printed records made from the DEVICE's results must equal the records the oracle's
    epilogue makes from the ORACLE's alignments of the same read -- three competing candidates of one read here
    (the read against its own locus and against two mutated copies of it, a repeat)."""
    from blasr_testlib import EpilogueChecker, ops_from_alignment
    rng = np.random.default_rng(31337)
    orc = EpilogueChecker("oracle")
    ctx = blasr_amd.Context(0)
    for trial in range(4):
        q, t0 = synth.make_pair(rng, int(rng.integers(1500, 5000)), flank=(0, 200))
        targets = [t0, synth.mutate(t0, rng, 0.004, 0.004, 0.01), synth.mutate(t0, rng, 0.01, 0.01, 0.03)]
        pairs = [(q, np.ascontiguousarray(t)) for t in targets]
        sdp = ctx.sdp(pairs)
        batch, keep = blasr_amd.RefineBatch(), []
        for (qq, tt), s in zip(pairs, sdp):
            last = s["blocks"][-1]
            qs = qq[s["qPos"]: s["qPos"] + last[0] + last[2]]
            ts = tt[s["tPos"]: s["tPos"] + last[1] + last[2]]
            batch.add(qs, ts, s["blocks"])
            keep.append((qs, ts, s))
        refined = ctx.refine(batch)
        dev_alns, orc_alns = [], []
        for (qs, ts, s), r in zip(keep, refined):
            o = oracle.affine_guided_align(qs, ts, s["blocks"])
            o_ops = np.array([(ln << 2) | ty for ty, ln in ops_from_alignment(o)], dtype=np.uint32)
            q_span = lambda ops: int(sum(int(w >> 2) for w in ops if w & 3 != 2))  # noqa: E731
            dev_alns.append(dict(q=qs, t=ts, ops=r.ops, q_pos=r.qPos, t_pos=r.tPos, q_align_start=s["qPos"] + r.qPos,
                                 q_align_end=s["qPos"] + r.qPos + q_span(r.ops), q_strand=0, q_length=len(q),
                                 n_match=r.nMatch))
            orc_alns.append(dict(q=qs, t=ts, ops=o_ops, q_pos=o["qPos"], t_pos=o["tPos"],
                                 q_align_start=s["qPos"] + o["qPos"], q_align_end=s["qPos"] + o["qPos"] + q_span(o_ops),
                                 q_strand=0, q_length=len(q), n_match=o["nMatch"]))
        got_qv, _ = blasr_amd.map_qvs(dev_alns)
        want_qv, _ = orc.store_map_qvs(orc_alns)
        assert got_qv == want_qv, trial
        for i, (d, o, r) in enumerate(zip(dev_alns, orc_alns, refined)):
            kw = dict(clipping=blasr_amd.hip.CLIP_SOFT, read_length=len(q), q_align_start=d["q_align_start"],
                      q_align_end=d["q_align_end"])
            cig, clips = blasr_amd.cigar(d["q"], d["t"], d["q_pos"], d["t_pos"], ops=d["ops"], **kw)
            want_cig, want_clips = orc.cigar(o["q"], o["t"], o["q_pos"], o["t_pos"], ops=o["ops"], clipping=1,
                                             read_length=len(q), q_align_start=o["q_align_start"],
                                             q_align_end=o["q_align_end"])
            assert (cig, clips) == (want_cig, want_clips), (trial, i)
            t_span = int(sum(int(w >> 2) for w in d["ops"] if w & 3 != 1))
            line = blasr_amd.m4_line("read/%d" % trial, "ref", r.score, float(r.pctSimilarity), 0, d["q_align_start"],
                                     d["q_align_end"], len(q), 0, d["t_pos"], d["t_pos"] + t_span, len(d["t"]), got_qv[i])
            want_line, _ = orc.m4_line("read/%d" % trial, "ref", r.score, float(r.pctSimilarity), 0, 0, d["q_align_start"],
                                       len(q), 0, 0, d["t_pos"], len(d["t"]), want_qv[i],
                                       [(0, 0, 1), (d["q_align_end"] - d["q_align_start"] - 1, t_span - 1, 1)])
            assert line == want_line, (trial, i)
