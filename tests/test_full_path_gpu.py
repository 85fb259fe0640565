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
