"""next-1, host half: blasr_hip_find_intervals -- RemoveOverlappingAnchors, FindMaxIncreasingInterval for
both strands, the WeightedIntervalSet -- against the reference's OWN functions through oracle/_ref (every interval
and its p-value, bit for bit, several parameter sets, random "stale" elements past the lists' ends), against committed
golden intervals of the reference (tests/golden/intervals_golden.npz), and END TO END against the candidate intervals
the reference BINARY printed with -V 1 for the 80 noisy reads of tests/golden/m4_noisy_golden.npz (anchors by the
oracle's MapReadToGenome on a suffix array the reference's sawriter builds here)."""
import ctypes as C
import os

import numpy as np
import pytest

from blasr_amd import hip
import blasr_testlib as T

HERE = os.path.dirname(os.path.abspath(__file__))
have_ref = os.path.exists(T.SAWRITER) and os.path.exists(T.REFMAPPER_SO)
DEFAULT_ANCHORS = (14, 0, 1, 1000, 0, 0, 1)  # MappingParameters.h:244, :309; AnchorParameters.h:34-37


def as_rows(ivs):
    if not len(ivs):
        return np.zeros((0, 5), np.uint32)
    return np.stack([ivs["start"], ivs["end"], ivs["q_start"], ivs["q_end"], ivs["strand"].astype(np.uint32)], axis=1)


def pack_lists(lists):
    flat = [x for pair in lists for x in pair]
    off = np.concatenate([[0], np.cumsum([len(x) for x in flat])]).astype(np.int64)
    tri = np.concatenate(flat) if sum(len(x) for x in flat) else np.zeros((0, 3), np.uint32)
    a = np.zeros(len(tri), hip.ANCHOR_DTYPE)
    a["t"], a["q"], a["l"] = tri[:, 0], tri[:, 1], tri[:, 2]
    return a, off


def test_intervals_against_the_golden_intervals_of_the_reference():
    z = np.load(os.path.join(HERE, "golden", "intervals_golden.npz"))
    genome, starts, reads = z["genome"], z["starts"], z["reads"]
    a = np.zeros(len(z["anchors"]), hip.ANCHOR_DTYPE)
    a["t"], a["q"], a["l"] = z["anchors"][:, 0], z["anchors"][:, 1], z["anchors"][:, 2]
    st = np.zeros(len(z["stale"]), hip.ANCHOR_DTYPE)
    st["t"], st["q"], st["l"] = z["stale"][:, 0], z["stale"][:, 1], z["stale"][:, 2]
    ct = hip.count_tuples(genome, int(z["tuple_size"]))
    k = c = 0
    for kw in (dict(), dict(n_candidates=3), dict(min_interval=30), dict(remove_contained_intervals=1),
               dict(n_candidates=1, min_interval=0)):
        ivs, off, pv = hip.find_intervals(genome, starts, ct, reads, a, z["list_off"], hip.interval_params(threads=1 + c % 3, **kw), st)
        for r in range(len(reads)):
            n = int(z["counts"][c])
            assert np.array_equal(as_rows(ivs[int(off[r]): int(off[r + 1])]), z["intervals"][k: k + n]), (kw, r)
            assert np.array_equal(pv[int(off[r]): int(off[r + 1])], z["p_values"][k: k + n]), (kw, r)
            k += n
            c += 1
    assert k == len(z["intervals"]) and k > 100


def test_count_table_and_edges():
    g = np.frombuffer(b"ACGTNACGTACGTTTNNACGN", dtype=np.uint8)
    ct = hip.count_tuples(g, 2)
    want = np.zeros(16, np.int32)
    s = g.tobytes()
    for i in range(len(s) - 1):
        if all(c in b"ACGT" for c in s[i:i + 2]):
            want[b"ACGT".index(s[i]) * 4 + b"ACGT".index(s[i + 1])] += 1
    assert np.array_equal(ct, want)
    # no reads; reads without anchors; a single anchor per list (StoreLargestIntervals' (0, 0) range chains nothing)
    starts = np.array([0, len(g)], np.uint32)
    ivs, off, pv = hip.find_intervals(g, starts, ct, np.zeros(0, hip.READ_DTYPE), np.zeros(0, hip.ANCHOR_DTYPE), np.zeros(1, np.int64))
    assert len(ivs) == 0 and off.tolist() == [0]
    reads = np.zeros(2, hip.READ_DTYPE)
    reads[0] = (0, 500, 0, 500, 0)
    reads[1] = (500, 400, 10, 390, 0)
    one = np.zeros(1, hip.ANCHOR_DTYPE)
    one[0] = (3, 5, 4)
    ivs, off, pv = hip.find_intervals(g, starts, ct, reads, one, np.array([0, 0, 0, 1, 1], np.int64))
    assert len(ivs) == 0 and off.tolist() == [0, 0, 0]
    with pytest.raises(hip.BlasrHipError):
        hip.find_intervals(g, starts, ct, reads, one, np.array([0, 1, 0, 1, 1], np.int64))  # offsets that go backwards


@pytest.mark.skipif(not have_ref, reason="reference tools not built")
def test_intervals_against_the_reference(tmp_path):
    oracle = T.load_oracle()
    ref = C.CDLL(T.REFMAPPER_SO)
    ref.ref_ct_open.restype = C.c_void_p
    rng = np.random.default_rng(21)
    total = 0
    for it, (blt, min_prefix, ps, glen, rlen) in enumerate([(8, 8, (12, 0, 1, 1000, 0, 0, 1), 150000, (500, 4000)),
                                                           (None, 8, DEFAULT_ANCHORS, 200000, (3000, 9000)),
                                                           (8, 8, (10, 1, 1, 50, 0, 0, 0), 90000, (60, 900))]):
        fasta, reads, pool = T.anchor_genome_case(rng, 20, genome_len=glen, read_len=rlen)
        genome, starts, sa, _ = T.write_suffix_array(tmp_path, fasta, blt)
        lists = []
        for r in range(len(reads)):
            off, L = int(reads[r]["off"]), int(reads[r]["length"])
            rd = np.ascontiguousarray(pool[off: off + L])
            s0, s1 = int(reads[r]["subread_start"]), int(reads[r]["subread_end"])
            lists.append([T.oracle_anchors(oracle, genome, sa, rd, s0, s1, st, min_prefix, ps, True) for st in (0, 1)])
        if it == 2:  # tiny lists too: nothing for RemoveOverlappingAnchors to remove, so the stale element is the caller's
            lists[3] = [lists[3][0][:2], lists[3][1][:1]]
            lists[5] = [lists[5][0][:0], lists[5][1][:3]]
        a, off = pack_lists(lists)
        ct = hip.count_tuples(genome, 8)
        h = ref.ref_ct_open(genome.ctypes.data_as(C.c_void_p), len(genome), 8)
        for kw in (dict(), dict(n_candidates=3), dict(min_interval=30), dict(remove_contained_intervals=1),
                   dict(n_candidates=1), dict(indel_rate=0.05), dict(max_p_value=-40.0)):
            stale = np.zeros(2 * len(reads), hip.ANCHOR_DTYPE)
            stale["t"] = rng.integers(0, len(genome), len(stale))
            stale["q"] = rng.integers(0, 3000, len(stale))
            stale["l"] = rng.integers(8, 30, len(stale))
            p = hip.interval_params(**kw)
            ivs, ioff, pv = hip.find_intervals(genome, starts, ct, reads, a, off, p, stale)
            for r in range(len(reads)):
                st3 = np.stack([stale["t"][2 * r: 2 * r + 2], stale["q"][2 * r: 2 * r + 2], stale["l"][2 * r: 2 * r + 2]], axis=1)
                want, wpv = T.ref_find_intervals(ref, h, genome, starts, int(reads[r]["length"]), int(reads[r]["subread_start"]),
                                                 int(reads[r]["subread_end"]), lists[r][0], lists[r][1], st3,
                                                 n_candidates=p.n_candidates, min_interval=p.min_interval,
                                                 remove_contained=p.remove_contained_intervals, indel_rate=p.indel_rate,
                                                 max_p_value=p.max_p_value)
                assert np.array_equal(as_rows(ivs[int(ioff[r]): int(ioff[r + 1])]), want), (it, kw, r)
                assert np.array_equal(pv[int(ioff[r]): int(ioff[r + 1])], wpv), (it, kw, r)
                total += len(want)
        ref.ref_ct_close(C.c_void_p(h))
    assert total > 600


@pytest.mark.skipif(not os.path.exists(T.SAWRITER), reason="oracle/_ref/sawritermc not built")
def test_anchors_and_intervals_reproduce_the_reference_binarys_candidate_intervals(tmp_path):
    """blasrmc -V 1 printed 125 candidate intervals for the 80 noisy reads of the end-to-end fixture (its own suffix
    array, built on the fly: lookup prefix 8; -minMatch 14).  MapReadToGenome (oracle) + blasr_hip_find_intervals give
    the same 125, field for field."""
    oracle = T.load_oracle()
    z = np.load(os.path.join(HERE, "golden", "m4_noisy_golden.npz"))
    fasta = b">contig0\n" + bytes(z["contig0"]) + b"\n>contig1\n" + bytes(z["contig1"]) + b"\n"
    genome, starts, sa, _ = T.write_suffix_array(tmp_path, fasta, 8)
    lens = z["read_lens"].astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)])
    reads = np.zeros(len(lens), hip.READ_DTYPE)
    lists = []
    for r in range(len(lens)):
        reads[r] = (offs[r], lens[r], 0, lens[r], 0)
        rd = np.ascontiguousarray(z["reads"][offs[r]: offs[r + 1]])
        lists.append([T.oracle_anchors(oracle, genome, sa, rd, 0, len(rd), st, 8, DEFAULT_ANCHORS, True) for st in (0, 1)])
    a, off = pack_lists(lists)
    ivs, ioff, _ = hip.find_intervals(genome, starts, hip.count_tuples(genome, 8), reads, a, off)
    assert len(ivs) == len(z["intervals"]) == 125
    for r in range(len(reads)):
        assert np.array_equal(as_rows(ivs[int(ioff[r]): int(ioff[r + 1])]).astype(np.int64), z["intervals"][z["owner"] == r]), r


def test_the_chains_of_the_intervals():
    """blasr_hip_find_intervals_chains: the same intervals, and with each its chained anchors (WeightedInterval::matches,
    what -rbao reads): a subsequence of the strand's anchor list that advances in q and in t, from the interval's first
    base to its last (FindMaxInterval.h:826-831), and whose lengths add up to at least minInterval."""
    z = np.load(os.path.join(HERE, "golden", "intervals_golden.npz"))
    genome, starts, reads = z["genome"], z["starts"], z["reads"]
    a = np.zeros(len(z["anchors"]), hip.ANCHOR_DTYPE)
    a["t"], a["q"], a["l"] = z["anchors"][:, 0], z["anchors"][:, 1], z["anchors"][:, 2]
    st = np.zeros(len(z["stale"]), hip.ANCHOR_DTYPE)
    st["t"], st["q"], st["l"] = z["stale"][:, 0], z["stale"][:, 1], z["stale"][:, 2]
    ct = hip.count_tuples(genome, int(z["tuple_size"]))
    list_off = z["list_off"]
    seen = 0
    for kw in (dict(), dict(n_candidates=3), dict(remove_contained_intervals=1)):
        p = hip.interval_params(threads=2, **kw)
        ivs, off, pv = hip.find_intervals(genome, starts, ct, reads, a, list_off, p, st)
        ivs2, off2, pv2, chains, chain_off = hip.find_intervals(genome, starts, ct, reads, a, list_off, p, st, chains=True)
        assert np.array_equal(ivs, ivs2) and np.array_equal(off, off2) and np.array_equal(pv, pv2)
        assert len(chain_off) == len(ivs) + 1 and chain_off[0] == 0 and chain_off[-1] == len(chains)
        for r in range(len(reads)):
            for k in range(int(off[r]), int(off[r + 1])):
                c = chains[int(chain_off[k]): int(chain_off[k + 1])].astype(np.int64)
                assert len(c) >= 1
                iv = ivs[k]
                assert c[0, 0] == iv["start"] and c[-1, 0] + c[-1, 2] == iv["end"]
                assert c[0, 1] == iv["q_start"] and c[-1, 1] + c[-1, 2] == iv["q_end"]
                assert np.all(np.diff(c[:, 0]) > 0) and np.all(np.diff(c[:, 1]) > 0)
                assert c[:, 2].sum() >= int(p.min_interval) or len(c) > 1
                lst = 2 * r + int(iv["strand"])
                mine = {(int(t), int(q)) for t, q in zip(a["t"][int(list_off[lst]): int(list_off[lst + 1])],
                                                         a["q"][int(list_off[lst]): int(list_off[lst + 1])])}
                assert all((int(t), int(q)) in mine for t, q, _ in c)
                seen += 1
    assert seen > 60
