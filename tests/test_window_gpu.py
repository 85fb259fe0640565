"""a-1 on the device: blasr_hip_upload_genome + blasr_hip_windows_* against the golden vectors of the
reference's AlignIntervals and against the oracle, and the windows -> SDP chain on the device."""
import os

import numpy as np
import pytest

import blasr_amd
from blasr_testlib import WINDOW_FIELDS, MapperOracle, random_interval_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WIN_KEYS = ("t_aligned_seq_pos", "t_aligned_seq_length", "t_strand", "t_index", "t_length", "q_aligned_seq_pos",
            "q_aligned_seq_length")


@pytest.fixture(scope="module")
def ctx():
    c = blasr_amd.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def mg():
    return np.load(os.path.join(ROOT, "tests", "golden", "mapper_golden.npz"))


def pack(cases):
    """cases: list of (read, subread, iv) -> INTERVAL / READ arrays and the read pool."""
    reads = np.zeros(len(cases), dtype=blasr_amd.READ_DTYPE)
    ivs = np.zeros(len(cases), dtype=blasr_amd.INTERVAL_DTYPE)
    pool, at = [], 0
    for i, (read, subread, iv) in enumerate(cases):
        reads[i] = (at, len(read), subread[0], subread[1], 0)
        ivs[i] = (iv[0], iv[1], iv[2], iv[3], i, iv[4])
        pool.append(read)
        at += len(read)
    return ivs, reads, np.concatenate(pool)


def check(out, cases, want):
    seq, q_off, q_len, t_off, t_len, win = out
    for i, (read, subread, iv) in enumerate(cases):
        w, tseq = want[i]
        assert [int(win[i][k]) for k in WIN_KEYS] == [w[f] for f in WINDOW_FIELDS], i
        assert int(win[i]["q_length"]) == len(read)
        q = seq[q_off[i]: q_off[i] + q_len[i]]
        t = seq[t_off[i]: t_off[i] + t_len[i]]
        assert np.array_equal(q, read[w["qAlignedSeqPos"]: w["qAlignedSeqPos"] + w["qAlignedSeqLength"]]), i
        assert np.array_equal(t, tseq), i


def test_windows_match_golden(ctx, mg):
    genome = mg["genome"]
    for with_contigs in (False, True):
        idx = [i for i in range(int(mg["n_windows"])) if (len(mg["w%d_contigs" % i]) > 0) == with_contigs]
        groups = {}
        for i in idx:  # one upload per contig table
            groups.setdefault(mg["w%d_contigs" % i].tobytes(), []).append(i)
        for key, members in groups.items():
            contigs = mg["w%d_contigs" % members[0]]
            ctx.upload_genome(genome, contigs if len(contigs) else None)
            cases, want = [], []
            for i in members:
                inp = mg["w%d_in" % i]
                cases.append((mg["w%d_read" % i], (int(inp[0]), int(inp[1])), tuple(int(v) for v in inp[2:7])))
                want.append((dict(zip(WINDOW_FIELDS, (int(v) for v in mg["w%d_win" % i]))), mg["w%d_tseq" % i]))
            ivs, reads, pool = pack(cases)
            check(ctx.windows(ivs, reads, pool), cases, want)


def test_windows_match_oracle_random_and_chain_into_sdp(ctx, oracle):
    from blasr_amd import synth
    mo = MapperOracle()
    rng = np.random.default_rng(99)
    genome = synth.random_genome(50000, rng)
    ctx.upload_genome(genome)
    cases, want = [], []
    for it in range(48):
        read, subread, iv, _ = random_interval_case(rng, genome, 1 + 3 * it)  # never a contig table here
        cases.append((read, subread, iv))
        want.append(mo.window(genome, len(read), subread, iv, None))
    ivs, reads, pool = pack(cases)
    out = ctx.windows(ivs, reads, pool)
    check(out, cases, want)
    # the arrays are an SDP batch: run it and compare with the oracle's SDPAlign on the oracle's windows
    seq, q_off, q_len, t_off, t_len, _ = out
    pairs = [(seq[q_off[i]: q_off[i] + q_len[i]], seq[t_off[i]: t_off[i] + t_len[i]]) for i in range(0, 48, 4)]
    res = ctx.sdp(pairs)
    for (q, t), r in zip(pairs, res):
        o = oracle.sdp_align(q, t)
        assert (r["qPos"], r["tPos"], r["score"]) == (o["qPos"], o["tPos"], o["score"])
        assert np.array_equal(r["blocks"], o["blocks"])


def test_windows_on_a_genome_beyond_two_gigabases():
    """Coordinates are DNALength = 32-bit unsigned in the reference (Types.h); a human-sized genome puts most of
    them above 2^31.  3.2 G bases (a random block of prime length, repeated, so that an offset wrong by a power of
    two lands on different bases), windows all over it, with and without a contig table, both strands."""
    from blasr_amd import synth
    mo = MapperOracle()
    rng = np.random.default_rng(2024)
    block = synth.random_genome(67108859, 7)
    genome = np.tile(block, 48)
    assert 2 ** 31 < len(genome) < 2 ** 32
    big = blasr_amd.Context(0)
    cuts = np.array([0, 900_000_011, 2_147_483_000, 2_147_484_500, 3_000_000_019, len(genome)], dtype=np.uint32)
    try:
        for contigs in (None, cuts):
            big.upload_genome(genome, contigs)
            cases, want = [], []
            # (its that are multiples of 3 / 7 / 11 make contig tables or genome-long intervals: not here)
            for it, kind in enumerate([i for i in range(1, 200) if i % 3 and i % 7 and i % 11][:40]):
                read, subread, iv, _ = random_interval_case(rng, genome, kind)
                if it % 5 == 0:  # straddle 2^31 (and, with the table, a contig end next to it)
                    shift = 2 ** 31 - (iv[0] + iv[1]) // 2
                    iv = (iv[0] + shift, iv[1] + shift, iv[2], iv[3], iv[4])
                elif it % 5 == 1:  # the top of the coordinate range
                    shift = 3_150_000_000 + 1000 * it - iv[0]
                    iv = (iv[0] + shift, iv[1] + shift, iv[2], iv[3], iv[4])
                cases.append((read, subread, iv))
                want.append(mo.window(genome, len(read), subread, iv, contigs))
            assert sum(1 for _, _, iv in cases if iv[0] > 2 ** 31) >= 10 and sum(1 for _, _, iv in cases if iv[0] < 2 ** 31 < iv[1]) >= 8
            ivs, reads, pool = pack(cases)
            check(big.windows(ivs, reads, pool), cases, want)
    finally:
        big.close()


def test_windows_errors(ctx):
    from blasr_amd import synth
    rng = np.random.default_rng(5)
    genome = synth.random_genome(5000, rng)
    fresh = blasr_amd.Context(0)
    read, subread, iv, _ = random_interval_case(rng, genome, 1)
    ivs, reads, pool = pack([(read, subread, iv)])
    with pytest.raises(blasr_amd.BlasrHipError) as e:  # no genome uploaded
        fresh.windows(ivs, reads, pool)
    assert e.value.code == blasr_amd.hip.ERR_INVALID
    fresh.upload_genome(genome)
    with pytest.raises(blasr_amd.BlasrHipError) as e:  # output arena too small
        fresh.windows(ivs, reads, pool, seq_capacity=100)
    assert e.value.code == blasr_amd.hip.ERR_CAPACITY
    assert len(fresh.windows(ivs[:0], reads, pool)[1]) == 0
    fresh.close()
