"""The C++ MapRead driver of the library (blasr_hip_map_reads_*, blasr_amd/csrc/mapread_api.hip) and the in-library
trimming (blasr_hip_refine_from_sdp_device) against MapRead composed from the ORACLE's pieces in the reference's
order (tests/blasr_testlib.py:oracle_map_reads): windows -> SDP + statistics -> sort -> RemoveLowQualityAlignments
-> refine the survivors -> sort -> RemoveLowQualityAlignments -> RemoveOverlappingAlignments, on reads with
competing candidates (a genome with planted repeat families), every field of every surviving alignment, its aligned
sequences and its op list."""
import numpy as np
import pytest

import blasr_amd
from blasr_amd import hip
from blasr_testlib import HIT_FIELDS, MapperOracle, oracle_map_reads, repeat_genome_case

pytestmark = pytest.mark.gpu


def check_hits(hits, off, seq, ops, want):
    assert len(off) == len(want) + 1
    n_multi = 0
    for r, w in enumerate(want):
        mine = hits[int(off[r]): int(off[r + 1])]
        assert len(mine) == len(w), (r, len(mine), len(w))
        n_multi += len(w) > 1
        for h, x in zip(mine, w):
            assert int(h["read"]) == r
            for f in HIT_FIELDS:
                assert h[f] == x[f], (r, f, h[f], x[f])
            q = seq[int(h["q_off"]): int(h["q_off"]) + int(h["q_aligned_seq_length"])]
            t = seq[int(h["t_off"]): int(h["t_off"]) + int(h["t_aligned_seq_length"])]
            assert np.array_equal(q, x["q"]) and np.array_equal(t, x["t"]), r
            assert ops[int(h["ops_off"]): int(h["ops_off"]) + int(h["n_ops"])].tolist() == x["ops"], r
    return n_multi


@pytest.mark.parametrize("kw", [dict(), dict(max_gap=40), dict(n_candidates=2), dict(max_score=-2000, min_pct_identity=80.0),
                                dict(extend_alignments=1), dict(extend_alignments=1, extend_band_size=4, max_extend_dropoff=3)])
def test_map_reads_against_oracle_composition(oracle, kw):
    rng = np.random.default_rng(77)
    genome, pool, reads, ivs = repeat_genome_case(rng, 48)
    ctx = blasr_amd.Context(0)
    ctx.upload_genome(genome)
    params = ctx.map_params(**kw)
    hits, off, seq, ops = ctx.map_reads(reads, pool, ivs, params=params)
    want = oracle_map_reads(oracle, MapperOracle(), genome, None, pool, reads, ivs, n_candidates=params.n_candidates,
                            max_gap=params.max_gap, max_score=params.max_score, min_pct=params.min_pct_identity,
                            extend=bool(params.extend_alignments), extend_band=params.extend_band_size,
                            extend_drops=params.max_extend_dropoff)
    n_multi = check_hits(hits, off, seq, ops, want)
    t = ctx.map_last_timing()
    assert t["intervals"] == len(ivs) and t["hits"] == len(hits)
    assert t["aligned_reads"] == sum(1 for w in want if w)
    assert t["aligned_bases"] == sum(int(reads[r]["length"]) for r, w in enumerate(want) if w)
    if not kw:
        assert n_multi >= 3  # reads that keep several alignments do occur
        assert t["refined"] < len(ivs) or all(len(w) for w in want)
    ctx.close()


@pytest.mark.parametrize("kw", [dict(band=0), dict(band=0, n_candidates=2), dict(band=25), dict(band=0, indel=4, max_score=-1000)])
def test_map_reads_without_guided_align_against_oracle_composition(oracle, kw):
    """use_guided_align = 0 (-noUseGuidedAlign): RefineAlignment's k-band branch (Blasr.cpp:917-1046) inside the mapper --
    ComputeDrift, the trimming, k from the drift or from bandSize, AffineKBandAlign(k * 1.2), the non-affine statistics --
    against MapRead composed from the oracle's pieces; every field, sequence and op list of every hit."""
    rng = np.random.default_rng(80)
    genome, pool, reads, ivs = repeat_genome_case(rng, 40)
    ctx = blasr_amd.Context(0)
    ctx.upload_genome(genome)
    params = ctx.map_params(use_guided_align=0, **kw)
    hits, off, seq, ops = ctx.map_reads(reads, pool, ivs, params=params)
    want = oracle_map_reads(oracle, MapperOracle(), genome, None, pool, reads, ivs, band=params.band,
                            n_candidates=params.n_candidates, max_score=params.max_score, guided=False, indel=params.indel)
    check_hits(hits, off, seq, ops, want)
    assert sum(len(w) for w in want) >= 30 and all(int(h["q_pos"]) == 0 and int(h["t_pos"]) == 0 for h in hits)
    ctx.close()


def test_map_reads_without_guided_align_matches_the_reference_RefineAlignment():
    """The same branch against what the reference's OWN RefineAlignment (useGuidedAlign == false) left in 40 candidates
    (tests/golden/nonguided_golden.npz): positions, lengths, blocks, gap lists, counts, identity.  (Not the score: the
    reference computes it with a matrix it never initialised, SURVEY a-11.)"""
    from blasr_testlib import nonguided_golden_cases
    genome, cases = nonguided_golden_cases()
    ctx = blasr_amd.Context(0)
    n_wide = 0
    for i, c in enumerate(cases):
        ctx.upload_genome(genome, c["contigs"])
        reads = np.zeros(1, dtype=blasr_amd.READ_DTYPE)
        reads[0] = (0, len(c["read"]), c["subread"][0], c["subread"][1], 0)
        ivs = np.zeros(1, dtype=blasr_amd.INTERVAL_DTYPE)
        ivs[0] = (c["iv"][0], c["iv"][1], c["iv"][2], c["iv"][3], 0, c["iv"][4])
        params = ctx.map_params(use_guided_align=0, band=c["band"], max_score=1 << 30, n_candidates=10)
        hits, off, seq, ops = ctx.map_reads(reads, c["read"], ivs, params=params)
        assert len(hits) == 1, i
        h, w = hits[0], c["want"]
        for f, g in (("t_aligned_seq_pos", "tAlignedSeqPos"), ("t_aligned_seq_length", "tAlignedSeqLength"),
                     ("q_aligned_seq_pos", "qAlignedSeqPos"), ("q_aligned_seq_length", "qAlignedSeqLength"), ("q_pos", "qPos"),
                     ("t_pos", "tPos"), ("n_match", "nMatch"), ("n_mismatch", "nMismatch"), ("n_ins", "nIns"), ("n_del", "nDel")):
            assert int(h[f]) == w[g], (i, f, int(h[f]), w[g])
        assert np.float32(h["pct_similarity"]) == w["pct"], i
        blocks, gc, gi = hip.ops_to_alignment(ops[int(h["ops_off"]): int(h["ops_off"]) + int(h["n_ops"])])
        assert np.array_equal(blocks, w["blocks"]) and np.array_equal(gc, w["gapCounts"]) and np.array_equal(gi, w["gapItems"]), i
        n_wide += int(h["n_cells"]) > 200 * int(h["q_aligned_seq_length"])
    assert n_wide >= 2  # bands of more than 200 columns occur
    ctx.close()


def test_map_reads_with_contigs_and_empty_inputs(oracle):
    rng = np.random.default_rng(78)
    genome, pool, reads, ivs = repeat_genome_case(rng, 20, genome_len=120000)
    contigs = np.array([0, 30000, 75000, len(genome)], dtype=np.uint32)
    ctx = blasr_amd.Context(0)
    ctx.upload_genome(genome, contigs)
    hits, off, seq, ops = ctx.map_reads(reads, pool, ivs)
    want = oracle_map_reads(oracle, MapperOracle(), genome, contigs, pool, reads, ivs)
    check_hits(hits, off, seq, ops, want)
    hits_x, off_x, seq_x, ops_x = ctx.map_reads(reads, pool, ivs, params=ctx.map_params(extend_alignments=1))
    check_hits(hits_x, off_x, seq_x, ops_x, oracle_map_reads(oracle, MapperOracle(), genome, contigs, pool, reads, ivs, extend=True))
    # a read without intervals keeps an empty slice; no intervals at all is not an error
    some = ivs[ivs["read"] != 3]
    hits2, off2, _, _ = ctx.map_reads(reads, pool, some)
    assert off2[3] == off2[4] and len(hits2) == len(hits) - (off[4] - off[3])
    hits3, off3, _, _ = ctx.map_reads(reads, pool, ivs[:0])
    assert len(hits3) == 0 and off3.tolist() == [0] * (len(reads) + 1)
    # an interval naming a read that does not exist is refused
    bad = ivs.copy()
    bad[0]["read"] = len(reads)
    with pytest.raises(blasr_amd.BlasrHipError) as e:
        ctx.map_reads(reads, pool, bad)
    assert e.value.code == hip.ERR_INVALID
    ctx.close()


def test_refine_from_sdp_equals_host_trimming(oracle):
    """blasr_hip_refine_from_sdp_device on the SDP stage's device records = trimming on the host + blasr_hip_refine_host,
    for all candidates and for a selection (out of order, with a candidate that has no blocks)."""
    rng = np.random.default_rng(79)
    from blasr_amd import synth
    from blasr_testlib import DeviceMemory
    pairs = [synth.make_pair(rng, int(rng.integers(300, 2500))) for _ in range(12)]
    pairs.append((np.frombuffer(b"ACGTACGTAC" * 8, np.uint8), np.frombuffer(b"TTTTTTTTTT" * 9, np.uint8)))  # no blocks
    ctx = blasr_amd.Context(0)
    mem = DeviceMemory()
    seqs, q_off, q_len, t_off, t_len, at = [], [], [], [], [], 0
    for q, t in pairs:
        q_off.append(at); q_len.append(len(q)); at += len(q)
        t_off.append(at); t_len.append(len(t)); at += len(t)
        seqs += [q, t]
    n = len(pairs)
    ptrs = dict(seq=mem.upload(np.concatenate(seqs)), q_off=mem.upload(np.array(q_off, np.int64)),
                q_len=mem.upload(np.array(q_len, np.int32)), t_off=mem.upload(np.array(t_off, np.int64)),
                t_len=mem.upload(np.array(t_len, np.int32)))
    sdp_res, blocks = mem.zeros(n * 48), mem.zeros(12 * at)
    ctx.sdp_device(ptrs, n, blasr_amd.SdpParams(), blasr_amd.ScoreParams(), sdp_res, blocks, at)
    host_sdp = ctx.sdp(pairs)
    batch = blasr_amd.RefineBatch()
    for (q, t), s in zip(pairs, host_sdp):
        bl = s["blocks"]
        if len(bl):
            batch.add(q[s["qPos"]: s["qPos"] + bl[-1, 0] + bl[-1, 2]], t[s["tPos"]: s["tPos"] + bl[-1, 1] + bl[-1, 2]], bl)
        else:
            batch.add(q[:0], t[:0], bl)
    want = ctx.refine(batch)
    assert want[-1].status == hip.TASK_EMPTY_GUIDE
    ptrs["n"] = n
    for sel in (None, [5, 0, 12, 7, 7, 3]):
        m = n if sel is None else len(sel)
        sel_p = 0 if sel is None else mem.upload(np.array(sel, np.int32))
        res, ops = mem.zeros(m * 56), mem.zeros(8 * at)
        ctx.refine_from_sdp_device(ptrs, sdp_res, blocks, m, sel_p, 16, blasr_amd.ScoreParams(), res, ops, 2 * at)
        rec = mem.download(res, hip.RESULT_DTYPE, m)
        ops_h = mem.download(ops, np.uint32, 2 * at)
        for i in range(m):
            w = want[i if sel is None else sel[i]]
            g = blasr_amd.RefineResult(rec[i], ops_h[int(rec[i]["ops_off"]): int(rec[i]["ops_off"]) + int(rec[i]["n_ops"])])
            assert g.status == w.status
            assert (g.qPos, g.tPos, g.dpScore, g.nCells, g.score, g.nMatch, g.nMismatch, g.nIns, g.nDel) == \
                   (w.qPos, w.tPos, w.dpScore, w.nCells, w.score, w.nMatch, w.nMismatch, w.nIns, w.nDel), i
            assert g.ops.tolist() == w.ops.tolist(), i
    mem.free()
    ctx.close()


def test_two_mapping_threads_give_the_single_thread_results(oracle):
    """The C ABI is one context per host thread (the reference's -nproc): two threads map the two halves of a block of
    reads at the same time, each through its own context and streams, over several rounds -- every hit record, aligned
    sequence and op list must equal what one thread produces for the same reads (those results are checked against
    the oracle by the tests above).  A missing stream / event dependency inside the library would show up here."""
    import threading
    rng = np.random.default_rng(80)
    genome, pool, reads, ivs = repeat_genome_case(rng, 160, genome_len=400000, read_len=(1500, 6000))
    half = len(reads) // 2

    def subset(lo, hi):
        m = (ivs["read"] >= lo) & (ivs["read"] < hi)
        s = ivs[m].copy()
        s["read"] -= lo
        r = reads[lo:hi].copy()
        base = int(r["off"][0])
        r["off"] -= base
        return r, pool[base: base + int(r["off"][-1] + r["length"][-1])], s

    parts = [subset(0, half), subset(half, len(reads))]
    one = blasr_amd.Context(0)
    one.upload_genome(genome)
    want = [one.map_reads(r, p, s) for r, p, s in parts]
    one.close()
    ctxs = [blasr_amd.Context(0) for _ in parts]
    for c in ctxs:
        c.upload_genome(genome)
    n_candidates = 0
    for rnd in range(6):
        got = [None, None]

        def work(i):
            got[i] = ctxs[i].map_reads(*parts[i])

        ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        for i in range(2):
            for a, b in zip(got[i], want[i]):
                assert np.array_equal(a, b), (rnd, i)
        n_candidates += len(ivs)
    assert n_candidates >= 2000
    for c in ctxs:
        c.close()


def test_bench_with_two_ranks_on_the_devices_there_are():
    """The N > 1 path of bench.py with the REAL device step: `bench.py --gpus 2 --share-gpus` starts its two ranks itself;
    on a one-GPU box both map their own shard on that GPU (each through its own context) and the four counters are
    reduced across the ranks.  The line must say n_gpus = 2, carry both shards' counters, and say that the GPUs were
    shared (it is a run of the path, not a scaling figure)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpus", "--config", "1", "--reads", "600",
                        "--read-len", "3000", "--genome-len", "400000", "--steps", "1", "--warmup", "1", "--cpu-sample", "0",
                        "--blasrmc-reads", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [json.loads(x) for x in p.stdout.decode().splitlines() if x.startswith("{")]
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 2 and d["config"]["reads"] == 1200 and d["config"]["aligned_reads"] == 1200
    assert d["scaling"] == "weak" and d["value"] > 0
    import torch
    if torch.cuda.device_count() < 2:
        assert d["physical_gpus"] == torch.cuda.device_count() and "SHARE" in d["note"]


def test_bench_with_two_ranks_over_rccl_when_the_box_has_two_gpus():
    """`bench.py --gpus 2` as the driver launches it on a multi-GPU node: one process per GPU, `torch.distributed` with the
    `nccl` backend (= RCCL), the distinct-device check, one all-reduce of the counters.  Skipped on a one-GPU box -- the
    first box with two GPUs runs the branch."""
    import json
    import os
    import subprocess
    import sys

    import pytest
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU: the RCCL branch needs two devices (the gloo form of the same path runs above)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "1", "--reads", "600", "--read-len", "3000",
                        "--genome-len", "400000", "--steps", "1", "--warmup", "1", "--cpu-sample", "0", "--blasrmc-reads", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [json.loads(x) for x in p.stdout.decode().splitlines() if x.startswith("{")]
    assert len(lines) == 1 and len(p.stdout.decode().splitlines()[-1]) < 4096
    d = lines[0]
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["counters_backend"] == "nccl"
    assert d["config"]["reads"] == 1200 and d["config"]["aligned_reads"] == 1200 and len(d["per_rank_ms_per_step"]) == 2
