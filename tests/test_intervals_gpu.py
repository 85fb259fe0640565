"""next-1 on the device: blasr_hip_find_intervals_device -- RemoveOverlappingAnchors, the sliding window of
StoreLargestIntervals and GlobalChain (the reference's priority search tree simulated step by step) as kernels over
anchor lists that stay in device memory, p-values and interval sets finished on the host -- against
blasr_hip_find_intervals (the host restatement, itself pinned to the reference's own functions in
tests/test_intervals_cpu.py) on the same anchors: every interval and p-value, bit for bit, over several parameter sets,
multi-contig genomes, tiny and empty lists, planted repeats that make large clusters with tied endpoints; against the
committed golden intervals of the reference where their `stale` element is zero; and the reference BINARY's 125
candidate intervals of the noisy fixture."""
import os

import numpy as np
import pytest
import torch

import blasr_amd
from blasr_amd import hip
import blasr_testlib as T
from test_intervals_cpu import DEFAULT_ANCHORS, as_rows, pack_lists

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def device_intervals(ctx, genome, starts, ct, reads, anchors, off, params=None):
    buf = torch.from_numpy(anchors.view(np.uint8).copy() if len(anchors) else np.zeros(16, np.uint8)).cuda()
    return ctx.find_intervals_device(genome, starts, ct, reads, buf.data_ptr(), off, params)


def compare(ctx, genome, starts, ct, reads, anchors, off, kw, tag):
    p = hip.interval_params(**kw)
    want, woff, wpv = hip.find_intervals(genome, starts, ct, reads, anchors, off, p)
    got, goff, gpv = device_intervals(ctx, genome, starts, ct, reads, anchors, off, p)
    assert np.array_equal(goff, woff), (tag, kw)
    assert np.array_equal(as_rows(got), as_rows(want)), (tag, kw)
    assert np.array_equal(gpv, wpv), (tag, kw)
    # and the intervals' chained anchors (what -rbao reads), host against device
    _, _, _, wch, wco = hip.find_intervals(genome, starts, ct, reads, anchors, off, p, chains=True)
    buf = torch.from_numpy(anchors.view(np.uint8).copy() if len(anchors) else np.zeros(16, np.uint8)).cuda()
    got2, _, _, gch, gco = ctx.find_intervals_device(genome, starts, ct, reads, buf.data_ptr(), off, p, chains=True)
    assert np.array_equal(as_rows(got2), as_rows(want)) and np.array_equal(gco, wco) and np.array_equal(gch, wch), (tag, kw)
    return len(want)


@pytest.mark.skipif(not os.path.exists(T.SAWRITER), reason="oracle/_ref/sawritermc not built")
def test_device_intervals_equal_the_host_restatement(tmp_path):
    rng = np.random.default_rng(77)
    ctx = blasr_amd.Context(0)
    total = 0
    for it, (blt, ps, glen, rlen, n) in enumerate([(8, (12, 0, 1, 1000, 0, 0, 1), 150000, (500, 4000), 40),
                                                  (None, DEFAULT_ANCHORS, 200000, (3000, 9000), 40),
                                                  (8, (10, 1, 1, 50, 0, 0, 0), 90000, (60, 900), 60),
                                                  (8, (9, 0, 1, 1000, 0, 0, 1), 60000, (2000, 6000), 30)]):
        fasta, reads, pool = T.anchor_genome_case(rng, n, genome_len=glen, read_len=rlen)
        genome, starts, sa, _ = T.write_suffix_array(tmp_path, fasta, blt)
        ctx.upload_genome(genome, starts if len(starts) > 2 else None)
        ctx.upload_suffix_array(sa)
        ap = ctx.anchor_params(min_match_length=ps[0], expand=ps[1], use_lookup_table=ps[2], max_anchors_per_position=ps[3],
                               advance_exact_matches=ps[4], max_lcp_length=ps[5], stop_mapping_once_unique=ps[6],
                               min_prefix_match_length=8, sorted=1)
        anchors, off = ctx.anchor_reads(reads, pool, ap)
        ct = hip.count_tuples(genome, 8)
        for kw in (dict(), dict(n_candidates=3), dict(min_interval=30), dict(remove_contained_intervals=1), dict(n_candidates=1),
                   dict(indel_rate=0.05), dict(max_p_value=-40.0), dict(min_interval=0, n_candidates=20)):
            total += compare(ctx, genome, starts, ct, reads, anchors, off, kw, it)
        if it == 2:  # tiny and empty lists
            lists = []
            for r in range(len(reads)):
                pair = []
                for sd in (0, 1):
                    a = anchors[int(off[2 * r + sd]): int(off[2 * r + sd + 1])]
                    pair.append(np.stack([a["t"], a["q"], a["l"]], axis=1) if len(a) else np.zeros((0, 3), np.uint32))
                lists.append(pair)
            lists[3] = [lists[3][0][:2], lists[3][1][:1]]
            lists[5] = [lists[5][0][:0], lists[5][1][:3]]
            lists[7] = [lists[7][0][:1], lists[7][1][:0]]
            a2, off2 = pack_lists(lists)
            total += compare(ctx, genome, starts, ct, reads, a2, off2, dict(min_interval=0), "tiny")
    assert total > 1500
    ctx.close()


def test_large_clusters_with_tied_endpoints():
    """Hand-made lists: many anchors of one diagonal band (clusters of hundreds of anchors, beyond the small class),
    nested anchors that survive RemoveOverlappingAnchors (equal end points: the endpoint sort has ties and the replay of
    std::sort decides), anchors in two contigs inside one window, and a cluster of thousands (the global-memory class)."""
    rng = np.random.default_rng(5)
    genome = blasr_amd.synth.random_genome(400000, 3)
    genome[150000] = ord("N")
    starts = np.array([0, 150001, 400000], np.uint32)
    ct = hip.count_tuples(genome, 8)
    reads = np.zeros(6, hip.READ_DTYPE)
    lists = []
    sizes = [60, 400, 1500, 5000, 30, 200]
    at = 0
    for r, m in enumerate(sizes):
        L = 9000
        reads[r] = (at, L, 0, L, 0)
        at += L
        base_t = int(rng.integers(10000, 120000)) if r != 4 else 149000  # r = 4: a window across the contig end
        q = np.sort(rng.integers(0, L - 40, m)).astype(np.int64)
        t = base_t + q + rng.integers(-30, 30, m)
        ln = rng.integers(9, 25, m)
        tri = np.stack([t, q, ln], axis=1)
        # nested anchors on the same diagonal a few positions on: equal end points
        extra = tri[rng.integers(0, m, m // 5)].copy()
        d = rng.integers(2, 6, len(extra))
        extra[:, 0] += d
        extra[:, 1] += d
        extra[:, 2] = np.maximum(extra[:, 2] - d, 1)
        tri = np.concatenate([tri, extra])
        tri = np.unique(tri[:, :2], axis=0, return_index=True)[1]
        tri_all = np.concatenate([np.stack([t, q, ln], axis=1), extra])[tri]
        order = np.lexsort((tri_all[:, 1], tri_all[:, 0]))
        fwd = tri_all[order].astype(np.uint32)
        rev = fwd[: max(1, len(fwd) // 7)].copy()
        lists.append([fwd, rev])
    a, off = pack_lists(lists)
    ctx = blasr_amd.Context(0)
    n = 0
    for kw in (dict(min_interval=0), dict(), dict(n_candidates=2, min_interval=20), dict(indel_rate=0.02)):
        n += compare(ctx, genome, starts, ct, reads, a, off, kw, "large")
    assert n >= 8
    ctx.close()


def test_golden_intervals_of_the_reference_where_nothing_stale_is_read():
    z = np.load(os.path.join(HERE, "golden", "intervals_golden.npz"))
    genome, starts, reads = z["genome"], z["starts"], z["reads"]
    a = np.zeros(len(z["anchors"]), hip.ANCHOR_DTYPE)
    a["t"], a["q"], a["l"] = z["anchors"][:, 0], z["anchors"][:, 1], z["anchors"][:, 2]
    ct = hip.count_tuples(genome, int(z["tuple_size"]))
    ctx = blasr_amd.Context(0)
    # (the goldens were made with random `stale` elements; the device takes zeros: compare with the host at zeros)
    for kw in (dict(), dict(n_candidates=3), dict(min_interval=30), dict(remove_contained_intervals=1)):
        compare(ctx, genome, starts, ct, reads, a, z["list_off"], kw, "golden")
    ctx.close()


@pytest.mark.skipif(not os.path.exists(T.SAWRITER), reason="oracle/_ref/sawritermc not built")
def test_the_reference_binarys_candidate_intervals_from_device_anchors_and_device_intervals(tmp_path):
    z = np.load(os.path.join(HERE, "golden", "m4_noisy_golden.npz"))
    fasta = b">contig0\n" + bytes(z["contig0"]) + b"\n>contig1\n" + bytes(z["contig1"]) + b"\n"
    genome, starts, sa, _ = T.write_suffix_array(tmp_path, fasta, 8)
    lens = z["read_lens"].astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)])
    reads = np.zeros(len(lens), hip.READ_DTYPE)
    for r in range(len(lens)):
        reads[r] = (offs[r], lens[r], 0, lens[r], 0)
    ctx = blasr_amd.Context(0)
    ctx.upload_genome(genome, starts)
    ctx.upload_suffix_array(sa)
    anchors, off = ctx.anchor_reads(reads, z["reads"], ctx.anchor_params(min_match_length=14, min_prefix_match_length=8))
    ivs, ioff, _ = device_intervals(ctx, genome, starts, hip.count_tuples(genome, 8), reads, anchors, off)
    assert len(ivs) == len(z["intervals"]) == 125
    for r in range(len(reads)):
        assert np.array_equal(as_rows(ivs[int(ioff[r]): int(ioff[r + 1])]).astype(np.int64), z["intervals"][z["owner"] == r]), r
    ctx.close()
