"""next-1, device half: blasr_hip_anchor_reads_* -- MapReadToGenome for both strands of every read
plus SortMatchPosList, on the device -- against the oracle's restatement (oracle/blasr_oracle.c:bo_map_read_to_genome,
itself pinned to the reference in tests/test_anchor_cpu.py), against the reference's own MapReadToGenome through
oracle/_ref where that build is present, and against the committed golden lists of the reference
(tests/golden/anchor_golden.npz).  Every anchor of every list, bit for bit: t, q, l, in the reference's order."""
import ctypes as C
import os

import numpy as np
import pytest

import blasr_amd
from blasr_amd import hip
import blasr_testlib as T

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "anchor_golden.npz")
have_sawriter = os.path.exists(T.SAWRITER)


def read_of(reads, pool, r):
    off, L = int(reads[r]["off"]), int(reads[r]["length"])
    return np.ascontiguousarray(pool[off: off + L]), int(reads[r]["subread_start"]), int(reads[r]["subread_end"])


def as_triples(anchors):
    return np.stack([anchors["t"], anchors["q"], anchors["l"]], axis=1) if len(anchors) else np.zeros((0, 3), np.uint32)


def params_of(ctx, ps, min_prefix, sort, **kw):
    return ctx.anchor_params(min_match_length=ps[0], expand=ps[1], use_lookup_table=ps[2], max_anchors_per_position=ps[3],
                             advance_exact_matches=ps[4], max_lcp_length=ps[5], stop_mapping_once_unique=ps[6],
                             min_prefix_match_length=min_prefix, sorted=int(sort), **kw)


def golden_case():
    z = np.load(GOLDEN)
    sa = dict(index=z["sa_index"], start_pos_table=z["start_pos_table"], end_pos_table=z["end_pos_table"],
              lookup_prefix_length=int(z["lookup_prefix_length"]))
    return z, sa


def test_anchors_against_the_golden_lists_of_the_reference():
    z, sa = golden_case()
    genome, reads, pool = z["genome"], z["reads"], z["pool"]
    ctx = blasr_amd.Context(0)
    ctx.upload_genome(genome)
    ctx.upload_suffix_array(sa)
    lists, k = {}, 0
    for ci, case in enumerate(z["cases"]):
        n = int(z["counts"][ci])
        lists[tuple(int(v) for v in case)] = z["anchors"][k: k + n]
        k += n
    groups = sorted({key[:8] + (key[10],) for key in lists})
    checked = 0
    for g in groups:
        ps, min_prefix, sort = g[:7], g[7], g[8]
        anchors, off = ctx.anchor_reads(reads, pool, params_of(ctx, ps, min_prefix, sort))
        got = as_triples(anchors)
        for r in range(len(reads)):
            for strand in (0, 1):
                want = lists.get(ps + (min_prefix, r, strand, sort))
                if want is None:
                    continue
                mine = got[int(off[2 * r + strand]): int(off[2 * r + strand + 1])]
                assert np.array_equal(mine, want), (g, r, strand)
                checked += len(want)
    assert checked == len(z["anchors"])
    ctx.close()


@pytest.mark.skipif(not have_sawriter, reason="oracle/_ref/sawritermc not built")
@pytest.mark.parametrize("blt", [None, 8])
def test_anchors_against_oracle_and_reference(oracle, tmp_path, blt):
    rng = np.random.default_rng(31 if blt is None else 32)
    fasta, reads, pool = T.anchor_genome_case(rng, 24, genome_len=100000 if blt else 150000)
    genome, _, sa, sa_path = T.write_suffix_array(tmp_path, fasta, blt)
    ref, h = None, None
    if os.path.exists(T.REFMAPPER_SO):
        ref = C.CDLL(T.REFMAPPER_SO)
        ref.ref_sa_open.restype = C.c_void_p
        h = ref.ref_sa_open(sa_path.encode())
    ctx = blasr_amd.Context(0)
    ctx.upload_genome(genome)
    ctx.upload_suffix_array(sa)
    total = failing = 0
    for ps in T.ANCHOR_PARAM_SETS:
        for min_prefix in ([sa["lookup_prefix_length"], 8 if blt is None else 6] if ps[2] else [1]):
            for sort in (0, 1):
                want = [T.oracle_anchors(oracle, genome, sa, *read_of(reads, pool, r), strand, min_prefix, ps, bool(sort))
                        for r in range(len(reads)) for strand in (0, 1)]
                p = params_of(ctx, ps, min_prefix, sort)
                if any(w is None for w in want):  # the reference asserts on such a batch; the library reports it
                    with pytest.raises(hip.BlasrHipError) as e:
                        ctx.anchor_reads(reads, pool, p)
                    assert "asserts" in str(e.value)
                    failing += 1
                    continue
                anchors, off = ctx.anchor_reads(reads, pool, p)
                got = as_triples(anchors)
                assert off[-1] == len(got) == sum(len(w) for w in want)
                for i, w in enumerate(want):
                    assert np.array_equal(got[int(off[i]): int(off[i + 1])], w), (ps, min_prefix, sort, i)
                total += len(got)
                if ref is not None and sort:  # the reference itself on a few of the lists
                    for i in range(0, len(want), 7):
                        rd, s0, s1 = read_of(reads, pool, i // 2)
                        w = T.ref_anchors(ref, h, genome, rd, s0, s1, i % 2, min_prefix, ps, True)
                        assert np.array_equal(got[int(off[i]): int(off[i + 1])], w), (ps, min_prefix, i)
    t = ctx.anchor_last_timing()
    assert t["positions"] > 0 and t["pieces"] >= 1
    assert total > 50000 and failing <= 4
    if ref is not None:
        ref.ref_sa_close(C.c_void_p(h))
    ctx.close()


def test_anchor_pieces_forward_only_capacity_and_device_entry(oracle, monkeypatch):
    z, sa = golden_case()
    genome, reads, pool = z["genome"], z["reads"], z["pool"]
    ctx = blasr_amd.Context(0)
    ctx.upload_genome(genome)
    ctx.upload_suffix_array(sa)
    ps, min_prefix = T.ANCHOR_PARAM_SETS[1], 6
    p = params_of(ctx, ps, min_prefix, 1)
    whole, off = ctx.anchor_reads(reads, pool, p)
    assert ctx.anchor_last_timing()["pieces"] == 1
    # the same batch cut into many pieces (one or two lists each)
    monkeypatch.setenv("BLASR_HIP_ANCHOR_PIECE", "300")
    pieces, off2 = ctx.anchor_reads(reads, pool, p)
    assert ctx.anchor_last_timing()["pieces"] > len(reads) // 2
    assert np.array_equal(off, off2) and np.array_equal(whole, pieces)
    # a capacity that is too small: the count needed comes back with the error
    need = C.c_int64(0)
    small = np.zeros(10, dtype=hip.ANCHOR_DTYPE)
    lo = np.zeros(2 * len(reads) + 1, np.int64)
    rc = ctx.lib.blasr_hip_anchor_reads_host(ctx.handle, len(reads), reads.ctypes.data, pool.ctypes.data, C.c_int64(len(pool)),
                                             C.byref(p), small.ctypes.data, C.c_int64(10), C.byref(need), lo.ctypes.data)
    assert rc == hip.ERR_CAPACITY and need.value == len(whole) and np.array_equal(lo, off)
    monkeypatch.delenv("BLASR_HIP_ANCHOR_PIECE")
    # forwardOnly: the reverse lists stay empty, the forward ones do not change
    fwd, off3 = ctx.anchor_reads(reads, pool, params_of(ctx, ps, min_prefix, 1, forward_only=1))
    for r in range(len(reads)):
        assert off3[2 * r + 2] == off3[2 * r + 1]
        assert np.array_equal(fwd[int(off3[2 * r]): int(off3[2 * r + 1])], whole[int(off[2 * r]): int(off[2 * r + 1])])
    # device-resident pool and output
    mem = T.DeviceMemory()
    d_pool = mem.upload(pool)
    d_out = mem.zeros(len(whole) * hip.ANCHOR_DTYPE.itemsize)
    n, off4 = ctx.anchor_reads_device(reads, d_pool, p, d_out, len(whole))
    assert n == len(whole) and np.array_equal(off4, off)
    assert np.array_equal(mem.download(d_out, hip.ANCHOR_DTYPE, n), whole)
    mem.free()
    ctx.close()


def test_anchors_without_the_companion_array(monkeypatch):
    """BLASR_HIP_ANCHOR_NO_EXT: every probe through index[] and genome[] (the reference's own access pattern)."""
    monkeypatch.setenv("BLASR_HIP_ANCHOR_NO_EXT", "1")
    test_anchors_against_the_golden_lists_of_the_reference()


def test_anchor_edges():
    z, sa = golden_case()
    genome, reads, pool = z["genome"], z["reads"], z["pool"]
    ctx = blasr_amd.Context(0)
    p = ctx.anchor_params()
    with pytest.raises(hip.BlasrHipError):  # nothing uploaded
        ctx.anchor_reads(reads, pool, p)
    ctx.upload_genome(genome[:-5])
    ctx.upload_suffix_array(sa)
    with pytest.raises(hip.BlasrHipError):  # the array belongs to another genome
        ctx.anchor_reads(reads, pool, p)
    ctx.upload_genome(genome)
    a, off = ctx.anchor_reads(reads[:0], pool, p)  # no reads
    assert len(a) == 0 and off.tolist() == [0]
    # reads shorter than minMatchLength, and subreads that leave nothing to map: empty lists (MapBySuffixArray.h:227-230)
    short = np.zeros(3, dtype=hip.READ_DTYPE)
    short[0] = (0, 9, 0, 9, 0)
    short[1] = (100, 300, 140, 150, 0)
    short[2] = (500, 13, 0, 13, 0)
    a, off = ctx.anchor_reads(short, pool, ctx.anchor_params(min_prefix_match_length=6))
    assert len(a) == 0 and not off.any()
    for bad in (dict(expand=8), dict(max_anchors_per_position=-1), dict(min_prefix_match_length=40)):
        with pytest.raises(hip.BlasrHipError):
            ctx.anchor_reads(reads, pool, ctx.anchor_params(**bad))
    wrong = reads.copy()
    wrong[1]["subread_end"] = wrong[1]["length"] + 1
    with pytest.raises(hip.BlasrHipError):
        ctx.anchor_reads(wrong, pool, p)
    ctx.close()


def test_a_second_context_anchors_in_a_borrowed_index():
    """blasr_hip_borrow_index: a context with its own streams and buffers that searches the genome and the suffix array
    another context of the same device holds -- the golden lists again, from both contexts, and the refusals."""
    z, sa = golden_case()
    genome, reads, pool = z["genome"], z["reads"], z["pool"]
    owner = blasr_amd.Context(0)
    owner.upload_genome(genome)
    owner.upload_suffix_array(sa)
    other = blasr_amd.Context(0)
    with pytest.raises(hip.BlasrHipError):  # nothing to anchor in yet
        other.anchor_reads(reads, pool)
    other.borrow_index(owner)
    held = other.held_bytes()
    assert held < len(genome)  # (neither the genome nor the 4 bytes per base of the index are counted twice)
    p = owner.anchor_params(min_match_length=12)
    a0, off0 = owner.anchor_reads(reads, pool, p)
    a1, off1 = other.anchor_reads(reads, pool, p)
    assert len(a0) > 1000 and np.array_equal(off0, off1) and np.array_equal(as_triples(a0), as_triples(a1))
    with pytest.raises(hip.BlasrHipError):
        other.borrow_index(other)
    empty = blasr_amd.Context(0)
    with pytest.raises(hip.BlasrHipError):
        other.borrow_index(empty)
    empty.close()
    # the owner replaces its index: the borrower's pointers are stale, and it is told so instead of searching freed memory
    # (the owner counts its changes; an address handed out again by the allocator would compare equal)
    owner.upload_suffix_array(sa)
    with pytest.raises(hip.BlasrHipError, match="borrow again"):
        other.anchor_reads(reads, pool, p)
    other.borrow_index(owner)
    a3, off3 = other.anchor_reads(reads, pool, p)
    assert np.array_equal(off0, off3) and np.array_equal(as_triples(a0), as_triples(a3))
    other.close()  # (frees nothing of the owner's)
    a2, off2 = owner.anchor_reads(reads, pool, p)
    assert np.array_equal(as_triples(a0), as_triples(a2))
    owner.close()


def test_sort_values_of_either_width(monkeypatch):
    """SortMatchPosList on the device: t is the key of a stable sort, (length, q) its value -- 32 bits where every read of the
    batch is shorter than 65 536 bases, 64 otherwise (BLASR_HIP_ANCHOR_WIDE_VALUES forces those): the same lists, and the
    golden ones."""
    z, sa = golden_case()
    genome, reads, pool = z["genome"], z["reads"], z["pool"]
    ctx = blasr_amd.Context(0)
    ctx.upload_genome(genome)
    ctx.upload_suffix_array(sa)
    p = ctx.anchor_params(min_match_length=10, max_anchors_per_position=50)
    a0, off0 = ctx.anchor_reads(reads, pool, p)
    monkeypatch.setenv("BLASR_HIP_ANCHOR_WIDE_VALUES", "1")
    a1, off1 = ctx.anchor_reads(reads, pool, p)
    assert len(a0) > 1000 and np.array_equal(off0, off1) and np.array_equal(as_triples(a0), as_triples(a1))
    # sorted by (t, q) inside every list
    for i in range(len(off0) - 1):
        seg = a0[off0[i]: off0[i + 1]]
        key = seg["t"].astype(np.int64) << 32 | seg["q"].astype(np.int64)
        assert np.all(key[1:] > key[:-1])
    ctx.close()
