"""next-1, device half: the oracle's restatement of MapReadToGenome (StoreLCPBounds, SearchLeftBound /
SearchRightBound, LocateAnchorBoundsInSuffixArray, the emission loop) against the reference's own templates through
oracle/_ref/libblasr_refmapper.so on suffix arrays the reference's sawriter builds here, and against the committed
golden lists (tests/golden/anchor_golden.npz, written by tests/golden/make_anchor_golden.py from the reference)."""
import ctypes as C
import os

import numpy as np
import pytest

import blasr_testlib as T

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "anchor_golden.npz")
have_ref = os.path.exists(T.SAWRITER) and os.path.exists(T.REFMAPPER_SO)


def read_of(reads, pool, r):
    off, L = int(reads[r]["off"]), int(reads[r]["length"])
    return np.ascontiguousarray(pool[off: off + L]), int(reads[r]["subread_start"]), int(reads[r]["subread_end"])


@pytest.mark.skipif(not have_ref, reason="reference tools not built")
@pytest.mark.parametrize("blt", [None, 8])
def test_oracle_anchors_against_the_reference(tmp_path, blt):
    oracle = T.load_oracle()
    ref = C.CDLL(T.REFMAPPER_SO)
    ref.ref_sa_open.restype = C.c_void_p
    rng = np.random.default_rng(11 if blt is None else 12)
    fasta, reads, pool = T.anchor_genome_case(rng, 20, genome_len=90000 if blt else 120000)
    genome, _, sa, sa_path = T.write_suffix_array(tmp_path, fasta, blt)
    assert sa["lookup_prefix_length"] == (blt or 13)
    h = ref.ref_sa_open(sa_path.encode())
    assert h
    total = asserting = 0
    for ps in T.ANCHOR_PARAM_SETS:
        # Blasr passes params.lookupTableLength (8) whatever the table's own prefix is (Blasr.cpp:2282-2285); a value
        # above the table's prefix over-reports the lengths and runs into the reference's asserts
        for min_prefix in ([sa["lookup_prefix_length"], 8 if blt is None else 6] if ps[2] else [1]):
            for r in range(len(reads)):
                rd, s0, s1 = read_of(reads, pool, r)
                for strand in (0, 1):
                    for sort in (False, True):
                        got = T.oracle_anchors(oracle, genome, sa, rd, s0, s1, strand, min_prefix, ps, sort)
                        if got is None:  # e.g. a poly-A word of the table's length next to an N, with minMatchLength
                            asserting += 1  # no longer than the table's prefix: the reference aborts on it
                            continue
                        want = T.ref_anchors(ref, h, genome, rd, s0, s1, strand, min_prefix, ps, sort)
                        assert np.array_equal(got, want), (ps, min_prefix, r, strand, sort)
                        total += len(want)
    assert total > 20000 and asserting <= 4
    ref.ref_sa_close(C.c_void_p(h))


def test_oracle_anchors_against_the_golden_lists():
    oracle = T.load_oracle()
    z = np.load(GOLDEN)
    sa = dict(index=z["sa_index"], start_pos_table=z["start_pos_table"], end_pos_table=z["end_pos_table"],
              lookup_prefix_length=int(z["lookup_prefix_length"]))
    genome, reads, pool = z["genome"], z["reads"], z["pool"]
    k = 0
    for ci, case in enumerate(z["cases"]):
        ps, min_prefix, r, strand, sort = case[:7], int(case[7]), int(case[8]), int(case[9]), int(case[10])
        rd, s0, s1 = read_of(reads, pool, r)
        got = T.oracle_anchors(oracle, genome, sa, rd, s0, s1, strand, min_prefix, ps, bool(sort))
        assert got is not None, case
        n = int(z["counts"][ci])
        assert np.array_equal(got, z["anchors"][k: k + n]), case
        k += n
    assert k == len(z["anchors"]) and k > 3000
