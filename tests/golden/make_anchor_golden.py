"""Writes tests/golden/anchor_golden.npz: anchor lists of the REFERENCE's MapReadToGenome (through
oracle/_ref/libblasr_refmapper.so) on a small genome whose suffix array the reference's sawriter built (-blt 6, so the
lookup table stays small).  Run in the container that has /root/reference built under oracle/_ref:
    python tests/golden/make_anchor_golden.py"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import blasr_testlib as T  # noqa: E402


def main():
    rng = np.random.default_rng(2024)
    fasta, reads, pool = T.anchor_genome_case(rng, 10, genome_len=24000, read_len=(200, 700))
    ref = C.CDLL(T.REFMAPPER_SO)
    ref.ref_sa_open.restype = C.c_void_p
    with tempfile.TemporaryDirectory() as tmp:
        genome, _, sa, sa_path = T.write_suffix_array(tmp, fasta, 6)
        h = ref.ref_sa_open(sa_path.encode())
        cases, counts, anchors = [], [], []
        for ps in T.ANCHOR_PARAM_SETS:
            for min_prefix in ([6, 5] if ps[2] else [1]):
                for r in range(len(reads)):
                    off, L = int(reads[r]["off"]), int(reads[r]["length"])
                    rd = np.ascontiguousarray(pool[off: off + L])
                    s0, s1 = int(reads[r]["subread_start"]), int(reads[r]["subread_end"])
                    for strand in (0, 1):
                        sort = (r + strand) % 2
                        a = T.ref_anchors(ref, h, genome, rd, s0, s1, strand, min_prefix, ps, bool(sort))
                        cases.append(list(ps) + [min_prefix, r, strand, sort])
                        counts.append(len(a))
                        anchors.append(a)
        ref.ref_sa_close(C.c_void_p(h))
    out = os.path.join(HERE, "anchor_golden.npz")
    np.savez_compressed(out, genome=genome, sa_index=sa["index"], start_pos_table=sa["start_pos_table"],
                        end_pos_table=sa["end_pos_table"], lookup_prefix_length=np.int32(6), reads=reads, pool=pool,
                        cases=np.array(cases, np.int32), counts=np.array(counts, np.int64),
                        anchors=np.concatenate(anchors).astype(np.uint32))
    print(out, os.path.getsize(out), "bytes;", sum(counts), "anchors in", len(cases), "lists")


if __name__ == "__main__":
    main()
