"""Writes tests/golden/intervals_golden.npz: candidate intervals and p-values of the REFERENCE's
RemoveOverlappingAnchors + FindMaxIncreasingInterval (Forward, Reverse) + WeightedIntervalSet (through
oracle/_ref/libblasr_refmapper.so) for anchor lists of the reference's MapReadToGenome, on a small two-contig genome
with planted repeats (suffix array by the reference's sawriter, -blt 6).  Run where oracle/_ref is built:
    python tests/golden/make_intervals_golden.py"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import blasr_testlib as T  # noqa: E402

PARAMS = [dict(), dict(n_candidates=3), dict(min_interval=30), dict(remove_contained=1), dict(n_candidates=1, min_interval=0)]


def main():
    rng = np.random.default_rng(909)
    fasta, reads, pool = T.anchor_genome_case(rng, 16, genome_len=60000, read_len=(400, 2500))
    ref = C.CDLL(T.REFMAPPER_SO)
    ref.ref_sa_open.restype = C.c_void_p
    ref.ref_ct_open.restype = C.c_void_p
    with tempfile.TemporaryDirectory() as tmp:
        genome, starts, sa, sa_path = T.write_suffix_array(tmp, fasta, 6)
        h = ref.ref_sa_open(sa_path.encode())
        ct = ref.ref_ct_open(genome.ctypes.data_as(C.c_void_p), len(genome), 6)
        ps = (10, 0, 1, 1000, 0, 0, 1)
        anchors, off = [], [0]
        for r in range(len(reads)):
            o, L = int(reads[r]["off"]), int(reads[r]["length"])
            rd = np.ascontiguousarray(pool[o:o + L])
            for strand in (0, 1):
                a = T.ref_anchors(ref, h, genome, rd, int(reads[r]["subread_start"]), int(reads[r]["subread_end"]), strand,
                                  6, ps, True)
                anchors.append(a)
                off.append(off[-1] + len(a))
        flat = np.concatenate(anchors).astype(np.uint32)
        stale = np.stack([rng.integers(0, len(genome), 2 * len(reads)), rng.integers(0, 2500, 2 * len(reads)),
                          rng.integers(6, 30, 2 * len(reads))], axis=1).astype(np.uint32)
        want, want_p, counts = [], [], []
        for kw in PARAMS:
            for r in range(len(reads)):
                iv, pv = T.ref_find_intervals(ref, ct, genome, starts, int(reads[r]["length"]), int(reads[r]["subread_start"]),
                                              int(reads[r]["subread_end"]), anchors[2 * r], anchors[2 * r + 1],
                                              stale[2 * r: 2 * r + 2], **kw)
                want.append(iv)
                want_p.append(pv)
                counts.append(len(iv))
        ref.ref_sa_close(C.c_void_p(h))
        ref.ref_ct_close(C.c_void_p(ct))
    out = os.path.join(HERE, "intervals_golden.npz")
    np.savez_compressed(out, genome=genome, starts=starts, tuple_size=np.int32(6), reads=reads, anchors=flat,
                        list_off=np.array(off, np.int64), stale=stale, counts=np.array(counts, np.int32),
                        intervals=np.concatenate(want).astype(np.uint32), p_values=np.concatenate(want_p).astype(np.float32))
    print(out, os.path.getsize(out), "bytes;", sum(counts), "intervals,", len(flat), "anchors")


if __name__ == "__main__":
    main()
