#!/usr/bin/env python3
"""Generate tests/golden/nonguided_golden.npz from the REFERENCE's own AlignIntervals + RefineAlignment with
useGuidedAlign == false (alignment/Blasr.cpp:917-1046, through oracle/_ref/libblasr_refmapper.so, see
oracle/ref_mapper_harness.cpp): a seeded genome, reads with their anchor intervals, and what the reference left in each
candidate -- positions, lengths, blocks, gap lists, counts, identity; not the score (uninitialised memory there,
SURVEY a-11).  Data only.

    python tests/golden/make_nonguided_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from blasr_testlib import load_refmapper, random_interval_case  # noqa: E402
from blasr_amd import synth  # noqa: E402


def main():
    refm = load_refmapper()
    assert refm is not None, "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(20261005)
    genome = synth.random_genome(50000, 17)
    out = dict(genome=genome)
    n = 0
    for it in range(40):
        read, subread, iv, contigs = random_interval_case(rng, genome, it)
        band = [0, 0, 0, 24][it % 4]
        win, tseq, sdp, ref = refm.align_interval(genome, read, subread, iv, contigs, nonguided_band=band)
        if len(sdp["blocks"]) == 0:
            continue
        p = "c%d_" % n
        out[p + "read"] = read
        out[p + "par"] = np.array([subread[0], subread[1], iv[0], iv[1], iv[2], iv[3], iv[4], band], np.int64)
        out[p + "contigs"] = np.zeros(0, np.uint32) if contigs is None else contigs
        out[p + "fields"] = np.array([ref["tAlignedSeqPos"], ref["tAlignedSeqLength"], ref["qAlignedSeqPos"],
                                      ref["qAlignedSeqLength"], ref["qPos"], ref["tPos"], ref["nMatch"], ref["nMismatch"],
                                      ref["nIns"], ref["nDel"]], np.int64)
        out[p + "pct"] = np.float32(ref["pct"])
        out[p + "blocks"], out[p + "gapCounts"], out[p + "gapItems"] = ref["blocks"], ref["gapCounts"], ref["gapItems"]
        n += 1
    out["n"] = n
    np.savez_compressed(os.path.join(HERE, "nonguided_golden.npz"), **out)
    print("wrote %d candidates" % n)


if __name__ == "__main__":
    main()
