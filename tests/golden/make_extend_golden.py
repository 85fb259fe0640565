#!/usr/bin/env python3
"""Generate tests/golden/extend_golden.npz from the REFERENCE's own ExtendAlignmentForward / ExtendAlignmentReverse
templates (oracle/_ref/libblasr_ref.so, see oracle/ref_harness.cpp): seeded synthetic read ends and the genome next to
them, and what the reference returned (return value, qPos, tPos, blocks, gap lists).  Data only.

    python tests/golden/make_extend_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from blasr_testlib import extend_align, load_ref, random_extend_case  # noqa: E402
from blasr_amd import synth  # noqa: E402


def main():
    ref = load_ref()
    assert ref is not None, "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(20261003)
    cases = [random_extend_case(rng, it) for it in range(120)]
    acgt = np.frombuffer(b"ACGT", np.uint8)
    # the shapes AlignIntervals produces (Blasr.cpp:1805-1897): up to 500 bases of read end against up to 500 of
    # genome, band 10, 10 drops; Forward from (0, 0), Reverse from (length - 1, length - 1)
    for ql in (1, 7, 60, 250, 500):
        q, t = synth.make_pair(rng, ql, flank=(0, 0))
        if len(q) == 0 or len(t) == 0:
            q, t = acgt[:1].copy(), acgt[:2].copy()
        q, t = q[:500], t[:500]
        cases.append((q, 0, t, 0, 10, 10, False))
        cases.append((q[::-1].copy(), len(q) - 1, t[::-1].copy(), len(t) - 1, 10, 10, True))
    q, t = synth.make_pair(rng, 400, ins=0, dele=0, sub=0, flank=(0, 0))
    cases.append((q, 0, t, 0, 10, 10, False))           # identical: runs to the end of the shorter sequence
    cases.append((q, len(q), t, len(t), 10, 10, True))
    cases.append((q, 0, t[:37], 0, 10, 10, False))      # the genome ends first (contig end)
    cases.append((q[:23], 0, t, 0, 10, 10, False))      # the read ends first
    cases.append((q, 0, acgt[rng.integers(0, 4, 300)], 0, 10, 10, False))  # unrelated: a handful of rows
    cases.append((q, 0, t, 0, 10, 0, False))            # no drop allowed: no row is scored
    cases.append((q, len(q) + 3, t, 0, 10, 10, False))  # nothing left of the read: returns 0
    out = dict(n=np.array(len(cases)))
    for i, (q, qp, t, tp, k, drops, rev) in enumerate(cases):
        r = extend_align(ref, q, qp, t, tp, k, drops, rev)
        out["e%d_q" % i], out["e%d_t" % i] = np.asarray(q, np.uint8), np.asarray(t, np.uint8)
        out["e%d_par" % i] = np.array([qp, tp, k, drops, int(rev)], np.int32)
        out["e%d_res" % i] = np.array([r["score"], r["qPos"], r["tPos"]], np.int64)
        out["e%d_blocks" % i], out["e%d_gapCounts" % i], out["e%d_gapItems" % i] = r["blocks"], r["gapCounts"], r["gapItems"]
    path = os.path.join(HERE, "extend_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(cases), "cases")


if __name__ == "__main__":
    main()
