#!/usr/bin/env python3
"""Generate tests/golden/kband_golden.npz from the REFERENCE's own AffineKBandAlign template
(oracle/_ref/libblasr_ref.so, see oracle/ref_harness.cpp): seeded synthetic pairs and what the reference
returned for them (score, blocks, gap lists).  Data only.

    python tests/golden/make_kband_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from blasr_testlib import affine_kband_align, kband_align, load_ref  # noqa: E402
from blasr_amd import synth  # noqa: E402


def main():
    ref = load_ref()
    assert ref is not None, "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(20261004)
    cases = []
    for ql, k in [(1, 10), (9, 10), (40, 12), (64, 12), (150, 12), (400, 14), (1000, 18), (2500, 30), (2500, 120),
                  (6000, 40), (300, 300), (800, 543)]:
        q, t = synth.make_pair(rng, ql, flank=(0, 0))
        cases.append((q, t, k))
    q, t = synth.make_pair(rng, 600, flank=(0, 0))
    cases.append((q, t[:-35], 12))            # target much shorter: SetKBoundedLengths trims the query
    cases.append((q[:-50], t, 12))            # query much shorter: the target is trimmed
    q, t = synth.make_pair(rng, 500, ins=0, dele=0, sub=0, flank=(0, 0))
    cases.append((q, t, 10))                  # identical
    q, t = synth.make_pair(rng, 900, flank=(0, 0))
    q = q.copy()
    q[300:330] = ord("A")                     # long homopolymer in the query
    cases.append((q, t, 24))
    q, t = synth.make_pair(rng, 700, n_run=12, flank=(0, 0))
    cases.append((q, t, 16))                  # N run in the target
    q, t = synth.make_pair(rng, 1200, big_indel=("ins", 20), flank=(0, 0))
    cases.append((q, t, 30))
    q, t = synth.make_pair(rng, 1200, big_indel=("del", 20), flank=(0, 0))
    cases.append((q, t, 30))
    out = dict(n=np.int64(len(cases)))
    for i, (q, t, k) in enumerate(cases):
        r = affine_kband_align(ref, q, t, k)
        out["c%d_q" % i], out["c%d_t" % i], out["c%d_k" % i] = np.ascontiguousarray(q), np.ascontiguousarray(t), np.int64(k)
        out["c%d_score" % i] = np.int64(r["score"])
        out["c%d_blocks" % i], out["c%d_gapCounts" % i], out["c%d_gapItems" % i] = r["blocks"], r["gapCounts"], r["gapItems"]
    # KBandAlign (one matrix, linear gaps) in its four alignment types
    lin = []
    for i, (ql, k) in enumerate([(15, 10), (80, 12), (300, 12), (1200, 20), (1200, 60), (2500, 30)]):
        for at in (1, 2, 3, 7):
            q, t = synth.make_pair(rng, ql, flank=(0, int(rng.choice([0, 12]))))
            if (i + at) % 3 == 0:
                t = t[: max(12, len(t) - 9)]
            if (i + at) % 4 == 0:
                q = q[: max(12, len(q) - 11)]
            lin.append((q, t, k, at, int(rng.choice([5, 7])), int(rng.choice([5, 7])), 5, int(rng.choice([4, 5, 6]))))
    out["n_linear"] = np.int64(len(lin))
    for i, (q, t, k, at, ins, dele, fi, fd) in enumerate(lin):
        r = kband_align(ref, q, t, k, at, ins=ins, dele=dele, fn_ins=fi, fn_del=fd)
        out["l%d_q" % i], out["l%d_t" % i] = np.ascontiguousarray(q), np.ascontiguousarray(t)
        out["l%d_par" % i] = np.array([k, at, ins, dele, fi, fd], dtype=np.int64)
        out["l%d_res" % i] = np.array([r["score"], r["qPos"], r["tPos"]], dtype=np.int64)
        out["l%d_blocks" % i], out["l%d_gapCounts" % i], out["l%d_gapItems" % i] = r["blocks"], r["gapCounts"], r["gapItems"]
    path = os.path.join(HERE, "kband_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
