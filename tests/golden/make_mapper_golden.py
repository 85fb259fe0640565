#!/usr/bin/env python3
"""Generate tests/golden/mapper_golden.npz from the REFERENCE mapper itself (a-1 and a-10).

Needs oracle/_ref/libblasr_refmapper.so (`make -C oracle`, reference mounted): the reference's own
alignment/Blasr.cpp behind the C shim of oracle/ref_mapper_harness.cpp.  The fixture holds only data:
seeded synthetic inputs and what the reference's AlignIntervals / filters produced for them.

    python tests/golden/make_mapper_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from blasr_testlib import WINDOW_FIELDS, load_refmapper, random_candidates, random_interval_case  # noqa: E402
from blasr_amd import synth  # noqa: E402

N_WINDOWS, N_FILTER = 36, 40


def main():
    ref = load_refmapper()
    assert ref is not None, "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(20261003)
    genome = synth.random_genome(40000, rng)
    genome[12000:12030] = ord("N")
    genome[20000:20010] = np.frombuffer(b"acgtnACGTN", dtype=np.uint8)  # lower case through CopyAsRC
    out = dict(genome=genome, n_windows=np.int64(N_WINDOWS), n_filter=np.int64(N_FILTER))
    for i in range(N_WINDOWS):
        read, subread, iv, contigs = random_interval_case(rng, genome, i)
        win, tseq, sdp = ref.align_interval(genome, read, subread, iv, contigs)
        out["w%d_read" % i] = read
        out["w%d_in" % i] = np.array(list(subread) + list(iv), dtype=np.int64)
        out["w%d_contigs" % i] = contigs if contigs is not None else np.zeros(0, np.uint32)
        out["w%d_win" % i] = np.array([win[f] for f in WINDOW_FIELDS], dtype=np.int64)
        out["w%d_tseq" % i] = tseq
        out["w%d_sdp" % i] = np.array([sdp["qPos"], sdp["tPos"], sdp["score"], sdp["nMatch"], sdp["nMismatch"],
                                       sdp["nIns"], sdp["nDel"]], dtype=np.int64)
        out["w%d_blocks" % i] = sdp["blocks"]
    for i in range(N_FILTER):
        n = int(rng.choice([0, 1, 2, 5, 10, 12, 40]))
        c = random_candidates(rng, n)
        params = dict(n_candidates=int(rng.choice([10, 3])), max_gap=int(rng.choice([0, 0, 25])), max_score=-200,
                      min_pct=float(rng.choice([0.0, 70.0])))
        for k, v in c.items():
            out["f%d_%s" % (i, k)] = v
        out["f%d_params" % i] = np.array([params["n_candidates"], params["max_gap"], params["max_score"],
                                          params["min_pct"]], dtype=np.float64)
        for op in range(3):
            kept, score = ref.post_filter(op, c["score"], c["t_pos"], c["t_aligned_seq_pos"], c["t_index"],
                                          c["t_strand"], c["pct"], c["blocks"], c["blk_off"], c["n_blocks"],
                                          params["n_candidates"], params["max_gap"], params["max_score"],
                                          params["min_pct"])
            out["f%d_op%d_kept" % (i, op)] = kept
            out["f%d_op%d_score" % (i, op)] = score
    path = os.path.join(HERE, "mapper_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
