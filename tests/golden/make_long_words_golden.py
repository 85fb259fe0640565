#!/usr/bin/env python3
"""Generate tests/golden/long_words_golden.npz: the REFERENCE's own SDPAlign (algorithms/alignment/SDPAlign.h:95-626 through
oracle/_ref/libblasr_ref.so) with -sdpTupleSize 14 .. 32 on the pairs blasr_testlib.long_word_cases makes -- the words the
device keeps in a 16-base sort key + a comparison of the rest.  Data only: sequences, parameters, and what the reference
returned (score, qPos, tPos, blocks).

    python tests/golden/make_long_words_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from blasr_testlib import load_ref, long_word_cases  # noqa: E402


def main():
    ref = load_ref()
    assert ref is not None, "build oracle/_ref first (make -C oracle)"
    out, n, with_blocks = {}, 0, 0
    for q, t, sp in long_word_cases(np.random.default_rng(20261006), 48):
        r = ref.sdp_align(q, t, None, sp)
        p = "c%d_" % n
        out[p + "q"], out[p + "t"] = q, t
        out[p + "par"] = np.array([sp.k, sp.align_type, sp.recurse, sp.recurse_over, sp.max_per_pos, sp.prefix, sp.detailed,
                                   r["score"], r["qPos"], r["tPos"]], np.int64)
        out[p + "blocks"] = np.asarray(r["blocks"], np.int32).reshape(-1)
        with_blocks += len(r["blocks"]) > 0
        n += 1
    out["n"] = n
    np.savez_compressed(os.path.join(HERE, "long_words_golden.npz"), **out)
    print("wrote %d cases, %d with blocks" % (n, with_blocks))


if __name__ == "__main__":
    main()
