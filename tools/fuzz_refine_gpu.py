"""Randomised parity fuzz of the refine stage's fill kernels on guides with jumps (the hybrid fillp_kernel's domain) against
the oracle's AffineGuidedAlign: guides of 2-8 short exact blocks with gaps of 0-6000 columns / rows in either direction,
back to back or in the first rows, mixed with plain noisy pairs so that the candidates of a wavefront differ.
    python tools/fuzz_refine_gpu.py SEED BATCHES
    python tools/fuzz_refine_gpu.py SEED BATCHES narrow
"narrow" is fill8_kernel's domain instead: batches of up to 200 noisy pairs of 30-4000 bases whose guides have gaps of at most
~75 columns (rows of up to 112 cells: one to three passes of 40 columns, candidates of one wavefront at different rows), with N
bases in either sequence, targets that end inside the band, and random bands."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import blasr_amd  # noqa: E402
from blasr_amd import synth  # noqa: E402
import blasr_testlib as T  # noqa: E402
from golden_cases import REFINE_KEYS, first_difference  # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    batches = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    narrow = len(sys.argv) > 3 and sys.argv[3] == "narrow"
    rng = np.random.default_rng(seed)
    oracle = T.load_oracle()
    ctx = blasr_amd.Context(0)
    bad = total = 0
    t0 = time.time()
    for _ in range(batches):
        b, want = blasr_amd.RefineBatch(), []
        band = 16
        if narrow:
            band = int(rng.choice([16, 16, 8, 12, 20]))
            for it in range(int(rng.integers(1, 200))):
                big = None
                if rng.integers(0, 3) == 0:
                    big = (str(rng.choice(["del", "ins"])), int(rng.integers(5, 76)))
                q, t = synth.make_pair(rng, int(rng.integers(30, 4000)), flank=(0, int(rng.integers(0, 60))), big_indel=big)
                for seq in (q, t):
                    if rng.integers(0, 6) == 0:
                        seq[rng.integers(0, len(seq), size=int(rng.integers(1, 6)))] = ord("N")
                s = oracle.sdp_align(q, t)
                bl = s["blocks"]
                if len(bl) == 0:
                    continue
                qs = np.ascontiguousarray(q[s["qPos"]: s["qPos"] + bl[-1, 0] + bl[-1, 2]])
                cut = int(rng.integers(0, 3)) == 0  # the reference's window, or a target that ends with the last block
                ts = np.ascontiguousarray(t[s["tPos"]: (s["tPos"] + bl[-1, 1] + bl[-1, 2]) if cut else len(t)])
                b.add(qs, ts, bl)
                want.append(oracle.affine_guided_align(qs, ts, bl, band=band))
        for it in range(0 if narrow else int(rng.integers(1, 40))):
            if rng.integers(0, 4) == 0:  # a plain noisy pair with the SDP blocks of the oracle as its guide
                q, t = synth.make_pair(rng, int(rng.integers(100, 3000)))
                s = oracle.sdp_align(q, t)
                bl = s["blocks"]
                if len(bl) == 0:
                    continue
                qs = q[s["qPos"]: s["qPos"] + bl[-1, 0] + bl[-1, 2]]
                ts = t[s["tPos"]: s["tPos"] + bl[-1, 1] + bl[-1, 2]]
            else:
                nb = int(rng.integers(2, 9))
                qp = tp = int(rng.integers(0, 30))
                blocks = []
                for k in range(nb):
                    length = int(rng.integers(6, 50))
                    blocks.append([qp, tp, length])
                    step = int(rng.integers(0, 300))
                    gap = int(rng.choice([0, 0, 20, 100, 150, 400, 1000, 3000, 6000])) if k + 1 < nb else 0
                    kind = int(rng.integers(0, 3))
                    qp += length + step + (gap if kind == 1 else 0) + (gap // 4 if kind == 2 else 0)
                    tp += length + step + (gap if kind == 0 else 0) + (gap // 3 if kind == 2 else 0)
                qs = synth.random_genome(qp + 20, int(rng.integers(1, 1 << 30)))
                ts = synth.random_genome(tp + 20, int(rng.integers(1, 1 << 30)))
                for (bq, bt, ln) in blocks:
                    ts[bt: bt + ln] = qs[bq: bq + ln]
                bl = np.array(blocks, dtype=np.int32)
                qs, ts = qs[: blocks[-1][0] + blocks[-1][2]], ts[: blocks[-1][1] + blocks[-1][2]]
            b.add(np.ascontiguousarray(qs), np.ascontiguousarray(ts), bl)
            want.append(oracle.affine_guided_align(np.ascontiguousarray(qs), np.ascontiguousarray(ts), bl))
        if not want:
            continue
        got = ctx.refine(b, band=band)
        for i, (g, w) in enumerate(zip(got, want)):
            total += 1
            if g.status != 0 or first_difference(g.as_dict(), w, REFINE_KEYS) is not None:
                bad += 1
                print("DIFF seed", seed, "candidate", i, g.status)
    print("seed", seed, "candidates", total, "bad", bad, "seconds", round(time.time() - t0, 1))
    ctx.close()


if __name__ == "__main__":
    main()
