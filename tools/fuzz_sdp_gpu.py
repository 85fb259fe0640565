"""Randomised parity fuzz of the SDP stage's sweep kernels against the oracle's SDPAlign: Global calls with short words on
rectangles of unrelated or low-complexity sequence -- what the recursion issues on the gaps of a chance chain, and what
routes an invocation to one of the scattered sweep's kernels (sdp_chain_bits4_kernel<false> / <true>, sdp_chain_bits_kernel<W>,
sdp_chain_rows_kernel) -- in shapes drawn around the routing bounds (targets of 4096 / 16384 / 65536 columns, 1024 / 4096
fragments, four fragments per query row), mixed with Local calls on noisy pairs (sdp_chain4_kernel).
    python tools/fuzz_sdp_gpu.py SEED BATCHES
Prints one JSON line: invocations, how many differ, which sweep kernels ran and on how many invocations."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import blasr_amd  # noqa: E402
from blasr_amd import synth, hip  # noqa: E402
import blasr_testlib as T  # noqa: E402


def lowcomplex(rng, n, letters, run):
    out = np.empty(n, dtype=np.uint8)
    at = 0
    while at < n:
        c = ord("ACGT"[int(rng.integers(0, letters))])
        r = int(rng.integers(1, run + 1))
        out[at: at + r] = c
        at += r
    return out


def same(got, want, stats):
    if got["status"] != 0:
        return False
    if (got["score"], got["qPos"], got["tPos"]) != (want["score"], want["qPos"], want["tPos"]):
        return False
    if not np.array_equal(got["blocks"], want["blocks"]):
        return False
    s = got["stats"]
    return [s["score"], s["nMatch"], s["nMismatch"], s["nIns"], s["nDel"]] == stats


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    batches = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    rng = np.random.default_rng(seed)
    oracle = T.load_oracle()
    ctx = blasr_amd.Context(0)
    total = bad = 0
    ran = {}
    examples = []
    t0 = time.time()
    for b in range(batches):
        k = int(rng.choice([5, 7, 7, 6, 8]))
        rect = []
        for i in range(int(rng.integers(6, 16))):
            kind = int(rng.integers(0, 6))
            s1, s2 = int(rng.integers(1, 1 << 30)), int(rng.integers(1, 1 << 30))
            if kind == 0:    # a target just around the four-per-wavefront form's column bounds
                t = int(rng.choice([4095, 4096, 4097, 4100, 16383, 16384, 16385, 9000, 12000]))
                q = int(rng.integers(30, 3000))
            elif kind == 1:  # fragment counts around 1024 / 4096: q x t / 4^k
                want_f = int(rng.choice([900, 1100, 3900, 4300, 2000]))
                t = int(rng.integers(2000, 16000))
                q = max(20, min(6000, want_f * (4 ** k) // t))
            elif kind == 2:  # dense rows (the row sweep's domain) and sparse ones
                t = int(rng.integers(3000, 20000))
                q = int(rng.integers(200, 1500))
            elif kind == 3:  # the larger bitmap classes
                t = int(rng.choice([17000, 30000, 66000]))
                q = int(rng.integers(300, 2500))
            elif kind == 4:  # low-complexity sequence: rows of dozens of fragments, ties among costs
                letters, run = [(2, 3), (3, 4), (4, 6)][int(rng.integers(0, 3))]
                rect.append((lowcomplex(rng, int(rng.integers(100, 700)), letters, run), lowcomplex(rng, int(rng.integers(200, 6000)), letters, run)))
                continue
            else:            # small ones: the fused kernels and the four-per-wavefront sweeps' short invocations
                t = int(rng.integers(60, 4000))
                q = int(rng.integers(30, 1200))
            rect.append((synth.random_genome(q, s1), synth.random_genome(t, s2)))
        osp = T.SdpParams(k=k, align_type=T.GLOBAL, prefix=0, recurse=int(rng.integers(0, 3)))
        sp = blasr_amd.SdpParams(k=k, align_type=1, prefix=0, recurse=osp.recurse)
        want = [oracle.sdp_align(q, t, None, osp) for q, t in rect]
        got = ctx.sdp(rect, sdp=sp)
        for r in ctx.sdp_last_kernels():
            if "chain" in r["name"]:
                ran[r["name"].split("::")[-1]] = ran.get(r["name"].split("::")[-1], 0) + r["invocations"]
        for i, ((q, t), g, w) in enumerate(zip(rect, got, want)):
            st = oracle.compute_alignment_stats(q, t, w["qPos"], w["tPos"], w["blocks"])
            ok = same(g, w, [st["score"], st["nMatch"], st["nMismatch"], st["nIns"], st["nDel"]])
            total += 1
            if not ok:
                bad += 1
                if len(examples) < 5:
                    examples.append(dict(batch=b, i=i, k=k, q=len(q), t=len(t), recurse=osp.recurse))
        # Local calls on noisy pairs (the top level's near-diagonal sweep), default parameters
        pairs = []
        for i in range(int(rng.integers(2, 8))):
            t = synth.random_genome(int(rng.integers(500, 6000)), int(rng.integers(1, 1 << 30)))
            pairs.append((np.ascontiguousarray(synth.mutate(t, rng)), t))
        wantL = [oracle.sdp_align(q, t) for q, t in pairs]
        gotL = ctx.sdp(pairs)
        for r in ctx.sdp_last_kernels():
            if "chain" in r["name"]:
                ran[r["name"].split("::")[-1]] = ran.get(r["name"].split("::")[-1], 0) + r["invocations"]
        for (q, t), g, w in zip(pairs, gotL, wantL):
            st = oracle.compute_alignment_stats(q, t, w["qPos"], w["tPos"], w["blocks"])
            total += 1
            if not same(g, w, [st["score"], st["nMatch"], st["nMismatch"], st["nIns"], st["nDel"]]):
                bad += 1
                if len(examples) < 5:
                    examples.append(dict(batch=b, local=True, q=len(q), t=len(t)))
    ctx.close()
    print(json.dumps(dict(seed=seed, batches=batches, calls=total, differ=bad, sweep_kernel_invocations=ran, examples=examples,
                          seconds=round(time.time() - t0, 1), build_id=hip.build_id())))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
