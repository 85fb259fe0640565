#!/usr/bin/env python3
"""Generate tests/golden/refine_golden.npz from the REFERENCE itself.

Run in a container that has /root/reference mounted, after `make -C oracle`
built oracle/_ref/libblasr_ref.so (the reference's own SDPAlign /
AffineGuidedAlign / SWAlign / ComputeAlignmentStats templates behind a C shim,
see oracle/ref_harness.cpp).  The fixture holds only data: seeded synthetic
inputs and the reference's outputs for them.

    python tests/golden/make_golden.py

Cases: PacBio-like pairs of several lengths, identical
sequences, corner-aligned pairs (prefix/suffix 5-mer ties), runs of N, a
planted long insertion and deletion (uncapped guide rows / straight-down
rows), query shorter than k, no shared k-mer, 30 kb at 15 % indel, an
asymmetric score matrix, non-default SDP parameters, Global SDP.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from blasr_testlib import LOCAL, GLOBAL, SMRT_MATRIX, ScoreParams, SdpParams, load_ref  # noqa: E402
from blasr_amd import synth  # noqa: E402


def main():
    ref = load_ref()
    assert ref is not None, "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(20261002)
    cases = []

    def add(name, q, t, sc=None, sp=None, band=16):
        cases.append((name, np.ascontiguousarray(q), np.ascontiguousarray(t), sc or ScoreParams(),
                      sp or SdpParams(), band))

    for i, qlen in enumerate([150, 400, 1000, 2500, 6000, 10000]):
        q, t = synth.make_pair(rng, qlen)
        add("pacbio_%d" % qlen, q, t)
    q, t = synth.make_pair(rng, 1200, ins=0, dele=0, sub=0, flank=(0, 0))
    add("identical_1200", q, t)
    q, t = synth.make_pair(rng, 1000, ins=0, dele=0, sub=0, flank=(30, 60))
    add("errorfree_flanked", q, t)
    q, t = synth.make_pair(rng, 2000, flank=(0, 0))
    add("corner_aligned", q, t)
    q, t = synth.make_pair(rng, 3000, n_run=40)
    add("n_run_target", q, t)
    q, t = synth.make_pair(rng, 2000)
    q = q.copy()
    q[700:730] = ord("N")
    add("n_run_query", q, t)
    q, t = synth.make_pair(rng, 4000, big_indel=("ins", 350))
    add("big_insertion", q, t)
    q, t = synth.make_pair(rng, 4000, big_indel=("del", 420))
    add("big_deletion", q, t)
    q, t = synth.make_pair(rng, 3000, big_indel=("del", 1500))
    add("huge_deletion", q, t)
    add("short_query", synth._ACGT[rng.integers(0, 4, size=8)], synth._ACGT[rng.integers(0, 4, size=300)])
    add("no_shared_kmer", np.frombuffer(b"A" * 200, dtype=np.uint8), np.frombuffer(b"C" * 260, dtype=np.uint8))
    q, t = synth.make_pair(rng, 30000, ins=0.10, dele=0.05, sub=0.01, flank=(0, 900))
    add("pacbio_30k_15pct", q, t)
    m = SMRT_MATRIX.copy()
    m[0, 1], m[1, 0], m[2, 3], m[3, 2], m[4, 0], m[0, 4] = 3, 7, 2, 9, 1, -1
    q, t = synth.make_pair(rng, 1500, n_run=10)
    add("asymmetric_matrix", q, t, sc=ScoreParams(matrix=m, ins=4, dele=7, affine_open=20, affine_extend=1))
    q, t = synth.make_pair(rng, 2500)
    add("k8_band30", q, t, sp=SdpParams(k=8, sdp_ins=3, sdp_del=6, indel_rate=0.6, recurse=3, recurse_over=400), band=30)
    q, t = synth.make_pair(rng, 1500, flank=(5, 40))
    add("global_sdp", q, t, sp=SdpParams(align_type=GLOBAL, extend_front=1))
    q, t = synth.make_pair(rng, 2500, ins=0.02, dele=0.2, sub=0.01)
    add("deletion_heavy", q, t)
    q, t = synth.make_pair(rng, 2500, ins=0.25, dele=0.02, sub=0.01)
    add("insertion_heavy", q, t)

    out = {"names": np.array([c[0] for c in cases])}
    for idx, (name, q, t, sc, sp, band) in enumerate(cases):
        p = "c%02d_" % idx
        out[p + "q"], out[p + "t"] = q, t
        out[p + "score"] = np.array([*sc.matrix.reshape(-1), sc.ins, sc.dele, sc.affine_open, sc.affine_extend], dtype=np.int32)
        out[p + "sdp_params"] = np.array([sp.k, sp.sdp_ins, sp.sdp_del, sp.align_type, sp.detailed, sp.extend_front,
                                          sp.prefix, sp.recurse, sp.recurse_over, sp.max_per_pos, band], dtype=np.int32)
        out[p + "indel_rate"] = np.array([sp.indel_rate], dtype=np.float32)
        s = ref.sdp_align(q, t, sc, sp)
        out[p + "sdp"] = np.array([s["score"], s["qPos"], s["tPos"]], dtype=np.int32)
        out[p + "sdp_blocks"] = s["blocks"].astype(np.int32)
        st = ref.compute_alignment_stats(q, t, s["qPos"], s["tPos"], s["blocks"], sc=sc)
        out[p + "sdp_stats"] = np.array([st["score"], st["nMatch"], st["nMismatch"], st["nIns"], st["nDel"]], dtype=np.int32)
        out[p + "sdp_pct"] = np.array([st["pct"]], dtype=np.float32)
        if len(s["blocks"]):
            # RefineAlignment's trimming (Blasr.cpp:850-859)
            bl = s["blocks"]
            qs = q[s["qPos"]: s["qPos"] + bl[-1, 0] + bl[-1, 2]]
            ts = t[s["tPos"]: s["tPos"] + bl[-1, 1] + bl[-1, 2]]
            rows, ncells = ref.alignment_to_guide(bl, band)
            out[p + "guide_digest"] = np.array([len(rows), ncells, int(rows[:, 2].max()), int(rows[:, 3].max()),
                                                int((rows[:, 1:4].astype(np.int64) * [3, 5, 7]).sum() % (1 << 31))], dtype=np.int64)
            a = ref.affine_guided_align(qs, ts, bl, sc, band)
            out[p + "aga"] = np.array([a["qPos"], a["tPos"], a["dpScore"], a["nCells"], a["score"], a["nMatch"],
                                       a["nMismatch"], a["nIns"], a["nDel"]], dtype=np.int32)
            out[p + "aga_pct"] = np.array([a["pct"]], dtype=np.float32)
            out[p + "aga_blocks"] = a["blocks"].astype(np.int32)
            out[p + "aga_gap_counts"] = a["gapCounts"].astype(np.int32)
            out[p + "aga_gap_items"] = a["gapItems"].astype(np.int32)
        print("%-20s q=%d t=%d sdp_blocks=%d" % (name, len(q), len(t), len(s["blocks"])))

    # small SWAlign(Global) and chain vectors
    sw = []
    for i in range(24):
        q = synth._ACGT[rng.integers(0, 4, size=int(rng.integers(1, 30)))]
        t = synth._ACGT[rng.integers(0, 4, size=int(rng.integers(1, 30)))]
        if i % 6 == 0:
            t = t.copy()
            t[0] = ord("N")
        r = ref.sw_align(q, t)
        out["sw%02d_q" % i], out["sw%02d_t" % i] = q, t
        out["sw%02d_score" % i] = np.array([r["score"]], dtype=np.int32)
        out["sw%02d_blocks" % i] = r["blocks"].astype(np.int32)
    out["n_sw"] = np.array([24])
    for i in range(12):
        L, k = int(rng.integers(40, 400)), int(rng.integers(5, 12))
        n = int(rng.integers(5, 300))
        xy = np.stack([rng.integers(0, L, size=n), rng.integers(0, L + 30, size=n)], 1)
        run = np.arange(int(rng.integers(0, L // 2)))
        xy = np.unique(np.concatenate([xy, np.stack([run, run + int(rng.integers(0, 8))], 1)]), axis=0)
        ln = np.where(rng.random(len(xy)) < 0.2, min(k, 5), k)
        xyl = np.concatenate([xy, ln[:, None]], 1).astype(np.int32)
        xyl = xyl[np.lexsort((xyl[:, 1], xyl[:, 0]))]
        out["ch%02d_xyl" % i] = xyl
        out["ch%02d_par" % i] = np.array([L + k, k], dtype=np.int32)
        out["ch%02d_local" % i] = ref.sdp_chain(L + k, xyl, k, align_type=LOCAL)
        out["ch%02d_global" % i] = ref.sdp_chain(L + k, xyl, k, align_type=GLOBAL)
    out["n_chain"] = np.array([12])
    path = os.path.join(HERE, "refine_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
