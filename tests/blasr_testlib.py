"""ctypes access to the two checkers used by the tests:

* ``oracle/libblasr_oracle.so``  -- the plain-C restatement (prefix ``bo_``)
* ``oracle/_ref/libblasr_ref.so`` -- the reference's own templates behind a C
  shim (prefix ``ref_``); present only where ``make -C oracle`` ran with
  /root/reference mounted (the prebuilt file travels to the GPU box).

Both export the same flat-array signatures, so one wrapper class drives both.
Nothing in blasr_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.environ.get("BLASR_ORACLE_SO") or os.path.join(ROOT, "oracle", "libblasr_oracle.so")  # (the sanitizer build)
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libblasr_ref.so")

LOCAL, GLOBAL = 0, 1
GAP_QUERY, GAP_TARGET = 0, 1

# ScoreMatrices.h:22-28 (SMRTDistanceMatrix) -- data, the default score matrix.
SMRT_MATRIX = np.array(
    [[-5, 6, 6, 6, 0], [6, -5, 6, 6, 0], [6, 6, -5, 6, 0], [6, 6, 6, -5, 0], [0, 0, 0, 0, 0]],
    dtype=np.int32,
)


class ScoreParams:
    """DistanceMatrixScoreFunction as Blasr.cpp:1175-1177 / :807-809 set it up."""

    def __init__(self, matrix=None, ins=5, dele=5, affine_open=50, affine_extend=0):
        self.matrix = np.ascontiguousarray(SMRT_MATRIX if matrix is None else matrix, dtype=np.int32)
        self.ins, self.dele = ins, dele
        self.affine_open, self.affine_extend = affine_open, affine_extend


class SdpParams:
    """Defaults of MappingParameters.h:192-361 as they reach SDPAlign (Blasr.cpp:1716-1722)."""

    def __init__(self, k=11, sdp_ins=5, sdp_del=10, indel_rate=0.9, align_type=LOCAL, detailed=1,
                 extend_front=0, prefix=50, recurse=2, recurse_over=1000, max_per_pos=0):
        self.k, self.sdp_ins, self.sdp_del, self.indel_rate = k, sdp_ins, sdp_del, indel_rate
        self.align_type, self.detailed, self.extend_front = align_type, detailed, extend_front
        self.prefix, self.recurse, self.recurse_over, self.max_per_pos = prefix, recurse, recurse_over, max_per_pos


def _u8(b):
    if isinstance(b, (bytes, bytearray)):
        return np.frombuffer(bytes(b), dtype=np.uint8)
    return np.ascontiguousarray(b, dtype=np.uint8)


def _p(a, t=C.c_int):
    return a.ctypes.data_as(C.POINTER(t))


class Checker:
    """One of the two CPU checkers."""

    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        self.name = prefix.rstrip("_")

    def fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def sdp_align(self, q, t, sc=None, sp=None):
        sc, sp = sc or ScoreParams(), sp or SdpParams()
        q, t = _u8(q), _u8(t)
        cap = len(q) + len(t) + 16
        blocks = np.zeros(3 * cap, dtype=np.int32)
        qpos, tpos, nb = C.c_int(0), C.c_int(0), C.c_int(0)
        f = self.fn("sdp_align")
        f.restype = C.c_int
        score = f(_p(q, C.c_ubyte), len(q), _p(t, C.c_ubyte), len(t), _p(sc.matrix), sc.ins, sc.dele,
                  sc.affine_open, sc.affine_extend, sp.k, sp.sdp_ins, sp.sdp_del, C.c_float(sp.indel_rate),
                  sp.align_type, sp.detailed, sp.extend_front, sp.prefix, sp.recurse, sp.recurse_over,
                  sp.max_per_pos, C.byref(qpos), C.byref(tpos), _p(blocks), cap, C.byref(nb))
        return dict(score=score, qPos=qpos.value, tPos=tpos.value,
                    blocks=blocks[: 3 * nb.value].reshape(-1, 3).copy())

    def sdp_chain(self, query_length, xyl, k, ins=5, dele=10, match=-5, align_type=LOCAL):
        xyl = np.ascontiguousarray(xyl, dtype=np.int32).reshape(-1, 3)
        chain = np.zeros(max(len(xyl), 1), dtype=np.int32)
        f = self.fn("sdp_chain")
        f.restype = C.c_int
        n = f(query_length, _p(xyl), len(xyl), k, ins, dele, match, align_type, _p(chain), len(chain))
        return chain[:n].copy()

    def sw_align(self, q, t, sc=None):
        sc = sc or ScoreParams()
        q, t = _u8(q), _u8(t)
        cap = len(q) + len(t) + 4
        blocks = np.zeros(3 * cap, dtype=np.int32)
        qpos, tpos, nb = C.c_int(0), C.c_int(0), C.c_int(0)
        f = self.fn("sw_align")
        f.restype = C.c_int
        score = f(_p(q, C.c_ubyte), len(q), _p(t, C.c_ubyte), len(t), _p(sc.matrix), sc.ins, sc.dele, GLOBAL,
                  C.byref(qpos), C.byref(tpos), _p(blocks), cap, C.byref(nb))
        return dict(score=score, blocks=blocks[: 3 * nb.value].reshape(-1, 3).copy())

    def alignment_to_guide(self, blocks, band=16):
        blocks = np.ascontiguousarray(blocks, dtype=np.int32).reshape(-1, 3)
        cap = int(blocks[-1, 0] + blocks[-1, 2] - blocks[0, 0] + 2) if len(blocks) else 1
        rows = np.zeros(5 * cap, dtype=np.int32)
        ncells = C.c_int(0)
        f = self.fn("alignment_to_guide")
        f.restype = C.c_int
        n = f(_p(blocks), len(blocks), band, _p(rows), cap, C.byref(ncells))
        return rows[: 5 * n].reshape(-1, 5).copy(), ncells.value

    def affine_guided_align(self, q, t, guide_blocks, sc=None, band=16):
        sc = sc or ScoreParams()
        q, t = _u8(q), _u8(t)
        guide_blocks = np.ascontiguousarray(guide_blocks, dtype=np.int32).reshape(-1, 3)
        cap = len(q) + len(t) + 16
        blocks = np.zeros(3 * cap, dtype=np.int32)
        gap_counts = np.zeros(cap, dtype=np.int32)
        gap_items = np.zeros(2 * cap, dtype=np.int32)
        stats = np.zeros(7, dtype=np.int32)
        pct = C.c_float(0)
        qpos, tpos, nb, ngl, ngi = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        f = self.fn("affine_guided_align")
        f.restype = C.c_int
        rc = f(_p(q, C.c_ubyte), len(q), _p(t, C.c_ubyte), len(t), _p(guide_blocks), len(guide_blocks),
               _p(sc.matrix), sc.ins, sc.dele, sc.affine_open, sc.affine_extend, band, C.byref(qpos),
               C.byref(tpos), _p(blocks), cap, C.byref(nb), _p(gap_counts), cap, C.byref(ngl), _p(gap_items),
               cap, C.byref(ngi), _p(stats), C.byref(pct))
        assert rc == 0, rc
        return dict(qPos=qpos.value, tPos=tpos.value, dpScore=int(stats[0]), nCells=int(stats[1]),
                    score=int(stats[2]), nMatch=int(stats[3]), nMismatch=int(stats[4]), nIns=int(stats[5]),
                    nDel=int(stats[6]), pct=pct.value,
                    blocks=blocks[: 3 * nb.value].reshape(-1, 3).copy(),
                    gapCounts=gap_counts[: ngl.value].copy(),
                    gapItems=gap_items[: 2 * ngi.value].reshape(-1, 2).copy())

    def compute_drift(self, blocks):
        """ComputeDrift (AlignmentUtils.h:586-618)."""
        blocks = np.ascontiguousarray(blocks, dtype=np.int32).reshape(-1, 3)
        f = self.fn("compute_drift")
        f.restype = C.c_int
        return f(_p(blocks), len(blocks))

    def refine_nonguided(self, q, t, sdp, sc=None, indel=5, band=0):
        """RefineAlignment under -noUseGuidedAlign (Blasr.cpp:917-1046) for the candidate of one window: q / t = the
        window's query / target, sdp = its SDP alignment (qPos, tPos, blocks).  Returns drift, k, kband (the k handed to
        AffineKBandAlign), dpScore, score (ComputeAlignmentStats' non-affine form WITH the matrix in place -- the
        reference's own value is uninitialised memory), counts, pct, qs / ts (the trimmed sequences), blocks, gap lists;
        the candidate's qPos / tPos are 0 afterwards."""
        sc = sc or ScoreParams()
        q, t = _u8(q), _u8(t)
        sb = np.ascontiguousarray(sdp["blocks"], dtype=np.int32).reshape(-1, 3)
        if not hasattr(self.lib, self.prefix + "refine_nonguided"):
            # the reference build (oracle/_ref/libblasr_ref.so) exports its templates one by one: the branch composed
            # from its own AffineKBandAlign and ComputeAlignmentStats, the drift and the band rule as Blasr.cpp:943-1003
            if len(sb) == 0:
                return dict(drift=0, k=0, kband=0, dpScore=0, score=0, nMatch=0, nMismatch=0, nIns=0, nDel=0, pct=0.0, qPos=0,
                            tPos=0, qs=q[:0], ts=t[:0], nCells=0, blocks=sb, gapCounts=np.zeros(0, np.int32),
                            gapItems=np.zeros((0, 2), np.int32))
            gaps = (sb[1:, 1] - sb[:-1, 1] - sb[:-1, 2]).astype(np.int64) - (sb[1:, 0] - sb[:-1, 0] - sb[:-1, 2])
            drift = int(np.abs(np.cumsum(gaps)).max()) if len(gaps) else 0
            k = max(10, int(drift * 1.5) if band == 0 else band)
            kk = int(k * 1.2)
            qs = q[int(sdp["qPos"]): int(sdp["qPos"]) + int(sb[-1, 0] + sb[-1, 2])]
            ts = t[int(sdp["tPos"]): int(sdp["tPos"]) + int(sb[-1, 1] + sb[-1, 2] - sb[0, 1])]
            a = affine_kband_align(self, qs, ts, kk, matrix=sc.matrix, indel=indel)
            st = self.compute_alignment_stats(qs, ts, 0, 0, a["blocks"], a["gapCounts"], a["gapItems"], sc=sc, affine=0)
            q_len = min(len(qs), len(ts) + kk) if len(ts) < len(qs) else len(qs)
            return dict(drift=drift, k=k, kband=kk, dpScore=a["score"], score=st["score"], nMatch=st["nMatch"],
                        nMismatch=st["nMismatch"], nIns=st["nIns"], nDel=st["nDel"], pct=st["pct"], qPos=0, tPos=0, qs=qs, ts=ts,
                        nCells=(q_len + 1) * (2 * kk + 1), blocks=a["blocks"], gapCounts=a["gapCounts"], gapItems=a["gapItems"])
        cap = len(q) + len(t) + 16
        blocks = np.zeros(3 * cap, dtype=np.int32)
        gap_counts, gap_items = np.zeros(cap, dtype=np.int32), np.zeros(2 * cap, dtype=np.int32)
        out = np.zeros(8, dtype=np.int32)
        pct = C.c_float(0)
        score, ql, tl, nb, ngl, ngi = (C.c_int(0) for _ in range(6))
        f = self.fn("refine_nonguided")
        f.restype = C.c_int
        rc = f(_p(q, C.c_ubyte), _p(t, C.c_ubyte), int(sdp["qPos"]), int(sdp["tPos"]), _p(sb), len(sb), _p(sc.matrix), sc.ins,
               sc.dele, int(indel), int(band), _p(out), C.byref(score), C.byref(pct), C.byref(ql), C.byref(tl), _p(blocks), cap,
               C.byref(nb), _p(gap_counts), cap, C.byref(ngl), _p(gap_items), cap, C.byref(ngi))
        assert rc == 0, rc
        qs = q[int(sdp["qPos"]): int(sdp["qPos"]) + ql.value]
        ts = t[int(sdp["tPos"]): int(sdp["tPos"]) + tl.value]
        kk = int(out[2])
        q_len = min(len(qs), len(ts) + kk) if len(ts) < len(qs) else len(qs)  # SetKBoundedLengths (KBandAlign.h:36-56)
        return dict(drift=int(out[0]), k=int(out[1]), kband=kk, dpScore=int(out[3]), score=score.value, nMatch=int(out[4]),
                    nMismatch=int(out[5]), nIns=int(out[6]), nDel=int(out[7]), pct=pct.value, qPos=0, tPos=0, qs=qs, ts=ts,
                    nCells=(q_len + 1) * (2 * kk + 1) if len(sb) else 0,
                    blocks=blocks[: 3 * nb.value].reshape(-1, 3).copy(), gapCounts=gap_counts[: ngl.value].copy(),
                    gapItems=gap_items[: 2 * ngi.value].reshape(-1, 2).copy())

    def refine_candidate(self, q, t, sc=None, sp=None, band=16):
        """SDPAlign(Local) -> stats -> AffineGuidedAlign -> stats for one candidate window (oracle only)."""
        sc, sp = sc or ScoreParams(), sp or SdpParams()
        q, t = _u8(q), _u8(t)
        cap = len(q) + len(t) + 16
        blocks = np.zeros(3 * cap, dtype=np.int32)
        gap_counts = np.zeros(cap, dtype=np.int32)
        gap_items = np.zeros(2 * cap, dtype=np.int32)
        out = np.zeros(12, dtype=np.int32)
        pct = C.c_float(0)
        qpos, tpos, nb, ngl, ngi = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        f = self.fn("refine_candidate")
        f.restype = C.c_int
        rc = f(_p(q, C.c_ubyte), len(q), _p(t, C.c_ubyte), len(t), _p(sc.matrix), sc.ins, sc.dele, sc.affine_open,
               sc.affine_extend, sp.k, sp.sdp_ins, sp.sdp_del, C.c_float(sp.indel_rate), sp.prefix, sp.recurse,
               sp.recurse_over, band, _p(out), C.byref(pct), C.byref(qpos), C.byref(tpos), _p(blocks), cap,
               C.byref(nb), _p(gap_counts), cap, C.byref(ngl), _p(gap_items), cap, C.byref(ngi))
        assert rc == 0, rc
        return dict(sdpScore=int(out[0]), sdpStatsScore=int(out[1]), sdpQPos=int(out[2]), sdpTPos=int(out[3]),
                    nSdpBlocks=int(out[4]), dpScore=int(out[5]), nCells=int(out[6]), score=int(out[7]),
                    nMatch=int(out[8]), nMismatch=int(out[9]), nIns=int(out[10]), nDel=int(out[11]), pct=pct.value,
                    qPos=qpos.value, tPos=tpos.value, blocks=blocks[: 3 * nb.value].reshape(-1, 3).copy(),
                    gapCounts=gap_counts[: ngl.value].copy(), gapItems=gap_items[: 2 * ngi.value].reshape(-1, 2).copy())

    def compute_alignment_stats(self, q, t, qpos, tpos, blocks, gap_counts=None, gap_items=None, sc=None,
                                affine=1):
        sc = sc or ScoreParams()
        q, t = _u8(q), _u8(t)
        blocks = np.ascontiguousarray(blocks, dtype=np.int32).reshape(-1, 3)
        gc = np.ascontiguousarray(gap_counts if gap_counts is not None else [], dtype=np.int32)
        gi = np.ascontiguousarray(gap_items if gap_items is not None else [], dtype=np.int32).reshape(-1)
        stats = np.zeros(5, dtype=np.int32)
        pct = C.c_float(0)
        f = self.fn("compute_alignment_stats")
        f.restype = None
        f(_p(q, C.c_ubyte), _p(t, C.c_ubyte), qpos, tpos, _p(blocks), len(blocks),
          _p(gc) if len(gc) else None, len(gc), _p(gi) if len(gi) else None, _p(sc.matrix), sc.ins, sc.dele,
          sc.affine_open, sc.affine_extend, affine, _p(stats), C.byref(pct))
        return dict(score=int(stats[0]), nMatch=int(stats[1]), nMismatch=int(stats[2]), nIns=int(stats[3]),
                    nDel=int(stats[4]), pct=pct.value)


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libblasr_oracle.so"])


def load_oracle():
    if not os.path.exists(ORACLE_SO):
        build_oracle()
    return Checker(ORACLE_SO, "bo_")


def load_ref():
    """The reference build, or None where it is absent."""
    if not os.path.exists(REF_SO):
        return None
    return Checker(REF_SO, "ref_")


def ops_from_alignment(res):
    """(type,len) op list (0=M, 1=I/Gap::Target, 2=D/Gap::Query) from blocks + gap lists."""
    ops = []
    gc, gi = res["gapCounts"], res["gapItems"]
    starts = np.concatenate([[0], np.cumsum(gc)])
    for b in range(len(res["blocks"])):
        ops.append((0, int(res["blocks"][b, 2])))
        if b + 1 < len(gc) and b + 1 < len(res["blocks"]):
            for j in range(starts[b + 1], starts[b + 2]):
                ops.append((1 if gi[j, 0] == GAP_TARGET else 2, int(gi[j, 1])))
    return ops


def ops_with_leading_gaps(res):
    """As ops_from_alignment, for an alignment that keeps its first gap list (AffineKBandAlign's: ArrowPathToAlignment
    without RemoveAlignmentPrefixGaps; the gap run behind the last block is dropped there, Alignment.h:190-254)."""
    ops = []
    gc, gi = res["gapCounts"], res["gapItems"]
    starts = np.concatenate([[0], np.cumsum(gc)])
    if len(gc) and len(res["blocks"]):
        for j in range(starts[0], starts[1]):
            ops.append((1 if gi[j, 0] == GAP_TARGET else 2, int(gi[j, 1])))
    return ops + ops_from_alignment(res)


# ---------------------------------------------------------------------------------------------
# a-1 / a-10: window set-up of AlignIntervals and the candidate filters.  The oracle restates them
# (oracle/blasr_oracle.c: bo_interval_window, bo_copy_window, bo_sort_by_score, bo_remove_*); the
# reference's own code is reached through oracle/_ref/libblasr_refmapper.so (ref_mapper_harness.cpp).
# ---------------------------------------------------------------------------------------------
REFMAPPER_SO = os.path.join(ROOT, "oracle", "_ref", "libblasr_refmapper.so")
WINDOW_FIELDS = ("tAlignedSeqPos", "tAlignedSeqLength", "tStrand", "tIndex", "tLength", "qAlignedSeqPos",
                 "qAlignedSeqLength")


def _pu(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_uint))


class MapperOracle:
    """bo_interval_window / bo_copy_window / filters of the plain-C oracle."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        self.lib = C.CDLL(ORACLE_SO)

    def window(self, genome, read_len, subread, iv, contigs=None, rate=1.3, extend_ends=1, do_global=0):
        """iv = (start, end, qStart, qEnd, strand).  Returns (fields dict, target window bytes)."""
        genome = _u8(genome)
        cs = None if contigs is None else np.ascontiguousarray(contigs, dtype=np.uint32)
        out = np.zeros(8, dtype=np.int32)
        self.lib.bo_interval_window(len(genome), int(read_len), int(subread[0]), int(subread[1]), int(iv[0]), int(iv[1]),
                                    int(iv[2]), int(iv[3]), int(iv[4]), _pu(cs), 0 if cs is None else len(cs) - 1,
                                    C.c_float(rate), int(extend_ends), int(do_global), _p(out))
        t = np.zeros(max(int(out[1]), 1), dtype=np.uint8)
        self.lib.bo_copy_window(_p(genome, C.c_ubyte), C.c_uint(int(out[7]) & 0xffffffff), int(out[1]), int(out[2]),
                                _p(t, C.c_ubyte))
        return dict(zip(WINDOW_FIELDS, (int(v) & 0xffffffff for v in out[:7]))), t[: int(out[1])]

    def sort_by_score(self, score, t_pos, t_aligned_seq_pos, order=None):
        score = np.ascontiguousarray(score, dtype=np.int32)
        order = np.arange(len(score), dtype=np.int32) if order is None else np.ascontiguousarray(order, np.int32).copy()
        self.lib.bo_sort_by_score(len(order), _p(score), _pu(np.ascontiguousarray(t_pos, np.uint32)),
                                  _pu(np.ascontiguousarray(t_aligned_seq_pos, np.uint32)), _p(order))
        return order

    def remove_low_quality(self, order, score, blocks, blk_off, n_blocks, n_candidates, max_gap, max_score):
        order = np.ascontiguousarray(order, np.int32).copy()
        score = np.ascontiguousarray(score, np.int32).copy()
        blocks = np.ascontiguousarray(blocks, np.int32).reshape(-1)
        if blocks.size == 0:
            blocks = np.zeros(3, np.int32)
        n = self.lib.bo_remove_low_quality(len(order), _p(order), _p(score), _p(blocks),
                                           _p(np.ascontiguousarray(blk_off, np.int32)),
                                           _p(np.ascontiguousarray(n_blocks, np.int32)), n_candidates, max_gap, max_score)
        return order[:n], score

    def remove_overlapping(self, order, score, t_pos, t_aligned_seq_pos, t_end, t_index, t_strand, pct, min_pct):
        order = np.ascontiguousarray(order, np.int32).copy()
        pct = np.ascontiguousarray(pct, np.float32)
        n = self.lib.bo_remove_overlapping(len(order), _p(order), _p(np.ascontiguousarray(score, np.int32)),
                                           _pu(np.ascontiguousarray(t_pos, np.uint32)),
                                           _pu(np.ascontiguousarray(t_aligned_seq_pos, np.uint32)),
                                           _pu(np.ascontiguousarray(t_end, np.uint32)),
                                           _p(np.ascontiguousarray(t_index, np.int32)),
                                           _p(np.ascontiguousarray(t_strand, np.int32)),
                                           pct.ctypes.data_as(C.POINTER(C.c_float)), C.c_float(min_pct))
        return order[:n]


class MapperRef:
    """The reference mapper's own AlignIntervals / filters (ref_mapper_harness.cpp)."""

    def __init__(self):
        self.lib = C.CDLL(REFMAPPER_SO)

    def align_interval(self, genome, read, subread, iv, contigs=None, rate=1.3, extend_ends=1, k=11, extend=None, rbao_chain=None,
                       nonguided_band=None):
        """extend: None, or (extendBandSize, maxExtendDropoff) to run the reference's -extend step as well.
        rbao_chain: None, or the interval's chained anchors ((m, 3) uint32: t, q, l) to run AlignIntervals' -rbao branch
        (Blasr.cpp:1354-1712) on them instead of SDPAlign over the window."""
        genome, read = _u8(genome), _u8(read)
        chain = None
        if rbao_chain is not None:
            chain = np.ascontiguousarray(rbao_chain, dtype=np.uint32).reshape(-1)
            self.lib.ref_set_interval_matches(_pu(chain), len(chain) // 3, 1)
        cs = None if contigs is None else np.ascontiguousarray(contigs, dtype=np.uint32)
        out = np.zeros(16, dtype=np.int32)
        cap = len(genome) + 16
        tseq = np.zeros(cap, dtype=np.uint8)
        blocks = np.zeros(3 * (len(read) + 16), dtype=np.int32)
        sw = None if extend is None else np.array([1, int(extend[0]), int(extend[1])], np.int32)
        if nonguided_band is not None:  # RefineAlignment with useGuidedAlign == false on the candidate afterwards
            r_out, r_pct = np.zeros(12, np.int32), C.c_float(0)
            r_cap = len(read) + 16
            r_blocks, r_gc, r_gi = np.zeros(3 * r_cap, np.int32), np.zeros(r_cap, np.int32), np.zeros(2 * 2 * r_cap, np.int32)
            self.lib.ref_request_nonguided_refine(int(nonguided_band), _p(r_out), C.byref(r_pct), _p(r_blocks), r_cap, _p(r_gc),
                                                  r_cap, _p(r_gi), 2 * r_cap)
        rc = self.lib.ref_align_interval_ex(_p(genome, C.c_ubyte), len(genome), _p(read, C.c_ubyte), len(read),
                                            int(subread[0]), int(subread[1]), int(iv[0]), int(iv[1]), int(iv[2]),
                                            int(iv[3]), int(iv[4]), _pu(cs), 0 if cs is None else len(cs) - 1,
                                            C.c_float(rate), int(extend_ends), int(k), None if sw is None else _p(sw),
                                            _p(out), _p(tseq, C.c_ubyte), cap, _p(blocks), len(read) + 16)
        if chain is not None:
            self.lib.ref_set_interval_matches(None, 0, 0)
        assert rc == 0
        win = dict(zip(WINDOW_FIELDS, (int(v) & 0xffffffff for v in out[:7])))
        sdp = dict(qPos=int(out[7]), tPos=int(out[8]), blocks=blocks[: 3 * int(out[9])].reshape(-1, 3).copy(),
                   score=int(out[10]), nMatch=int(out[11]), nMismatch=int(out[12]), nIns=int(out[13]), nDel=int(out[14]))
        if nonguided_band is not None:
            ngi = C.c_int(0)
            assert self.lib.ref_nonguided_refine_status(C.byref(ngi)) == 0
            self.lib.ref_request_nonguided_refine(0, None, None, None, 0, None, 0, None, 0)
            o = [int(v) for v in r_out]
            refined = dict(tAlignedSeqPos=o[0] & 0xffffffff, tAlignedSeqLength=o[1], qAlignedSeqPos=o[2], qAlignedSeqLength=o[3],
                           qPos=o[4], tPos=o[5], nMatch=o[8], nMismatch=o[9], nIns=o[10], nDel=o[11], pct=r_pct.value,
                           blocks=r_blocks[: 3 * o[6]].reshape(-1, 3).copy(), gapCounts=r_gc[: o[7]].copy(),
                           gapItems=r_gi[: 2 * ngi.value].reshape(-1, 2).copy())
            # (out[0..] were written before the refinement; tseq is the candidate's target afterwards: the trimmed one)
            return win, tseq[: int(out[1])].copy(), sdp, refined
        return win, tseq[: int(out[1])].copy(), sdp

    def post_filter(self, op, score, t_pos, t_aligned_seq_pos, t_index, t_strand, pct, blocks, blk_off, n_blocks,
                    n_candidates=10, max_gap=0, max_score=-200, min_pct=0.0):
        n = len(score)
        kept_idx, kept_score = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
        blocks = np.ascontiguousarray(blocks, np.int32).reshape(-1)
        if blocks.size == 0:
            blocks = np.zeros(3, np.int32)
        pct = np.ascontiguousarray(pct, np.float32)
        k = self.lib.ref_post_filter(op, n, _p(np.ascontiguousarray(score, np.int32)),
                                     _pu(np.ascontiguousarray(t_pos, np.uint32)),
                                     _pu(np.ascontiguousarray(t_aligned_seq_pos, np.uint32)),
                                     _p(np.ascontiguousarray(t_index, np.int32)),
                                     _p(np.ascontiguousarray(t_strand, np.int32)),
                                     pct.ctypes.data_as(C.POINTER(C.c_float)), _p(blocks),
                                     _p(np.ascontiguousarray(blk_off, np.int32)),
                                     _p(np.ascontiguousarray(n_blocks, np.int32)), n_candidates, max_gap, max_score,
                                     C.c_float(min_pct), _p(kept_idx), _p(kept_score))
        return kept_idx[:k].copy(), kept_score[:k].copy()


def load_refmapper():
    return MapperRef() if os.path.exists(REFMAPPER_SO) else None


def random_interval_case(rng, genome, it):
    """A read drawn from `genome` and a plausible anchor interval for it (all the corner cases of
    Blasr.cpp:1243-1298 come up over `it`: both strands, windows clipped at the genome / contig ends)."""
    from blasr_amd import synth
    rl = int(rng.integers(300, 3000))
    pos = int(rng.integers(0, len(genome) - rl))
    strand = int(rng.integers(0, 2))
    tmpl = genome[pos:pos + rl]
    if strand:
        tmpl = synth.revcomp(tmpl)
    read = np.ascontiguousarray(synth.mutate(tmpl, rng, 0.11, 0.04, 0.01))
    L = len(read)
    ss, se = int(rng.integers(0, 20)), L - int(rng.integers(0, 20))
    sS, sE = (ss, se) if strand == 0 else (L - se, L - ss)
    qS, qE = sS + int(rng.integers(0, 100)), sE - int(rng.integers(0, 100))
    g0, g1 = pos + int(rng.integers(0, 120)), pos + rl - int(rng.integers(0, 120))
    if it % 7 == 0:
        g0 = int(rng.integers(0, 50))
    if it % 11 == 0:
        g1 = len(genome) - int(rng.integers(0, 50))
        g0 = min(g0, g1 - 10)
    contigs = None
    if it % 3 == 0:
        cuts = sorted(set([0, len(genome)] + [int(x) for x in rng.integers(1000, len(genome) - 1000, 3)]))
        contigs = np.array(cuts, dtype=np.uint32)
    return read, (ss, se), (g0, g1, qS, qE, strand), contigs


def random_candidates(rng, n):
    """n candidates with the collisions the filters care about: equal scores, equal positions,
    containment on the same / different strand and contig, empty block lists, large gaps."""
    score = rng.choice([-900, -700, -700, -450, -200, -199, -50], size=n).astype(np.int32)
    base = rng.choice([1000, 1000, 5000, 9000], size=n).astype(np.uint32)
    t_aligned_seq_pos = (base + rng.integers(0, 3, size=n)).astype(np.uint32)
    t_pos = rng.integers(0, 4, size=n).astype(np.uint32)
    t_index = rng.choice([0, 0, 0, 1], size=n).astype(np.int32)
    t_strand = rng.choice([0, 0, 1], size=n).astype(np.int32)
    pct = rng.choice([60.0, 75.0, 88.5], size=n).astype(np.float32)
    blocks, blk_off, n_blocks, t_end = [], [], [], []
    for i in range(n):
        nb = int(rng.choice([0, 1, 3, 6]))
        blk_off.append(len(blocks))
        n_blocks.append(nb)
        q = t = 0
        for b in range(nb):
            ln = int(rng.integers(5, 40))
            blocks.append((q, t, ln))
            q += ln + int(rng.choice([0, 1, 2, 30]))
            t += ln + int(rng.choice([0, 1, 2, 30]))
        t_end.append(blocks[-1][1] + blocks[-1][2] if nb else 0)
    blocks = np.array(blocks, dtype=np.int32).reshape(-1, 3)
    return dict(score=score, t_pos=t_pos, t_aligned_seq_pos=t_aligned_seq_pos, t_index=t_index, t_strand=t_strand,
                pct=pct, blocks=blocks, blk_off=np.array(blk_off, np.int32), n_blocks=np.array(n_blocks, np.int32),
                t_end=np.array(t_end, np.uint32))


def affine_kband_align(checker, q, t, k, matrix=None, indel=5):
    """AffineKBandAlign, Global, with the penalties PairwiseLocalAlign derives from params.indel
    (Blasr.cpp:695-703).  `checker` = load_oracle() or load_ref().  Returns score, blocks, gap lists."""
    q, t = _u8(q), _u8(t)
    m = np.ascontiguousarray(SMRT_MATRIX if matrix is None else matrix, dtype=np.int32).reshape(-1)
    cap = len(q) + len(t) + 2 * k + 16
    blocks = np.zeros(3 * cap, dtype=np.int32)
    gap_counts, gap_items = np.zeros(cap, dtype=np.int32), np.zeros(2 * cap, dtype=np.int32)
    score, nb, ngl, ngi = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    f = checker.fn("affine_kband_align")
    f.restype = C.c_int
    rc = f(_p(q, C.c_ubyte), len(q), _p(t, C.c_ubyte), len(t), _p(m), indel + 2, indel - 3, indel + 2, indel - 1, indel,
           int(k), C.byref(score), _p(blocks), cap, C.byref(nb), _p(gap_counts), cap, C.byref(ngl), _p(gap_items), cap,
           C.byref(ngi))
    assert rc == 0
    return dict(score=score.value, blocks=blocks[: 3 * nb.value].reshape(-1, 3).copy(),
                gapCounts=gap_counts[: ngl.value].copy(), gapItems=gap_items[: 2 * ngi.value].reshape(-1, 2).copy())


KBAND_GLOBAL, KBAND_QUERYFIT, KBAND_TARGETFIT, KBAND_FIT = 1, 2, 3, 7  # AlignmentType, AlignmentUtils.h:12-30


def kband_align(checker, q, t, k, align_type=KBAND_GLOBAL, matrix=None, ins=5, dele=5, fn_ins=5, fn_del=5):
    """KBandAlign with a DistanceMatrixScoreFunction (KBandAlign.h:74-403).  ins/dele: boundary penalties,
    fn_ins/fn_del: the score function's.  Returns score, qPos, tPos, blocks, gap lists."""
    q, t = _u8(q), _u8(t)
    m = np.ascontiguousarray(SMRT_MATRIX if matrix is None else matrix, dtype=np.int32).reshape(-1)
    cap = len(q) + len(t) + 2 * k + 16
    blocks = np.zeros(3 * cap, dtype=np.int32)
    gap_counts, gap_items = np.zeros(cap, dtype=np.int32), np.zeros(2 * cap, dtype=np.int32)
    out = np.zeros(3, dtype=np.int32)
    nb, ngl, ngi = C.c_int(0), C.c_int(0), C.c_int(0)
    f = checker.fn("kband_align")
    f.restype = C.c_int
    rc = f(_p(q, C.c_ubyte), len(q), _p(t, C.c_ubyte), len(t), _p(m), int(ins), int(dele), int(fn_ins), int(fn_del),
           int(k), int(align_type), _p(out), _p(blocks), cap, C.byref(nb), _p(gap_counts), cap, C.byref(ngl),
           _p(gap_items), cap, C.byref(ngi))
    assert rc == 0
    return dict(score=int(out[0]), qPos=int(out[1]), tPos=int(out[2]), blocks=blocks[: 3 * nb.value].reshape(-1, 3).copy(),
                gapCounts=gap_counts[: ngl.value].copy(), gapItems=gap_items[: 2 * ngi.value].reshape(-1, 2).copy())


def extend_align(checker, q, query_pos, t, ref_pos, k=10, max_n_drops=10, reverse=False, matrix=None, ins=5, dele=5,
                 min_extend_n_bases=1):
    """ExtendAlignmentForward / ExtendAlignmentReverse (ExtendAlign.h:365-445) with a DistanceMatrixScoreFunction,
    as AlignIntervals calls them under -extend (Blasr.cpp:1826-1838, 1885-1897).  `checker` = load_oracle() or
    load_ref().  Returns score (the return value), qPos, tPos, blocks, gap lists."""
    q, t = _u8(q), _u8(t)
    m = np.ascontiguousarray(SMRT_MATRIX if matrix is None else matrix, dtype=np.int32).reshape(-1)
    cap = len(q) + len(t) + 16
    blocks = np.zeros(3 * cap, dtype=np.int32)
    gap_counts, gap_items = np.zeros(cap, dtype=np.int32), np.zeros(2 * cap, dtype=np.int32)
    out = np.zeros(5, dtype=np.int32)
    nb, ngl, ngi = C.c_int(0), C.c_int(0), C.c_int(0)
    f = checker.fn("extend_align")
    f.restype = C.c_int
    rc = f(_p(q, C.c_ubyte), len(q), int(query_pos), _p(t, C.c_ubyte), len(t), int(ref_pos), _p(m), int(ins), int(dele),
           int(k), int(min_extend_n_bases), int(max_n_drops), int(bool(reverse)), _p(out), _p(blocks), cap, C.byref(nb),
           _p(gap_counts), cap, C.byref(ngl), _p(gap_items), cap, C.byref(ngi))
    assert rc == 0
    return dict(score=int(out[0]), qPos=int(out[1]), tPos=int(out[2]), blocks=blocks[: 3 * nb.value].reshape(-1, 3).copy(),
                gapCounts=gap_counts[: ngl.value].copy(), gapItems=gap_items[: 2 * ngi.value].reshape(-1, 2).copy())


def random_extend_case(rng, it):
    """One ExtendAlignment input: a read end and the genome next to it, related by PacBio-like noise for a stretch
    and unrelated after it (so that the drop-off rule ends the extension), plus the degenerate shapes."""
    from blasr_amd import synth
    ql = int(rng.choice([1, 2, 5, 11, 40, 120, 500, 650]))
    q, t = synth.make_pair(rng, ql, flank=(0, 0))
    if len(q) == 0 or len(t) == 0:
        q, t = np.frombuffer(b"A", np.uint8).copy(), np.frombuffer(b"AC", np.uint8).copy()
    junk = int(rng.choice([0, 0, 30, 200]))
    if junk:
        q = np.concatenate([q, np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, junk)]])
        t = np.concatenate([t, np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, junk)]])
    if it % 5 == 0:
        t = t[: max(1, len(t) - int(rng.integers(0, 60)))]
    if it % 7 == 0:
        q = q[: max(1, len(q) - int(rng.integers(0, 60)))]
    if it % 11 == 0:
        q = q.copy()
        q[int(rng.integers(0, len(q)))] = ord("N")
    if it % 13 == 0:
        t = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, len(t))]  # nothing to extend into
    reverse = bool(it % 2)
    if reverse:
        q, t = q[::-1].copy(), t[::-1].copy()
        qp, tp = len(q) - int(rng.choice([0, 0, 1, 3])), len(t) - int(rng.choice([0, 0, 1, 2]))
        qp, tp = max(qp, 0), max(tp, 0)
    else:
        qp, tp = int(rng.choice([0, 0, 1, 4])), int(rng.choice([0, 0, 2]))
        qp, tp = min(qp, len(q)), min(tp, len(t))
    k = int(rng.choice([0, 1, 3, 10, 10, 15, 16, 31]))
    drops = int(rng.choice([0, 1, 2, 10, 10, 25]))
    return q, qp, t, tp, k, drops, reverse


# --- next-2: the epilogue pieces, oracle (plain C) and reference (ref_mapper_harness.cpp) --------------------
class EpilogueChecker:
    """bo_* of the oracle (kind="oracle") or ref_* of the reference harness (kind="ref"), same Python surface."""

    def __init__(self, kind):
        self.kind = kind
        if kind == "oracle":
            if not os.path.exists(ORACLE_SO):
                build_oracle()
            self.lib = C.CDLL(ORACLE_SO)
        else:
            self.lib = C.CDLL(REFMAPPER_SO)

    def ref_contig_location(self, pos, length, strand, seq_start_pos, genome_length):
        sp = np.ascontiguousarray(seq_start_pos, dtype=np.uint32)
        out = np.zeros(2, dtype=np.uint32)
        fn = self.lib.bo_ref_contig_location if self.kind == "oracle" else self.lib.ref_contig_location
        fn(C.c_uint(pos), C.c_uint(length), int(strand), _pu(sp), len(sp), C.c_uint(genome_length), _pu(out))
        return int(out[0]), int(out[1])

    def cigar(self, q, t, q_pos, t_pos, ops=None, blocks=None, clipping=3, t_strand=0, read_length=0, lqp=0, lqs=0,
              subread_start=0, subread_end=0, q_align_start=0, q_align_end=0):
        q, t = _u8(q), _u8(t)
        o = None if ops is None else np.ascontiguousarray(ops, dtype=np.uint32)
        b = None if blocks is None else np.ascontiguousarray(blocks, dtype=np.int32).reshape(-1)
        cap = 16 * (len(q) + len(t) + 8)
        out = C.create_string_buffer(cap)
        clips = np.zeros(5, dtype=np.uint32)
        args = [_p(q, C.c_ubyte), _p(t, C.c_ubyte), int(q_pos), int(t_pos), _pu(o), C.c_long(0 if o is None else len(o)),
                None if b is None else _p(b), C.c_long(0 if b is None else len(b) // 3), int(clipping), int(t_strand),
                C.c_uint(read_length), C.c_uint(lqp), C.c_uint(lqs), C.c_uint(subread_start), C.c_uint(subread_end),
                C.c_uint(q_align_start)]
        if self.kind == "oracle":
            n = self.lib.bo_cigar(*args, C.c_uint(q_align_end), out, cap, _pu(clips))
        else:
            n = self.lib.ref_cigar(*args, out, cap, _pu(clips))
            assert int(clips[4]) == q_align_end, (int(clips[4]), q_align_end)  # the harness derives QAlignEnd() itself
        assert n >= 0
        return out.value.decode(), tuple(int(v) for v in clips[:4])

    def m4_line(self, q_name, t_name, score, pct, q_strand, q_aligned_seq_pos, q_pos, q_length, t_strand,
                t_aligned_seq_pos, t_pos, t_length, map_qv, blocks):
        """blocks: (n,3) relative to q_pos / t_pos.  Returns (line, (QAlignStart, QAlignEnd, TAlignStart, tEnd))."""
        b = np.ascontiguousarray(blocks, dtype=np.int32).reshape(-1, 3)
        q_start, t_start = q_pos + q_aligned_seq_pos, t_pos + t_aligned_seq_pos
        q_end = int(b[-1, 0] + b[-1, 2]) + q_start
        t_end = t_aligned_seq_pos + t_pos + int(b[-1, 1] + b[-1, 2])
        out = C.create_string_buffer(len(q_name) + len(t_name) + 256)
        if self.kind == "oracle":
            n = self.lib.bo_m4_line(q_name.encode(), t_name.encode(), int(score), C.c_float(pct), int(q_strand),
                                    C.c_uint(q_start), C.c_uint(q_end), C.c_uint(q_length), int(t_strand),
                                    C.c_uint(t_start), C.c_uint(t_end), C.c_uint(t_length), int(map_qv), out, len(out))
        else:
            derived = np.zeros(3, dtype=np.uint32)
            n = self.lib.ref_m4_line(q_name.encode(), t_name.encode(), int(score), C.c_float(pct), int(q_strand),
                                     C.c_uint(q_aligned_seq_pos), C.c_uint(q_pos), C.c_uint(q_length), int(t_strand),
                                     C.c_uint(t_aligned_seq_pos), C.c_uint(t_pos), C.c_uint(t_length), int(map_qv),
                                     _p(b.reshape(-1)), C.c_long(len(b)), out, len(out), _pu(derived))
            assert (int(derived[0]), int(derived[1]), int(derived[2])) == (q_start, q_end, t_start)
        assert n >= 0
        return out.value.decode(), (q_start, q_end, t_start, t_end)


def random_epilogue_cases(rng, n):
    """Alignments for the printers: op lists as the refine stage emits them (block first and last), sequences with
    mixed case and N so that the = / X split is exercised, every clipping mode and both strands."""
    cases = []
    letters = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)
    for it in range(n):
        n_runs = int(rng.integers(1, 12)) * 2 - 1
        ops, qlen, tlen = [], 0, 0
        for r in range(n_runs):
            if r % 2 == 0:
                ln = int(rng.integers(1, 40))
                ops.append((ln << 2) | 0)
                qlen += ln
                tlen += ln
            else:
                for g in range(int(rng.integers(1, 4))):  # several gaps in one slot, alternating kinds
                    ty = int(rng.integers(1, 3)) if g == 0 else 3 - (ops[-1] & 3)
                    ln = int(rng.integers(1, 12))
                    ops.append((ln << 2) | ty)
                    if ty == 1:
                        qlen += ln
                    else:
                        tlen += ln
        q_pos, t_pos = int(rng.integers(0, 30)), int(rng.integers(0, 30))
        t = letters[rng.integers(0, len(letters), t_pos + tlen + 5)].copy()
        q = letters[rng.integers(0, len(letters), q_pos + qlen + 5)].copy()
        # make most aligned columns agree
        qi, ti = q_pos, t_pos
        for w in ops:
            ln, ty = w >> 2, w & 3
            if ty == 0:
                keep = rng.random(ln) < 0.85
                q[qi:qi + ln] = np.where(keep, t[ti:ti + ln], q[qi:qi + ln])
                qi += ln
                ti += ln
            elif ty == 1:
                qi += ln
            else:
                ti += ln
        q_aligned_seq_pos = int(rng.integers(0, 50))
        q_align_start = q_pos + q_aligned_seq_pos
        q_align_end = q_align_start + qlen
        lqp, lqs = int(rng.integers(0, q_align_start + 1)), int(rng.integers(0, 20))
        read_length = q_align_end + lqs + int(rng.integers(0, 40))
        sub_start = int(rng.integers(0, q_align_start + 1))
        sub_end = int(rng.integers(q_align_end, read_length + 1))
        # blocks of the same alignment (for the no-gap-list form of the printer)
        blocks, qq, tt = [], 0, 0
        for w in ops:
            ln, ty = w >> 2, w & 3
            if ty == 0:
                blocks.append((qq, tt, ln))
                qq += ln
                tt += ln
            elif ty == 1:
                qq += ln
            else:
                tt += ln
        cases.append(dict(q=q, t=t, q_pos=q_pos, t_pos=t_pos, ops=np.array(ops, dtype=np.uint32),
                          blocks=np.array(blocks, dtype=np.int32), clipping=it % 4, t_strand=(it // 4) % 2,
                          read_length=read_length, lqp=lqp, lqs=lqs, subread_start=sub_start, subread_end=sub_end,
                          q_align_start=q_align_start, q_align_end=q_align_end))
    return cases


def _store_map_qvs(checker, alignments, substitution_qv=None, ins=5, dele=5, affine_open=50, affine_extend=0,
                   affine_align=1, min_fraction=0.75):
    """StoreMapQVs through the oracle (bo_store_map_qvs) or the reference harness (ref_store_map_qvs)."""
    n = len(alignments)
    qs = [_u8(a["q"]) for a in alignments]
    ts = [_u8(a["t"]) for a in alignments]
    qptr = (C.POINTER(C.c_ubyte) * max(n, 1))(*[_p(x, C.c_ubyte) for x in qs])
    tptr = (C.POINTER(C.c_ubyte) * max(n, 1))(*[_p(x, C.c_ubyte) for x in ts])
    ops = np.concatenate([np.asarray(a["ops"], dtype=np.uint32) for a in alignments]) if n else np.zeros(1, np.uint32)
    op_off = np.zeros(n + 1, dtype=np.int64)
    for i, a in enumerate(alignments):
        op_off[i + 1] = op_off[i] + len(a["ops"])
    ia = lambda k: np.array([a[k] for a in alignments] or [0], dtype=np.int32)  # noqa: E731
    ua = lambda k: np.array([a[k] for a in alignments] or [0], dtype=np.uint32)  # noqa: E731
    q_pos, t_pos, q_strand, n_match = ia("q_pos"), ia("t_pos"), ia("q_strand"), ia("n_match")
    q_start, q_end, q_length = ua("q_align_start"), ua("q_align_end"), ua("q_length")
    qv = None if substitution_qv is None else _u8(substitution_qv)
    map_qv, prob = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float32)
    if checker.kind == "oracle":
        from blasr_amd.hip import SMRT_LOG_PROB_MATRIX
        m = np.array(SMRT_LOG_PROB_MATRIX, dtype=np.int32)
        checker.lib.bo_store_map_qvs(n, qptr, tptr, _p(q_pos), _p(t_pos), _pu(ops), _p(op_off, C.c_long), _pu(q_start),
                                     _pu(q_end), _p(q_strand), _pu(q_length), _p(n_match),
                                     None if qv is None else _p(qv, C.c_ubyte), _p(m), ins, dele, affine_open,
                                     affine_extend, affine_align, C.c_float(min_fraction), _p(map_qv), _p(prob, C.c_float))
    else:
        read_length = int(max([a["q_length"] for a in alignments] or [0]))
        checker.lib.ref_store_map_qvs(n, qptr, tptr, _p(q_pos), _p(t_pos), _pu(ops), _p(op_off, C.c_long), _pu(q_start),
                                      _p(q_strand), _pu(q_length), _p(n_match), C.c_uint(read_length),
                                      None if qv is None else _p(qv, C.c_ubyte), ins, dele, affine_open, affine_extend,
                                      affine_align, C.c_float(min_fraction), _p(map_qv), _p(prob, C.c_float))
    return [int(v) for v in map_qv[:n]], [float(v) for v in prob[:n]]


EpilogueChecker.store_map_qvs = _store_map_qvs


def random_mapqv_case(rng, it):
    """The alignments of one read for StoreMapQVs: one to six, mostly overlapping on the read (the competing
    candidates of a repeat), sometimes disjoint (a chimeric read), both strands, with and without substitution QVs."""
    read_length = int(rng.integers(400, 3000))
    n = int(rng.integers(1, 7))
    letters = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)
    alns = []
    base_start = int(rng.integers(0, read_length // 4))
    base_end = int(rng.integers(3 * read_length // 4, read_length + 1))
    for a in range(n):
        if rng.random() < 0.75:  # a competitor over (nearly) the same part of the read
            s = max(0, base_start + int(rng.integers(-40, 60)))
            e = min(read_length, base_end + int(rng.integers(-60, 40)))
        else:  # somewhere else on the read
            s = int(rng.integers(0, read_length - 50))
            e = int(rng.integers(s + 30, read_length + 1))
        span = e - s
        ops, qlen, tlen = [], 0, 0
        while qlen < span:
            ln = min(int(rng.integers(1, 60)), span - qlen)
            ops.append((ln << 2) | 0)
            qlen += ln
            tlen += ln
            if qlen < span and rng.random() < 0.8:
                ty = int(rng.integers(1, 3))
                gl = int(rng.integers(1, 60 if rng.random() < 0.1 else 6))
                if ty == 1:
                    gl = min(gl, span - qlen - 1)
                    if gl <= 0:
                        continue
                    qlen += gl
                else:
                    tlen += gl
                ops.append((gl << 2) | ty)
        if (ops[-1] & 3) != 0:  # the list ends with a block
            ops.append((1 << 2) | 0)
            qlen += 1
            tlen += 1
            e = s + qlen
            if e > read_length:
                read_length = e
        q_pos, t_pos = int(rng.integers(0, 20)), int(rng.integers(0, 20))
        t = letters[rng.integers(0, 9, t_pos + tlen + 4)].copy()
        q = letters[rng.integers(0, 9, q_pos + qlen + 4)].copy()
        ident = 0.70 + 0.29 * rng.random()
        qi, ti, n_match = q_pos, t_pos, 0
        for w in ops:
            ln, ty = w >> 2, w & 3
            if ty == 0:
                keep = rng.random(ln) < ident
                q[qi:qi + ln] = np.where(keep, t[ti:ti + ln], q[qi:qi + ln])
                n_match += int(keep.sum())
                qi += ln
                ti += ln
            elif ty == 1:
                qi += ln
            else:
                ti += ln
        q_strand = int(rng.integers(0, 2))
        alns.append(dict(q=q, t=t, ops=np.array(ops, dtype=np.uint32), q_pos=q_pos, t_pos=t_pos, q_align_start=s,
                         q_align_end=s + qlen, q_strand=q_strand, q_length=0, n_match=n_match))
    # near-ties: copies of the first alignment with a few bases changed, so that the Phred branch sees scores a
    # fraction of a unit apart (a repeat with a handful of differences)
    for c in range(int(rng.integers(0, 4))):
        a = dict(alns[0])
        q = a["q"].copy()
        span = a["q_align_end"] - a["q_align_start"]
        for k in range(int(rng.integers(0, 7))):
            q[a["q_pos"] + int(rng.integers(0, span))] = letters[int(rng.integers(0, 9))]
        a["q"] = q
        a["n_match"] = max(0, a["n_match"] - int(rng.integers(0, 7)))
        a["q_strand"] = int(rng.integers(0, 2)) if rng.random() < 0.2 else a["q_strand"]
        alns.insert(int(rng.integers(0, len(alns) + 1)), a)
    for a in alns:
        a["q_length"] = read_length
    qv = rng.integers(0, 60, read_length).astype(np.uint8) if it % 3 == 0 else None
    return alns, qv


SAM_INT_FIELDS = ("q_pos", "t_pos", "clipping", "q_strand", "t_strand", "is_primary", "read_length", "low_quality_prefix",
                  "low_quality_suffix", "subread_start", "subread_end", "q_aligned_seq_pos", "q_aligned_seq_length",
                  "t_aligned_seq_pos", "t_length", "q_length", "map_qv", "score", "edit_dist", "hole_number", "qs", "qe",
                  "num_passes", "subread_context")


def _sam_line(checker, c, cigar=None, clips=None, derived=None):
    """SAMOutput::PrintAlignment through the reference harness (returns (line, derived)) or the oracle (needs the
    CIGAR, the clips and the derived positions)."""
    iv = np.array([c[k] for k in SAM_INT_FIELDS] + ([0] * 4 if derived is None else list(derived)), dtype=np.int32)
    fv = np.array([c["read_quality"]] + list(c["snr"]), dtype=np.float32)
    seq = _u8(c["read_seq"])
    qual = None if c["read_qual"] is None else _u8(c["read_qual"])
    cap = 2 * len(seq) + 4096
    out = C.create_string_buffer(cap)
    if checker.kind == "ref":
        b = np.ascontiguousarray(c["blocks"], dtype=np.int32).reshape(-1)
        der = np.zeros(4, dtype=np.uint32)
        n = checker.lib.ref_sam_line(c["q_name"].encode(), c["t_name"].encode(), c["read_group_id"].encode(),
                                     _p(seq, C.c_ubyte), None if qual is None else _p(qual, C.c_ubyte), None, C.c_long(0),
                                     _p(b), C.c_long(len(b) // 3), _p(iv), _p(fv, C.c_float), out, cap, _pu(der))
        assert n >= 0
        return out.value.decode(), tuple(int(v) for v in der)
    k4 = np.array(clips, dtype=np.uint32)
    n = checker.lib.bo_sam_line(c["q_name"].encode(), c["t_name"].encode(), c["read_group_id"].encode(),
                                _p(seq, C.c_ubyte), None if qual is None else _p(qual, C.c_ubyte), cigar.encode(), _pu(k4),
                                _p(iv), _p(fv, C.c_float), out, cap)
    assert n >= 0
    return out.value.decode()


EpilogueChecker.sam_line = _sam_line


def random_sam_case(rng, it):
    """One alignment and its read for the SAM printer.  The asserts of PrintAlignment (:508, :512) hold: the hard
    clips plus the printed sequence make up the read."""
    letters = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)
    nb = int(rng.integers(1, 6))
    blocks, q, t = [], 0, 0
    for b in range(nb):
        ln = int(rng.integers(1, 60))
        blocks.append((q, t, ln))
        q += ln + int(rng.integers(0, 5))
        t += ln + int(rng.integers(0, 5))
    q_end_rel = blocks[-1][0] + blocks[-1][2]
    q_pos, t_pos = int(rng.integers(0, 10)), int(rng.integers(0, 10))
    q_aligned_seq_pos = int(rng.integers(0, 60))
    q_align_start = q_pos + q_aligned_seq_pos
    q_align_end = q_align_start + q_end_rel
    clipping = it % 4
    lqp = int(rng.integers(0, q_align_start + 1))
    lqs = int(rng.integers(0, 15))
    read_length = q_align_end + lqs + int(rng.integers(0, 30))
    sub_start = int(rng.integers(lqp, q_align_start + 1))         # subread clipping: lqp <= subreadStart
    sub_end = int(rng.integers(q_align_end, read_length - lqs + 1))  # and lqs <= length - subreadEnd
    t_aligned_seq_pos = int(rng.integers(0, 5000))
    t_end_rel = blocks[-1][1] + blocks[-1][2]
    t_length = t_aligned_seq_pos + t_pos + t_end_rel + int(rng.integers(0, 3000))
    return dict(q_name="m%d/%d/0_%d" % (it, it * 3, read_length), t_name="ctg%d" % (it % 4), read_group_id="rg%x" % (it * 2654435761 % 2**32),
                read_seq=letters[rng.integers(0, 9, read_length)].copy(),
                read_qual=None if it % 3 == 0 else rng.integers(0, 60, read_length).astype(np.uint8),
                blocks=np.array(blocks, dtype=np.int32), q_pos=q_pos, t_pos=t_pos, clipping=clipping,
                q_strand=int(rng.integers(0, 2)), t_strand=int(rng.integers(0, 2)), is_primary=int(rng.integers(0, 2)),
                read_length=read_length, low_quality_prefix=lqp, low_quality_suffix=lqs, subread_start=sub_start,
                subread_end=sub_end, q_aligned_seq_pos=q_aligned_seq_pos, q_aligned_seq_length=int(rng.integers(q_end_rel, q_end_rel + 50)),
                t_aligned_seq_pos=t_aligned_seq_pos, t_length=t_length, q_length=read_length,
                map_qv=int(rng.integers(0, 255)), score=int(rng.integers(-50000, 10)), edit_dist=int(rng.integers(0, 900)),
                hole_number=int(rng.integers(-1, 160000)), qs=int(rng.integers(0, 500)), qe=int(rng.integers(500, 40000)),
                num_passes=int(rng.integers(0, 9)), subread_context=int(rng.integers(0, 4)),
                read_quality=float(np.float32([0.0, 0.8, 0.85331, 1.0, 0.123456][it % 5])),
                snr=[float(np.float32(v)) for v in rng.random(4) * [12, 9, 7, 100]])


def _order_gaps_by_type(checker, ops):
    o = np.ascontiguousarray(ops, dtype=np.uint32)
    out = np.zeros(len(o) + 1, dtype=np.uint32)
    fn = checker.lib.bo_order_gaps_by_type if checker.kind == "oracle" else checker.lib.ref_order_gaps_by_type
    fn.restype = C.c_long
    n = fn(_pu(o), C.c_long(len(o)), _pu(out))
    return out[:n].copy()


EpilogueChecker.order_gaps_by_type = _order_gaps_by_type


# ---------------------------------------------------------------------------------------------
# MapRead from the intervals on, composed from the oracle's pieces in the reference's order
# (Blasr.cpp:2530-2644): the checker for blasr_hip_map_reads_* (blasr_amd/csrc/mapread_api.hip).
# ---------------------------------------------------------------------------------------------
_RC = np.full(256, 127, dtype=np.uint8)  # ReverseComplementNuc (NucConversion.h:337-370)
for _a, _b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _RC[_a] = _b


def oracle_extend_candidate(oracle, genome, contigs, read, w, q, tseq, s, sc, band=10, drops=10, max_len=500):
    """The -extend step of AlignIntervals (Blasr.cpp:1749-1940) on one candidate, composed from the oracle's
    ExtendAlignment and plain index arithmetic (DNALength = 32-bit unsigned, kept with & M): the candidate cut to its SDP
    extent, up to 500 read / genome bases after it (Forward) and before it (Reverse: called one base early, and on the
    reverse strand against MakeRC of a FORWARD position, :1876, :1885), AppendAlignment without gap lists.  Returns the
    new (window fields, query string, target string, sdp record)."""
    M = 0xffffffff
    blocks = np.asarray(s["blocks"], np.int64).reshape(-1, 3)
    if len(blocks) == 0:
        return w, q, tseq, s
    L = len(genome)
    q_end, t_end = int(blocks[-1, 0] + blocks[-1, 2]), int(blocks[-1, 1] + blocks[-1, 2])
    q_asp, t_asp = (w["qAlignedSeqPos"] + s["qPos"]) & M, (w["tAlignedSeqPos"] + s["tPos"]) & M
    tw = np.asarray(tseq)[s["tPos"]: s["tPos"] + t_end]
    cs, ce = (0, L) if contigs is None else (int(contigs[w["tIndex"]]), int(contigs[w["tIndex"] + 1]) - 1)
    rev_strand = w["tStrand"] == 1
    c_start, c_end = ((L - ce) & M, (L - cs) & M) if rev_strand else (cs, ce)
    last_q, last_t = (q_end + q_asp) & M, (t_end + t_asp) & M
    rsl, gsl = min((len(read) - last_q) & M, max_len), min((c_end - last_t) & M, max_len)
    rpl, gpl = min(q_asp, max_len), min((t_asp - c_start) & M, max_len)

    def strand_piece(pos, n):  # CopyAsRC: n bases of the strand of the alignment from `pos` on it
        return genome[pos: pos + n] if not rev_strand else _RC[genome[L - pos - n: L - pos][::-1]]

    m = sc.matrix.reshape(-1)
    new_blocks = [tuple(int(v) for v in b) for b in blocks]
    q_len = q_end
    fwd_t = np.zeros(0, np.uint8)
    if rsl > 0 and gsl > 0:
        g_suffix = strand_piece(last_t, gsl)
        f = extend_align(oracle, read[last_q: last_q + rsl], 0, g_suffix, 0, band, drops, False, matrix=m, ins=sc.ins, dele=sc.dele)
        if f["score"] < 0:  # :1840-1855
            fb = f["blocks"]
            fq, ft = int(fb[-1, 0] + fb[-1, 2]), int(fb[-1, 1] + fb[-1, 2])
            new_blocks += [(int(b[0]) + q_end, int(b[1]) + t_end, int(b[2])) for b in fb]
            q_len += fq
            fwd_t = g_suffix[:ft]
    q_pos = t_pos = 0
    pre_t = np.zeros(0, np.uint8)
    if rpl > 0 and gpl > 0:
        r_prefix = read[q_asp - rpl: q_asp]
        if not rev_strand:
            g_prefix = genome[t_asp - gpl: t_asp]
        else:  # MakeRC of the forward stretch at the reverse-strand position (:1876)
            g_prefix = _RC[genome[t_asp - gpl: t_asp][::-1]]
        v = extend_align(oracle, r_prefix, rpl - 1, g_prefix, gpl - 1, band, drops, True, matrix=m, ins=sc.ins, dele=sc.dele)
        if v["score"] < 0:  # :1899-1935
            q_pos, t_pos = v["qPos"], v["tPos"]
            qs, ts = rpl - q_pos, gpl - t_pos
            new_blocks = [tuple(int(x) for x in b) for b in v["blocks"]] + [(b[0] + qs, b[1] + ts, b[2]) for b in new_blocks]
            q_asp, t_asp = (q_asp - rpl) & M, (t_asp - gpl) & M
            q_len += rpl
            pre_t = g_prefix
    new_t = np.concatenate([pre_t, tw, fwd_t]).astype(np.uint8)
    new_q = np.asarray(read[q_asp: q_asp + q_len], np.uint8)
    w2 = dict(w)
    w2.update(qAlignedSeqPos=q_asp, qAlignedSeqLength=len(new_q), tAlignedSeqPos=t_asp, tAlignedSeqLength=len(new_t))
    s2 = dict(s)
    s2.update(qPos=int(q_pos), tPos=int(t_pos), blocks=np.array(new_blocks, np.int32).reshape(-1, 3))
    return w2, new_q, new_t, s2


def oracle_map_reads(oracle, mo, genome, contigs, pool, reads, intervals, band=16, n_candidates=10, max_gap=0,
                     max_score=-200, min_pct=0.0, sp=None, sc=None, extend=False, extend_band=10, extend_drops=10,
                     guided=True, indel=5, with_ops=True):
    """reads / intervals: the product's READ_DTYPE / INTERVAL_DTYPE arrays.  Returns, per read, the alignments
    MapRead keeps (best first) as dicts: interval, the hit record's fields, q / t (aligned sequences), ops."""
    sp, sc = sp or SdpParams(), sc or ScoreParams()
    by_read = [[] for _ in range(len(reads))]
    for k, iv in enumerate(intervals):
        by_read[int(iv["read"])].append(k)
    out = []
    for r, ks in enumerate(by_read):
        rd = reads[r]
        read = pool[int(rd["off"]): int(rd["off"]) + int(rd["length"])]
        c = []
        for k in ks:
            iv = intervals[k]
            w, tseq = mo.window(genome, len(read), (int(rd["subread_start"]), int(rd["subread_end"])),
                                (int(iv["start"]), int(iv["end"]), int(iv["q_start"]), int(iv["q_end"]), int(iv["strand"])),
                                contigs)
            q = read[w["qAlignedSeqPos"]: w["qAlignedSeqPos"] + w["qAlignedSeqLength"]]
            s = oracle.sdp_align(q, tseq, sc, sp)
            if extend:
                w, q, tseq, s = oracle_extend_candidate(oracle, np.asarray(genome), contigs, read, w, q, tseq, s, sc,
                                                        extend_band, extend_drops)
            st = oracle.compute_alignment_stats(q, tseq, s["qPos"], s["tPos"], s["blocks"], sc=sc)
            c.append(dict(interval=k, win=w, q=q, t=tseq, sdp=s, score=st["score"], pct=np.float32(st["pct"]),
                          t_pos=s["tPos"], t_aligned_seq_pos=w["tAlignedSeqPos"], t_end=0, blocks=s["blocks"],
                          refined=None))

        def filters(order, low_quality_only):
            score = np.array([x["score"] for x in c], np.int32)
            blocks = [np.asarray(x["blocks"], np.int32).reshape(-1, 3) for x in c]
            n_blocks = np.array([len(b) for b in blocks], np.int32)
            blk_off = np.concatenate([[0], np.cumsum(n_blocks)])[:-1].astype(np.int32)
            flat = np.concatenate(blocks).reshape(-1) if n_blocks.sum() else np.zeros(0, np.int32)
            t_pos = np.array([x["t_pos"] for x in c], np.uint32)
            tasp = np.array([x["t_aligned_seq_pos"] for x in c], np.uint32)
            order = mo.sort_by_score(score, t_pos, tasp, order)
            order, score = mo.remove_low_quality(order, score, flat, blk_off, n_blocks, n_candidates, max_gap, max_score)
            for j, x in enumerate(c):
                x["score"] = int(score[j])
            if not low_quality_only:
                order = mo.remove_overlapping(order, score, t_pos, tasp, np.array([x["t_end"] for x in c], np.uint32),
                                              np.array([x["win"]["tIndex"] for x in c], np.int32),
                                              np.array([x["win"]["tStrand"] for x in c], np.int32),
                                              np.array([x["pct"] for x in c], np.float32), min_pct)
            return order

        res = []
        out.append(res)
        if not c:
            continue
        order = filters(np.arange(len(c), dtype=np.int32), True)                       # :2547, :2594-2596
        for j in order:                                                                # RefineAlignments :2164-2181
            x = c[int(j)]
            bl = x["sdp"]["blocks"]
            if len(bl) == 0:
                continue
            qs = x["q"][x["sdp"]["qPos"]: x["sdp"]["qPos"] + bl[-1, 0] + bl[-1, 2]]
            ts = x["t"][x["sdp"]["tPos"]: x["sdp"]["tPos"] + bl[-1, 1] + bl[-1, 2]]
            if guided:
                rr = oracle.affine_guided_align(qs, ts, bl, sc, band)
            else:  # -noUseGuidedAlign: RefineAlignment's k-band branch (Blasr.cpp:917-1046); `band` = params.bandSize
                rr = oracle.refine_nonguided(x["q"], x["t"], x["sdp"], sc, indel, band)
                qs, ts = rr["qs"], rr["ts"]
            x.update(refined=rr, qs=qs, ts=ts, score=rr["score"], pct=np.float32(rr["pct"]), t_pos=rr["tPos"],
                     t_aligned_seq_pos=x["win"]["tAlignedSeqPos"] + x["sdp"]["tPos"], blocks=rr["blocks"],
                     t_end=(int(rr["blocks"][-1, 1] + rr["blocks"][-1, 2]) if len(rr["blocks"]) else 0))
        order = filters(order, False)                                                  # :2178-2180, :2615-2616
        for j in order:
            x = c[int(j)]
            rr = x["refined"]
            if rr is None:
                continue
            res.append(dict(interval=x["interval"], t_strand=x["win"]["tStrand"], t_index=x["win"]["tIndex"],
                            t_aligned_seq_pos=x["t_aligned_seq_pos"] & 0xffffffff, t_aligned_seq_length=len(x["ts"]),
                            t_length=x["win"]["tLength"],
                            q_aligned_seq_pos=x["win"]["qAlignedSeqPos"] + x["sdp"]["qPos"],
                            q_aligned_seq_length=len(x["qs"]), q_length=len(read), q_pos=rr["qPos"], t_pos=rr["tPos"],
                            score=x["score"],  # the refined score, or maxScore where the maxGap rule flagged it (:2041)
                            n_match=rr["nMatch"], n_mismatch=rr["nMismatch"], n_ins=rr["nIns"],
                            n_del=rr["nDel"], n_cells=rr["nCells"], pct_similarity=np.float32(rr["pct"]),
                            q=x["qs"], t=x["ts"],
                            ops=[(l << 2) | ty for ty, l in (ops_from_alignment(rr) if guided else ops_with_leading_gaps(rr))]
                            if with_ops else None))  # (a Python loop per op: bench.py's many-threaded CPU leg leaves it out)
    return out


HIT_FIELDS = ("interval", "t_strand", "t_index", "t_aligned_seq_pos", "t_aligned_seq_length", "t_length",
              "q_aligned_seq_pos", "q_aligned_seq_length", "q_length", "q_pos", "t_pos", "score", "n_match",
              "n_mismatch", "n_ins", "n_del", "n_cells", "pct_similarity")


def repeat_genome_case(rng, n_reads, genome_len=300000, read_len=(800, 3000), copies=(2, 5), divergence=(0.01, 0.06)):
    """A genome with planted repeat families and reads drawn half from the repeats: every read gets its true
    locus and the loci of the other copies of its family as candidate intervals (what anchoring reports on a
    repeat-bearing genome: 1..copies competing candidates), on either strand.  Returns genome, pool, reads,
    intervals (product dtypes)."""
    import blasr_amd
    from blasr_amd import synth
    genome = synth.random_genome(genome_len, int(rng.integers(1, 1 << 30)))
    fams = []
    for f in range(4):
        unit = int(rng.integers(3000, 6000))
        src = int(rng.integers(0, genome_len - unit))
        k = int(rng.integers(copies[0], copies[1] + 1))
        locs = [(src, 0)]
        for _ in range(k - 1):
            d = float(rng.uniform(*divergence))
            piece = np.ascontiguousarray(synth.mutate(genome[src:src + unit], rng, d / 3, d / 3, d / 3))
            strand = int(rng.integers(0, 2))
            if strand:
                piece = synth.revcomp(piece)
            at = int(rng.integers(0, genome_len - len(piece)))
            genome[at:at + len(piece)] = piece
            locs.append((at, strand))
        fams.append((unit, locs))
    reads = np.zeros(n_reads, dtype=blasr_amd.READ_DTYPE)
    ivs, pool, off = [], [], 0
    for r in range(n_reads):
        rl = int(rng.integers(*read_len))
        strand = int(rng.integers(0, 2))
        if r % 2 == 0:
            unit, locs = fams[(r // 2) % len(fams)]
            rl = min(rl, unit - 200)
            inner = int(rng.integers(0, unit - rl))
            pos = locs[0][0] + inner
            others = [(a + inner, s) for a, s in locs[1:]]
        else:
            pos, others = int(rng.integers(0, genome_len - rl)), []
        tmpl = genome[pos:pos + rl]
        if strand:
            tmpl = synth.revcomp(tmpl)
        read = np.ascontiguousarray(synth.mutate(tmpl, rng, 0.11, 0.04, 0.01))
        L = len(read)
        reads[r] = (off, L, 0, L, 0)
        pool.append(read)
        off += L
        cands = [(pos, strand)] + [(min(max(a, 0), genome_len - rl), strand ^ s) for a, s in others]
        order = rng.permutation(len(cands))
        for ci in order:
            a, s = cands[int(ci)]
            ua0, ua1 = int(rng.integers(0, 60)), int(rng.integers(0, 60))
            g0 = a + int(rng.integers(0, 60))
            g1 = a + rl - int(rng.integers(0, 60))
            ivs.append((g0, g1, ua0, L - ua1, r, s))
    return genome, np.concatenate(pool), reads, np.array(ivs, dtype=blasr_amd.INTERVAL_DTYPE)


# ---------------------------------------------------------------------------------------------
# Raw device buffers for the *_device entry points, through the HIP runtime the library itself is linked
# against (torch bundles another copy of the runtime, which cannot take the device after this one has).
# ---------------------------------------------------------------------------------------------
class DeviceMemory:
    def __init__(self):
        self.rt = C.CDLL("libamdhip64.so.7")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.held = []

    def zeros(self, nbytes):
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), max(int(nbytes), 16)) == 0
        assert self.rt.hipMemset(p, 0, max(int(nbytes), 16)) == 0
        self.held.append(p)
        return p.value

    def upload(self, array):
        a = np.ascontiguousarray(array)
        p = self.zeros(a.nbytes)
        if a.nbytes:
            assert self.rt.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0  # hipMemcpyHostToDevice
        return p

    def download(self, ptr, dtype, count):
        out = np.zeros(count, dtype=dtype)
        if out.nbytes:
            assert self.rt.hipMemcpy(out.ctypes.data, ptr, out.nbytes, 2) == 0  # hipMemcpyDeviceToHost
        return out

    def free(self):
        for p in self.held:
            self.rt.hipFree(p)
        self.held = []


# ---------------------------------------------------------------------------------------------
# next-1, device half: anchoring in the suffix array (MapReadToGenome)
# ---------------------------------------------------------------------------------------------
SAWRITER = os.path.join(ROOT, "oracle", "_ref", "sawritermc")

# (minMatchLength, expand, useLookupTable, maxAnchorsPerPosition, advanceExactMatches, maxLCPLength,
#  stopMappingOnceUnique): the defaults of Blasr.cpp, then every switch away from them
ANCHOR_PARAM_SETS = [
    (12, 0, 1, 1000, 0, 0, 1), (12, 1, 1, 1000, 0, 0, 1), (14, 2, 1, 1000, 0, 0, 0), (10, 0, 1, 3, 0, 0, 1),
    (12, 1, 1, 1000, 2, 0, 1), (12, 0, 1, 1000, 0, 20, 0), (8, 3, 1, 50, 0, 15, 1), (12, 0, 0, 1000, 0, 0, 1),
    (12, 1, 1, 1000, 5, 0, 0), (9, 5, 1, 1000, 0, 0, 0), (13, 7, 1, 8, 30, 0, 1),
]


def anchor_genome_case(rng, n_reads, genome_len=120000, read_len=(300, 1500)):
    """A two-contig genome with planted repeat families, a run of N and a tandem repeat, and reads drawn from it
    (half from the repeats), some with N inside and some with a subread shorter than the read.  Returns the FASTA
    text, reads (READ_DTYPE), pool."""
    genome, pool, reads, _ = repeat_genome_case(rng, n_reads, genome_len=genome_len, read_len=read_len)
    g = bytearray(bytes(genome))
    g[5000:5040] = b"N" * 40
    g[genome_len // 2:genome_len // 2 + 600] = (b"ACGGTCA" * 100)[:600]
    cut = (2 * genome_len) // 3
    fasta = b">a\n" + bytes(g[:cut]) + b"\n>b\n" + bytes(g[cut:]) + b"\n"
    pool = np.ascontiguousarray(pool).copy()
    reads = reads.copy()
    for r in range(len(reads)):
        off, L = int(reads[r]["off"]), int(reads[r]["length"])
        if r % 5 == 0:
            pool[off + L // 3: off + L // 3 + 3] = ord("N")
        if r % 3 == 0:
            reads[r]["subread_start"], reads[r]["subread_end"] = L // 10, L - L // 7
    return fasta, reads, pool


def write_suffix_array(tmp_dir, fasta, blt=None):
    """FASTA text -> (genome, contig starts, suffix array dict, .sa path) through the reference's sawriter."""
    import subprocess
    from blasr_amd import hip
    fa = os.path.join(str(tmp_dir), "g%s.fasta" % (blt or ""))
    with open(fa, "wb") as f:
        f.write(fasta)
    cmd = [SAWRITER, fa + ".sa", fa] + (["-blt", str(blt)] if blt else [])
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1200)
    genome, starts, _ = hip.read_genome(fa)
    return genome, starts, hip.read_suffix_array(fa + ".sa"), fa + ".sa"


def strand_view(read, s0, s1, strand):
    """The sequence and subread MapRead hands MapReadToGenome for a strand (Blasr.cpp:3449-3450)."""
    from blasr_amd import synth
    L = len(read)
    if strand:
        return np.ascontiguousarray(synth.revcomp(read)), L - s1, L - s0
    return np.ascontiguousarray(read), s0, s1


def oracle_anchors(oracle, genome, sa, read, s0, s1, strand, min_prefix, ap, sort=False):
    """bo_map_read_to_genome (+ bo_sort_match_pos) on one strand -> (n, 3) uint32 array of (t, q, l); None where the
    reference would assert."""
    oracle = getattr(oracle, "lib", oracle)  # a Checker or the library itself
    oracle.bo_map_read_to_genome.restype = C.c_long
    q, a0, a1 = strand_view(read, s0, s1, strand)
    ap = np.asarray(ap, dtype=np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    cap = 1 << 16
    while True:
        out = np.zeros(3 * cap, np.uint32)
        n = oracle.bo_map_read_to_genome(vp(genome), len(genome), vp(sa["index"]), vp(sa["start_pos_table"]),
                                         vp(sa["end_pos_table"]), int(sa["lookup_prefix_length"]), vp(q), len(q), a0, a1,
                                         int(min_prefix), vp(ap), vp(out), C.c_long(cap))
        if n <= cap:
            break
        cap = n
    if n < 0:  # the reference's asserts (MapBySuffixArray.h:283, :301-302) would end the process
        return None
    out = out[: 3 * n].reshape(n, 3)
    if sort and n:
        oracle.bo_sort_match_pos(vp(out), C.c_long(n))
    return out


def ref_anchors(ref, handle, genome, read, s0, s1, strand, min_prefix, ap, sort=False):
    """The reference's MapReadToGenome (+ SortMatchPosList) through oracle/_ref/libblasr_refmapper.so."""
    ref.ref_map_read_to_genome.restype = C.c_long
    ap = np.asarray(ap, dtype=np.int32)
    read = np.ascontiguousarray(read)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    cap = 1 << 16
    while True:
        out = np.zeros(3 * cap, np.uint32)
        n = ref.ref_map_read_to_genome(C.c_void_p(handle), vp(genome), len(genome), vp(read), len(read), s0, s1, strand,
                                       int(min_prefix), vp(ap), int(sort), vp(out), C.c_long(cap))
        if n <= cap:
            break
        cap = n
    return out[: 3 * n].reshape(n, 3)


def ref_find_intervals(ref, ct_handle, genome, starts, read_len, s0, s1, fwd, rc, stale, n_candidates=10, min_interval=100,
                       remove_contained=0, indel_rate=0.3, max_p_value=None):
    """The reference's RemoveOverlappingAnchors + FindMaxIncreasingInterval x 2 (+ RemoveContained) through
    oracle/_ref/libblasr_refmapper.so.  fwd / rc / stale: (n, 3) uint32 (t, q, l).  Returns ((k, 5) uint32 array of
    start, end, qStart, qEnd, strand; float32 p-values)."""
    import math
    ref.ref_find_intervals.restype = C.c_long
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    fwd, rc, stale = (np.ascontiguousarray(x, dtype=np.uint32) for x in (fwd, rc, stale))
    starts = np.ascontiguousarray(starts, dtype=np.uint32)
    iv = np.array([n_candidates, min_interval, remove_contained, 0, 0], np.int32)
    fv = np.array([indel_rate, math.log(0.5) if max_p_value is None else max_p_value], np.float32)
    cap = max(n_candidates, 1) + 4
    out = np.zeros(5 * cap, np.uint32)
    pv = np.zeros(cap, np.float32)
    k = ref.ref_find_intervals(C.c_void_p(ct_handle), vp(genome), len(genome), vp(starts), len(starts), read_len, s0, s1,
                               vp(fwd), C.c_long(len(fwd)), vp(rc), C.c_long(len(rc)), vp(stale), vp(iv), vp(fv), vp(out),
                               vp(pv), C.c_long(cap))
    assert k <= cap
    return out[: 5 * k].reshape(k, 5), pv[:k]


def nonguided_golden_cases():
    """tests/golden/nonguided_golden.npz (made by the reference's RefineAlignment with useGuidedAlign == false):
    the genome and, per candidate, read / subread / interval / contigs / bandSize and what the reference left behind."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "nonguided_golden.npz"))
    cases = []
    for i in range(int(g["n"])):
        p = "c%d_" % i
        par = [int(v) for v in g[p + "par"]]
        f = [int(v) for v in g[p + "fields"]]
        cases.append(dict(read=g[p + "read"], subread=(par[0], par[1]), iv=tuple(par[2:7]), band=par[7],
                          contigs=g[p + "contigs"] if len(g[p + "contigs"]) else None,
                          want=dict(tAlignedSeqPos=f[0], tAlignedSeqLength=f[1], qAlignedSeqPos=f[2], qAlignedSeqLength=f[3],
                                    qPos=f[4], tPos=f[5], nMatch=f[6], nMismatch=f[7], nIns=f[8], nDel=f[9],
                                    pct=np.float32(g[p + "pct"]), blocks=g[p + "blocks"], gapCounts=g[p + "gapCounts"],
                                    gapItems=g[p + "gapItems"])))
    return g["genome"], cases


def long_word_cases(rng, n=60):
    """(query, target, SdpParams) for -sdpTupleSize 14 .. 32 (a tuple is 64 bits, tuples/DNATuple.h:55-94): pairs clean enough
    to share words of that length; targets of every size class of the device's matchers (under 129 bases, under 256, up to a few
    thousand); tandem repeats whose copies agree in their first sixteen bases and differ behind them, or not at all (many equal
    words per position, maxMatchesPerPosition); runs of N; both alignment types; recursion (the children's words are k - 4
    and k / 2 bases: above sixteen too)."""
    from blasr_amd import synth
    acgt = np.frombuffer(b"ACGT", np.uint8)
    cases = []
    for it in range(n):
        kind = it % 6
        k = int(rng.integers(14, 33)) if it >= 8 else (16, 17, 19, 20, 24, 31, 32, 32)[it]
        qlen = int({0: rng.integers(20, 120), 1: rng.integers(100, 250), 2: rng.integers(200, 1400)}.get(kind, rng.integers(300, 5000)))
        er = dict(ins=float(rng.uniform(0, 0.03)), dele=float(rng.uniform(0, 0.02)), sub=float(rng.uniform(0, 0.01)))
        flank = (0, 0) if kind in (0, 1) and it % 2 else (0, int(rng.integers(0, 60)))
        q, t = synth.make_pair(rng, qlen, flank=flank, n_run=int(rng.integers(1, 12)) if kind == 4 else None, **er)
        if kind in (3, 5):  # a tandem repeat planted in both: units of 18 .. 40 bases
            unit = acgt[rng.integers(0, 4, size=int(rng.integers(18, 41)))]
            copies = int(rng.integers(3, 12))
            rep = np.tile(unit, copies)
            if kind == 5:  # every other copy differs in one base behind the sixteenth
                for c in range(1, copies, 2):
                    at = c * len(unit) + int(rng.integers(16, len(unit)))
                    rep[at] = acgt[(int(np.searchsorted(acgt, rep[at])) + 1) % 4]
            q, t = q.copy(), t.copy()
            aq, at_ = int(rng.integers(0, max(1, len(q) - len(rep)))), int(rng.integers(0, max(1, len(t) - len(rep))))
            q[aq: aq + len(rep)] = rep[: len(q) - aq]
            t[at_: at_ + len(rep)] = rep[: len(t) - at_]
        sp = SdpParams(k=k, align_type=int(rng.integers(0, 2)), recurse=int(rng.integers(0, 3)),
                       recurse_over=int(rng.choice([100, 1000])), max_per_pos=int(rng.choice([0, 0, 2, 5])),
                       prefix=int(rng.choice([0, 20, 50])), detailed=int(rng.integers(0, 2)) if it % 5 == 0 else 1)
        cases.append((np.ascontiguousarray(q), np.ascontiguousarray(t), sp))
    return cases


def long_word_golden_cases():
    """tests/golden/long_words_golden.npz (the reference's own SDPAlign on long_word_cases; make_long_words_golden.py)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "long_words_golden.npz"))
    out = []
    for i in range(int(g["n"])):
        p = "c%d_" % i
        v = [int(x) for x in g[p + "par"]]
        sp = SdpParams(k=v[0], align_type=v[1], recurse=v[2], recurse_over=v[3], max_per_pos=v[4], prefix=v[5], detailed=v[6])
        out.append((g[p + "q"], g[p + "t"], sp, dict(score=v[7], qPos=v[8], tPos=v[9], blocks=g[p + "blocks"].reshape(-1, 3))))
    return out
