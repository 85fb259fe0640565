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
ORACLE_SO = os.path.join(ROOT, "oracle", "libblasr_oracle.so")
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
