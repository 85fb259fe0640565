"""Decode tests/golden/refine_golden.npz (written by tests/golden/make_golden.py)."""
import numpy as np

from blasr_testlib import ScoreParams, SdpParams


class GoldenCase:
    def __init__(self, g, idx):
        p = "c%02d_" % idx
        self.name = str(g["names"][idx])
        self.q, self.t = g[p + "q"], g[p + "t"]
        s = g[p + "score"]
        self.sc = ScoreParams(matrix=s[:25].reshape(5, 5), ins=int(s[25]), dele=int(s[26]),
                              affine_open=int(s[27]), affine_extend=int(s[28]))
        sp = g[p + "sdp_params"]
        self.sp = SdpParams(k=int(sp[0]), sdp_ins=int(sp[1]), sdp_del=int(sp[2]), indel_rate=float(g[p + "indel_rate"][0]),
                            align_type=int(sp[3]), detailed=int(sp[4]), extend_front=int(sp[5]), prefix=int(sp[6]),
                            recurse=int(sp[7]), recurse_over=int(sp[8]), max_per_pos=int(sp[9]))
        self.band = int(sp[10])
        self.sdp_score, self.sdp_qpos, self.sdp_tpos = (int(v) for v in g[p + "sdp"])
        self.sdp_blocks = g[p + "sdp_blocks"].reshape(-1, 3)
        self.sdp_stats = g[p + "sdp_stats"]
        self.sdp_pct = float(g[p + "sdp_pct"][0])
        self.has_refine = (p + "aga") in g.files
        if self.has_refine:
            self.guide_digest = g[p + "guide_digest"]
            self.aga = g[p + "aga"]
            self.aga_pct = float(g[p + "aga_pct"][0])
            self.aga_blocks = g[p + "aga_blocks"].reshape(-1, 3)
            self.aga_gap_counts = g[p + "aga_gap_counts"]
            self.aga_gap_items = g[p + "aga_gap_items"].reshape(-1, 2)
            bl = self.sdp_blocks
            self.qs = self.q[self.sdp_qpos: self.sdp_qpos + bl[-1, 0] + bl[-1, 2]]
            self.ts = self.t[self.sdp_tpos: self.sdp_tpos + bl[-1, 1] + bl[-1, 2]]

    def expected_refine(self):
        a = self.aga
        return dict(qPos=int(a[0]), tPos=int(a[1]), dpScore=int(a[2]), nCells=int(a[3]), score=int(a[4]),
                    nMatch=int(a[5]), nMismatch=int(a[6]), nIns=int(a[7]), nDel=int(a[8]), pct=self.aga_pct,
                    blocks=self.aga_blocks, gapCounts=self.aga_gap_counts, gapItems=self.aga_gap_items)


def golden_cases(g):
    return [GoldenCase(g, i) for i in range(len(g["names"]))]


def guide_digest(rows, ncells):
    return np.array([len(rows), ncells, int(rows[:, 2].max()), int(rows[:, 3].max()),
                     int((rows[:, 1:4].astype(np.int64) * [3, 5, 7]).sum() % (1 << 31))], dtype=np.int64)


REFINE_KEYS = ["qPos", "tPos", "dpScore", "nCells", "score", "nMatch", "nMismatch", "nIns", "nDel", "pct",
               "blocks", "gapCounts", "gapItems"]


def first_difference(a, b, keys):
    for k in keys:
        if isinstance(a[k], np.ndarray) or isinstance(b[k], np.ndarray):
            x, y = np.asarray(a[k]), np.asarray(b[k])
            if x.shape != y.shape or not (x == y).all():
                return k
        elif a[k] != b[k]:
            return k
    return None
