"""A lane-level numpy model of kband_wide_kernel + kband_wide_traceback_kernel (blasr_amd/csrc/kband_api.hip):
the same strips, anti-diagonal steps, masks, boundary columns, arrow addressing and closed-form boundary walk, run
on the CPU against the oracle's AffineKBandAlign.  A design check for the index arithmetic -- not product code and
not a test of the library: `python tools/kband_wide_model.py` (needs oracle/libblasr_oracle.so).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

INF = 1 << 28
C, LANES = 4, 64
W = C * LANES
DIAG, LEFT, INS_CLOSE, HP_CLOSE, I_OPEN, H_OPEN = 0, 1, 2, 3, 4, 8


def base_code(b):
    return {65: 0, 67: 1, 71: 2, 84: 3}.get(int(b) & 0xdf, 4)


def shr1(v, edge):
    out = np.empty_like(v)
    out[1:] = v[:-1]
    out[0] = edge
    return out


def model(q_seq, t_seq, k, m, hpo, hpe, io, ie, dele):
    q_length, t_length = len(q_seq), len(t_seq)
    q_len, t_len = q_length, t_length  # SetKBoundedLengths
    if t_length < q_length:
        q_len = min(q_length, t_length + k)
    elif q_length < t_length:
        t_len = min(t_length, q_length + k)
    n_strips = (t_len + W - 1) // W
    steps_pad = (min(q_len, W + 2 * k) + 63 + 7) // 8 * 8
    arrows = np.zeros((n_strips, steps_pad, LANES, C), dtype=np.uint8)  # nibble per (strip, step, lane, column)
    bound = np.zeros((2, q_len + 2), dtype=np.int64)
    lane = np.arange(LANES)
    dp_score = None
    for s in range(n_strips):
        t_a = 1 + s * W
        q_lo, q_hi = max(1, t_a - k), min(q_len, t_a + W - 1 + k)
        n_rows = q_hi - q_lo + 1
        n_steps = n_rows + 63
        tcol = t_a + C * lane[:, None] + np.arange(C)[None, :]  # [lane][c]
        tc = np.array([[base_code(t_seq[t - 1]) if t <= t_len else 4 for t in row] for row in tcol])
        e_top = tcol - k - q_lo + lane[:, None]  # upperOk <=> d > e_top
        e_bot = tcol + k - q_lo + lane[:, None]  # leftOk  <=> d < e_bot
        # virtual row q_lo - 1 (row 0 when q_lo == 1)
        if q_lo == 1:
            pS = (tcol * dele).astype(np.int64)
        else:
            pS = np.full((LANES, C), INF, dtype=np.int64)
        pI = np.full((LANES, C), INF, dtype=np.int64)
        pH = np.full((LANES, C), INF, dtype=np.int64)
        last_s = pS[:, C - 1].copy()
        b_in, b_out = bound[(s + 1) & 1], bound[s & 1]

        def left_boundary(qq):  # S(qq, t_a - 1)
            if s == 0:
                return 0 if qq == 0 else qq * ie + io
            return int(b_in[qq]) if 0 <= qq <= q_len else INF

        # S(q_lo - 1, t_a - 1): row 0 by its formula, else the column the strip before left behind
        left_prev = np.zeros(LANES, dtype=np.int64)
        first_diag = (t_a - 1) * dele if q_lo == 1 else left_boundary(q_lo - 1)
        # lanes that have not started hold row q_lo-1: lane l's diagonal at its first step comes from lane l-1's
        # initial last_s, shifted twice
        left_in = shr1(last_s, first_diag)  # what the shift "one step before d = 0" produced
        info = np.zeros(LANES, dtype=np.int64)
        out_b = np.zeros(LANES, dtype=np.int64)
        for d in range(n_steps):
            qrow0 = q_lo + d  # the row lane 0 works on
            # row info enters at lane 0 and moves one lane per step
            if qrow0 <= q_len:
                raw = q_seq[qrow0 - 1]
                same = qrow0 > 1 and q_seq[qrow0 - 1] == q_seq[qrow0 - 2]
                new_info = base_code(raw) * 5 + (256 if same else 0)
            else:
                new_info = 0
            info = shr1(info, new_info)
            left_prev = left_in
            left_in = shr1(last_s, left_boundary(qrow0))
            active = (d - lane >= 0) & (d - lane < n_rows)
            qc5 = info & 0xff
            hp_ext_add = np.where(info & 256, hpe, INF)
            oS, oI, oH = pS.copy(), pI.copy(), pH.copy()
            sres = np.zeros((LANES, C), dtype=np.int64)
            for c in range(C):
                diag = left_prev if c == 0 else oS[:, c - 1]
                left = left_in if c == 0 else sres[:, c - 1]
                match = diag + m[qc5 + tc[:, c]]
                upper_ok = d > e_top[:, c]
                left_ok = d < e_bot[:, c]
                hp_open = oS[:, c] + hpo
                hp_ext = oH[:, c] + hp_ext_add
                h_opens = hp_open < hp_ext
                min_hp = np.where(upper_ok, np.minimum(hp_open, hp_ext), INF)
                ins_open = oS[:, c] + io
                ins_ext = oI[:, c] + ie
                i_opens = ins_open < ins_ext
                min_ins = np.where(upper_ok, np.minimum(ins_open, ins_ext), INF)
                del_score = np.where(left_ok, left + dele, INF)
                sc = np.minimum(np.minimum(match, del_score), np.minimum(min_ins, min_hp))
                arrow = np.full(LANES, HP_CLOSE)
                arrow = np.where(sc == min_ins, INS_CLOSE, arrow)
                arrow = np.where(sc == del_score, LEFT, arrow)
                arrow = np.where(sc == match, DIAG, arrow)
                nib = arrow | np.where(i_opens, I_OPEN, 0) | np.where(h_opens, H_OPEN, 0)
                sres[:, c] = sc
                # exec-masked update (ramp variant): inactive lanes keep their state
                pS[:, c] = np.where(active, sc, pS[:, c])
                pI[:, c] = np.where(active, min_ins, pI[:, c])
                pH[:, c] = np.where(active, min_hp, pH[:, c])
                arrows[s, d, :, c] = np.where(active, nib, arrows[s, d, :, c])
            last_s = pS[:, C - 1].copy()
            r = d - 63  # the row lane 63 has just finished
            if 0 <= r < n_rows:
                out_b[r % 64] = last_s[63]
                if r % 64 == 63 or r == n_rows - 1:
                    base = q_lo + (r & ~63)
                    for i in range((r % 64) + 1):
                        b_out[base + i] = out_b[i]
        if s == n_strips - 1:
            l_star, c_star = (t_len - t_a) // C, (t_len - t_a) % C
            dp_score = int(pS[l_star, c_star])
    # ---- traceback
    ops = []

    def nibble(qq, tt):
        s = (tt - 1) // W
        col = (tt - 1) % W
        l, cc = col // C, col % C
        q_lo = max(1, 1 + s * W - k)
        return int(arrows[s, qq - q_lo + l, l, cc])

    qq, tt, mat = q_len, t_len, 0
    while qq > 0 and tt > 0:
        nb = nibble(qq, tt)
        if mat == 0:
            ar = nb & 3
            if ar == DIAG:
                ops.append("M"); qq -= 1; tt -= 1
            elif ar == LEFT:
                ops.append("D"); tt -= 1
            else:
                mat = 1 if ar == INS_CLOSE else 2
        else:
            if nb & (I_OPEN if mat == 1 else H_OPEN):
                mat = 0
            ops.append("I"); qq -= 1
    if tt == 0:
        ops += ["I"] * qq
    else:
        ops += ["D"] * tt
    ops.reverse()
    return dp_score, ops, q_len, t_len


def ops_to_blocks(ops):
    """ArrowPathToAlignment (Alignment.h:190-254) + the trailing gap list cleared, as the oracle returns it."""
    blocks, q, t, i = [], 0, 0, 0
    while i < len(ops):
        if ops[i] == "M":
            j = i
            while j < len(ops) and ops[j] == "M":
                j += 1
            blocks.append((q, t, j - i))
            q += j - i
            t += j - i
            i = j
        else:
            if ops[i] == "I":
                q += 1
            else:
                t += 1
            i += 1
    return np.array(blocks, dtype=np.int32).reshape(-1, 3)


def main():
    from blasr_amd import synth
    from blasr_amd.hip import SMRT_MATRIX
    from blasr_testlib import affine_kband_align, load_oracle
    oracle = load_oracle()
    rng = np.random.default_rng(3)
    m = SMRT_MATRIX.reshape(-1).astype(np.int64)
    bad = 0
    for it in range(40):
        ql = int(rng.choice([3, 30, 200, 300, 700]))
        q, t = synth.make_pair(rng, ql, flank=(0, 0))
        if it % 3 == 0:
            t = t[: max(1, len(t) - int(rng.integers(0, 40)))]
        if it % 4 == 0:
            q = q[: max(1, len(q) - int(rng.integers(0, 40)))]
        if it % 5 == 0:
            q = q.copy()
            q[len(q) // 3: len(q) // 3 + 9] = ord("T")
        k = int(rng.choice([10, 12, 40, 100, 150, 400]))
        want = affine_kband_align(oracle, q, t, k)
        score, ops, q_len, t_len = model(q, t, k, m, 7, 2, 7, 4, 5)
        # the oracle's blocks are relative to the first block's start (qPos / tPos folded by ArrowPathToAlignment)
        blocks = ops_to_blocks(ops)
        if len(blocks):
            blocks = blocks - np.array([blocks[0][0], blocks[0][1], 0], dtype=np.int32)
        ok = score == want["score"] and np.array_equal(blocks, want["blocks"])
        if not ok:
            bad += 1
            print("DIFF", it, len(q), len(t), k, score, want["score"], len(blocks), len(want["blocks"]))
    print("model vs oracle: %d differ of 40" % bad)


if __name__ == "__main__":
    main()
