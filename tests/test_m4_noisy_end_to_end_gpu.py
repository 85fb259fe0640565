"""End to end on NOISY reads with competing candidates: PacBio-like reads against a two-contig genome with planted
repeats.  tests/golden/make_m4_noisy_golden.py ran the reference BINARY (oracle/_ref/blasrmc -m 4 -bestn 1 -V 1) and
kept, besides the -m 4 lines, the candidate intervals it derived from its anchors.  Given those intervals -- anchoring
(SURVEY next-1) is the one stage that is not built -- everything else of MapRead is done here, in the reference's
order (Blasr.cpp:2530-2644, 3509-3516, 3086): windows -> SDPAlign + statistics per interval -> sort by score ->
RemoveLowQualityAlignments -> AffineGuidedAlign + statistics -> sort -> RemoveLowQualityAlignments ->
RemoveOverlappingAlignments -> AssignRefContigLocation -> StoreMapQVs -> the -m 4 line of the best one.
The lines must equal the reference's byte for byte, and so must the SAM records of the same alignments (-sam:
CreateCIGARString + SAMOutput::PrintAlignment)."""
import os

import numpy as np
import pytest

import blasr_amd

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "m4_noisy_golden.npz")
MAX_SCORE = -200  # MappingParameters::maxScore


def test_m4_lines_of_noisy_reads_with_the_reference_intervals():
    g = np.load(GOLDEN)
    contigs = [g["contig0"], g["contig1"]]
    want = bytes(g["lines"]).decode().splitlines()
    want_sam = bytes(g["sam_lines"]).decode().splitlines()
    read_group = [x for x in bytes(g["sam_header"]).decode().splitlines() if x.startswith("@RG")][0].split("ID:")[1].split("\t")[0]
    read_lens, pool, intervals, owner = g["read_lens"], g["reads"], g["intervals"], g["owner"]
    genome = np.concatenate([contigs[0], [ord("N")], contigs[1], [ord("N")]]).astype(np.uint8)
    starts = np.array([0, len(contigs[0]) + 1, len(contigs[0]) + 1 + len(contigs[1]) + 1], dtype=np.uint32)
    ctx = blasr_amd.Context(0)
    ctx.upload_genome(genome, starts)
    n_reads, n_iv = len(read_lens), len(intervals)
    reads = np.zeros(n_reads, dtype=blasr_amd.READ_DTYPE)
    off = 0
    for i in range(n_reads):
        reads[i] = (off, int(read_lens[i]), 0, int(read_lens[i]), 0)
        off += int(read_lens[i])
    ivs = np.zeros(n_iv, dtype=blasr_amd.INTERVAL_DTYPE)
    for k in range(n_iv):
        s, e, qs, qe, strand = (int(v) for v in intervals[k])
        ivs[k] = (s, e, qs, qe, int(owner[k]), strand)
    # ---- AlignIntervals for every interval of every read: one batch
    seq, q_off, q_len, t_off, t_len, win = ctx.windows(ivs, reads, pool)
    pairs = [(seq[q_off[k]: q_off[k] + q_len[k]], seq[t_off[k]: t_off[k] + t_len[k]]) for k in range(n_iv)]
    sdp = ctx.sdp(pairs)
    # ---- refine everything that has blocks (a superset of what survives the first filter; candidates are
    # independent, so refining the ones the filter drops changes nothing)
    batch, slot = blasr_amd.RefineBatch(), {}
    for k, ((q, t), s) in enumerate(zip(pairs, sdp)):
        if len(s["blocks"]) == 0:
            continue
        last = s["blocks"][-1]
        slot[k] = len(slot)
        batch.add(q[s["qPos"]: s["qPos"] + last[0] + last[2]], t[s["tPos"]: s["tPos"] + last[1] + last[2]], s["blocks"])
    refined = ctx.refine(batch)
    got, got_sam = [], []
    for r in range(n_reads):
        ks = [k for k in range(n_iv) if owner[k] == r]
        L = int(read_lens[r])
        # candidates as AlignIntervals leaves them
        cands = np.zeros(len(ks), dtype=blasr_amd.CANDIDATE_DTYPE)
        blocks, at = [], 0
        for j, k in enumerate(ks):
            s, bl = sdp[k], sdp[k]["blocks"]
            cols = s["stats"]["nMatch"] + s["stats"]["nMismatch"] + s["stats"]["nIns"] + s["stats"]["nDel"]
            pct = np.float32((s["stats"]["nMatch"] * 2.0) / (cols + cols) * 100) if cols else np.float32(0)
            cands[j] = (s["stats"]["score"], s["tPos"], int(win[k]["t_aligned_seq_pos"]),
                        int(bl[-1, 1] + bl[-1, 2]) if len(bl) else 0, int(win[k]["t_index"]), int(win[k]["t_strand"]),
                        pct, len(bl), at)
            blocks.append(np.asarray(bl, dtype=np.int32).reshape(-1, 3))
            at += len(bl)
        flat = np.concatenate(blocks).reshape(-1) if at else np.zeros(0, np.int32)
        order = blasr_amd.sort_candidates(cands)                                    # :2547
        order, cands = blasr_amd.remove_low_quality(cands, flat, order)             # :2594-2596
        # RefineAlignments (:2164-2181): the survivors take their refined fields (copy-back :880-914), then sort
        ref_of = {}
        for j in order:
            k = ks[int(j)]
            if k not in slot:
                cands[j]["score"] = 0                                               # no blocks: score 0 (:941-944 style)
                continue
            rr, s = refined[slot[k]], sdp[k]
            t_span = int(sum(int(x >> 2) for x in rr.ops if x & 3 != 1))
            cands[j]["score"] = rr.score
            cands[j]["t_aligned_seq_pos"] = int(win[k]["t_aligned_seq_pos"]) + s["tPos"]
            cands[j]["t_pos"] = rr.tPos
            cands[j]["t_end"] = t_span
            cands[j]["pct_similarity"] = rr.pctSimilarity
            ref_of[int(j)] = (k, rr, s)
        # the refined blocks for the maxGap test of the second RemoveLowQualityAlignments
        blocks2, at2 = [], 0
        for j in range(len(ks)):
            bl = np.asarray(ref_of[j][1].blocks_and_gaps()[0], dtype=np.int32).reshape(-1, 3) if j in ref_of else np.zeros((0, 3), np.int32)
            cands[j]["n_blocks"], cands[j]["blk_off"] = len(bl), at2
            blocks2.append(bl)
            at2 += len(bl)
        flat2 = np.concatenate(blocks2).reshape(-1) if at2 else np.zeros(0, np.int32)
        order = blasr_amd.sort_candidates(cands, order)                             # :2178-2180
        order, cands = blasr_amd.remove_low_quality(cands, flat2, order)            # :2615
        order = blasr_amd.remove_overlapping(cands, order)                          # :2616
        if len(order) == 0:
            continue
        # StoreMapQVs over what is left (:3509-3516), then the first one is printed (:3086)
        alns = []
        for j in order:
            k, rr, s = ref_of[int(j)]
            q, t = pairs[k]
            last = s["blocks"][-1]
            qs = q[s["qPos"]: s["qPos"] + last[0] + last[2]]
            ts = t[s["tPos"]: s["tPos"] + last[1] + last[2]]
            q_span = int(sum(int(x >> 2) for x in rr.ops if x & 3 != 2))
            q_start = int(win[k]["q_aligned_seq_pos"]) + s["qPos"] + rr.qPos
            alns.append(dict(q=qs, t=ts, ops=rr.ops, q_pos=rr.qPos, t_pos=rr.tPos, q_align_start=q_start,
                             q_align_end=q_start + q_span, q_strand=0, q_length=L, n_match=rr.nMatch))
        j0 = int(order[0])
        k, rr, s = ref_of[j0]
        map_qv = blasr_amd.map_qvs(alns)[0][0] if cands[j0]["score"] < MAX_SCORE else 50
        ts_len = len(alns[0]["t"])
        contig, contig_pos = blasr_amd.ref_contig_location(int(cands[j0]["t_aligned_seq_pos"]), ts_len,
                                                           int(win[k]["t_strand"]), starts, len(genome))
        t_span = int(cands[j0]["t_end"])
        line = blasr_amd.m4_line("read%d/%d/0_%d/0_%d" % (r, r, L, L), "contig%d" % contig, rr.score,
                                 float(rr.pctSimilarity), 0, alns[0]["q_align_start"], alns[0]["q_align_end"], L,
                                 int(win[k]["t_strand"]), contig_pos + rr.tPos, contig_pos + rr.tPos + t_span,
                                 int(win[k]["t_length"]), map_qv)
        got.append(line.rstrip("\n"))
        # ---- the SAM record of the same alignment (-sam, clipping none): CIGAR + PrintAlignment
        a0, strand = alns[0], int(win[k]["t_strand"])
        # -sam switches forPicard on (MappingParameters.h:488-491), so MapRead has run OrderGapsByType over the
        # alignment (Blasr.cpp:2621-2626) before it is printed
        sam_ops = blasr_amd.order_gaps_by_type(rr.ops)
        if cands[j0]["score"] < MAX_SCORE:  # StoreMapQVs comes after it and scores the re-ordered gap lists
            map_qv = blasr_amd.map_qvs([dict(x, ops=blasr_amd.order_gaps_by_type(x["ops"])) for x in alns])[0][0]
        cig, clips = blasr_amd.cigar(a0["q"], a0["t"], rr.qPos, rr.tPos, ops=sam_ops, clipping=blasr_amd.hip.CLIP_NONE,
                                     t_strand=strand, read_length=L, q_align_start=a0["q_align_start"],
                                     q_align_end=a0["q_align_end"])
        read_seq = pool[int(reads[r]["off"]): int(reads[r]["off"]) + L]
        sam = blasr_amd.sam_line("read%d/%d/0_%d/0_%d" % (r, r, L, L), "contig%d" % contig, read_group, cig, clips,
                                 read_seq, None, clipping=blasr_amd.hip.CLIP_NONE, q_strand=0, t_strand=strand,
                                 is_primary=1, read_length=L, subread_start=0, subread_end=L,
                                 q_align_start=a0["q_align_start"], q_align_end=a0["q_align_end"],
                                 t_align_start=contig_pos + rr.tPos, t_end=t_span, t_length=int(win[k]["t_length"]),
                                 q_length=L, q_aligned_seq_pos=int(win[k]["q_aligned_seq_pos"]) + s["qPos"],
                                 q_aligned_seq_length=len(a0["q"]), map_qv=map_qv, score=rr.score, edit_dist=0,
                                 hole_number=-1)
        got_sam.append(sam.rstrip("\n"))
    bad = [(a, b) for a, b in zip(got, want) if a != b]
    assert len(got) == len(want) and not bad, (len(got), len(want), bad[:3])
    bad = [(a[:300], b[:300]) for a, b in zip(got_sam, want_sam) if a != b]
    assert len(got_sam) == len(want_sam) and not bad, (len(bad), bad[:2])
