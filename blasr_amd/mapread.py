"""MapRead from the candidate intervals on: the host-side mirror of the reference's per-read driver, batched over
all reads of a call so the device sees every interval at once.

The reference runs, per read (Blasr.cpp:2530-2644, 3509-3516, 3086):
  AlignIntervals (windows -> SDPAlign + statistics per interval)         :2539, :605-1105
  sort by score, RemoveLowQualityAlignments                              :2547, :2594-2596
  RefineAlignments (AffineGuidedAlign + statistics, copy-back), sort     :2607, :2164-2181, :880-914
  RemoveLowQualityAlignments, RemoveOverlappingAlignments                :2615-2616
  OrderGapsByType when the output is for Picard (-sam)                   :2621-2626
  AssignRefContigLocation, StoreMapQVs                                   :3509-3516
  print the first bestn                                                  :3086
Here the per-interval stages run once for the whole batch on the device (Context.windows / sdp / refine) and the
per-read bookkeeping runs through the library's host entry points.  Anchoring -- where the intervals come from -- is
outside this path (SURVEY next-1): the caller brings them.

Everything goes through libblasr_hip.so; there is no CPU fallback."""
import numpy as np

from . import hip

MAX_SCORE = -200  # MappingParameters::maxScore (MappingParameters.h)


class Hit(dict):
    """One surviving alignment of a read, in print order.  Keys: read, interval, strand, contig, contig_pos, score,
    pct_similarity, q_align_start, q_align_end, t_start, t_end, t_length, map_qv, ops, q, t, q_pos, t_pos,
    q_aligned_seq_pos, n_match, read_length."""


def map_reads(ctx, genome_len, contig_starts, pool, reads, intervals, for_picard=False, bestn=None):
    """reads: READ_DTYPE array over `pool`; intervals: INTERVAL_DTYPE array (any order, each naming its read).
    The context must hold the genome.  Returns, per read, the list of Hits the reference would keep, best first."""
    n_reads, n_iv = len(reads), len(intervals)
    owner = np.asarray(intervals["read"], dtype=np.int64) if n_iv else np.zeros(0, np.int64)
    # ---- AlignIntervals for every interval of every read: one batch
    seq, q_off, q_len, t_off, t_len, win = ctx.windows(intervals, reads, pool)
    pairs = [(seq[q_off[k]: q_off[k] + q_len[k]], seq[t_off[k]: t_off[k] + t_len[k]]) for k in range(n_iv)]
    sdp = ctx.sdp(pairs)
    # ---- refine everything that has blocks (a superset of what survives the first filter; candidates are
    # independent, so refining the ones the filter drops changes nothing)
    batch, slot = hip.RefineBatch(), {}
    for k, ((q, t), s) in enumerate(zip(pairs, sdp)):
        if len(s["blocks"]) == 0:
            continue
        last = s["blocks"][-1]
        slot[k] = len(slot)
        batch.add(q[s["qPos"]: s["qPos"] + last[0] + last[2]], t[s["tPos"]: s["tPos"] + last[1] + last[2]], s["blocks"])
    refined = ctx.refine(batch) if len(slot) else []
    by_read = [[] for _ in range(n_reads)]
    for k in range(n_iv):
        by_read[int(owner[k])].append(k)
    out = []
    for r in range(n_reads):
        ks = by_read[r]
        L = int(reads[r]["length"])
        hits = []
        out.append(hits)
        if not ks:
            continue
        # candidates as AlignIntervals leaves them
        cands = np.zeros(len(ks), dtype=hip.CANDIDATE_DTYPE)
        blocks, at = [], 0
        for j, k in enumerate(ks):
            s, bl = sdp[k], sdp[k]["blocks"]
            st = s["stats"]
            cols = st["nMatch"] + st["nMismatch"] + st["nIns"] + st["nDel"]
            pct = np.float32((st["nMatch"] * 2.0) / (cols + cols) * 100) if cols else np.float32(0)
            cands[j] = (st["score"], s["tPos"], int(win[k]["t_aligned_seq_pos"]),
                        int(bl[-1, 1] + bl[-1, 2]) if len(bl) else 0, int(win[k]["t_index"]), int(win[k]["t_strand"]),
                        pct, len(bl), at)
            blocks.append(np.asarray(bl, dtype=np.int32).reshape(-1, 3))
            at += len(bl)
        flat = np.concatenate(blocks).reshape(-1) if at else np.zeros(0, np.int32)
        order = hip.sort_candidates(cands)                                          # :2547
        order, cands = hip.remove_low_quality(cands, flat, order)                   # :2594-2596
        # RefineAlignments (:2164-2181): the survivors take their refined fields (copy-back :880-914), then sort
        ref_of = {}
        for j in order:
            k = ks[int(j)]
            if k not in slot:
                cands[j]["score"] = 0
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
            bl = (np.asarray(ref_of[j][1].blocks_and_gaps()[0], dtype=np.int32).reshape(-1, 3) if j in ref_of
                  else np.zeros((0, 3), np.int32))
            cands[j]["n_blocks"], cands[j]["blk_off"] = len(bl), at2
            blocks2.append(bl)
            at2 += len(bl)
        flat2 = np.concatenate(blocks2).reshape(-1) if at2 else np.zeros(0, np.int32)
        order = hip.sort_candidates(cands, order)                                   # :2178-2180
        order, cands = hip.remove_low_quality(cands, flat2, order)                  # :2615
        order = hip.remove_overlapping(cands, order)                                # :2616
        order = [int(j) for j in order if int(j) in ref_of]
        if not order:
            continue
        # StoreMapQVs over what is left (:3509-3516); -sam has re-ordered the gap lists before it (:2621-2626)
        alns = []
        for j in order:
            k, rr, s = ref_of[j]
            q, t = pairs[k]
            last = s["blocks"][-1]
            qs = q[s["qPos"]: s["qPos"] + last[0] + last[2]]
            ts = t[s["tPos"]: s["tPos"] + last[1] + last[2]]
            ops = hip.order_gaps_by_type(rr.ops) if for_picard else rr.ops
            q_span = int(sum(int(x >> 2) for x in rr.ops if x & 3 != 2))
            q_start = int(win[k]["q_aligned_seq_pos"]) + s["qPos"] + rr.qPos
            alns.append(dict(q=qs, t=ts, ops=ops, q_pos=rr.qPos, t_pos=rr.tPos, q_align_start=q_start,
                             q_align_end=q_start + q_span, q_strand=0, q_length=L, n_match=rr.nMatch))
        qvs = hip.map_qvs(alns)[0]
        for rank, j in enumerate(order[: bestn if bestn else len(order)]):
            k, rr, s = ref_of[j]
            a = alns[rank]
            strand = int(win[k]["t_strand"])
            contig, contig_pos = hip.ref_contig_location(int(cands[j]["t_aligned_seq_pos"]), len(a["t"]), strand,
                                                         contig_starts, genome_len)
            t_span = int(cands[j]["t_end"])
            hits.append(Hit(read=r, interval=k, strand=strand, contig=contig, contig_pos=contig_pos, score=rr.score,
                            pct_similarity=float(rr.pctSimilarity), q_align_start=a["q_align_start"],
                            q_align_end=a["q_align_end"], t_start=contig_pos + rr.tPos,
                            t_end=contig_pos + rr.tPos + t_span, t_span=t_span, t_length=int(win[k]["t_length"]),
                            map_qv=int(qvs[rank]) if cands[j]["score"] < MAX_SCORE else 50, ops=a["ops"], q=a["q"],
                            t=a["t"], q_pos=rr.qPos, t_pos=rr.tPos,
                            q_aligned_seq_pos=int(win[k]["q_aligned_seq_pos"]) + s["qPos"], n_match=rr.nMatch,
                            read_length=L))
    return out


def m4_record(hit, read_name, contig_name):
    """IntervalAlignmentPrinter::Print (-m 4) of one Hit, with the trailing newline."""
    return hip.m4_line(read_name, contig_name, hit["score"], hit["pct_similarity"], 0, hit["q_align_start"],
                       hit["q_align_end"], hit["read_length"], hit["strand"], hit["t_start"], hit["t_end"],
                       hit["t_length"], hit["map_qv"])


def sam_record(hit, read_name, contig_name, read_group, read_seq, quality=None, clipping=hip.CLIP_NONE, is_primary=1,
               subread_start=0, subread_end=None, hole_number=-1):
    """CreateCIGARString + SAMOutput::PrintAlignment of one Hit (map the reads with for_picard=True for -sam)."""
    L = hit["read_length"]
    cig, clips = hip.cigar(hit["q"], hit["t"], hit["q_pos"], hit["t_pos"], ops=hit["ops"], clipping=clipping,
                           t_strand=hit["strand"], read_length=L, q_align_start=hit["q_align_start"],
                           q_align_end=hit["q_align_end"])
    return hip.sam_line(read_name, contig_name, read_group, cig, clips, read_seq, quality, clipping=clipping,
                        q_strand=0, t_strand=hit["strand"], is_primary=is_primary, read_length=L,
                        subread_start=subread_start, subread_end=L if subread_end is None else subread_end,
                        q_align_start=hit["q_align_start"], q_align_end=hit["q_align_end"],
                        t_align_start=hit["t_start"], t_end=hit["t_span"], t_length=hit["t_length"], q_length=L,
                        q_aligned_seq_pos=hit["q_aligned_seq_pos"], q_aligned_seq_length=len(hit["q"]),
                        map_qv=hit["map_qv"], score=hit["score"], edit_dist=0, hole_number=hole_number)
