"""MapRead from the candidate intervals on, and the epilogue behind it, for a block of reads.  MapRead proper lives in
the library (C++: blasr_amd/csrc/mapread_api.hip, blasr_hip_map_reads_*); this module is the thin Python caller
plus the printing helpers the tests use.

The reference runs, per read (Blasr.cpp:2530-2644, 3509-3516, 3086):
  AlignIntervals (windows -> SDPAlign + statistics per interval)         :2539, :605-1105
  sort by score, RemoveLowQualityAlignments                              :2547, :2594-2596
  RefineAlignments (AffineGuidedAlign + statistics, copy-back), sort     :2607, :2164-2181, :880-914
  RemoveLowQualityAlignments, RemoveOverlappingAlignments                :2615-2616
  OrderGapsByType when the output is for Picard (-sam)                   :2621-2626
  AssignRefContigLocation, StoreMapQVs                                   :3509-3516
  print the first bestn                                                  :3086
Everything up to OrderGapsByType is inside blasr_hip_map_reads_host: the per-interval stages as device batches over
all reads of the call, the per-read bookkeeping in C++ between them.  Anchoring -- where the intervals come from --
is outside this path (SURVEY next-1): the caller brings them.

Everything goes through libblasr_hip.so; there is no CPU fallback."""
import numpy as np

from . import hip



class Hit(dict):
    """One surviving alignment of a read, in print order.  Keys: read, interval, strand, contig, contig_pos, score,
    pct_similarity, q_align_start, q_align_end, t_start, t_end, t_length, map_qv, ops, q, t, q_pos, t_pos,
    q_aligned_seq_pos, n_match, read_length, rank (its index among the read's alignments: 0 is the primary one)."""


def candidate_intervals(ctx, genome, contig_starts, count_table, pool, reads, anchor_params=None, interval_params=None,
                        chains=False):
    """MapRead up to AlignIntervals (Blasr.cpp:2279-2500): MapReadToGenome for both strands + SortMatchPosList on the
    device (the context holds the genome and its suffix array), then RemoveOverlappingAnchors /
    FindMaxIncreasingInterval / the interval set on the host.  Returns (intervals INTERVAL_DTYPE in the order
    AlignIntervals walks them read by read, read_interval_off, p_values); with chains=True also the intervals' chained
    anchors (chains, chain_off), which -rbao reads."""
    anchors, list_off = ctx.anchor_reads(reads, pool, anchor_params)
    return hip.find_intervals(genome, contig_starts, count_table, reads, anchors, list_off, interval_params, chains=chains)


def map_reads(ctx, genome_len, contig_starts, pool, reads, intervals, for_picard=False, bestn=None, params=None,
              min_align_length=0, min_map_qv=0, unaligned=None, chains=None):
    """reads: READ_DTYPE array over `pool`; intervals: INTERVAL_DTYPE array (any order, each naming its read).
    The context must hold the genome.  Returns, per read, the list of Hits the reference would print, best first
    (`unaligned`, a list, receives the reads MapRead left without any alignment).

    MapRead itself -- the device stages and the per-read ordering / pruning between them -- is ONE call into the
    library's C++ driver (blasr_hip_map_reads_host, blasr_amd/csrc/mapread_api.hip); what follows here is the
    epilogue of MapReads (:3509-3516, :3086) through the library's host entry points."""
    n_reads = len(reads)
    params = params or ctx.map_params()
    max_score = int(params.max_score)
    hits, off, seq, ops = ctx.map_reads(reads, pool, intervals, params=params, fetch=True, chains=chains)
    out = []
    for r in range(n_reads):
        mine = hits[int(off[r]): int(off[r + 1])]
        res = []
        out.append(res)
        if len(mine) == 0:
            if unaligned is not None:  # -sam prints a record for these (hip.sam_unaligned_line; Blasr.cpp:3848-3866)
                unaligned.append(r)
            continue
        # StoreMapQVs over what is left (:3509-3516); -sam has re-ordered the gap lists before it (:2621-2626)
        alns = []
        for h in mine:
            qs = seq[int(h["q_off"]): int(h["q_off"]) + int(h["q_aligned_seq_length"])]
            ts = seq[int(h["t_off"]): int(h["t_off"]) + int(h["t_aligned_seq_length"])]
            o = ops[int(h["ops_off"]): int(h["ops_off"]) + int(h["n_ops"])]
            o = hip.order_gaps_by_type(o) if for_picard else o.copy()
            q_start = int(h["q_aligned_seq_pos"]) + int(h["q_pos"])
            q_span = int(h["n_match"]) + int(h["n_mismatch"]) + int(h["n_ins"])
            alns.append(dict(q=qs, t=ts, ops=o, q_pos=int(h["q_pos"]), t_pos=int(h["t_pos"]), q_align_start=q_start,
                             q_align_end=q_start + q_span, q_strand=0, q_length=int(h["q_length"]),
                             n_match=int(h["n_match"])))
        # the mapQVs are stored only when the best alignment beats maxScore (:3511-3516); else they stay 50
        qvs = (hip.map_qvs(alns, ins=int(params.score.ins), dele=int(params.score.del_),
                           affine_open=int(params.score.affine_open), affine_extend=int(params.score.affine_extend))[0]
               if int(mine[0]["score"]) < max_score else [50] * len(mine))
        for rank, h in enumerate(mine[: bestn if bestn else len(mine)]):
            # PrintAlignment's gates (:2957-2978): an alignment held back is not replaced by the next one
            if not hip.alignment_is_printed(int(h["score"]), float(h["pct_similarity"]), int(h["t_aligned_seq_length"]),
                                            int(qvs[rank]), max_score, float(params.min_pct_identity), min_align_length,
                                            min_map_qv):
                continue
            a = alns[rank]
            strand = int(h["t_strand"])
            contig, contig_pos = hip.ref_contig_location(int(h["t_aligned_seq_pos"]), len(a["t"]), strand,
                                                         contig_starts, genome_len)
            t_span = int(h["n_match"]) + int(h["n_mismatch"]) + int(h["n_del"])
            res.append(Hit(read=r, interval=int(h["interval"]), strand=strand, contig=contig, contig_pos=contig_pos,
                           score=int(h["score"]), pct_similarity=float(h["pct_similarity"]),
                           q_align_start=a["q_align_start"], q_align_end=a["q_align_end"],
                           t_start=contig_pos + int(h["t_pos"]), t_end=contig_pos + int(h["t_pos"]) + t_span,
                           t_span=t_span, t_length=int(h["t_length"]),
                           map_qv=int(qvs[rank]), ops=a["ops"], q=a["q"],
                           t=a["t"], q_pos=int(h["q_pos"]), t_pos=int(h["t_pos"]),
                           q_aligned_seq_pos=int(h["q_aligned_seq_pos"]), n_match=int(h["n_match"]),
                           read_length=int(h["q_length"]), rank=rank))
    return out


def m4_record(hit, read_name, contig_name):
    """IntervalAlignmentPrinter::Print (-m 4) of one Hit, with the trailing newline."""
    return hip.m4_line(read_name, contig_name, hit["score"], hit["pct_similarity"], 0, hit["q_align_start"],
                       hit["q_align_end"], hit["read_length"], hit["strand"], hit["t_start"], hit["t_end"],
                       hit["t_length"], hit["map_qv"])


def sam_record(hit, read_name, contig_name, read_group, read_seq, quality=None, clipping=hip.CLIP_NONE, is_primary=None,
               subread_start=0, subread_end=None, hole_number=-1):
    """CreateCIGARString + SAMOutput::PrintAlignment of one Hit (map the reads with for_picard=True for -sam).
    is_primary: by default what PrintAlignments sets (Blasr.cpp:3109-3114): the first alignment of the read."""
    L = hit["read_length"]
    if is_primary is None:
        is_primary = 1 if hit.get("rank", 0) == 0 else 0
    cig, clips = hip.cigar(hit["q"], hit["t"], hit["q_pos"], hit["t_pos"], ops=hit["ops"], clipping=clipping,
                           t_strand=hit["strand"], read_length=L, q_align_start=hit["q_align_start"],
                           q_align_end=hit["q_align_end"], subread_start=subread_start,
                           subread_end=L if subread_end is None else subread_end)
    return hip.sam_line(read_name, contig_name, read_group, cig, clips, read_seq, quality, clipping=clipping,
                        q_strand=0, t_strand=hit["strand"], is_primary=is_primary, read_length=L,
                        subread_start=subread_start, subread_end=L if subread_end is None else subread_end,
                        q_align_start=hit["q_align_start"], q_align_end=hit["q_align_end"],
                        t_align_start=hit["t_start"], t_end=hit["t_span"], t_length=hit["t_length"], q_length=L,
                        q_aligned_seq_pos=hit["q_aligned_seq_pos"], q_aligned_seq_length=len(hit["q"]),
                        map_qv=hit["map_qv"], score=hit["score"], edit_dist=0, hole_number=hole_number,
                        # the mapped subread's qs / qe (SMRTSequence.h:102-103), printed under -clipping soft / hard / subread
                        qs=subread_start + 1, qe=(L if subread_end is None else subread_end) + 1)
