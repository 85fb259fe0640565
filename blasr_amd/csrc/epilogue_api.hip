// epilogue_api.hip -- the first pieces of the output epilogue: what MapRead and the
// printers do with a refined alignment AFTER the hot path.  Host code, as in the reference; no GPU work.
//
//   blasr_hip_ref_contig_location   AssignRefContigLocation             alignment/Blasr.cpp:2183-2220
//   blasr_hip_cigar                 CreateCIGARString and its helpers   printers/SAMPrinter.h:122-416
//   blasr_hip_m4_line               IntervalAlignmentPrinter::Print     printers/IntervalAlignmentPrinter.h:9-38
//   blasr_hip_map_qvs               StoreMapQVs and its helpers         alignment/Blasr.cpp:2647-2950
//   blasr_hip_sam_line              SAMOutput::PrintAlignment           printers/SAMPrinter.h:29-105, 418-548
//   blasr_hip_order_gaps_by_type    Alignment::OrderGapsByType          datastructures/alignment/Alignment.h:255-317
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <set>
#include <string>
#include <vector>

#include "../../include/blasr_hip.h"
#include "abi_guard.h"

namespace {

struct CigarOps {
  std::vector<int> size;
  std::vector<char> op;
  void push(int n, char c) {
    size.push_back(n);
    op.push_back(c);
  }
};

// AddUngappedOperations / AddUnmatchedOperations (SAMPrinter.h:138-201): runs of differing bytes are X, of equal
// bytes =; the comparison is on the raw characters, case included
void match_runs(const uint8_t *q, const uint8_t *t, int len, CigarOps &out) {
  int i = 0;
  while (i < len) {
    int start = i;
    while (i < len && q[i] != t[i]) i++;
    if (i > start) out.push(i - start, 'X');
    start = i;
    while (i < len && q[i] == t[i]) i++;
    if (i > start) out.push(i - start, '=');
  }
}

}  // namespace

extern "C" {

int blasr_hip_ref_contig_location(uint32_t t_aligned_seq_pos, uint32_t t_aligned_seq_length, int32_t t_strand,
                                  const uint32_t *seq_start_pos, int32_t n_seq_pos, uint32_t genome_length,
                                  int32_t *contig, uint32_t *contig_pos) try {
  if (!seq_start_pos || n_seq_pos < 1 || !contig || !contig_pos) return BLASR_HIP_ERR_INVALID;
  // SequenceIndexDatabase::SearchForIndex (SequenceIndexDatabase.h:108-118)
  auto search = [&](uint32_t pos) -> int {
    if (n_seq_pos == 1) return 0;
    const uint32_t *it = std::upper_bound(seq_start_pos + 1, seq_start_pos + n_seq_pos, pos);
    return (int)(it - seq_start_pos) - 1;
  };
  if (t_strand == 0) {
    const int idx = search(t_aligned_seq_pos);
    *contig = idx;
    *contig_pos = t_aligned_seq_pos - seq_start_pos[idx];
  } else {
    if (t_aligned_seq_length == 0) return BLASR_HIP_ERR_INVALID;  // the reference asserts
    // the forward coordinate of the last aligned base finds the contig; its last base (the start of the next
    // contig - 2: an N separates contigs) flipped back gives the offset (Blasr.cpp:2197-2218)
    const uint32_t forward = genome_length - (t_aligned_seq_pos + t_aligned_seq_length - 1) - 1;
    const int idx = search(forward);
    if (idx + 1 >= n_seq_pos) return BLASR_HIP_ERR_INVALID;
    const uint32_t reverseOffset = genome_length - (seq_start_pos[idx + 1] - 2) - 1;
    *contig = idx;
    *contig_pos = t_aligned_seq_pos - reverseOffset;
  }
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_cigar(const uint8_t *q_aligned_seq, const uint8_t *t_aligned_seq, int32_t q_pos, int32_t t_pos,
                    const uint32_t *ops, int64_t n_ops, const int32_t *blocks, int64_t n_blocks,
                    const blasr_hip_cigar_params *p, char *cigar, int64_t cigar_cap, int64_t *cigar_len,
                    uint32_t clips[4]) try {
  if (!q_aligned_seq || !t_aligned_seq || !p || !cigar_len || (!ops && !blocks)) return BLASR_HIP_ERR_INVALID;
  if ((ops && n_ops <= 0) || (!ops && n_blocks <= 0)) return BLASR_HIP_ERR_INVALID;  // the printers skip empty ones
  uint32_t prefixSoft = 0, suffixSoft = 0, prefixHard = 0, suffixHard = 0;
  if (p->clipping == BLASR_HIP_CLIP_HARD) {  // SetHardClip :320-335
    prefixHard = p->q_align_start;
    suffixHard = p->read_length - p->q_align_end;
    if (p->t_strand == 1) std::swap(prefixHard, suffixHard);
  }
  if (p->clipping == BLASR_HIP_CLIP_SOFT || p->clipping == BLASR_HIP_CLIP_SUBREAD) {  // :366-382
    if (p->clipping == BLASR_HIP_CLIP_SOFT) {
      prefixHard = p->low_quality_prefix;
      suffixHard = p->low_quality_suffix;
    } else {
      prefixHard = std::max(p->subread_start, p->low_quality_prefix);
      suffixHard = std::max(p->read_length - p->subread_end, p->low_quality_suffix);
    }
    // SetSoftClip :300-313 (blocks[0].qPos is 0 once the prefix gaps are folded into qPos)
    const uint32_t first = blocks ? (uint32_t)blocks[0] : 0u;
    prefixSoft = p->q_align_start + first - prefixHard;
    suffixSoft = p->read_length - suffixHard - p->q_align_end;
  }
  CigarOps c;
  if (prefixHard > 0) c.push((int)prefixHard, 'H');
  if (prefixSoft > 0) c.push((int)prefixSoft, 'S');
  int qp = q_pos, tp = t_pos;
  if (ops) {
    // CreateNoClippingCigarOps with gap lists (:289-294): a block, then the gaps behind it, in path order
    for (int64_t i = 0; i < n_ops; i++) {
      const int len = (int)(ops[i] >> 2), ty = (int)(ops[i] & 3u);
      if (ty == 0) {
        match_runs(q_aligned_seq + qp, t_aligned_seq + tp, len, c);
        qp += len;
        tp += len;
      } else if (ty == 2) {  // Gap::Query: bases of the target the query lacks
        c.push(len, 'D');
        tp += len;
      } else {               // Gap::Target
        c.push(len, 'I');
        qp += len;
      }
    }
  } else {
    // ... without gap lists (:228-287): the gaps are what lies between consecutive blocks; a stretch unaligned
    // on both sides is an insertion or deletion for the difference and mismatches for the rest
    qp = q_pos + blocks[0];
    tp = t_pos + blocks[1];
    for (int64_t b = 0; b < n_blocks; b++) {
      const int len = blocks[3 * b + 2];
      int qGap = 0, tGap = 0;
      if (b + 1 < n_blocks) {
        qGap = blocks[3 * (b + 1)] - blocks[3 * b] - len;
        tGap = blocks[3 * (b + 1) + 1] - blocks[3 * b + 1] - len;
      }
      match_runs(q_aligned_seq + qp, t_aligned_seq + tp, len, c);
      qp += len;
      tp += len;
      int unmatched = 0;
      if (qGap > 0 && tGap > 0) {
        const int common = std::min(qGap, tGap);
        qGap -= common;
        tGap -= common;
        unmatched = common;
      }
      if (qGap > 0) {
        c.push(qGap, 'I');
        qp += qGap;
      }
      if (tGap > 0) {
        c.push(tGap, 'D');
        tp += tGap;
      }
      match_runs(q_aligned_seq + qp, t_aligned_seq + tp, unmatched, c);
      qp += unmatched;
      tp += unmatched;
    }
  }
  if (suffixSoft > 0) c.push((int)suffixSoft, 'S');
  if (suffixHard > 0) c.push((int)suffixHard, 'H');
  if (p->t_strand == 1) {
    std::reverse(c.size.begin(), c.size.end());
    std::reverse(c.op.begin(), c.op.end());
  }
  std::string s;
  char buf[24];
  for (size_t i = 0; i < c.size.size(); i++) {
    snprintf(buf, sizeof buf, "%d%c", c.size[i], c.op[i]);
    s += buf;
  }
  if (clips) {
    clips[0] = prefixSoft;
    clips[1] = suffixSoft;
    clips[2] = prefixHard;
    clips[3] = suffixHard;
  }
  *cigar_len = (int64_t)s.size();
  if (!cigar || (int64_t)s.size() + 1 > cigar_cap) return BLASR_HIP_ERR_CAPACITY;
  memcpy(cigar, s.c_str(), s.size() + 1);
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_alignment_is_printed(int32_t score, float pct_similarity, uint32_t t_aligned_seq_length, int32_t map_qv,
                                   const blasr_hip_print_params *p) try {
  if (!p) return BLASR_HIP_ERR_INVALID;
  if (score > p->max_score) return 0;                                // :2958
  if (pct_similarity < p->min_pct_identity) return 0;                // :2964 (float against float)
  if (t_aligned_seq_length < (uint32_t)p->min_align_length) return 0;  // :2970 (DNALength against int: unsigned)
  if (map_qv < p->min_map_qv) return 0;                              // :2976
  return 1;
} BLASR_HIP_CATCH_CODE

int blasr_hip_m4_line(const blasr_hip_m4_fields *f, char *line, int64_t line_cap, int64_t *line_len) try {
  if (!f || !f->q_name || !f->t_name || !line_len) return BLASR_HIP_ERR_INVALID;
  // ostream << float prints like %g with six significant digits
  char buf[96];
  std::string s = std::string(f->q_name) + " " + f->t_name + " ";
  snprintf(buf, sizeof buf, "%d %g %d %u %u %u %d %u %u %u %d\n", f->score, (double)f->pct_similarity, f->q_strand,
           f->q_align_start, f->q_align_end, f->q_length, f->t_strand, f->t_align_start, f->t_align_end, f->t_length,
           f->map_qv);
  s += buf;
  *line_len = (int64_t)s.size();
  if (!line || (int64_t)s.size() + 1 > line_cap) return BLASR_HIP_ERR_CAPACITY;
  memcpy(line, s.c_str(), s.size() + 1);
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

}  // extern "C"

namespace {

int three_bit(uint8_t c) {  // ThreeBit (NucConversion.h): A C G T in either case, everything else 4
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
  }
}

// LogSumOfTwo (utils/LogUtils.h:11-41), float intermediates and all
double log_sum_of_two(double value1, double value2) {
  const double kLog10 = 2.3025850929;
  double minValue = value1, maxValue = value2;
  if (maxValue < minValue) {
    minValue = value2;
    maxValue = value1;
  }
  minValue *= kLog10;
  maxValue *= kLog10;
  const double difference = minValue - maxValue;
  if (difference < -30) return maxValue / kLog10;
  if (difference < -8) return (maxValue + exp(difference)) / kLog10;
  const float expv = exp(difference);
  const float log1pv = log1p(expv);
  return (maxValue + log1pv) / kLog10;
}

struct QInterval {
  int start, end;
};
// AlignmentCandidate::GetQInterval (AlignmentCandidate.h:162-174)
QInterval q_interval(const blasr_hip_mapqv_alignment &a, bool forwardStrand) {
  QInterval r;
  r.start = (int)a.q_align_start;
  r.end = (int)a.q_align_end;
  if (forwardStrand && a.q_strand == 1) {
    const int fs = (int)a.q_length - r.end, fe = (int)a.q_length - r.start;
    r.start = fs;
    r.end = fe;
  }
  return r;
}

// AlignmentsOverlap (Blasr.cpp:2679-2722)
bool alignments_overlap(const blasr_hip_mapqv_alignment &a, const blasr_hip_mapqv_alignment &b, float minPercentOverlap) {
  const QInterval A = q_interval(a, true), B = q_interval(b, true);
  int ovp = 0;
  if (A.start <= B.start && A.end >= B.end) return true;
  if (B.start <= A.start && B.end >= A.end) return true;
  if (A.end >= B.start && A.end <= B.end) ovp = A.end - B.start;
  else if (A.start >= B.start && A.start <= B.end) ovp = B.end - A.start;
  float ovpPercent;
  if (A.end - A.start > 0 && B.end - B.start > 0)
    ovpPercent = std::max(float(ovp) / float(A.end - A.start), float(ovp) / float(B.end - B.start));
  else
    ovpPercent = 0;
  return ovpPercent > minPercentOverlap;
}

}  // namespace

extern "C" int blasr_hip_map_qvs(const blasr_hip_mapqv_alignment *alns, int32_t n, const blasr_hip_mapqv_params *p,
                                 const uint8_t *substitution_qv, int32_t *map_qv, float *prob_score_out) {
  if (n < 0 || (n > 0 && (!alns || !map_qv)) || !p) return BLASR_HIP_ERR_INVALID;
  const int kMaxPhred = 254, kWiggle = 5;  // MAX_PHRED_SCORE, MAPQV_END_ALIGN_WIGGLE (Blasr.cpp:79-80)
  std::vector<float> probScore((size_t)n);
  // ---- rescoring with the log-probability matrix (:2777-2785, ComputeAlignmentScore AlignmentUtils.h:127-170)
  for (int a = 0; a < n; a++) {
    const blasr_hip_mapqv_alignment &al = alns[a];
    if (!al.q_aligned_seq || !al.t_aligned_seq || !al.ops || al.n_ops <= 0) return BLASR_HIP_ERR_INVALID;
    int score = 0, q = al.q_pos, t = al.t_pos;
    for (int64_t i = 0; i < al.n_ops; i++) {
      const int len = (int)(al.ops[i] >> 2), ty = (int)(al.ops[i] & 3u);
      if (ty == 0) {
        for (int l = 0; l < len; l++) score += p->matrix[three_bit(al.t_aligned_seq[t + l]) * 5 + three_bit(al.q_aligned_seq[q + l])];
        q += len;
        t += len;
      } else {
        const int unit = ty == 1 ? p->ins : p->del;
        if (p->affine_align && p->affine_open + len * p->affine_extend < len * unit) score += p->affine_open + len * p->affine_extend;
        else score += len * unit;
        if (ty == 1) q += len; else t += len;
      }
    }
    probScore[a] = -score / 10.0;
  }
  // ---- PartitionOverlappingAlignments (:2726-2759)
  std::vector<std::set<int> > partitions;
  for (int i = 0; i < n; i++) {
    bool found = false;
    for (size_t pi = 0; pi < partitions.size() && !found; pi++)
      for (std::set<int>::iterator it = partitions[pi].begin(); it != partitions[pi].end() && !found; ++it)
        if (alignments_overlap(alns[i], alns[*it], p->min_fraction_overlapping) ||
            (alns[i].q_align_start <= alns[*it].q_align_start && alns[i].q_align_end > alns[*it].q_align_end)) {
          partitions[pi].insert(i);
          found = true;
        }
    if (!found) {
      partitions.push_back(std::set<int>());
      partitions.back().insert(i);
    }
  }
  // ---- the widest alignment of every partition (:2797-2826)
  std::vector<int> begin(partitions.size(), -1), end(partitions.size(), -1), pscore(partitions.size(), 0);
  for (size_t pi = 0; pi < partitions.size(); pi++) {
    std::set<int>::iterator it = partitions[pi].begin();
    QInterval iv = q_interval(alns[*it], false);
    begin[pi] = iv.start;
    end[pi] = iv.end;
    pscore[pi] = alns[*it].n_match;
    for (++it; it != partitions[pi].end(); ++it) {
      iv = q_interval(alns[*it], false);
      if (iv.end - iv.start > end[pi] - begin[pi] && alns[*it].n_match * 1.20 >= pscore[pi]) {
        begin[pi] = iv.start;
        end[pi] = iv.end;
        pscore[pi] = alns[*it].n_match;
      }
    }
  }
  // ---- what an alignment would lose if it were extended to the partition's ends (:2851-2873, SumMismatches :2647-2671)
  for (size_t pi = 0; pi < partitions.size(); pi++)
    for (std::set<int>::iterator it = partitions[pi].begin(); it != partitions[pi].end(); ++it) {
      const QInterval iv = q_interval(alns[*it], true);
      int mismatchSum = 0;
      if (iv.start - begin[pi] > kWiggle || end[pi] - iv.end > kWiggle) {
        if (substitution_qv) {
          for (int x = begin[pi]; x < iv.start; x++) mismatchSum += substitution_qv[x];
          for (int x = iv.end; x < end[pi]; x++) mismatchSum += substitution_qv[x];
        } else {
          mismatchSum += 15 * ((iv.start - begin[pi]) + (end[pi] - iv.end));
        }
      }
      probScore[*it] += -(mismatchSum) * 0.5;
    }
  // ---- mapQV from the scores of a partition (:2878-2944)
  for (size_t pi = 0; pi < partitions.size(); pi++) {
    std::set<int>::iterator it = partitions[pi].begin();
    float denominator = probScore[*it];
    int maxNMatch = alns[*it].n_match;
    for (++it; it != partitions[pi].end(); ++it) {
      if (alns[*it].n_match > maxNMatch) maxNMatch = alns[*it].n_match;
      if (alns[*it].n_match * 1.2 >= maxNMatch) denominator = log_sum_of_two(denominator, probScore[*it]);
    }
    for (it = partitions[pi].begin(); it != partitions[pi].end(); ++it) {
      if (partitions[pi].size() == 1) {
        map_qv[*it] = kMaxPhred;
      } else if (probScore[*it] - denominator < -20) {
        map_qv[*it] = 0;
      } else {
        const double log10v = log(10);
        const double sub = probScore[*it] - denominator;
        const double expo = exp(log10v * sub);
        const double diff = 1.0 - expo;
        int phred;
        if (expo == 0) phred = 0;
        else if (diff == 0) phred = kMaxPhred;
        else {  // Phred (utils/PhredUtils.h:10-13), truncated.  diff can come out negative (exp rounds above 1): the
          // reference then converts a NaN to int -- what x86's cvttsd2si gives for it (INT_MIN), spelled out
          const double v = -10 * log10(diff);
          phred = (v != v || v >= 2147483648.0 || v < -2147483648.0) ? INT_MIN : (int)v;
        }
        if (phred > kMaxPhred) phred = kMaxPhred;
        if (phred < 0) phred = 0;
        map_qv[*it] = phred;
      }
    }
  }
  if (prob_score_out)
    for (int a = 0; a < n; a++) prob_score_out[a] = probScore[a];
  return BLASR_HIP_OK;
}

namespace {
uint8_t rc_nuc_host(uint8_t c) {  // ReverseComplementNuc (NucConversion.h:337-370)
  switch (c) {
    case 0: return 3;
    case 1: return 2;
    case 2: return 1;
    case 3: return 0;
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    case 'N': return 'N';
    case 'a': return 't';
    case 'c': return 'g';
    case 'g': return 'c';
    case 't': return 'a';
    case 'n': return 'n';
    default: return 127;
  }
}
}  // namespace

extern "C" int blasr_hip_sam_line(const blasr_hip_sam_fields *f, char *line, int64_t line_cap, int64_t *line_len) {
  if (!f || !f->q_name || !f->t_name || !f->read_group_id || !f->cigar || !f->read_seq || !line_len)
    return BLASR_HIP_ERR_INVALID;
  // SetAlignedSequence (:58-105): which part of the read is printed
  uint32_t start = 0, length = 0;
  if (f->clipping == BLASR_HIP_CLIP_NONE || f->clipping == BLASR_HIP_CLIP_HARD) {
    start = f->q_align_start;
    length = f->q_align_end - f->q_align_start;
  } else if (f->clipping == BLASR_HIP_CLIP_SOFT) {
    length = f->read_length - f->low_quality_prefix - f->low_quality_suffix;
    start = f->low_quality_prefix;
  } else if (f->clipping == BLASR_HIP_CLIP_SUBREAD) {
    length = f->subread_end - f->subread_start;
    start = f->subread_start;
  } else {
    return BLASR_HIP_ERR_INVALID;
  }
  if ((uint64_t)start + length > f->read_length) return BLASR_HIP_ERR_INVALID;
  std::string seq((size_t)length, ' '), qual;
  if (f->t_strand == 0) {
    for (uint32_t i = 0; i < length; i++) seq[i] = (char)f->read_seq[start + i];
  } else {  // MakeRC: reverse complement, qualities reversed
    for (uint32_t i = 0; i < length; i++) seq[i] = (char)rc_nuc_host(f->read_seq[start + length - 1 - i]);
  }
  if (f->read_qual) {
    qual.resize(length);
    for (uint32_t i = 0; i < length; i++)
      qual[i] = (char)(f->read_qual[f->t_strand == 0 ? start + i : start + length - 1 - i] + 33);  // charToQuality
  } else {
    qual = "*";
  }
  unsigned flag = 0;  // BuildFlag (:29-50)
  if (f->t_strand != f->q_strand) flag |= 0x10;  // SEQ_REVERSED
  if (!f->is_primary) flag |= 0x100;             // SECONDARY_ALIGNMENT
  const uint32_t pos = f->t_strand == 0 ? f->t_align_start + 1 : f->t_length - (f->t_align_start + f->t_end) + 1;
  char buf[256];
  std::string s = std::string(f->q_name) + "\t";
  snprintf(buf, sizeof buf, "%u\t", flag);
  s += buf;
  s += std::string(f->t_name) + "\t";
  snprintf(buf, sizeof buf, "%u\t%d\t", pos, f->map_qv);
  s += buf;
  s += std::string(f->cigar) + "\t*\t0\t";
  snprintf(buf, sizeof buf, "%u\t", f->t_end);  // TLEN = GenomicTEnd() - GenomicTBegin()
  s += buf;
  s += seq + "\t" + qual + "\t";
  s += "RG:Z:" + std::string(f->read_group_id) + "\t";
  snprintf(buf, sizeof buf, "AS:i:%d\t", f->score);
  s += buf;
  if (f->clipping == BLASR_HIP_CLIP_NONE) {
    snprintf(buf, sizeof buf, "XS:i:%u\tXE:i:%u\t", f->q_align_start + 1, f->q_align_end + 1);
    s += buf;
  } else {
    uint32_t xs = f->clips[2], xe = f->read_length - f->clips[3];
    if (f->t_strand == 1) {
      xs = f->clips[3];
      xe = f->read_length - f->clips[2];
    }
    snprintf(buf, sizeof buf, "XS:i:%u\tXE:i:%u\tqs:i:%d\tqe:i:%d\t", xs + 1, xe + 1, f->qs, f->qe);
    s += buf;
  }
  snprintf(buf, sizeof buf, "zm:i:%d\tXL:i:%u\tXT:i:1\tNM:i:%d\tFI:i:%u\tXQ:i:%u\t", f->hole_number, f->q_aligned_seq_length,
           f->edit_dist, f->q_aligned_seq_pos + 1, f->q_length);
  s += buf;
  // samFile.precision(4) from here on
  snprintf(buf, sizeof buf, "rq:f:%.4g\tnp:i:%d\tcx:i:%d\tsn:B:f,%.4g,%.4g,%.4g,%.4g\n", (double)f->read_quality, f->num_passes,
           f->subread_context, (double)f->snr[0], (double)f->snr[1], (double)f->snr[2], (double)f->snr[3]);
  s += buf;
  *line_len = (int64_t)s.size();
  if (!line || (int64_t)s.size() + 1 > line_cap) return BLASR_HIP_ERR_CAPACITY;
  memcpy(line, s.c_str(), s.size() + 1);
  return BLASR_HIP_OK;
}

// SAMOutput::PrintUnalignedRead (printers/SAMPrinter.h:578-664): the record -sam prints for a (sub)read without an
// alignment (Blasr.cpp:3848-3866).  Uses q_name, read_group_id, read_seq, read_length, subread_start / subread_end,
// clipping, qs, qe, hole_number, read_quality, num_passes, subread_context, snr of the fields; *line_len = 0 where the
// reference prints nothing (subread out of bounds or empty, :591-600).
extern "C" int blasr_hip_sam_unaligned_line(const blasr_hip_sam_fields *f, char *line, int64_t line_cap, int64_t *line_len) {
  if (!f || !f->q_name || !f->read_group_id || !f->read_seq || !line_len) return BLASR_HIP_ERR_INVALID;
  *line_len = 0;
  uint32_t length = f->subread_end - f->subread_start;
  const uint32_t start = f->subread_start;
  if (start <= f->read_length && length == 0) length = f->read_length - start;  // ReferenceSubstring(read, pos, 0): to the end
  if (!(start <= f->read_length && length <= f->read_length) || length == 0) {
    if (line && line_cap > 0) line[0] = 0;
    return BLASR_HIP_OK;
  }
  if ((uint64_t)start + length > f->read_length) return BLASR_HIP_ERR_INVALID;  // the reference would read past the read
  char buf[256];
  std::string s = std::string(f->q_name) + "\t4\t*\t0\t0\t*\t*\t0\t0\t";  // SEGMENT_UNMAPPED, fixed defaults (:603)
  s.append((const char *)f->read_seq + start, length);
  s += "\t*\tRG:Z:" + std::string(f->read_group_id) + "\t";
  if (f->clipping == BLASR_HIP_CLIP_SUBREAD)
    snprintf(buf, sizeof buf, "XS:i:%u\tXE:i:%u\tqs:i:%d\tqe:i:%d\t", f->subread_start + 1, f->subread_end + 1, f->qs, f->qe);
  else
    snprintf(buf, sizeof buf, "XS:i:1\tXE:i:%u\tqs:i:1\tqe:i:%u\t", length, length);
  s += buf;
  snprintf(buf, sizeof buf, "zm:i:%d\tXL:i:0\trq:f:%.4g\tnp:i:%d\tcx:i:%d\tsn:B:f,%.4g,%.4g,%.4g,%.4g\tXT:i:1\tXQ:i:%u\n",
           f->hole_number, (double)f->read_quality, f->num_passes, f->subread_context, (double)f->snr[0], (double)f->snr[1],
           (double)f->snr[2], (double)f->snr[3], length);
  s += buf;
  *line_len = (int64_t)s.size();
  if (!line || (int64_t)s.size() + 1 > line_cap) return BLASR_HIP_ERR_CAPACITY;
  memcpy(line, s.c_str(), s.size() + 1);
  return BLASR_HIP_OK;
}

// -forPicard (Blasr.cpp:2621-2626): samtools / picard reject insertion-deletion-insertion patterns, so the gaps
// between two blocks are condensed to ONE gap of the longer kind, shortened by the other kind's bases, which become
// aligned columns of the block before (Alignment.h:255-317).  The blocks stay separate blocks.
extern "C" int blasr_hip_order_gaps_by_type(const uint32_t *ops, int64_t n_ops, uint32_t *out, int64_t out_cap,
                                            int64_t *n_out) {
  if (!ops || n_ops <= 0 || !n_out) return BLASR_HIP_ERR_INVALID;
  std::vector<uint32_t> r;
  int64_t i = 0;
  while (i < n_ops) {
    if ((ops[i] & 3u) == 0u) {
      r.push_back(ops[i]);
      i++;
      continue;
    }
    int64_t j = i;
    uint32_t target = 0, query = 0;  // Gap::Target = insertion (op 1), Gap::Query = deletion (op 2)
    while (j < n_ops && (ops[j] & 3u) != 0u) {
      if ((ops[j] & 3u) == 1u) target += ops[j] >> 2; else query += ops[j] >> 2;
      j++;
    }
    if (j - i <= 1 || r.empty()) {  // a single gap stays as it is
      for (int64_t k = i; k < j; k++) r.push_back(ops[k]);
    } else {
      uint32_t matchExtend;
      uint32_t gap = 0;
      if (target > query) {
        gap = ((target - query) << 2) | 1u;
        matchExtend = query;
      } else if (query > target) {
        gap = ((query - target) << 2) | 2u;
        matchExtend = target;
      } else {
        matchExtend = target;
      }
      r.back() += matchExtend << 2;  // blocks[g-1].length += matchExtend
      if (gap) r.push_back(gap);
    }
    i = j;
  }
  *n_out = (int64_t)r.size();
  if (!out || (int64_t)r.size() > out_cap) return BLASR_HIP_ERR_CAPACITY;
  memcpy(out, r.data(), sizeof(uint32_t) * r.size());
  return BLASR_HIP_OK;
}
