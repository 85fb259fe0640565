// epilogue_api.hip -- the first pieces of the output epilogue: what MapRead and the
// printers do with a refined alignment AFTER the hot path.  Host code, as in the reference; no GPU work.
//
//   blasr_hip_ref_contig_location   AssignRefContigLocation             alignment/Blasr.cpp:2183-2220
//   blasr_hip_cigar                 CreateCIGARString and its helpers   printers/SAMPrinter.h:122-416
//   blasr_hip_m4_line               IntervalAlignmentPrinter::Print     printers/IntervalAlignmentPrinter.h:9-38
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/blasr_hip.h"

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
                                  int32_t *contig, uint32_t *contig_pos) {
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
}

int blasr_hip_cigar(const uint8_t *q_aligned_seq, const uint8_t *t_aligned_seq, int32_t q_pos, int32_t t_pos,
                    const uint32_t *ops, int64_t n_ops, const int32_t *blocks, int64_t n_blocks,
                    const blasr_hip_cigar_params *p, char *cigar, int64_t cigar_cap, int64_t *cigar_len,
                    uint32_t clips[4]) {
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
}

int blasr_hip_m4_line(const blasr_hip_m4_fields *f, char *line, int64_t line_cap, int64_t *line_len) {
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
}

}  // extern "C"
