// extend_api.hip -- next-3, first piece: ExtendAlignmentForward / ExtendAlignmentReverse
// (algorithms/alignment/ExtendAlign.h:115-445), the -extend step AlignIntervals runs on both ends of every SDP
// alignment (alignment/Blasr.cpp:1749-1940).
//
// Band of 2k+1 diagonals around the main one (band column = k + 1 - q + t, ExtendAlign.h:13-41), linear gaps,
// one row per query base until maxNDrops rows have raised the row minimum or the shorter sequence (+ maxNDrops)
// is used up; the alignment ends at the first cell that holds the global minimum.
//   extend_plan_kernel   the T_Index fields (:365-445), the row / column limits (:228-250), workspace sizes
//   extend_kernel        one LANE per extension: a task is at most ~500 rows of 2k+1 cells and its length is
//                        data dependent (the drop-off rule), a mapped batch holds two tasks per alignment, so the
//                        parallelism is across tasks.  Two rows of scores per lane in LDS ([column][lane]: no
//                        bank conflicts), two arrow bits per cell packed into one or two 64-bit words per row in
//                        HBM, then the traceback of :339-374 by the same lane, run-length encoded into the op
//                        arena shared with the refiners.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/blasr_hip.h"
#include "ctx.h"
#include "refine_kernels.h"
#include "wave_utils.h"

namespace blasr_hip {

constexpr int kExtMaxCols = 64;  // nCols = 2k + 2 <= 64: k <= 31 (params.extendBandSize is 10)
enum { EX_DIAG = 0, EX_UP = 1, EX_LEFT = 2 };

struct ExtendGeo {
  int32_t idxQ, idxT;            // T_Index::queryPos / refPos
  int32_t qAlignLen, tAlignLen;  // T_Index::queryAlignLength / refAlignLength
  int32_t maxRows, maxT;         // last row / column the loops of :230-252 can reach
  int32_t status;
  int32_t early;                 // returns 0 before any work (:150-157)
  int64_t pathWords;             // 64-bit arrow words
  int64_t arenaWords;            // worst-case op count
};

__global__ __launch_bounds__(256) void extend_plan_kernel(int n, const int32_t *qLen, const int32_t *tLen,
                                                          const int32_t *qPos, const int32_t *tPos,
                                                          const uint8_t *reverse, int k, int minExtendNBases,
                                                          int maxNDrops, ExtendGeo *geo) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ExtendGeo g;
  memset(&g, 0, sizeof(g));
  const int64_t ql = qLen[i], tl = tLen[i], qp = qPos[i], tp = tPos[i];
  const bool rev = reverse[i] != 0;
  if (ql < 0 || tl < 0 || qp < 0 || tp < 0 || (rev && (qp > ql || tp > tl))) {
    g.status = TASK_BAD_GUIDE;  // the reference would read outside the sequences
    geo[i] = g;
    return;
  }
  int64_t QL, TL;
  if (rev) {  // ExtendAlignmentReverse :430-434
    g.idxQ = (int32_t)(qp - 1);
    g.idxT = (int32_t)(tp - 1);
    QL = qp;
    TL = tp;
  } else {  // ExtendAlignmentForward :386-393
    g.idxQ = (int32_t)qp;
    g.idxT = (int32_t)tp;
    QL = ql - qp;
    TL = tl - tp;
  }
  g.qAlignLen = (int32_t)QL;
  g.tAlignLen = (int32_t)TL;
  if (QL < minExtendNBases || TL < minExtendNBases) {
    g.early = 1;
    geo[i] = g;
    return;
  }
  const int64_t maxAlignLength = min(QL, TL) + maxNDrops;  // :228
  const int64_t maxRows = maxNDrops > 0 ? max((int64_t)0, min(QL, maxAlignLength - 1)) : 0;
  const int64_t maxT = max((int64_t)0, min(TL, maxAlignLength - 1));
  g.maxRows = (int32_t)maxRows;
  g.maxT = (int32_t)maxT;
  const int nW = (2 * k + 2) > 32 ? 2 : 1;
  g.pathWords = maxRows * nW;
  g.arenaWords = maxRows + maxT + 4;
  geo[i] = g;
}

struct ExtendArgs {
  int first, count;
  int k, ins, del, maxNDrops;
  const int32_t *matrix;
  const ExtendGeo *geo;
  const uint8_t *seq;
  const int64_t *qOff, *tOff;
  const int32_t *qPos, *tPos;
  const uint8_t *reverse;
  unsigned long long *path;
  const int64_t *pathOff;
  uint32_t *arena;
  const int64_t *arenaOff;
  int32_t *opsStart;
  RefineResult *results;
};

__global__ __launch_bounds__(64) void extend_kernel(ExtendArgs a) {
  __shared__ int rowImg[2][kExtMaxCols][64];
  __shared__ int m[25];
  if (threadIdx.x < 25) m[threadIdx.x] = a.matrix[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x;
  const int idx = blockIdx.x * 64 + lane;
  if (idx >= a.count) return;
  const int task = a.first + idx;
  const ExtendGeo g = a.geo[task];
  const bool rev = a.reverse[task] != 0;
  RefineResult res;
  memset(&res, 0, sizeof(res));
  if (g.status != 0) {
    res.status = g.status;
    a.results[task] = res;
    a.opsStart[task] = 0;
    return;
  }
  if (g.early) {  // empty alignment, return value 0; Forward still sets qPos / tPos (:401-402)
    if (!rev) {
      res.q_pos = a.qPos[task];
      res.t_pos = a.tPos[task];
    }
    a.results[task] = res;
    a.opsStart[task] = 0;
    return;
  }
  const uint8_t *qSeq = a.seq + a.qOff[task], *tSeq = a.seq + a.tOff[task];
  unsigned long long *path = a.path + a.pathOff[task];
  uint32_t *arena = a.arena + a.arenaOff[task];
  const int k = a.k, nCols = 2 * k + 2, ins = a.ins, del = a.del, QL = g.qAlignLen, TL = g.tAlignLen;
  const int nW = nCols > 32 ? 2 : 1;
  const int step = rev ? -1 : 1;
  // row 0 (:201-212): S[0][t] = t * del for t <= min(k, TL), band column k + 1 + t
  for (int c = 0; c < nCols; c++) {
    const int t = c - (k + 1);
    rowImg[0][c][lane] = (t >= 1 && t <= k && t <= TL) ? t * del : 0;
  }
  int nDrops = 0, prevRowMin = INT_MAX, globalMin = INT_MAX, gq = 0, gt = 0, nCells = 0;
  const int maxAlignLength = min(QL, TL) + a.maxNDrops;  // :228
  int q = 1;
  for (; q - 1 != QL && nDrops < a.maxNDrops && q < maxAlignLength; q++) {  // :230-233
    const int pb = (q - 1) & 1, cb = q & 1;
    const int base = k + 1 - q;  // band column of t = 0
    int left = 0;
    if (q <= k) {  // S[q][0] = q * ins (:186-195)
      left = q * ins;
      rowImg[cb][base][lane] = left;
    }
    const int tStart = max(1, q - k);
    const int tEnd = min(min(q + k + 1, TL + 1), maxAlignLength);
    const int qc = base_code((unsigned)qSeq[g.idxQ + step * (q - 1)]);
    int curRowMin = INT_MAX;
    unsigned long long w0 = 0, w1 = 0;
    for (int t = tStart; t < tEnd; t++) {  // :252-330
      const int c = base + t;
      const int tc = base_code((unsigned)tSeq[g.idxT + step * (t - 1)]);
      const int matchScore = rowImg[pb][c][lane] + m[tc * 5 + qc];                      // (q-1, t-1): always in the band
      const int insScore = c + 1 <= 2 * k + 1 ? rowImg[pb][c + 1][lane] + ins : INT_MAX;  // (q-1, t)
      const int delScore = c - 1 >= 1 ? left + del : INT_MAX;                             // (q, t-1)
      const int minScore = min(min(matchScore, delScore), insScore);
      int arrow = EX_UP;  // :306-310: the last test that holds wins
      if (minScore == delScore) arrow = EX_LEFT;
      if (minScore == matchScore) arrow = EX_DIAG;
      rowImg[cb][c][lane] = minScore;
      left = minScore;
      if (c < 32) w0 |= (unsigned long long)arrow << (2 * c);
      else w1 |= (unsigned long long)arrow << (2 * (c - 32));
      curRowMin = min(curRowMin, minScore);
      if (minScore < globalMin) {
        globalMin = minScore;
        gq = q;
        gt = t;
      }
    }
    nCells += max(0, tEnd - tStart);
    path[(int64_t)(q - 1) * nW] = w0;
    if (nW == 2) path[(int64_t)(q - 1) * nW + 1] = w1;
    if (curRowMin > prevRowMin) nDrops++;  // :333-336
    prevRowMin = curRowMin;
  }
  // ---- traceback from the global minimum (:339-374).  Forward: the walk runs against the alignment, ops go in
  // from the end of the arena; Reverse: the walk already is in sequence order (ReverseIndex::OrderArrowVector).
  int pos = rev ? 0 : (int)g.arenaWords, curType = -1, curLen = 0;
  auto flush = [&]() {
    if (!curLen) return;
    const uint32_t op = ((uint32_t)curLen << 2) | (uint32_t)curType;
    if (rev) arena[pos++] = op;
    else arena[--pos] = op;
  };
  q = gq;
  int t = gt;
  while (q > 0 || t > 0) {
    int arrow;
    if (q == 0) arrow = EX_LEFT;
    else if (t == 0) arrow = EX_UP;
    else {
      const int c = k + 1 - q + t;
      arrow = (int)((path[(int64_t)(q - 1) * nW + (c >> 5)] >> (2 * (c & 31))) & 3ull);
    }
    const int type = arrow == EX_DIAG ? BLASR_HIP_OP_M : arrow == EX_UP ? BLASR_HIP_OP_I : BLASR_HIP_OP_D;
    if (type == curType) curLen++;
    else {
      flush();
      curType = type;
      curLen = 1;
    }
    if (arrow != EX_LEFT) q--;
    if (arrow != EX_UP) t--;
  }
  flush();
  res.dp_score = globalMin;
  res.n_cells = nCells;
  if (rev) {  // ReverseIndex::GetQueryStartPos / GetRefStartPos (:97-103)
    res.q_pos = g.idxQ - (gq - 1);
    res.t_pos = g.idxT - (gt - 1);
    res.n_ops = pos;
    a.opsStart[task] = 0;
  } else {
    res.q_pos = a.qPos[task];
    res.t_pos = a.tPos[task];
    res.n_ops = (int32_t)(g.arenaWords - pos);
    a.opsStart[task] = pos;
  }
  a.results[task] = res;
}


// ---------------------------------------------------------------------------------------------
// The -extend step of AlignIntervals on the mapper's own device records (alignment/Blasr.cpp:1749-1940): every SDP
// alignment with blocks is cut to its aligned extent, up to 500 read bases after it are aligned to up to 500 genome
// bases (Forward), up to 500 before it to up to 500 before it (Reverse), and an extension that scored below 0 is
// joined to the candidate: blocks, aligned sequences, positions.
//   extend_tasks_kernel  the positions and lengths of :1755-1823 and :1865-1884 and the four strings of a candidate
//   extend_kernel        (above) 2 tasks per candidate
//   extend_merge_kernel  AppendAlignment (AlignmentCandidate.h:185-277; no gap lists here) in both directions, the
//                        new aligned sequences, the candidate's fields (:1840-1855, :1899-1935)
//   extend_stats_kernel  ComputeAlignmentStats of the result (:1943-1945), an alignment without gap lists
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kMaxExtendLength = 500;  // DNALength maximumExtendLength = 500 (Blasr.cpp:1782)
constexpr int kExtSlot = 512;               // bytes per string in the task pool (4 strings per candidate)

struct ExtendPlan {
  uint32_t qASP1, tASP1;       // qAlignedSeqPos / tAlignedSeqPos after :1758, :1769
  int32_t qEnd, tEnd;          // QEnd() / TEnd() of the SDP blocks
  int32_t rsl, gsl, rpl, gpl;  // readSuffixLength, genomeSuffixLength, readPrefixLength, genomePrefixLength
  int32_t active;              // the candidate has blocks
  int32_t pad;
};

// ReverseComplementNuc (NucConversion.h:337-370): codes 0..3, ACGTN in both cases, everything else 127
__device__ __forceinline__ uint8_t ext_rc_nuc(uint8_t c) {
  uint8_t r = 127;
  r = c < 4 ? (uint8_t)(3 - c) : r;
  r = c == 'A' ? 'T' : r;
  r = c == 'C' ? 'G' : r;
  r = c == 'G' ? 'C' : r;
  r = c == 'T' ? 'A' : r;
  r = c == 'N' ? 'N' : r;
  r = c == 'a' ? 't' : r;
  r = c == 'c' ? 'g' : r;
  r = c == 'g' ? 'c' : r;
  r = c == 't' ? 'a' : r;
  r = c == 'n' ? 'n' : r;
  return r;
}

struct ExtendTaskArrays {  // 2 tasks per candidate: 2k Forward, 2k + 1 Reverse
  int64_t *qOff, *tOff;
  int32_t *qLen, *tLen, *qPos, *tPos;
  uint8_t *reverse;
};

__global__ __launch_bounds__(256) void extend_tasks_kernel(int n, const blasr_hip_interval *intervals,
                                                           const blasr_hip_read *reads, const uint8_t *readPool,
                                                           const uint8_t *genome, uint32_t genomeLen,
                                                           const uint32_t *contigStarts, const blasr_hip_window *win,
                                                           const blasr_hip_sdp_result *res, const int32_t *blocks,
                                                           ExtendPlan *plan, ExtendTaskArrays ta, uint8_t *pool) {
  const int k = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (k >= n) return;
  const int lane = lane_id();
  const blasr_hip_sdp_result r = res[k];
  const blasr_hip_window w = win[k];
  const blasr_hip_read rd = reads[intervals[k].read];
  const int nb = r.status == 0 ? r.n_blocks : 0;
  ExtendPlan p;
  memset(&p, 0, sizeof(p));
  uint8_t *slot = pool + (int64_t)k * 4 * kExtSlot;
  bool callFwd = false, callRev = false;
  uint32_t lastQ = 0, lastT = 0;
  if (nb > 0) {
    const int32_t *last = blocks + 3 * (r.blk_off + nb - 1);
    p.active = 1;
    p.qEnd = last[0] + last[2];
    p.tEnd = last[1] + last[2];
    p.qASP1 = w.q_aligned_seq_pos + (uint32_t)r.q_pos;  // :1758
    p.tASP1 = w.t_aligned_seq_pos + (uint32_t)r.t_pos;  // :1769
    // the contig of the interval on the strand of the alignment (:1272-1279, :1301-1303, :1319-1321)
    uint32_t cs = 0, ce = genomeLen;
    if (contigStarts != nullptr) {
      cs = contigStarts[w.t_index];
      ce = contigStarts[w.t_index + 1] - 1u;
    }
    const uint32_t contigStart = w.t_strand ? genomeLen - ce : cs, contigEnd = w.t_strand ? genomeLen - cs : ce;
    lastQ = (uint32_t)p.qEnd + p.qASP1;  // :1786-1787
    lastT = (uint32_t)p.tEnd + p.tASP1;
    p.rsl = (int32_t)min(rd.length - lastQ, kMaxExtendLength);  // :1803
    p.gsl = (int32_t)min(contigEnd - lastT, kMaxExtendLength);  // :1814
    p.rpl = (int32_t)min(p.qASP1, kMaxExtendLength);            // :1861
    p.gpl = (int32_t)min(p.tASP1 - contigStart, kMaxExtendLength);  // :1869
    callFwd = p.rsl > 0 && p.gsl > 0;  // :1825
    callRev = p.rpl > 0 && p.gpl > 0;  // :1884
    // the four strings
    const uint8_t *read = readPool + rd.off;
    for (int j = lane; j < p.rsl; j += 64) slot[j] = read[lastQ + (uint32_t)j];
    for (int j = lane; j < p.rpl; j += 64) slot[2 * kExtSlot + j] = read[p.qASP1 - (uint32_t)p.rpl + (uint32_t)j];
    if (w.t_strand == 0) {
      for (int j = lane; j < p.gsl; j += 64) slot[kExtSlot + j] = genome[lastT + (uint32_t)j];
      for (int j = lane; j < p.gpl; j += 64) slot[3 * kExtSlot + j] = genome[p.tASP1 - (uint32_t)p.gpl + (uint32_t)j];
    } else {
      // CopyAsRC (DNASequence.h:219-236): the position is on the reverse strand
      for (int j = lane; j < p.gsl; j += 64) slot[kExtSlot + j] = ext_rc_nuc(genome[genomeLen - lastT - 1u - (uint32_t)j]);
      // MakeRC (DNASequence.h:238-250) takes a FORWARD position and is handed the reverse-strand one (:1876): the
      // prefix of a reverse-strand alignment is the reverse complement of forward[firstT - len, firstT)
      const uint32_t pos = p.tASP1 - (uint32_t)p.gpl;
      for (int j = lane; j < p.gpl; j += 64)
        slot[3 * kExtSlot + j] = ext_rc_nuc(genome[pos + (uint32_t)p.gpl - 1u - (uint32_t)j]);
    }
  }
  if (lane == 0) {
    plan[k] = p;
    const int64_t base = (int64_t)k * 4 * kExtSlot;
    const int f = 2 * k, v = 2 * k + 1;
    ta.qOff[f] = base;
    ta.tOff[f] = base + kExtSlot;
    ta.qLen[f] = callFwd ? p.rsl : 0;  // not called: forwardScore = 0 (:1824)
    ta.tLen[f] = callFwd ? p.gsl : 0;
    ta.qPos[f] = 0;
    ta.tPos[f] = 0;
    ta.reverse[f] = 0;
    ta.qOff[v] = base + 2 * kExtSlot;
    ta.tOff[v] = base + 3 * kExtSlot;
    ta.qLen[v] = callRev ? p.rpl : 0;
    ta.tLen[v] = callRev ? p.gpl : 0;
    ta.qPos[v] = callRev ? p.rpl - 1 : 0;  // readPrefix.length - 1, genomePrefixLength - 1 (:1885-1886)
    ta.tPos[v] = callRev ? p.gpl - 1 : 0;
    ta.reverse[v] = 1;
  }
}

struct ExtendMergeArgs {
  int n;
  const blasr_hip_interval *intervals;
  const blasr_hip_read *reads;
  const uint8_t *readPool;
  const ExtendPlan *plan;
  const RefineResult *ext;  // 2 per candidate
  const uint32_t *extOps;
  const uint8_t *pool;
  const uint8_t *oldSeq;
  const int32_t *oldBlocks;
  uint8_t *newSeq;
  int32_t *newBlocks;
  const int64_t *newQOff, *newTOff, *newBlkOff;
  int64_t *qOff, *tOff;  // the SDP batch arrays, rewritten in place
  int32_t *qLen, *tLen;
  blasr_hip_window *win;
  blasr_hip_sdp_result *res;
};

__global__ __launch_bounds__(256) void extend_merge_kernel(ExtendMergeArgs a) {
  __shared__ int sh[4][8];
  const int wv = threadIdx.x >> 6;
  const int k = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wv));
  if (k >= a.n) return;
  const int lane = lane_id();
  const ExtendPlan p = a.plan[k];
  if (!p.active) {  // no blocks: nothing reads its sequences again
    if (lane == 0) {
      a.qOff[k] = a.tOff[k] = 0;
      a.qLen[k] = a.tLen[k] = 0;
    }
    return;
  }
  const RefineResult fwd = a.ext[2 * k], rev = a.ext[2 * k + 1];
  const bool fwdOn = fwd.status == 0 && fwd.dp_score < 0;  // :1840
  const bool revOn = rev.status == 0 && rev.dp_score < 0;  // :1899
  blasr_hip_sdp_result r = a.res[k];
  const int nb0 = r.n_blocks;
  const int32_t *old = a.oldBlocks + 3 * r.blk_off;
  int32_t *out = a.newBlocks + 3 * a.newBlkOff[k];
  if (lane == 0) {
    int nbRev = 0, qShift = 0, tShift = 0;
    if (revOn) {  // extendedAlignmentReverse.AppendAlignment(*alignment) (:1909): its blocks come first
      int q = 0, t = 0;
      const uint32_t *ops = a.extOps + rev.ops_off;
      for (int i = 0; i < rev.n_ops; i++) {
        const int len = (int)(ops[i] >> 2), type = (int)(ops[i] & 3u);
        if (type == BLASR_HIP_OP_M) {
          out[3 * nbRev] = q, out[3 * nbRev + 1] = t, out[3 * nbRev + 2] = len;
          nbRev++;
          q += len, t += len;
        } else if (type == BLASR_HIP_OP_I) q += len;
        else t += len;
      }
      qShift = p.rpl - rev.q_pos;  // qOffset of AlignmentCandidate.h:210-211
      tShift = p.gpl - rev.t_pos;
    }
    int nbFwd = 0, fwdQ = 0, fwdT = 0;
    if (fwdOn) {  // alignment->AppendAlignment(extendedAlignmentForward) (:1854): offsets QEnd() / TEnd()
      int q = 0, t = 0;
      const uint32_t *ops = a.extOps + fwd.ops_off;
      int32_t *dst = out + 3 * (nbRev + nb0);
      for (int i = 0; i < fwd.n_ops; i++) {
        const int len = (int)(ops[i] >> 2), type = (int)(ops[i] & 3u);
        if (type == BLASR_HIP_OP_M) {
          dst[3 * nbFwd] = q + p.qEnd + qShift, dst[3 * nbFwd + 1] = t + p.tEnd + tShift, dst[3 * nbFwd + 2] = len;
          nbFwd++;
          q += len, t += len;
          fwdQ = q, fwdT = t;  // extendedAlignmentForward.QEnd() / TEnd() (its qPos / tPos are 0)
        } else if (type == BLASR_HIP_OP_I) q += len;
        else t += len;
      }
    }
    sh[wv][0] = nbRev, sh[wv][1] = nbFwd, sh[wv][2] = qShift, sh[wv][3] = tShift, sh[wv][4] = fwdQ, sh[wv][5] = fwdT;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the LDS writes of lane 0
  const int nbRev = sh[wv][0], nbFwd = sh[wv][1], qShift = sh[wv][2], tShift = sh[wv][3], fwdQ = sh[wv][4],
            fwdT = sh[wv][5];
  for (int b = lane; b < nb0; b += 64) {
    out[3 * (nbRev + b)] = old[3 * b] + qShift;
    out[3 * (nbRev + b) + 1] = old[3 * b + 1] + tShift;
    out[3 * (nbRev + b) + 2] = old[3 * b + 2];
  }
  // the aligned sequences: the read is contiguous (:1759-1760, :1852, :1923-1926); the target is the prefix
  // string (all of it, :1911), the SDP extent of the old window (:1771) and the used part of the suffix string (:1850-1851)
  const blasr_hip_read rd = a.reads[a.intervals[k].read];
  const uint32_t qASP2 = revOn ? p.qASP1 - (uint32_t)p.rpl : p.qASP1;
  const uint32_t tASP2 = revOn ? p.tASP1 - (uint32_t)p.gpl : p.tASP1;
  const int qLen2 = (revOn ? p.rpl : 0) + p.qEnd + (fwdOn ? fwdQ : 0);
  const int tPre = revOn ? p.gpl : 0;
  const int tLen2 = tPre + p.tEnd + (fwdOn ? fwdT : 0);
  uint8_t *dq = a.newSeq + a.newQOff[k], *dt = a.newSeq + a.newTOff[k];
  const uint8_t *sq = a.readPool + rd.off + qASP2;
  for (int j = lane; j < qLen2; j += 64) dq[j] = sq[j];
  const uint8_t *slot = a.pool + (int64_t)k * 4 * kExtSlot;
  const uint8_t *sw = a.oldSeq + a.tOff[k] + r.t_pos;
  for (int j = lane; j < tPre; j += 64) dt[j] = slot[3 * kExtSlot + j];
  for (int j = lane; j < p.tEnd; j += 64) dt[tPre + j] = sw[j];
  if (fwdOn)
    for (int j = lane; j < fwdT; j += 64) dt[tPre + p.tEnd + j] = slot[kExtSlot + j];
  if (lane == 0) {
    blasr_hip_window w = a.win[k];
    w.q_aligned_seq_pos = qASP2;
    w.q_aligned_seq_length = (uint32_t)qLen2;
    w.t_aligned_seq_pos = tASP2;
    w.t_aligned_seq_length = (uint32_t)tLen2;
    a.win[k] = w;
    r.q_pos = revOn ? rev.q_pos : 0;  // :1927, :1761
    r.t_pos = revOn ? rev.t_pos : 0;  // :1919, :1772
    r.n_blocks = nbRev + nb0 + nbFwd;
    r.blk_off = a.newBlkOff[k];
    a.res[k] = r;
    a.qOff[k] = a.newQOff[k];
    a.tOff[k] = a.newTOff[k];
    a.qLen[k] = qLen2;
    a.tLen[k] = tLen2;
  }
}

// ComputeAlignmentStats (AlignmentUtils.h:535-584) of an alignment WITHOUT gap lists (CreateAlignmentStrings
// :408-446, :494-529): the gap between two blocks is the query surplus, then the target surplus, then the common
// part as aligned columns; the offset of the first block is the common part, then target, then query surplus.
__global__ __launch_bounds__(256) void extend_stats_kernel(int n, const ExtendPlan *plan, const uint8_t *seq,
                                                           const int64_t *qOff, const int64_t *tOff,
                                                           const int32_t *blocks, const int32_t *matrix, int ins,
                                                           int del, int open, int ext, blasr_hip_sdp_result *res) {
  __shared__ int m[25];
  if (threadIdx.x < 25) m[threadIdx.x] = matrix[threadIdx.x];
  __syncthreads();
  const int k = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (k >= n) return;
  if (!plan[k].active) return;
  const int lane = lane_id();
  blasr_hip_sdp_result r = res[k];
  const uint8_t *qSeq = seq + qOff[k] + r.q_pos, *tSeq = seq + tOff[k] + r.t_pos;
  const int32_t *blk = blocks + 3 * r.blk_off;
  const int nb = r.n_blocks;
  int score = 0, nMatch = 0, nMismatch = 0, nIns = 0, nDel = 0;
  auto columns = [&](int q, int t, int len) {
    for (int j = 0; j < len; j++) {
      const int qc = base_code(qSeq[q + j]), tc = base_code(tSeq[t + j]);
      score += m[qc * 5 + tc];
      if (qc == tc) nMatch++;
      else nMismatch++;
    }
  };
  auto gapRun = [&](int len, bool deletion) {
    if (len <= 0) return;
    const int linear = (deletion ? del : ins) * len, affine = len * ext + open;
    score += linear < affine ? linear : affine;
    if (deletion) nDel += len;
    else nIns += len;
  };
  for (int b = lane; b < nb; b += 64) {
    const int q = blk[3 * b], t = blk[3 * b + 1], len = blk[3 * b + 2];
    columns(q, t, len);
    if (b + 1 < nb) {
      int qGap = blk[3 * b + 3] - (q + len), tGap = blk[3 * b + 4] - (t + len);
      if (qGap > 0 || tGap > 0) {
        const int common = qGap > tGap ? tGap : qGap;
        qGap -= common;
        tGap -= common;
        gapRun(qGap, false);
        gapRun(tGap, true);
        columns(q + len + qGap, t + len + tGap, common);
      }
    }
    if (b == 0 && (q > 0 || t > 0)) {
      const int common = q > t ? t : q;
      columns(0, 0, common);
      gapRun(t - common, true);
      gapRun(q - common, false);
    }
  }
  int vals[5] = {score, nMatch, nMismatch, nIns, nDel};
  for (int v = 0; v < 5; v++)
    for (int off = 32; off > 0; off >>= 1) vals[v] += __shfl_xor(vals[v], off);
  if (lane == 0) {
    r.stats_score = vals[0];
    r.n_match = vals[1];
    r.n_mismatch = vals[2];
    r.n_ins = vals[3];
    r.n_del = vals[4];
    res[k] = r;
  }
}

}  // namespace blasr_hip

using namespace blasr_hip;

namespace {
struct Dev {
  void *p = nullptr;
  ~Dev() {
    if (p) (void)hipFree(p);
  }
};

int extend_run(blasr_hip_ctx *ctx, const blasr_hip_extend_batch *b, const blasr_hip_extend_params *p,
               blasr_hip_refine_result *results, uint32_t *ops, int64_t ops_capacity, int64_t *ops_used) {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (!b || !p || !ops_used || b->n < 0 || ops_capacity < 0 ||
      (b->n > 0 && (!b->seq || !b->q_off || !b->q_len || !b->t_off || !b->t_len || !b->q_pos || !b->t_pos ||
                    !b->reverse || !results || !ops)))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "extend: bad arguments");
  *ops_used = 0;
  const int n = b->n;
  if (n == 0) return BLASR_HIP_OK;
  if (p->band < 0 || 2 * (int64_t)p->band + 2 > kExtMaxCols)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_UNSUPPORTED, "extend: band (extendBandSize) must be 0..%d",
                          (kExtMaxCols - 2) / 2);
  int64_t magnitude = std::max<int64_t>(std::abs((int64_t)p->ins), std::abs((int64_t)p->del));
  for (int i = 0; i < 25; i++) magnitude = std::max<int64_t>(magnitude, std::abs((int64_t)p->matrix[i]));
  if (magnitude >= (1 << 10) || p->max_n_drops < 0 || p->max_n_drops > (1 << 20))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_UNSUPPORTED, "extend: |penalties| < 1024, 0 <= max_n_drops <= 2^20");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  Dev dGeo, dMatrix, dPathOff, dArenaOff, dOpsStart, dTotal;
  HIP_TRY(ctx, hipMalloc(&dGeo.p, sizeof(ExtendGeo) * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dMatrix.p, sizeof(int32_t) * 32));
  HIP_TRY(ctx, hipMalloc(&dPathOff.p, sizeof(int64_t) * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dArenaOff.p, sizeof(int64_t) * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dOpsStart.p, sizeof(int32_t) * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dTotal.p, sizeof(int64_t)));
  HIP_TRY(ctx, hipMemcpyAsync(dMatrix.p, p->matrix, sizeof(int32_t) * 25, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemsetAsync(dTotal.p, 0, sizeof(int64_t), st));
  hipLaunchKernelGGL(extend_plan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, b->q_len, b->t_len,
                     b->q_pos, b->t_pos, b->reverse, p->band, p->min_extend_n_bases, p->max_n_drops,
                     static_cast<ExtendGeo *>(dGeo.p));
  std::vector<ExtendGeo> geo((size_t)n);
  HIP_TRY(ctx, hipMemcpyAsync(geo.data(), dGeo.p, sizeof(ExtendGeo) * (size_t)n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  // chunks of tasks whose arrow words + op arenas fit the workspace
  const int64_t limit = std::max<int64_t>(ctx->workspaceLimit, 1 << 20);
  std::vector<int64_t> pathOff((size_t)n), arenaOff((size_t)n);
  struct Chunk {
    int first, count;
    int64_t pathWords, arenaWords;
  };
  std::vector<Chunk> chunks;
  int64_t maxBytes = 0;
  for (int i = 0; i < n;) {
    Chunk c{i, 0, 0, 0};
    while (i < n) {
      const int64_t pw = (geo[i].pathWords + 7) & ~(int64_t)7, aw = (geo[i].arenaWords + 15) & ~(int64_t)15;
      if (c.count > 0 && 8 * (c.pathWords + pw) + 4 * (c.arenaWords + aw) > limit) break;
      if (8 * pw + 4 * aw > limit)
        return blasr_hip_fail(ctx, BLASR_HIP_ERR_NOMEM, "extend: task %d needs %lld bytes of workspace", i,
                              (long long)(8 * pw + 4 * aw));
      pathOff[i] = c.pathWords;
      arenaOff[i] = c.arenaWords;
      c.pathWords += pw;
      c.arenaWords += aw;
      c.count++;
      i++;
    }
    maxBytes = std::max(maxBytes, 8 * c.pathWords + 4 * c.arenaWords + 1024);
    chunks.push_back(c);
  }
  if (ctx->workspaceBytes < (size_t)maxBytes) {
    if (ctx->workspace) (void)hipFree(ctx->workspace);
    ctx->workspace = nullptr;
    ctx->workspaceBytes = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->workspace, (size_t)maxBytes));
    ctx->workspaceBytes = (size_t)maxBytes;
  }
  HIP_TRY(ctx, hipMemcpyAsync(dPathOff.p, pathOff.data(), sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dArenaOff.p, arenaOff.data(), sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, st));
  RefineResult *dResults = reinterpret_cast<RefineResult *>(results);
  for (size_t ci = 0; ci < chunks.size(); ci++) {
    const Chunk &c = chunks[ci];
    ExtendArgs a;
    a.first = c.first;
    a.count = c.count;
    a.k = p->band;
    a.ins = p->ins;
    a.del = p->del;
    a.maxNDrops = p->max_n_drops;
    a.matrix = static_cast<int32_t *>(dMatrix.p);
    a.geo = static_cast<ExtendGeo *>(dGeo.p);
    a.seq = b->seq;
    a.qOff = b->q_off;
    a.tOff = b->t_off;
    a.qPos = b->q_pos;
    a.tPos = b->t_pos;
    a.reverse = b->reverse;
    a.path = reinterpret_cast<unsigned long long *>(ctx->workspace);
    a.pathOff = static_cast<int64_t *>(dPathOff.p);
    a.arena = reinterpret_cast<uint32_t *>(ctx->workspace + 8 * c.pathWords);
    a.arenaOff = static_cast<int64_t *>(dArenaOff.p);
    a.opsStart = static_cast<int32_t *>(dOpsStart.p);
    a.results = dResults;
    hipLaunchKernelGGL(extend_kernel, dim3((unsigned)((c.count + 63) / 64)), dim3(64), 0, st, a);
    launch_scan_ops(dResults, c.first, c.count, ops_capacity, static_cast<int64_t *>(dTotal.p), st);
    launch_compact_ops(dResults, c.first, c.count, a.arena, a.arenaOff, a.opsStart, ops, st);
    if (ci + 1 < chunks.size()) HIP_TRY(ctx, hipStreamSynchronize(st));  // the next chunk reuses the workspace
  }
  int64_t total = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&total, dTotal.p, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  HIP_TRY(ctx, hipGetLastError());
  *ops_used = total;
  if (total > ops_capacity)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_CAPACITY, "extend: %lld ops produced, arena holds %lld", (long long)total,
                          (long long)ops_capacity);
  return BLASR_HIP_OK;
}
size_t ext_up256(size_t v) { return (v + 255) & ~(size_t)255; }
}  // namespace

// Internal (ctx.h): the -extend step on the SDP stage's device records, between blasr_hip_sdp_device and the first
// filter of blasr_hip_map_reads_device.  seq / blocks hold the stage's sequences and blocks; the extended candidates
// are written to newSeq / newBlocks (every candidate with blocks moves there), the batch arrays, the windows and the
// result records are rewritten in place.  *blocks_used receives the triplets in newBlocks.
int blasr_hip_extend_candidates(blasr_hip_ctx *ctx, int n, const blasr_hip_interval *intervals,
                                const blasr_hip_read *reads, const uint8_t *read_pool,
                                const blasr_hip_extend_params *ep, const blasr_hip_score_params *score,
                                const uint8_t *seq, int64_t *qOff, int32_t *qLen, int64_t *tOff, int32_t *tLen,
                                blasr_hip_window *win, blasr_hip_sdp_result *res, const int32_t *blocks,
                                DevBuf &newSeq, DevBuf &newBlocks, DevBuf &scratch, DevBuf &opsBuf,
                                int64_t *blocks_used) {
  *blocks_used = 0;
  if (n <= 0) return BLASR_HIP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t nn = (size_t)n, n2 = 2 * nn;
  // scratch: plan, the 2n tasks, their results, the three offset arrays, the score matrix, the string pool
  size_t at = 0;
  auto carve = [&](size_t bytes) {
    const size_t o = at;
    at += ext_up256(bytes);
    return o;
  };
  const size_t oPlan = carve(sizeof(ExtendPlan) * nn), oQOff = carve(8 * n2), oTOff = carve(8 * n2), oQLen = carve(4 * n2),
               oTLen = carve(4 * n2), oQPos = carve(4 * n2), oTPos = carve(4 * n2), oRev = carve(n2),
               oRes = carve(sizeof(RefineResult) * n2), oNewQ = carve(8 * nn), oNewT = carve(8 * nn),
               oNewB = carve(8 * nn), oMatrix = carve(128), oPool = carve(nn * 4 * kExtSlot);
  int rc = blasr_hip_ensure(ctx, scratch, at);
  if (rc) return rc;
  char *sp = scratch.p;
  ExtendTaskArrays ta;
  ta.qOff = (int64_t *)(sp + oQOff);
  ta.tOff = (int64_t *)(sp + oTOff);
  ta.qLen = (int32_t *)(sp + oQLen);
  ta.tLen = (int32_t *)(sp + oTLen);
  ta.qPos = (int32_t *)(sp + oQPos);
  ta.tPos = (int32_t *)(sp + oTPos);
  ta.reverse = (uint8_t *)(sp + oRev);
  ExtendPlan *dPlan = (ExtendPlan *)(sp + oPlan);
  uint8_t *dPool = (uint8_t *)(sp + oPool);
  hipLaunchKernelGGL(extend_tasks_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, n, intervals, reads,
                     read_pool, ctx->genome, (uint32_t)ctx->genomeLen, ctx->contigStarts, win, res, blocks, dPlan, ta,
                     dPool);
  blasr_hip_extend_batch eb;
  eb.n = (int32_t)n2;
  eb.seq = dPool;
  eb.q_off = ta.qOff;
  eb.q_len = ta.qLen;
  eb.t_off = ta.tOff;
  eb.t_len = ta.tLen;
  eb.q_pos = ta.qPos;
  eb.t_pos = ta.tPos;
  eb.reverse = ta.reverse;
  blasr_hip_refine_result *dExt = (blasr_hip_refine_result *)(sp + oRes);
  int64_t opsCap = (int64_t)n2 * 24 + 1024, opsUsed = 0;
  for (int attempt = 0; attempt < 2; attempt++) {
    if ((rc = blasr_hip_ensure(ctx, opsBuf, sizeof(uint32_t) * (size_t)opsCap))) return rc;
    opsCap = (int64_t)(opsBuf.bytes / sizeof(uint32_t));
    rc = extend_run(ctx, &eb, ep, dExt, (uint32_t *)opsBuf.p, opsCap, &opsUsed);
    if (rc != BLASR_HIP_ERR_CAPACITY) break;
    opsCap = opsUsed + 64;
  }
  if (rc) return rc;
  std::vector<ExtendPlan> plan(nn);
  std::vector<blasr_hip_refine_result> ext(n2);
  std::vector<blasr_hip_sdp_result> sdp(nn);
  HIP_TRY(ctx, hipMemcpyAsync(plan.data(), dPlan, sizeof(ExtendPlan) * nn, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(ext.data(), dExt, sizeof(blasr_hip_refine_result) * n2, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(sdp.data(), res, sizeof(blasr_hip_sdp_result) * nn, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  // room for every candidate with blocks: its read stretch, its target string, its blocks
  std::vector<int64_t> newQ(nn), newT(nn), newB(nn);
  int64_t seqBytes = 0, blk = 0;
  for (size_t k = 0; k < nn; k++) {
    newQ[k] = newT[k] = seqBytes;
    newB[k] = blk;
    const ExtendPlan &p = plan[k];
    if (!p.active) continue;
    const blasr_hip_refine_result &f = ext[2 * k], &v = ext[2 * k + 1];
    if (f.status != 0 || v.status != 0)
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_HIP, "extend: candidate %zu: task status %d / %d", k, f.status, v.status);
    const bool fwdOn = f.dp_score < 0, revOn = v.dp_score < 0;
    const int64_t qCap = (revOn ? p.rpl : 0) + (int64_t)p.qEnd + (fwdOn ? p.rsl : 0);
    const int64_t tCap = (revOn ? p.gpl : 0) + (int64_t)p.tEnd + (fwdOn ? p.gsl : 0);
    newT[k] = seqBytes + qCap;
    seqBytes += qCap + tCap;
    blk += (int64_t)sdp[k].n_blocks + (fwdOn ? f.n_ops : 0) + (revOn ? v.n_ops : 0);
  }
  if ((rc = blasr_hip_ensure(ctx, newSeq, (size_t)seqBytes + 64))) return rc;
  if ((rc = blasr_hip_ensure(ctx, newBlocks, sizeof(int32_t) * 3 * (size_t)(blk + 1)))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(sp + oNewQ, newQ.data(), 8 * nn, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(sp + oNewT, newT.data(), 8 * nn, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(sp + oNewB, newB.data(), 8 * nn, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(sp + oMatrix, score->matrix, sizeof(int32_t) * 25, hipMemcpyHostToDevice, st));
  ExtendMergeArgs ma;
  ma.n = n;
  ma.intervals = intervals;
  ma.reads = reads;
  ma.readPool = read_pool;
  ma.plan = dPlan;
  ma.ext = reinterpret_cast<const RefineResult *>(dExt);
  ma.extOps = (const uint32_t *)opsBuf.p;
  ma.pool = dPool;
  ma.oldSeq = seq;
  ma.oldBlocks = blocks;
  ma.newSeq = (uint8_t *)newSeq.p;
  ma.newBlocks = (int32_t *)newBlocks.p;
  ma.newQOff = (const int64_t *)(sp + oNewQ);
  ma.newTOff = (const int64_t *)(sp + oNewT);
  ma.newBlkOff = (const int64_t *)(sp + oNewB);
  ma.qOff = qOff;
  ma.tOff = tOff;
  ma.qLen = qLen;
  ma.tLen = tLen;
  ma.win = win;
  ma.res = res;
  hipLaunchKernelGGL(extend_merge_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, ma);
  hipLaunchKernelGGL(extend_stats_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, n, dPlan,
                     (const uint8_t *)newSeq.p, qOff, tOff, (const int32_t *)newBlocks.p,
                     (const int32_t *)(sp + oMatrix), score->ins, score->del, score->affine_open, score->affine_extend,
                     res);
  HIP_TRY(ctx, hipStreamSynchronize(st));
  HIP_TRY(ctx, hipGetLastError());
  *blocks_used = blk;
  return BLASR_HIP_OK;
}

extern "C" {

int blasr_hip_extend_default_params(blasr_hip_extend_params *p) try {
  if (!p) return BLASR_HIP_ERR_INVALID;
  memset(p, 0, sizeof(*p));
  static const int32_t smrt[25] = {-5, 6, 6, 6, 0, 6, -5, 6, 6, 0, 6, 6, -5, 6, 0, 6, 6, 6, -5, 0, 0, 0, 0, 0, 0};
  memcpy(p->matrix, smrt, sizeof(smrt));  // distScoreFn: SMRTDistanceMatrix (Blasr.cpp:3666-3668)
  p->ins = 5;                             // params.insertion
  p->del = 5;                             // params.deletion
  p->band = 10;                           // params.extendBandSize (MappingParameters.h:312)
  p->min_extend_n_bases = 1;              // Blasr.cpp:1833
  p->max_n_drops = 10;                    // params.maxExtendDropoff (MappingParameters.h:270)
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_extend_device(blasr_hip_ctx *ctx, const blasr_hip_extend_batch *b, const blasr_hip_extend_params *p,
                            blasr_hip_refine_result *results, uint32_t *ops, int64_t ops_capacity, int64_t *ops_used) try {
  return extend_run(ctx, b, p, results, ops, ops_capacity, ops_used);
} BLASR_HIP_CATCH_CTX(ctx)

int blasr_hip_extend_host(blasr_hip_ctx *ctx, const blasr_hip_extend_batch *b, const blasr_hip_extend_params *p,
                          blasr_hip_refine_result *results, uint32_t *ops, int64_t ops_capacity, int64_t *ops_used) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (!b || !p || !ops_used || b->n < 0 || ops_capacity < 0 ||
      (b->n > 0 && (!b->seq || !b->q_off || !b->q_len || !b->t_off || !b->t_len || !b->q_pos || !b->t_pos ||
                    !b->reverse || !results || !ops)))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "extend: bad arguments");
  *ops_used = 0;
  const int n = b->n;
  if (n == 0) return BLASR_HIP_OK;
  int64_t seqBytes = 0;
  for (int i = 0; i < n; i++) {
    if (b->q_len[i] < 0 || b->t_len[i] < 0 || b->q_off[i] < 0 || b->t_off[i] < 0)
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "extend: negative length or offset");
    seqBytes = std::max<int64_t>(seqBytes, std::max(b->q_off[i] + b->q_len[i], b->t_off[i] + b->t_len[i]));
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  Dev dSeq, dQOff, dTOff, dQLen, dTLen, dQPos, dTPos, dRev, dRes, dOps;
  HIP_TRY(ctx, hipMalloc(&dSeq.p, (size_t)std::max<int64_t>(seqBytes, 1)));
  HIP_TRY(ctx, hipMalloc(&dQOff.p, sizeof(int64_t) * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dTOff.p, sizeof(int64_t) * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dQLen.p, sizeof(int32_t) * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dTLen.p, sizeof(int32_t) * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dQPos.p, sizeof(int32_t) * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dTPos.p, sizeof(int32_t) * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dRev.p, (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dRes.p, sizeof(blasr_hip_refine_result) * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dOps.p, sizeof(uint32_t) * (size_t)std::max<int64_t>(ops_capacity, 1)));
  HIP_TRY(ctx, hipMemcpyAsync(dSeq.p, b->seq, (size_t)seqBytes, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dQOff.p, b->q_off, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dTOff.p, b->t_off, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dQLen.p, b->q_len, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dTLen.p, b->t_len, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dQPos.p, b->q_pos, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dTPos.p, b->t_pos, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dRev.p, b->reverse, (size_t)n, hipMemcpyHostToDevice, st));
  blasr_hip_extend_batch db;
  db.n = n;
  db.seq = static_cast<uint8_t *>(dSeq.p);
  db.q_off = static_cast<int64_t *>(dQOff.p);
  db.q_len = static_cast<int32_t *>(dQLen.p);
  db.t_off = static_cast<int64_t *>(dTOff.p);
  db.t_len = static_cast<int32_t *>(dTLen.p);
  db.q_pos = static_cast<int32_t *>(dQPos.p);
  db.t_pos = static_cast<int32_t *>(dTPos.p);
  db.reverse = static_cast<uint8_t *>(dRev.p);
  const int rc = extend_run(ctx, &db, p, static_cast<blasr_hip_refine_result *>(dRes.p),
                            static_cast<uint32_t *>(dOps.p), ops_capacity, ops_used);
  if (rc != BLASR_HIP_OK && rc != BLASR_HIP_ERR_CAPACITY) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(results, dRes.p, sizeof(blasr_hip_refine_result) * (size_t)n, hipMemcpyDeviceToHost, st));
  const int64_t copyOps = std::min(*ops_used, ops_capacity);
  if (copyOps > 0) HIP_TRY(ctx, hipMemcpyAsync(ops, dOps.p, sizeof(uint32_t) * (size_t)copyOps, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return rc;
} BLASR_HIP_CATCH_CTX(ctx)

}  // extern "C"
