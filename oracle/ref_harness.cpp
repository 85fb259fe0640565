// TEST INFRASTRUCTURE ONLY -- never linked into the product.
//
// Thin extern "C" shim that instantiates the *reference's own* header
// templates (found under $BLASR_REF/common, default /root/reference/common)
// with the same types and arguments alignment/Blasr.cpp uses:
//   SDPAlign            as called at alignment/Blasr.cpp:1716-1722
//   AffineGuidedAlign   as called at alignment/Blasr.cpp:862-867
//   ComputeAlignmentStats as called at alignment/Blasr.cpp:1723-1724, 875-878
// No reference source is copied here: this file only #includes the headers
// where they lie.  oracle/Makefile compiles it into oracle/_ref/ (git-ignored).
// The resulting libblasr_ref.so pins oracle/blasr_oracle.c (the CPU
// restatement) and generates tests/golden/*.  It exists only in a container
// that has /root/reference mounted.
#include <string.h>
#include <stdint.h>
#include <vector>

#include "FASTASequence.h"
#include "FASTQSequence.h"
#include "tuples/DNATuple.h"
#include "tuples/TupleMetrics.h"
#include "algorithms/alignment.h"
#include "algorithms/alignment/AffineKBandAlign.h"
#include "algorithms/alignment/GuidedAlign.h"
#include "algorithms/alignment/AffineGuidedAlign.h"
#include "algorithms/alignment/ExtendAlign.h"
#include "algorithms/alignment/AlignmentUtils.h"
#include "algorithms/alignment/DistanceMatrixScoreFunction.h"

typedef DistanceMatrixScoreFunction<DNASequence, FASTQSequence> ScoreFn;

namespace {

// Same member names as MappingBuffers in alignment/Blasr.cpp:318-369 (the
// subset SDPAlign / AffineGuidedAlign touch).
struct Buffers {
  std::vector<int> scoreMat;
  std::vector<Arrow> pathMat;
  std::vector<Fragment> sdpFragmentSet, sdpPrefixFragmentSet, sdpSuffixFragmentSet;
  TupleList<PositionDNATuple> sdpCachedTargetTupleList;
  TupleList<PositionDNATuple> sdpCachedTargetPrefixTupleList;
  TupleList<PositionDNATuple> sdpCachedTargetSuffixTupleList;
  std::vector<int> sdpCachedMaxFragmentChain;
  std::vector<double> probMat, optPathProbMat;
  std::vector<float> lnSubPValueMat, lnInsPValueMat, lnDelPValueMat, lnMatchPValueMat;
};

void make_score_fn(ScoreFn &fn, const int *matrix25, int ins, int del, int affOpen, int affExt) {
  int m[5][5];
  for (int i = 0; i < 5; i++)
    for (int j = 0; j < 5; j++) m[i][j] = matrix25[i * 5 + j];
  fn.InitializeScoreMatrix(m);
  fn.ins = ins;
  fn.del = del;
  fn.affineOpen = affOpen;
  fn.affineExtend = affExt;
}

void set_seq(DNASequence &s, const unsigned char *p, int len) {
  s.seq = (Nucleotide *)p;
  s.length = len;
  s.deleteOnExit = false;
}

// Flatten an Alignment: blocks as (q,t,len) triplets; gap lists as
// gapCounts[nGapLists] + (seq,len) pairs.
int flatten(Alignment &a, int *blocks, int blockCap, int *nBlocks, int *gapCounts, int gapListCap,
            int *nGapLists, int *gapItems, int gapItemCap, int *nGapItems) {
  *nBlocks = (int)a.blocks.size();
  if ((int)a.blocks.size() > blockCap) return -1;
  for (size_t b = 0; b < a.blocks.size(); b++) {
    blocks[3 * b] = (int)a.blocks[b].qPos;
    blocks[3 * b + 1] = (int)a.blocks[b].tPos;
    blocks[3 * b + 2] = (int)a.blocks[b].length;
  }
  if (nGapLists) {
    *nGapLists = (int)a.gaps.size();
    if ((int)a.gaps.size() > gapListCap) return -2;
    int n = 0;
    for (size_t g = 0; g < a.gaps.size(); g++) {
      gapCounts[g] = (int)a.gaps[g].size();
      for (size_t i = 0; i < a.gaps[g].size(); i++) {
        if (n >= gapItemCap) return -3;
        gapItems[2 * n] = (int)a.gaps[g][i].seq;
        gapItems[2 * n + 1] = a.gaps[g][i].length;
        n++;
      }
    }
    *nGapItems = n;
  }
  return 0;
}

void unflatten(Alignment &a, int qPos, int tPos, const int *blocks, int nBlocks, const int *gapCounts,
               int nGapLists, const int *gapItems) {
  a.qPos = qPos;
  a.tPos = tPos;
  a.blocks.resize(nBlocks);
  for (int b = 0; b < nBlocks; b++) {
    a.blocks[b].qPos = blocks[3 * b];
    a.blocks[b].tPos = blocks[3 * b + 1];
    a.blocks[b].length = blocks[3 * b + 2];
  }
  a.gaps.clear();
  int n = 0;
  for (int g = 0; g < nGapLists; g++) {
    a.gaps.push_back(GapList());
    for (int i = 0; i < gapCounts[g]; i++) {
      a.gaps[g].push_back(Gap((Gap::GapSeq)gapItems[2 * n], gapItems[2 * n + 1]));
      n++;
    }
  }
}

}  // namespace

extern "C" {

// SDPAlign through the buffer-cache overload (SDPAlign.h:66-93), argument
// order as Blasr.cpp:1716-1722.  Returns the int SDPAlign returns.
int ref_sdp_align(const unsigned char *q, int qlen, const unsigned char *t, int tlen,
                  const int *matrix25, int ins, int del, int affOpen, int affExt, int wordSize,
                  int sdpIns, int sdpDel, float indelRate, int alignType, int detailed,
                  int extendFront, int sdpPrefixLength, int recurse, int noRecurseUnder,
                  int maxPerPos, int *outQPos, int *outTPos, int *blocks, int blockCap,
                  int *nBlocks) {
  ScoreFn fn;
  make_score_fn(fn, matrix25, ins, del, affOpen, affExt);
  FASTQSequence qs;
  DNASequence ts;
  set_seq(qs, q, qlen);
  set_seq(ts, t, tlen);
  Buffers buffers;
  Alignment aln;
  int score = SDPAlign(qs, ts, fn, wordSize, sdpIns, sdpDel, indelRate, aln, buffers,
                       (AlignmentType)alignType, detailed != 0, extendFront != 0, sdpPrefixLength,
                       recurse, noRecurseUnder, maxPerPos);
  *outQPos = (int)aln.qPos;
  *outTPos = (int)aln.tPos;
  if (flatten(aln, blocks, blockCap, nBlocks, 0, 0, 0, 0, 0, 0)) return INT_MIN;
  return score;
}

// SDPLongestCommonSubsequence (sdp/SparseDynamicProgramming.h:70-322) on a
// caller-supplied fragment list (x,y,length triplets; weight = fragmentLength).
int ref_sdp_chain(int queryLength, const int *xyl, int nFrag, int fragmentLength, int ins, int del,
                  int match, int alignType, int *chain, int cap) {
  std::vector<Fragment> frags;
  for (int i = 0; i < nFrag; i++) {
    Fragment f(xyl[3 * i], xyl[3 * i + 1]);
    f.weight = fragmentLength;
    f.length = xyl[3 * i + 2];
    frags.push_back(f);
  }
  std::vector<int> out;
  SDPLongestCommonSubsequence((DNALength)queryLength, frags, (DNALength)fragmentLength, ins, del,
                              match, out, (AlignmentType)alignType);
  for (size_t i = 0; i < out.size() && (int)i < cap; i++) chain[i] = out[i];
  return (int)out.size();
}

// SWAlign (SWAlign.h:19-389) as SDPAlign calls it: scoreFn + alignType.
int ref_sw_align(const unsigned char *q, int qlen, const unsigned char *t, int tlen,
                 const int *matrix25, int ins, int del, int alignType, int *outQPos, int *outTPos,
                 int *blocks, int blockCap, int *nBlocks) {
  ScoreFn fn;
  make_score_fn(fn, matrix25, ins, del, 0, 0);
  FASTQSequence qs;
  DNASequence ts;
  set_seq(qs, q, qlen);
  set_seq(ts, t, tlen);
  std::vector<int> scoreMat;
  std::vector<Arrow> pathMat;
  Alignment aln;
  int score = SWAlign(qs, ts, scoreMat, pathMat, aln, fn, (AlignmentType)alignType);
  *outQPos = (int)aln.qPos;
  *outTPos = (int)aln.tPos;
  if (flatten(aln, blocks, blockCap, nBlocks, 0, 0, 0, 0, 0, 0)) return INT_MIN;
  return score;
}

// AlignmentToGuide + StoreMatrixOffsets (GuidedAlign.h:94-259).  rows = 5 ints
// per guide row: q, t, tPre, tPost, matrixOffset.  Returns row count.
int ref_alignment_to_guide(const int *blocks, int nBlocks, int bandSize, int *rows, int rowCap,
                           int *nCells) {
  Alignment a;
  unflatten(a, 0, 0, blocks, nBlocks, 0, 0, 0);
  Guide guide;
  AlignmentToGuide(a, guide, bandSize);
  StoreMatrixOffsets(guide);
  *nCells = ComputeMatrixNElem(guide);
  for (size_t r = 0; r < guide.size() && (int)r < rowCap; r++) {
    rows[5 * r] = guide[r].q;
    rows[5 * r + 1] = guide[r].t;
    rows[5 * r + 2] = guide[r].tPre;
    rows[5 * r + 3] = guide[r].tPost;
    rows[5 * r + 4] = (int)guide[r].matrixOffset;
  }
  return (int)guide.size();
}

// AffineGuidedAlign through the buffer-cache overload
// (AffineGuidedAlign.h:561-578) exactly as Blasr.cpp:862-867, followed by
// ComputeAlignmentStats as Blasr.cpp:875-878.  stats = {dpScore, nCells,
// statsScore, nMatch, nMismatch, nIns, nDel}; pct = pctSimilarity.
int ref_affine_guided_align(const unsigned char *q, int qlen, const unsigned char *t, int tlen,
                            const int *guideBlocks, int nGuideBlocks, const int *matrix25, int ins,
                            int del, int affOpen, int affExt, int bandSize, int *outQPos,
                            int *outTPos, int *blocks, int blockCap, int *nBlocks, int *gapCounts,
                            int gapListCap, int *nGapLists, int *gapItems, int gapItemCap,
                            int *nGapItems, int *stats, float *pct) {
  ScoreFn fn;
  make_score_fn(fn, matrix25, ins, del, affOpen, affExt);
  FASTQSequence qs;
  DNASequence ts;
  set_seq(qs, q, qlen);
  set_seq(ts, t, tlen);
  Alignment guide;
  unflatten(guide, 0, 0, guideBlocks, nGuideBlocks, 0, 0, 0);
  Buffers buffers;
  Alignment refined;
  int dp = AffineGuidedAlign(qs, ts, guide, fn, bandSize, buffers, refined, Global, false);
  stats[0] = dp;
  stats[1] = refined.nCells;
  ComputeAlignmentStats(refined, qs.seq, ts.seq, fn, true);
  stats[2] = refined.score;
  stats[3] = refined.nMatch;
  stats[4] = refined.nMismatch;
  stats[5] = refined.nIns;
  stats[6] = refined.nDel;
  *pct = refined.pctSimilarity;
  *outQPos = (int)refined.qPos;
  *outTPos = (int)refined.tPos;
  return flatten(refined, blocks, blockCap, nBlocks, gapCounts, gapListCap, nGapLists, gapItems,
                 gapItemCap, nGapItems);
}

// ComputeAlignmentStats (AlignmentUtils.h:535-584) on a caller-supplied
// alignment.  nGapLists == 0 -> the "no gap list" branch (SDP alignments).
void ref_compute_alignment_stats(const unsigned char *q, const unsigned char *t, int qPos, int tPos,
                                 const int *blocks, int nBlocks, const int *gapCounts,
                                 int nGapLists, const int *gapItems, const int *matrix25, int ins,
                                 int del, int affOpen, int affExt, int useAffine, int *stats,
                                 float *pct) {
  ScoreFn fn;
  make_score_fn(fn, matrix25, ins, del, affOpen, affExt);
  Alignment a;
  unflatten(a, qPos, tPos, blocks, nBlocks, gapCounts, nGapLists, gapItems);
  ComputeAlignmentStats(a, (Nucleotide *)q, (Nucleotide *)t, fn, useAffine != 0);
  stats[0] = a.score;
  stats[1] = a.nMatch;
  stats[2] = a.nMismatch;
  stats[3] = a.nIns;
  stats[4] = a.nDel;
  *pct = a.pctSimilarity;
}

// AffineKBandAlign (AffineKBandAlign.h:12-401) as PairwiseLocalAlign calls it (Blasr.cpp:695-703):
// Global, the caller's six matrices are fresh vectors.  Returns the flattened alignment; *score = the
// function's return value (S at the end cell).
int ref_affine_kband_align(const unsigned char *q, int qlen, const unsigned char *t, int tlen,
                           const int *matrix25, int hpInsOpen, int hpInsExtend, int insOpen, int insExtend,
                           int del, int k, int *score, int *blocks, int blockCap, int *nBlocks,
                           int *gapCounts, int gapListCap, int *nGapLists, int *gapItems, int gapItemCap,
                           int *nGapItems) {
  int m[5][5];
  for (int i = 0; i < 5; i++)
    for (int j = 0; j < 5; j++) m[i][j] = matrix25[i * 5 + j];
  FASTQSequence qs;
  DNASequence ts;
  set_seq(qs, q, qlen);
  set_seq(ts, t, tlen);
  std::vector<int> scoreMat, hpInsScoreMat, insScoreMat;
  std::vector<Arrow> pathMat, hpInsPathMat, insPathMat;
  Alignment a;
  *score = AffineKBandAlign(qs, ts, m, hpInsOpen, hpInsExtend, insOpen, insExtend, del, k, scoreMat, pathMat,
                            hpInsScoreMat, hpInsPathMat, insScoreMat, insPathMat, a, Global);
  return flatten(a, blocks, blockCap, nBlocks, gapCounts, gapListCap, nGapLists, gapItems, gapItemCap, nGapItems);
}

// KBandAlign (KBandAlign.h:74-403) with a DistanceMatrixScoreFunction, as RefineAlignment calls it for
// -doGlobalAlignment (Blasr.cpp:819-823, Global) and PairwiseLocalAlign for reads with quality values
// (Blasr.cpp:714-730, Fit).  ins/del: the boundary penalties (function arguments); fnIns/fnDel: the penalties
// of the score function used inside the recurrence.  out[3] = score, qPos, tPos.
int ref_kband_align(const unsigned char *q, int qlen, const unsigned char *t, int tlen, const int *matrix25, int ins,
                    int del, int fnIns, int fnDel, int k, int alignType, int *out, int *blocks, int blockCap,
                    int *nBlocks, int *gapCounts, int gapListCap, int *nGapLists, int *gapItems, int gapItemCap,
                    int *nGapItems) {
  int m[5][5];
  for (int i = 0; i < 5; i++)
    for (int j = 0; j < 5; j++) m[i][j] = matrix25[i * 5 + j];
  ScoreFn fn;
  make_score_fn(fn, matrix25, fnIns, fnDel, 0, 0);
  FASTQSequence qs;
  DNASequence ts;
  set_seq(qs, q, qlen);
  set_seq(ts, t, tlen);
  std::vector<int> scoreMat;
  std::vector<Arrow> pathMat;
  Alignment a;
  out[0] = KBandAlign(qs, ts, m, ins, del, k, scoreMat, pathMat, a, (AlignmentType)alignType, fn, false);
  out[1] = (int)a.qPos;
  out[2] = (int)a.tPos;
  return flatten(a, blocks, blockCap, nBlocks, gapCounts, gapListCap, nGapLists, gapItems, gapItemCap, nGapItems);
}

// ExtendAlignmentForward / ExtendAlignmentReverse (ExtendAlign.h:365-445) with the types and the score function
// AlignIntervals hands them (Blasr.cpp:1826-1838, 1885-1897).  The two matrices are the mapper's reusable buffers:
// they arrive here large and full of leftovers, which must not show in the result.  out[3] = score, qPos, tPos.
int ref_extend_align(const unsigned char *q, int qlen, int queryPos, const unsigned char *t, int tlen, int refPos,
                     const int *matrix25, int ins, int del, int k, int minExtendNBases, int maxNDrops, int reverse,
                     int *out, int *blocks, int blockCap, int *nBlocks, int *gapCounts, int gapListCap, int *nGapLists,
                     int *gapItems, int gapItemCap, int *nGapItems) {
  ScoreFn fn;
  make_score_fn(fn, matrix25, ins, del, 0, 0);
  FASTQSequence qs;
  DNASequence ts;
  set_seq(qs, q, qlen);
  set_seq(ts, t, tlen);
  std::vector<int> scoreMat((size_t)(qlen + 2) * (2 * k + 2) + 977, -123456);
  std::vector<Arrow> pathMat(scoreMat.size(), Left);
  Alignment a;
  if (reverse)
    out[0] = ExtendAlignmentReverse(qs, queryPos, ts, refPos, k, scoreMat, pathMat, a, fn, minExtendNBases, maxNDrops);
  else
    out[0] = ExtendAlignmentForward(qs, queryPos, ts, refPos, k, scoreMat, pathMat, a, fn, minExtendNBases, maxNDrops);
  out[1] = (int)a.qPos;
  out[2] = (int)a.tPos;
  out[3] = out[4] = -1;
  return flatten(a, blocks, blockCap, nBlocks, gapCounts, gapListCap, nGapLists, gapItems, gapItemCap, nGapItems);
}


}  // extern "C"
