// oracle/ref_mapper_harness.cpp -- TEST INFRASTRUCTURE, never on the product path.
//
// Compiles the reference mapper's own translation unit (alignment/Blasr.cpp, where it lies under
// $(BLASR_REF); nothing is copied) with its main() renamed, and exposes the two host-side pieces of
// SURVEY.md section 8 that only exist inside that file behind a C shim:
//   a-1  the window / strand set-up of AlignIntervals  (Blasr.cpp:1152-1336), followed by the
//        default SDPAlign + ComputeAlignmentStats call (:1716-1724)
//   a-10 SortAlignmentPointersByScore (:226-236), RemoveLowQualityAlignments (:2022-2079),
//        RemoveOverlappingAlignments (:2081-2161)
// Output: oracle/_ref/libblasr_refmapper.so (git-ignored, travels to the GPU box prebuilt).
#define main blasr_reference_main
#include "Blasr.cpp"
#undef main

#include <cstring>

extern "C" {

// One interval through the reference's AlignIntervals.  contigStarts: nContigs+1 offsets (seqdb
// seqStartPos) or NULL for a single-contig genome.  out[16]:
//  0 tAlignedSeqPos 1 tAlignedSeqLength 2 tStrand 3 tIndex 4 tLength 5 qAlignedSeqPos
//  6 qAlignedSeqLength 7 qPos 8 tPos 9 nBlocks 10 score 11 nMatch 12 nMismatch 13 nIns 14 nDel
//  15 qLength
// The interval's chained anchors for the entry point below (WeightedInterval::matches, what -rbao and -sdpbypass read):
// n triplets (t, q, l) in the coordinates FindMaxIncreasingInterval leaves them in, or none.  Thread-local: set, call, clear.
static __thread const unsigned *g_matches = NULL;
static __thread int g_nMatches = 0;
static __thread int g_refineBetweenAnchorsOnly = 0;
void ref_set_interval_matches(const unsigned *tql, int n, int refineBetweenAnchorsOnly) {
  g_matches = tql;
  g_nMatches = n;
  g_refineBetweenAnchorsOnly = refineBetweenAnchorsOnly;
}

// What RefineAlignment (Blasr.cpp:799-1047) leaves in the candidate when it runs with useGuidedAlign == false
// (-noUseGuidedAlign: ComputeDrift, the trimming, the k rule, PairwiseLocalAlign -> AffineKBandAlign).  Thread-local
// request for the entry point below: set, call ref_align_interval_ex, read.  out[12]: tAlignedSeqPos,
// tAlignedSeqLength, qAlignedSeqPos, qAlignedSeqLength, qPos, tPos, nBlocks, nGapLists, nMatch, nMismatch, nIns, nDel
// -- NOT the score: PairwiseLocalAlign's score function has no matrix (:689-690), the value is uninitialised memory.
struct RefineRequest {
  int enabled, bandSize;
  int *out;
  float *pct;
  int *blocks, blocksCap, *gapCounts, gapListCap, *gapItems, gapItemCap, nGapItems;
  int rc;
};
static __thread RefineRequest g_refine = {0, 0, NULL, NULL, NULL, 0, NULL, 0, NULL, 0, 0, 0};
void ref_request_nonguided_refine(int bandSize, int *out, float *pct, int *blocks, int blocksCap, int *gapCounts,
                                  int gapListCap, int *gapItems, int gapItemCap) {
  g_refine.enabled = out != NULL;
  g_refine.bandSize = bandSize;
  g_refine.out = out;
  g_refine.pct = pct;
  g_refine.blocks = blocks;
  g_refine.blocksCap = blocksCap;
  g_refine.gapCounts = gapCounts;
  g_refine.gapListCap = gapListCap;
  g_refine.gapItems = gapItems;
  g_refine.gapItemCap = gapItemCap;
  g_refine.nGapItems = 0;
  g_refine.rc = 0;
}
int ref_nonguided_refine_status(int *nGapItems) {
  *nGapItems = g_refine.nGapItems;
  return g_refine.rc;
}

// extendSwitches: NULL, or {extendAlignments, extendBandSize, maxExtendDropoff} (-extend, Blasr.cpp:1749-1940)
int ref_align_interval_ex(const unsigned char *genomeSeq, unsigned genomeLen, const unsigned char *readSeq,
                          unsigned readLen, unsigned subreadStart, unsigned subreadEnd, unsigned ivStart,
                          unsigned ivEnd, unsigned ivQStart, unsigned ivQEnd, int strand, const unsigned *contigStarts,
                          int nContigs, float approximateMaxInsertionRate, int extendEnds, int sdpTupleSize,
                          const int *extendSwitches, int *out, unsigned char *tSeqOut, int tSeqCap, int *blocksOut,
                          int blocksCap) {
  MappingParameters params;
  params.MakeSane();
  params.approximateMaxInsertionRate = approximateMaxInsertionRate;
  params.extendEnds = extendEnds != 0;
  params.sdpTupleSize = sdpTupleSize;
  if (extendSwitches != NULL) {
    params.extendAlignments = extendSwitches[0] != 0;
    params.extendBandSize = extendSwitches[1];
    params.maxExtendDropoff = extendSwitches[2];
  }

  T_GenomeSequence genome;
  genome.seq = (Nucleotide *)genomeSeq;
  genome.length = genomeLen;
  genome.deleteOnExit = false;
  genome.title = (char *)"genome";
  genome.titleLength = 6;

  T_Sequence read, readRC;
  read.seq = (Nucleotide *)readSeq;
  read.length = readLen;
  read.deleteOnExit = false;
  read.title = (char *)"read";
  read.titleLength = 4;
  read.subreadStart = subreadStart;
  read.subreadEnd = subreadEnd;
  read.MakeRC(readRC);
  readRC.subreadStart = read.MakeRCCoordinate(read.subreadEnd - 1);
  readRC.subreadEnd = read.MakeRCCoordinate(read.subreadStart) + 1;

  SequenceIndexDatabase<FASTASequence> seqdb;
  if (contigStarts != NULL) {
    seqdb.seqStartPos = const_cast<DNALength *>(contigStarts);
    seqdb.nSeqPos = nContigs + 1;
    seqdb.names = new char *[nContigs];
    seqdb.nameLengths = new int[nContigs];
    seqdb.deleteStructures = false;
    for (int i = 0; i < nContigs; i++) {
      seqdb.names[i] = (char *)"c";
      seqdb.nameLengths[i] = 1;
    }
  }

  WeightedIntervalSet<ChainedMatchPos> intervals;
  WeightedInterval<ChainedMatchPos> iv(ivEnd - ivStart, ivStart, ivEnd, strand, 0.0);
  iv.qStart = ivQStart;
  iv.qEnd = ivQEnd;
  iv.qLength = readLen;
  for (int m = 0; m < g_nMatches; m++)
    iv.matches.push_back(ChainedMatchPos(g_matches[3 * m], g_matches[3 * m + 1], g_matches[3 * m + 2]));
  if (g_refineBetweenAnchorsOnly) params.refineBetweenAnchorsOnly = true;  // -rbao (Blasr.cpp:1354-1712)
  intervals.insert(iv);

  vector<T_AlignmentCandidate *> alignments(1);
  alignments[0] = new T_AlignmentCandidate;
  MappingBuffers buffers;
  AlignIntervals(genome, read, readRC, intervals, SMRTDistanceMatrix, params.indel, params.indel, params.sdpTupleSize,
                 contigStarts != NULL, seqdb, alignments, params, params.useScoreCutoff, params.maxScore, buffers, 0);
  T_AlignmentCandidate &a = *alignments[0];
  out[0] = (int)a.tAlignedSeqPos;
  // (under -extend the field is left at the SDP extent unless the reverse extension was taken, :1770 / :1937: report
  //  the length of the string itself there)
  out[1] = extendSwitches != NULL ? (int)a.tAlignedSeq.length : (int)a.tAlignedSeqLength;
  out[2] = a.tStrand;
  out[3] = a.tIndex;
  out[4] = (int)a.tLength;
  out[5] = (int)a.qAlignedSeqPos;
  out[6] = (int)a.qAlignedSeqLength;
  out[7] = (int)a.qPos;
  out[8] = (int)a.tPos;
  out[9] = (int)a.blocks.size();
  out[10] = a.score;
  out[11] = a.nMatch;
  out[12] = a.nMismatch;
  out[13] = a.nIns;
  out[14] = a.nDel;
  out[15] = (int)a.qLength;
  int rc = 0;
  if ((int)a.tAlignedSeq.length > tSeqCap || (int)a.blocks.size() > blocksCap) rc = -1;
  else {
    memcpy(tSeqOut, a.tAlignedSeq.seq, a.tAlignedSeq.length);
    for (size_t b = 0; b < a.blocks.size(); b++) {
      blocksOut[3 * b] = (int)a.blocks[b].qPos;
      blocksOut[3 * b + 1] = (int)a.blocks[b].tPos;
      blocksOut[3 * b + 2] = (int)a.blocks[b].length;
    }
  }
  if (g_refine.enabled) {  // RefineAlignments :2164-2181 hands the forward read to RefineAlignment for every candidate
    params.useGuidedAlign = false;
    params.bandSize = g_refine.bandSize;
    RefineAlignment(read, genome, a, params, buffers);
    int *o = g_refine.out;
    o[0] = (int)a.tAlignedSeqPos;
    o[1] = (int)a.tAlignedSeqLength;
    o[2] = (int)a.qAlignedSeqPos;
    o[3] = (int)a.qAlignedSeqLength;
    o[4] = (int)a.qPos;
    o[5] = (int)a.tPos;
    o[6] = (int)a.blocks.size();
    o[7] = (int)a.gaps.size();
    o[8] = a.nMatch;
    o[9] = a.nMismatch;
    o[10] = a.nIns;
    o[11] = a.nDel;
    *g_refine.pct = a.pctSimilarity;
    size_t items = 0;
    for (size_t g = 0; g < a.gaps.size(); g++) items += a.gaps[g].size();
    if ((int)a.blocks.size() > g_refine.blocksCap || (int)a.gaps.size() > g_refine.gapListCap ||
        (int)items > g_refine.gapItemCap)
      g_refine.rc = -1;
    else {
      for (size_t b = 0; b < a.blocks.size(); b++) {
        g_refine.blocks[3 * b] = (int)a.blocks[b].qPos;
        g_refine.blocks[3 * b + 1] = (int)a.blocks[b].tPos;
        g_refine.blocks[3 * b + 2] = (int)a.blocks[b].length;
      }
      int at = 0;
      for (size_t g = 0; g < a.gaps.size(); g++) {
        g_refine.gapCounts[g] = (int)a.gaps[g].size();
        for (size_t i = 0; i < a.gaps[g].size(); i++) {
          g_refine.gapItems[2 * at] = (int)a.gaps[g][i].seq;
          g_refine.gapItems[2 * at + 1] = a.gaps[g][i].length;
          at++;
        }
      }
      g_refine.nGapItems = at;
    }
  }
  delete alignments[0];
  readRC.Free();
  if (contigStarts != NULL) {
    delete[] seqdb.names;
    delete[] seqdb.nameLengths;
    seqdb.names = NULL;
    seqdb.nameLengths = NULL;
    seqdb.seqStartPos = NULL;
  }
  return rc;
}

int ref_align_interval(const unsigned char *genomeSeq, unsigned genomeLen, const unsigned char *readSeq,
                       unsigned readLen, unsigned subreadStart, unsigned subreadEnd, unsigned ivStart,
                       unsigned ivEnd, unsigned ivQStart, unsigned ivQEnd, int strand, const unsigned *contigStarts,
                       int nContigs, float approximateMaxInsertionRate, int extendEnds, int sdpTupleSize, int *out,
                       unsigned char *tSeqOut, int tSeqCap, int *blocksOut, int blocksCap) {
  return ref_align_interval_ex(genomeSeq, genomeLen, readSeq, readLen, subreadStart, subreadEnd, ivStart, ivEnd, ivQStart,
                               ivQEnd, strand, contigStarts, nContigs, approximateMaxInsertionRate, extendEnds, sdpTupleSize,
                               NULL, out, tSeqOut, tSeqCap, blocksOut, blocksCap);
}

// The post filters on n candidates given as parallel arrays.  op: 0 = std::sort with
// SortAlignmentPointersByScore, 1 = RemoveLowQualityAlignments, 2 = RemoveOverlappingAlignments.
// keptIdx / keptScore receive the surviving candidates in their final order; returns their number.
int ref_post_filter(int op, int n, const int *score, const unsigned *tPos, const unsigned *tAlignedSeqPos,
                    const int *tIndex, const int *tStrand, const float *pctSimilarity, const int *blocks,
                    const int *blkOff, const int *nBlocks, int nCandidates, int maxGap, int maxScore,
                    float minPctIdentity, int *keptIdx, int *keptScore) {
  MappingParameters params;
  params.MakeSane();
  params.nCandidates = nCandidates;
  params.maxGap = maxGap;
  params.maxScore = maxScore;
  params.minPctIdentity = minPctIdentity;
  vector<T_AlignmentCandidate *> ptrs(n);
  for (int i = 0; i < n; i++) {
    T_AlignmentCandidate *a = new T_AlignmentCandidate;
    a->score = score[i];
    a->tPos = tPos[i];
    a->tAlignedSeqPos = tAlignedSeqPos[i];
    a->tIndex = tIndex[i];
    a->tStrand = tStrand[i];
    a->pctSimilarity = pctSimilarity[i];
    a->nCells = i;  // carries the candidate's identity through the filters
    for (int b = 0; b < nBlocks[i]; b++) {
      Block blk;
      blk.qPos = blocks[3 * (blkOff[i] + b)];
      blk.tPos = blocks[3 * (blkOff[i] + b) + 1];
      blk.length = blocks[3 * (blkOff[i] + b) + 2];
      a->blocks.push_back(blk);
    }
    ptrs[i] = a;
  }
  if (op == 0) std::sort(ptrs.begin(), ptrs.end(), SortAlignmentPointersByScore());
  else if (op == 1) {
    T_Sequence read;
    RemoveLowQualityAlignments(read, ptrs, params);
  } else
    RemoveOverlappingAlignments(ptrs, params);
  const int kept = (int)ptrs.size();
  for (int i = 0; i < kept; i++) {
    keptIdx[i] = ptrs[i]->nCells;
    keptScore[i] = ptrs[i]->score;
    delete ptrs[i];
  }
  return kept;
}

}  // extern "C"

// ---- output epilogue -------------------------------------------------------
// An alignment for the printers from the refine stage's op list ((len << 2) | M 0 / I 1 / D 2) or from blocks.
static void fill_alignment(T_AlignmentCandidate &a, const unsigned char *q, const unsigned char *t, int qPos, int tPos,
                           const unsigned *ops, long nOps, const int *blocks, long nBlocks) {
  a.qAlignedSeq.seq = (Nucleotide *)q;
  a.tAlignedSeq.seq = (Nucleotide *)t;
  a.qAlignedSeq.deleteOnExit = a.tAlignedSeq.deleteOnExit = false;
  a.qPos = qPos;
  a.tPos = tPos;
  if (ops != NULL) {
    int qq = 0, tt = 0;
    a.gaps.push_back(GapList());
    for (long i = 0; i < nOps; i++) {
      const int len = (int)(ops[i] >> 2), ty = (int)(ops[i] & 3u);
      if (ty == 0) {
        Block b;
        b.qPos = qq;
        b.tPos = tt;
        b.length = len;
        a.blocks.push_back(b);
        a.gaps.push_back(GapList());
        qq += len;
        tt += len;
      } else if (ty == 2) {
        a.gaps.back().push_back(Gap(Gap::Query, len));
        tt += len;
      } else {
        a.gaps.back().push_back(Gap(Gap::Target, len));
        qq += len;
      }
    }
  } else {
    for (long b = 0; b < nBlocks; b++) {
      Block blk;
      blk.qPos = blocks[3 * b];
      blk.tPos = blocks[3 * b + 1];
      blk.length = blocks[3 * b + 2];
      a.blocks.push_back(blk);
    }
  }
}

extern "C" {

// AssignRefContigLocation on an alignment with the given window; out = {contig, tAlignedSeqPos afterwards}
int ref_contig_location(unsigned tAlignedSeqPos, unsigned tAlignedSeqLength, int tStrand, const unsigned *seqStartPos,
                        int nSeqPos, unsigned genomeLength, unsigned *out) {
  SequenceIndexDatabase<FASTQSequence> seqdb;
  seqdb.seqStartPos = const_cast<DNALength *>(seqStartPos);
  seqdb.nSeqPos = nSeqPos;
  seqdb.deleteStructures = false;
  DNASequence genome;
  genome.length = genomeLength;
  T_AlignmentCandidate a;
  a.tAlignedSeqPos = tAlignedSeqPos;
  a.tAlignedSeqLength = tAlignedSeqLength;
  a.tStrand = tStrand;
  out[0] = (unsigned)seqdb.SearchForIndex(tStrand == 0 ? tAlignedSeqPos
                                                        : genome.MakeRCCoordinate(tAlignedSeqPos + tAlignedSeqLength - 1));
  AssignRefContigLocation(a, seqdb, genome);
  out[1] = a.tAlignedSeqPos;
  seqdb.seqStartPos = NULL;
  return 0;
}

// SAMOutput::CreateCIGARString; clips = {prefixSoft, suffixSoft, prefixHard, suffixHard}; returns the length
int ref_cigar(const unsigned char *q, const unsigned char *t, int qPos, int tPos, const unsigned *ops, long nOps,
              const int *blocks, long nBlocks, int clipping, int tStrand, unsigned readLength, unsigned lowQualityPrefix,
              unsigned lowQualitySuffix, unsigned subreadStart, unsigned subreadEnd, unsigned qAlignStart, char *out,
              int outCap, unsigned *clips) {
  T_AlignmentCandidate a;
  fill_alignment(a, q, t, qPos, tPos, ops, nOps, blocks, nBlocks);
  a.tStrand = tStrand;
  a.qAlignedSeqPos = qAlignStart - (unsigned)qPos;  // QAlignStart() = qPos + qAlignedSeqPos
  SMRTSequence read;
  read.length = readLength;
  read.lowQualityPrefix = lowQualityPrefix;
  read.lowQualitySuffix = lowQualitySuffix;
  read.subreadStart = subreadStart;
  read.subreadEnd = subreadEnd;
  string cigar;
  DNALength ps, ss, ph, sh;
  SAMOutput::CreateCIGARString(a, read, cigar, (SAMOutput::Clipping)clipping, ps, ss, ph, sh);
  clips[0] = ps;
  clips[1] = ss;
  clips[2] = ph;
  clips[3] = sh;
  clips[4] = a.QAlignEnd();
  if ((int)cigar.size() + 1 > outCap) return -1;
  memcpy(out, cigar.c_str(), cigar.size() + 1);
  return (int)cigar.size();
}

// IntervalAlignmentPrinter::Print
int ref_m4_line(const char *qName, const char *tName, int score, float pctSimilarity, int qStrand, unsigned qAlignedSeqPos,
                unsigned qPos, unsigned qLength, int tStrand, unsigned tAlignedSeqPos, unsigned tPos, unsigned tLength,
                int mapQV, const int *blocks, long nBlocks, char *out, int outCap, unsigned *derived) {
  T_AlignmentCandidate a;
  static unsigned char dummy[1] = {0};
  fill_alignment(a, dummy, dummy, (int)qPos, (int)tPos, NULL, 0, blocks, nBlocks);
  a.qName = qName;
  a.tName = tName;
  a.score = score;
  a.pctSimilarity = pctSimilarity;
  a.qStrand = qStrand;
  a.tStrand = tStrand;
  a.qAlignedSeqPos = qAlignedSeqPos;
  a.tAlignedSeqPos = tAlignedSeqPos;
  a.qLength = qLength;
  a.tLength = tLength;
  a.mapQV = mapQV;
  stringstream strm;
  IntervalAlignmentPrinter::Print(a, strm);
  const string s = strm.str();
  derived[0] = a.QAlignStart();
  derived[1] = a.QAlignEnd();
  derived[2] = a.TAlignStart();
  if ((int)s.size() + 1 > outCap) return -1;
  memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}

}  // extern "C"

extern "C" {

// StoreMapQVs on n alignments of one read given as parallel arrays (op lists concatenated, opOff[n+1]).
// qvs: read.substitutionQV or NULL.  mapQV / probScore receive the results.
int ref_store_map_qvs(int n, const unsigned char *const *qSeqs, const unsigned char *const *tSeqs, const int *qPos,
                      const int *tPos, const unsigned *ops, const long *opOff, const unsigned *qAlignStart,
                      const int *qStrand, const unsigned *qLength, const int *nMatch, unsigned readLength,
                      const unsigned char *qvs, int ins, int del, int affineOpen, int affineExtend, int affineAlign,
                      float minFraction, int *mapQV, float *probScore) {
  MappingParameters params;
  params.MakeSane();
  params.insertion = ins;
  params.deletion = del;
  params.affineOpen = affineOpen;
  params.affineExtend = affineExtend;
  params.affineAlign = affineAlign != 0;
  params.minFractionToBeConsideredOverlapping = minFraction;
  vector<T_AlignmentCandidate *> ptrs(n);
  for (int i = 0; i < n; i++) {
    T_AlignmentCandidate *a = new T_AlignmentCandidate;
    fill_alignment(*a, qSeqs[i], tSeqs[i], qPos[i], tPos[i], ops + opOff[i], opOff[i + 1] - opOff[i], NULL, 0);
    a->qAlignedSeqPos = qAlignStart[i] - (unsigned)qPos[i];
    a->qStrand = qStrand[i];
    a->qLength = qLength[i];
    a->nMatch = nMatch[i];
    ptrs[i] = a;
  }
  SMRTSequence read;
  read.length = readLength;
  if (qvs != NULL) {
    read.substitutionQV.data = const_cast<unsigned char *>(qvs);
  }
  MappingBuffers buffers;
  StoreMapQVs(read, ptrs, params, buffers);
  read.substitutionQV.data = NULL;
  for (int i = 0; i < n; i++) {
    mapQV[i] = ptrs[i]->mapQV;
    probScore[i] = ptrs[i]->probScore;
    delete ptrs[i];
  }
  return 0;
}

}  // extern "C"

extern "C" {

// SAMOutput::PrintAlignment.  iv: 0 qPos 1 tPos 2 clipping 3 qStrand 4 tStrand 5 isPrimary 6 readLength
// 7 lowQualityPrefix 8 lowQualitySuffix 9 subreadStart 10 subreadEnd 11 qAlignedSeqPos 12 qAlignedSeqLength
// 13 tAlignedSeqPos 14 tLength 15 qLength 16 mapQV 17 score 18 editDist 19 holeNumber 20 qs 21 qe 22 numPasses
// 23 subreadContext.  fv: readQuality, snr[4].  derived: QAlignStart, QAlignEnd, TAlignStart + blocks[0].tPos, TEnd.
int ref_sam_line(const char *qName, const char *tName, const char *readGroupId, const unsigned char *readSeq,
                 const unsigned char *readQual, const unsigned *ops, long nOps, const int *blocks, long nBlocks,
                 const int *iv, const float *fv, char *out, int outCap, unsigned *derived) {
  T_AlignmentCandidate a;
  static unsigned char dummy[1] = {0};
  fill_alignment(a, dummy, dummy, iv[0], iv[1], ops, nOps, blocks, nBlocks);
  // the CIGAR compares the aligned sequences: give it the read itself and the same bases as target (all '=')
  a.qName = qName;
  a.tName = tName;
  a.qStrand = iv[3];
  a.tStrand = iv[4];
  a.qAlignedSeqPos = iv[11];
  a.qAlignedSeq.length = iv[12];
  a.tAlignedSeqPos = iv[13];
  a.tLength = iv[14];
  a.qLength = iv[15];
  a.mapQV = iv[16];
  a.score = iv[17];
  SMRTSequence read;
  read.seq = const_cast<unsigned char *>(readSeq);
  read.length = iv[6];
  read.deleteOnExit = false;
  if (readQual != NULL) read.qual.data = const_cast<unsigned char *>(readQual);
  read.lowQualityPrefix = iv[7];
  read.lowQualitySuffix = iv[8];
  read.subreadStart = iv[9];
  read.subreadEnd = iv[10];
  read.holeNumber = iv[19];
  read.qs = iv[20];
  read.qe = iv[21];
  read.numPasses = iv[22];
  read.subreadContext = iv[23];
  read.readQuality = fv[0];
  for (int i = 0; i < 4; i++) read.snr[i] = fv[1 + i];
  // aligned sequences for the = / X split: the bases at the alignment's own coordinates
  vector<unsigned char> qa((size_t)iv[0] + a.QEnd() + 1, 'A'), ta((size_t)iv[1] + a.TEnd() + 1, 'A');
  a.qAlignedSeq.seq = &qa[0];
  a.tAlignedSeq.seq = &ta[0];
  AlignmentContext context;
  context.isPrimary = iv[5] != 0;
  context.readGroupId = readGroupId;
  context.editDist = iv[18];
  SupplementalQVList qvlist;
  stringstream strm;
  SAMOutput::PrintAlignment(a, read, strm, context, qvlist, (SAMOutput::Clipping)iv[2]);
  derived[0] = a.QAlignStart();
  derived[1] = a.QAlignEnd();
  derived[2] = a.TAlignStart() + a.blocks[0].tPos;
  derived[3] = a.TEnd();
  read.seq = NULL;
  read.qual.data = NULL;
  a.qAlignedSeq.seq = a.tAlignedSeq.seq = NULL;
  const string s = strm.str();
  if ((int)s.size() + 1 > outCap) return -1;
  memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}

// SAMOutput::PrintUnalignedRead (printers/SAMPrinter.h:578-664).  iv: 0 readLength 1 subreadStart 2 subreadEnd 3 clipping
// 4 holeNumber 5 qs 6 qe 7 numPasses 8 subreadContext; fv: readQuality, snr[4].
int ref_sam_unaligned_line(const char *title, const char *readGroupId, const unsigned char *readSeq, const int *iv,
                           const float *fv, char *out, int outCap) {
  SMRTSequence read;
  read.seq = const_cast<unsigned char *>(readSeq);
  read.length = iv[0];
  read.deleteOnExit = false;
  read.subreadStart = iv[1];
  read.subreadEnd = iv[2];
  read.holeNumber = iv[4];
  read.qs = iv[5];
  read.qe = iv[6];
  read.numPasses = iv[7];
  read.subreadContext = iv[8];
  read.readQuality = fv[0];
  for (int i = 0; i < 4; i++) read.snr[i] = fv[1 + i];
  read.CopyTitle(string(title));
  AlignmentContext context;
  context.readGroupId = readGroupId;
  SupplementalQVList qvlist;
  stringstream strm;
  SAMOutput::PrintUnalignedRead(read, strm, context, qvlist, (SAMOutput::Clipping)iv[3]);
  read.seq = NULL;
  const string s = strm.str();
  if ((int)s.size() + 1 > outCap) return -1;
  memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}

}  // extern "C"

extern "C" {

// Alignment::OrderGapsByType on the alignment of an op list, turned back into an op list; returns its length
long ref_order_gaps_by_type(const unsigned *ops, long nOps, unsigned *out) {
  T_AlignmentCandidate a;
  static unsigned char dummy[1] = {0};
  fill_alignment(a, dummy, dummy, 0, 0, ops, nOps, NULL, 0);
  a.qAlignedSeq.seq = a.tAlignedSeq.seq = NULL;
  a.OrderGapsByType();
  long n = 0;
  for (size_t b = 0; b < a.blocks.size(); b++) {
    out[n++] = ((unsigned)a.blocks[b].length << 2) | 0u;
    if (b + 1 < a.gaps.size())
      for (size_t g = 0; g < a.gaps[b + 1].size(); g++)
        out[n++] = ((unsigned)a.gaps[b + 1][g].length << 2) | (a.gaps[b + 1][g].seq == Gap::Target ? 1u : 2u);
  }
  return n;
}

}  // extern "C"

// ---- next-4: the reference's own readers (FASTAReader, SuffixArray::Read), for the parity of
// blasr_amd/csrc/formats_api.hip ----
extern "C" {

// FASTAReader::ReadAllSequencesIntoOne + genome.ToUpper() as Blasr.cpp reads a genome (:4329, :4354).
// Returns the length (or -1 when the buffers are too small); starts: nSeq + 1 entries; titles: the records' titles
// joined by '\n'.
long ref_read_genome(const char *path, unsigned char *seqOut, long seqCap, unsigned *starts, int startsCap, char *titles,
                     long titlesCap, int *nSeqOut) {
  string name(path);
  FASTAReader reader;
  reader.Initialize(name);
  FASTASequence genome;
  SequenceIndexDatabase<FASTASequence> seqdb;
  reader.ReadAllSequencesIntoOne(genome, &seqdb);
  genome.ToUpper();
  const int nSeq = seqdb.nSeqPos - 1;
  *nSeqOut = nSeq;
  if ((long)genome.length > seqCap || nSeq + 1 > startsCap) return -1;
  memcpy(seqOut, genome.seq, genome.length);
  string all;
  for (int i = 0; i <= nSeq; i++) starts[i] = seqdb.seqStartPos[i];
  for (int i = 0; i < nSeq; i++) {
    all += string(seqdb.names[i]);
    all += "\n";
  }
  if ((long)all.size() + 1 > titlesCap) return -1;
  memcpy(titles, all.c_str(), all.size() + 1);
  return genome.length;
}

// FASTAReader::GetNext over a file: sequence lengths and bytes back to back; returns the number of records
int ref_read_fasta_records(const char *path, unsigned char *seqOut, long seqCap, long *lengths, int maxRecords) {
  string name(path);
  FASTAReader reader;
  reader.Initialize(name);
  FASTASequence seq;
  int n = 0;
  long at = 0;
  while (reader.GetNext(seq)) {
    if (n >= maxRecords || at + (long)seq.length > seqCap) return -1;
    memcpy(seqOut + at, seq.seq, seq.length);
    at += seq.length;
    lengths[n++] = seq.length;
  }
  return n;
}

// FASTQReader::GetNext(FASTQSequence&) over a file (common/FASTQReader.h:27-66): titles joined by '\n', bases and quality
// characters back to back, lengths; returns the number of records (-1: buffers too small).  NB the reference exit(0)s
// on a record without its '+' line: the tests call this in a forked child.
int ref_read_fastq_records(const char *path, unsigned char *seqOut, unsigned char *qualOut, long seqCap, long *lengths,
                           int maxRecords, char *titles, long titlesCap) {
  string name(path);
  FASTQReader reader;
  reader.Initialize(name);
  int n = 0;
  long at = 0;
  string all;
  for (;;) {
    FASTQSequence seq;
    if (!reader.GetNext(seq)) break;
    if (n >= maxRecords || at + (long)seq.length > seqCap) return -1;
    memcpy(seqOut + at, seq.seq, seq.length);
    if (seq.length > 0) memcpy(qualOut + at, seq.qual.data, seq.length);
    at += seq.length;
    lengths[n++] = seq.length;
    all += seq.title ? string(seq.title, seq.titleLength) : string();
    all += "\n";
  }
  if ((long)all.size() + 1 > titlesCap) return -1;
  memcpy(titles, all.c_str(), all.size() + 1);
  return n;
}

// ReaderAgglomerate::GetNext(SMRTSequence&) over a `.sam` read file (files/ReaderAgglomerate.h:198-203, :308-318): titles
// joined by '\n', bases and STORED quality values back to back (0 where the read has none), lengths, hasQual flags; returns
// the number of records (-1: buffers too small).  NB the reference exit(1)s on a line it cannot parse: forked child.
int ref_read_sam_reads(const char *path, unsigned char *seqOut, unsigned char *qualOut, long seqCap, long *lengths, int *hasQual,
                       int maxRecords, char *titles, long titlesCap) {
  string name(path);
  ReaderAgglomerate reader;
  reader.SetReadFileName(name);
  if (reader.Initialize() == 0) return -3;
  int n = 0;
  long at = 0;
  string all;
  for (;;) {
    SMRTSequence seq;
    if (!reader.GetNext(seq)) break;
    if (n >= maxRecords || at + (long)seq.length > seqCap) return -1;
    memcpy(seqOut + at, seq.seq, seq.length);
    hasQual[n] = seq.qual.data != NULL;
    for (unsigned i = 0; i < seq.length; i++) qualOut[at + i] = seq.qual.data ? seq.qual.data[i] : 0;
    at += seq.length;
    if (seq.subreadStart != 0 || seq.subreadEnd != seq.length) return -2;
    lengths[n++] = seq.length;
    all += seq.title ? string(seq.title, seq.titleLength) : string();
    all += "\n";
  }
  if ((long)all.size() + 1 > titlesCap) return -1;
  memcpy(titles, all.c_str(), all.size() + 1);
  return n;
}

// SuffixArray::Read: the index array and the lookup table of a .sa file.  out[0] = length, out[1] = lookupTableLength,
// out[2] = lookupPrefixLength
int ref_read_sa(const char *path, unsigned *index, long indexCap, unsigned *startPos, unsigned *endPos, long tableCap,
                long *out) {
  DNASuffixArray sa;
  string name(path);
  if (!sa.Read(name)) return -1;
  out[0] = sa.length;
  out[1] = sa.lookupTableLength;
  out[2] = sa.lookupPrefixLength;
  if ((long)sa.length > indexCap || (long)sa.lookupTableLength > tableCap) return -2;
  memcpy(index, sa.index, sizeof(unsigned) * sa.length);
  memcpy(startPos, sa.startPosTable, sizeof(unsigned) * sa.lookupTableLength);
  memcpy(endPos, sa.endPosTable, sizeof(unsigned) * sa.lookupTableLength);
  return 0;
}

// next-1 (device half): the reference's MapReadToGenome (algorithms/anchoring/MapBySuffixArray.h:217-311) on one
// strand of one read, against a suffix array read by SuffixArray::Read.  ref_sa_open keeps the array for later calls.
void *ref_sa_open(const char *path) {
  DNASuffixArray *sa = new DNASuffixArray;
  string name(path);
  if (!sa->Read(name)) {
    delete sa;
    return NULL;
  }
  return sa;
}
void ref_sa_close(void *h) { delete (DNASuffixArray *)h; }
// The same object over arrays that are already in memory (the caller keeps them alive): what SuffixArray::Read leaves
// behind, without the file (:577-583 for the tuple metrics).
void *ref_sa_adopt(unsigned *index, unsigned length, unsigned *startPosTable, unsigned *endPosTable,
                   unsigned lookupTableLength, unsigned lookupPrefixLength) {
  DNASuffixArray *sa = new DNASuffixArray;
  sa->deleteStructures = false;
  sa->index = index;
  sa->length = length;
  sa->startPosTable = startPosTable;
  sa->endPosTable = endPosTable;
  sa->lookupTableLength = lookupTableLength;
  sa->lookupPrefixLength = lookupPrefixLength;
  sa->tm.Initialize(lookupPrefixLength);
  return sa;
}

// ap[7] = minMatchLength, expand, useLookupTable, maxAnchorsPerPosition, advanceExactMatches, maxLCPLength,
// stopMappingOnceUnique.  strand 1 maps the reverse complement with Blasr.cpp:3449-3450's subread coordinates.
// out: (t, q, l) triples in emission order; sorted != 0 applies SortMatchPosList.  Returns the count (the first cap
// triples are stored).
long ref_map_read_to_genome(void *h, const unsigned char *genomeSeq, unsigned genomeLen, const unsigned char *readSeq,
                            unsigned readLen, unsigned subreadStart, unsigned subreadEnd, int strand,
                            unsigned minPrefixMatchLength, const int *ap, int sorted, unsigned *out, long cap) {
  DNASuffixArray &sa = *(DNASuffixArray *)h;
  T_GenomeSequence genome;
  genome.seq = (Nucleotide *)genomeSeq;
  genome.length = genomeLen;
  genome.deleteOnExit = false;
  T_Sequence read, readRC;
  read.seq = (Nucleotide *)readSeq;
  read.length = readLen;
  read.deleteOnExit = false;
  read.subreadStart = subreadStart;
  read.subreadEnd = subreadEnd;
  read.MakeRC(readRC);
  readRC.subreadStart = readLen - subreadEnd;
  readRC.subreadEnd = readLen - subreadStart;
  AnchorParameters anchor;
  anchor.minMatchLength = ap[0];
  anchor.expand = ap[1];
  anchor.useLookupTable = ap[2] != 0;
  anchor.maxAnchorsPerPosition = ap[3];
  anchor.advanceExactMatches = ap[4];
  anchor.maxLCPLength = ap[5];
  anchor.stopMappingOnceUnique = ap[6] != 0;
  vector<ChainedMatchPos> matches;
  MapReadToGenome(genome, sa, strand ? readRC : read, minPrefixMatchLength, matches, anchor);
  if (sorted) SortMatchPosList(matches);
  for (long i = 0; i < (long)matches.size() && i < cap; i++) {
    out[3 * i] = matches[i].t;
    out[3 * i + 1] = matches[i].q;
    out[3 * i + 2] = matches[i].l;
  }
  return (long)matches.size();
}

// next-1, host half: the reference's own RemoveOverlappingAnchors + FindMaxIncreasingInterval (Forward, then Reverse)
// into one WeightedIntervalSet, as MapRead runs them with its default switches (Blasr.cpp:2342-2500).
// ref_ct_open builds the tuple count table as Blasr.cpp:4479-4482 does when no .ctab is given.
struct RefCountTable {
  TupleCountTable<T_GenomeSequence, DNATuple> ct;
};
void *ref_ct_open(const unsigned char *genomeSeq, unsigned genomeLen, int tupleSize) {
  RefCountTable *h = new RefCountTable;
  T_GenomeSequence genome;
  genome.seq = (Nucleotide *)genomeSeq;
  genome.length = genomeLen;
  genome.deleteOnExit = false;
  TupleMetrics tm;
  tm.Initialize(tupleSize);
  h->ct.InitCountTable(tm);
  h->ct.AddSequenceTupleCountsLR(genome);
  return h;
}
void ref_ct_close(void *h) { delete (RefCountTable *)h; }
// TupleCountTable::Write / Read on a file (the .ctab format)
int ref_ct_write(void *h, const char *path) {
  ofstream out(path, std::ios::out | std::ios::binary);
  if (!out.good()) return -1;
  ((RefCountTable *)h)->ct.Write(out);
  return 0;
}
void *ref_ct_read(const char *path) {
  ifstream in(path, std::ios::in | std::ios::binary);
  if (!in.good()) return NULL;
  RefCountTable *h = new RefCountTable;
  h->ct.Read(in);
  return h;
}
long ref_ct_table(void *h, int *out, long cap) {
  RefCountTable *c = (RefCountTable *)h;
  for (long i = 0; i < c->ct.countTableLength && i < cap; i++) out[i] = c->ct.countTable[i];
  return c->ct.countTableLength;
}

// fwd / rc: (t, q, l) triples sorted as SortMatchPosList leaves them; stale: 2 triples = the element the reference
// finds one past the end of each list's buffer (the vector is built one longer and shrunk, so that element is defined).
// iv[5] = nCandidates, minInterval, removeContainedIntervals, -, -; fv[2] = indelRate, maxPValue.
// out: (start, end, qStart, qEnd, strand) per interval in the set's order; pv: their p-values.  Returns the count.
long ref_find_intervals(void *ctHandle, const unsigned char *genomeSeq, unsigned genomeLen, const unsigned *seqStartPos,
                        int nSeqPos, unsigned readLen, unsigned subreadStart, unsigned subreadEnd, const unsigned *fwd,
                        long nFwd, const unsigned *rc, long nRc, const unsigned *stale, const int *iv, const float *fv,
                        unsigned *out, float *pv, long cap) {
  RefCountTable *c = (RefCountTable *)ctHandle;
  T_GenomeSequence genome;
  genome.seq = (Nucleotide *)genomeSeq;
  genome.length = genomeLen;
  genome.deleteOnExit = false;
  SequenceIndexDatabase<FASTASequence> seqdb;
  seqdb.seqStartPos = const_cast<DNALength *>(seqStartPos);
  seqdb.nSeqPos = nSeqPos;
  seqdb.deleteStructures = false;
  SeqBoundaryFtr<FASTASequence> seqBoundary(&seqdb);
  T_Sequence read;
  read.seq = NULL;
  read.length = readLen;
  read.deleteOnExit = false;
  read.subreadStart = subreadStart;
  read.subreadEnd = subreadEnd;

  vector<ChainedMatchPos> lists[2];
  const unsigned *src[2] = {fwd, rc};
  const long n[2] = {nFwd, nRc};
  for (int s = 0; s < 2; s++) {
    lists[s].reserve(n[s] + 1);
    for (long i = 0; i < n[s]; i++) lists[s].push_back(ChainedMatchPos(src[s][3 * i], src[s][3 * i + 1], src[s][3 * i + 2]));
    lists[s].push_back(ChainedMatchPos(stale[3 * s], stale[3 * s + 1], stale[3 * s + 2]));
    lists[s].resize(n[s]);  // the stale element stays in the buffer, as a longer earlier list would leave it
  }
  MappingParameters params;
  params.MakeSane();
  params.nCandidates = iv[0];
  params.minInterval = iv[1];
  params.removeContainedIntervals = iv[2] != 0;
  params.indelRate = fv[0];
  PValueWeightor lisPValue(read, genome, c->ct.tm, &c->ct);
  LISSizeWeightor<vector<ChainedMatchPos> > lisWeightFn;
  IntervalSearchParameters intervalSearchParameters;
  params.InitializeIntervalSearchParameters(intervalSearchParameters);
  intervalSearchParameters.maxPValue = fv[1];
  WeightedIntervalSet<ChainedMatchPos> topIntervals(params.nCandidates);
  MappingBuffers buffers;
  RemoveOverlappingAnchors(lists[0]);
  RemoveOverlappingAnchors(lists[1]);
  DNALength maxGapLength = 50000;  // Blasr.cpp:2402-2406
  DNALength intervalLength = (read.subreadEnd - read.subreadStart) * (1 + params.indelRate);
  if (intervalLength - (read.subreadEnd - read.subreadStart) > maxGapLength)
    intervalLength = (read.subreadEnd - read.subreadStart) + maxGapLength;
  FindMaxIncreasingInterval(Forward, lists[0], intervalLength, params.nCandidates, seqBoundary, lisPValue, lisWeightFn,
                            topIntervals, genome, read, intervalSearchParameters, &buffers.globalChainEndpointBuffer,
                            &buffers.sdpFragmentSet, (const char *)NULL);
  FindMaxIncreasingInterval(Reverse, lists[1], intervalLength, params.nCandidates, seqBoundary, lisPValue, lisWeightFn,
                            topIntervals, genome, read, intervalSearchParameters, &buffers.globalChainEndpointBuffer,
                            &buffers.sdpFragmentSet, (const char *)NULL);
  if (params.removeContainedIntervals) topIntervals.RemoveContained();
  long k = 0;
  for (WeightedIntervalSet<ChainedMatchPos>::iterator it = topIntervals.begin(); it != topIntervals.end(); ++it, ++k) {
    if (k >= cap) continue;
    out[5 * k] = it->start;
    out[5 * k + 1] = it->end;
    out[5 * k + 2] = it->qStart;
    out[5 * k + 3] = it->qEnd;
    out[5 * k + 4] = it->GetStrandIndex();
    pv[k] = it->pValue;
  }
  return k;
}

}  // extern "C"
