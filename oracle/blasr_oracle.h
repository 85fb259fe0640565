/* TEST INFRASTRUCTURE -- CPU restatement ("oracle") of BLASR's per-read refine
 * stage.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (blasr_amd/, include/blasr_hip.h) never
 * does.  Every function cites the reference file:line it restates (paths are
 * relative to the reference checkout, e.g. /root/reference).
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py checks every entry point
 * below against oracle/_ref/libblasr_ref.so (the reference's own templates,
 * compiled from /root/reference by oracle/Makefile) on seeded inputs, and
 * tests/golden/ holds vectors generated from that library.
 *
 * The flat-array entry points deliberately share their signatures with the
 * ref_* functions of oracle/ref_harness.cpp so that one ctypes prototype
 * drives both.
 */
#ifndef BLASR_ORACLE_H_
#define BLASR_ORACLE_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* AlignmentType values used on the path (AlignmentUtils.h:14-58). */
enum { BO_LOCAL = 0, BO_GLOBAL = 1, BO_END_ANCHORED = 6 };
/* Gap::GapSeq (AlignmentGapList.h:11): Query = gap in the query (deletion,
 * arrow Left), Target = gap in the target (insertion, arrow Up). */
enum { BO_GAP_QUERY = 0, BO_GAP_TARGET = 1 };

/* SDPAlign (SDPAlign.h:95-626) incl. recursion, SWAlign gap fills, chaining. */
int bo_sdp_align(const unsigned char *q, int qlen, const unsigned char *t, int tlen,
                 const int *matrix25, int ins, int del, int affOpen, int affExt, int wordSize,
                 int sdpIns, int sdpDel, float indelRate, int alignType, int detailed,
                 int extendFront, int sdpPrefixLength, int recurse, int noRecurseUnder,
                 int maxPerPos, int *outQPos, int *outTPos, int *blocks, int blockCap,
                 int *nBlocks);

/* SDPLongestCommonSubsequence (sdp/SparseDynamicProgramming.h:70-322). */
int bo_sdp_chain(int queryLength, const int *xyl, int nFrag, int fragmentLength, int ins, int del,
                 int match, int alignType, int *chain, int cap);

/* SWAlign (SWAlign.h:19-389), Global mode only. */
int bo_sw_align(const unsigned char *q, int qlen, const unsigned char *t, int tlen,
                const int *matrix25, int ins, int del, int alignType, int *outQPos, int *outTPos,
                int *blocks, int blockCap, int *nBlocks);

/* AlignmentToGuide + StoreMatrixOffsets + ComputeMatrixNElem (GuidedAlign.h:83-259). */
int bo_alignment_to_guide(const int *blocks, int nBlocks, int bandSize, int *rows, int rowCap,
                          int *nCells);

/* AffineGuidedAlign (AffineGuidedAlign.h:31-488) + ComputeAlignmentStats, as
 * RefineAlignment does (Blasr.cpp:862-878). */
int bo_affine_guided_align(const unsigned char *q, int qlen, const unsigned char *t, int tlen,
                           const int *guideBlocks, int nGuideBlocks, const int *matrix25, int ins,
                           int del, int affOpen, int affExt, int bandSize, int *outQPos,
                           int *outTPos, int *blocks, int blockCap, int *nBlocks, int *gapCounts,
                           int gapListCap, int *nGapLists, int *gapItems, int gapItemCap,
                           int *nGapItems, int *stats, float *pct);

/* ComputeAlignmentStats (AlignmentUtils.h:535-584). */
void bo_compute_alignment_stats(const unsigned char *q, const unsigned char *t, int qPos, int tPos,
                                const int *blocks, int nBlocks, const int *gapCounts,
                                int nGapLists, const int *gapItems, const int *matrix25, int ins,
                                int del, int affOpen, int affExt, int useAffine, int *stats,
                                float *pct);

/* libstdc++ (GCC 11.4) std::sort schedule on (x,y,len) triplets keyed by (x,y);
 * exposed so the tie behaviour of SDPAlign.h:256 can be tested directly. */
void bo_introsort_xy(int *xyl, int n);

/* The whole refine stage for one candidate, as AlignIntervals + RefineAlignment
 * run it with default parameters (Blasr.cpp:1716-1724, 850-914):
 * SDPAlign(Local) -> ComputeAlignmentStats -> AffineGuidedAlign -> stats.
 * out[0..11] = sdpScore, sdpStatsScore, sdpQPos, sdpTPos, nSdpBlocks, dpScore,
 * nCells, score, nMatch, nMismatch, nIns, nDel; qPos/tPos are window-relative. */
int bo_refine_candidate(const unsigned char *q, int qlen, const unsigned char *t, int tlen,
                        const int *matrix25, int ins, int del, int affOpen, int affExt,
                        int wordSize, int sdpIns, int sdpDel, float indelRate, int sdpPrefixLength,
                        int recurse, int noRecurseUnder, int bandSize, int *out, float *pct,
                        int *outQPos, int *outTPos, int *blocks, int blockCap, int *nBlocks,
                        int *gapCounts, int gapListCap, int *nGapLists, int *gapItems,
                        int gapItemCap, int *nGapItems);

/* a-1: window / strand set-up of AlignIntervals (alignment/Blasr.cpp:1208-1336).  out[8]:
 * tAlignedSeqPos (on the interval's strand), tAlignedSeqLength, tStrand, tIndex, tLength,
 * qAlignedSeqPos, qAlignedSeqLength, first base of the window on the forward genome. */
void bo_interval_window(unsigned genomeLen, unsigned readLen, unsigned subreadStart, unsigned subreadEnd,
                        unsigned ivStart, unsigned ivEnd, unsigned ivQStart, unsigned ivQEnd, int strand,
                        const unsigned *contigStarts, int nContigs, float approximateMaxInsertionRate,
                        int extendEnds, int doGlobalAlignment, int *out);
void bo_copy_window(const unsigned char *genome, unsigned fwdStart, unsigned len, int strand, unsigned char *out);

/* a-10: SortAlignmentPointersByScore, RemoveLowQualityAlignments, RemoveOverlappingAlignments
 * (alignment/Blasr.cpp:226-236, 2022-2079, 2081-2161) on parallel arrays; order[] lists the live
 * candidates and is rewritten in place, the return value is the new count. */
void bo_sort_by_score(int n, const int *score, const unsigned *tPos, const unsigned *tAlignedSeqPos, int *order);
int bo_remove_low_quality(int n, int *order, int *score, const int *blocks, const int *blkOff, const int *nBlocks,
                          int nCandidates, int maxGap, int maxScore);
int bo_remove_overlapping(int n, int *order, const int *score, const unsigned *tPos, const unsigned *tAlignedSeqPos,
                          const unsigned *tEnd, const int *tIndex, const int *tStrand, const float *pctSimilarity,
                          float minPctIdentity);

/* a-11: AffineKBandAlign, Global (AffineKBandAlign.h:12-401) as PairwiseLocalAlign calls it (Blasr.cpp:695-703). */
int bo_affine_kband_align(const unsigned char *q, int qlen, const unsigned char *t, int tlen, const int *matrix25,
                          int hpInsOpen, int hpInsExtend, int insOpen, int insExtend, int del, int k, int *score,
                          int *blocks, int blockCap, int *nBlocks, int *gapCounts, int gapListCap, int *nGapLists,
                          int *gapItems, int gapItemCap, int *nGapItems);

/* a-11: ComputeDrift (AlignmentUtils.h:586-618) and RefineAlignment's -noUseGuidedAlign branch (Blasr.cpp:917-1046 with
 * PairwiseLocalAlign :675-746) for one candidate; see blasr_oracle.c for the outputs. */
int bo_compute_drift(const int *blocks, int nBlocks);
int bo_refine_nonguided(const unsigned char *q, const unsigned char *t, int sdpQPos, int sdpTPos, const int *sdpBlocks,
                        int nSdpBlocks, const int *matrix25, int ins, int del, int indel, int bandSize, int *out,
                        int *statsScore, float *pct, int *qTrimLen, int *tTrimLen, int *blocks, int blockCap,
                        int *nBlocks, int *gapCounts, int gapListCap, int *nGapLists, int *gapItems, int gapItemCap,
                        int *nGapItems);

/* a-11: KBandAlign with a DistanceMatrixScoreFunction (KBandAlign.h:74-403); alignType Global 1, QueryFit 2,
 * TargetFit 3, Fit 7; out[3] = score, qPos, tPos. */
int bo_kband_align(const unsigned char *q, int qlen, const unsigned char *t, int tlen, const int *matrix25, int ins,
                   int del, int fnIns, int fnDel, int k, int alignType, int *out, int *blocks, int blockCap,
                   int *nBlocks, int *gapCounts, int gapListCap, int *nGapLists, int *gapItems, int gapItemCap,
                   int *nGapItems);

/* next-3 (first piece): ExtendAlignmentForward / Reverse (ExtendAlign.h:115-445), the -extend step of AlignIntervals
 * (Blasr.cpp:1749-1940).  out[5] = return value, qPos, tPos, globalMinScoreQPos, globalMinScoreTPos. */
int bo_extend_align(const unsigned char *q, int qlen, int queryPos, const unsigned char *t, int tlen, int refPos,
                    const int *matrix25, int ins, int del, int k, int minExtendNBases, int maxNDrops, int reverse,
                    int *out, int *blocks, int blockCap, int *nBlocks, int *gapCounts, int gapListCap, int *nGapLists,
                    int *gapItems, int gapItemCap, int *nGapItems);

/* next-2 (first pieces of the output epilogue): AssignRefContigLocation (alignment/Blasr.cpp:2183-2220),
 * CreateCIGARString (printers/SAMPrinter.h:122-416; clipping: hard 0, soft 1, subread 2, none 3),
 * IntervalAlignmentPrinter::Print (printers/IntervalAlignmentPrinter.h:9-38). */
void bo_ref_contig_location(unsigned tAlignedSeqPos, unsigned tAlignedSeqLength, int tStrand, const unsigned *seqStartPos,
                            int nSeqPos, unsigned genomeLength, unsigned *out2);
int bo_cigar(const unsigned char *q, const unsigned char *t, int qPos, int tPos, const unsigned *ops, long nOps,
             const int *blocks, long nBlocks, int clipping, int tStrand, unsigned readLength, unsigned lowQualityPrefix,
             unsigned lowQualitySuffix, unsigned subreadStart, unsigned subreadEnd, unsigned qAlignStart,
             unsigned qAlignEnd, char *out, int outCap, unsigned *clips4);
int bo_m4_line(const char *qName, const char *tName, int score, float pctSimilarity, int qStrand, unsigned qAlignStart,
               unsigned qAlignEnd, unsigned qLength, int tStrand, unsigned tAlignStart, unsigned tAlignEnd,
               unsigned tLength, int mapQV, char *out, int outCap);

/* next-2: StoreMapQVs (alignment/Blasr.cpp:2647-2950) on n alignments of one read given as parallel arrays; the op
 * lists ((len << 2) | M 0 / I 1 / D 2) are concatenated, opOff has n+1 entries; matrix25 is indexed
 * [target][query] as DistanceMatrixScoreFunction::Match does; qvs = read.substitutionQV or NULL. */
void bo_store_map_qvs(int n, const unsigned char *const *qSeqs, const unsigned char *const *tSeqs, const int *qPos,
                      const int *tPos, const unsigned *ops, const long *opOff, const unsigned *qAlignStart,
                      const unsigned *qAlignEnd, const int *qStrand, const unsigned *qLength, const int *nMatch,
                      const unsigned char *qvs, const int *matrix25, int ins, int del, int affineOpen, int affineExtend,
                      int affineAlign, float minFraction, int *mapQV, float *probScore);

/* next-2: SAMOutput::PrintAlignment (printers/SAMPrinter.h:418-548; BuildFlag :29-50, SetAlignedSequence :58-105),
 * no passthrough, no supplemental QVs.  iv / fv as in oracle/ref_mapper_harness.cpp ref_sam_line, with
 * iv[24..27] = QAlignStart, QAlignEnd, TAlignStart + blocks[0].tPos, TEnd and clips4 from bo_cigar. */
int bo_sam_line(const char *qName, const char *tName, const char *readGroupId, const unsigned char *readSeq,
                const unsigned char *readQual, const char *cigar, const unsigned *clips4, const int *iv, const float *fv,
                char *out, int outCap);

/* next-2: Alignment::OrderGapsByType (Alignment.h:255-317) on an op list; returns the new number of ops */
long bo_order_gaps_by_type(const unsigned *ops, long nOps, unsigned *out);

/* next-1, device half: MapReadToGenome over the suffix array (algorithms/anchoring/MapBySuffixArray.h:24-311,
 * SuffixArray::StoreLCPBounds datastructures/suffixarray/SuffixArray.h:928-1070, SearchLeftBound / SearchRightBound
 * :733-816) on one strand (the caller reverse-complements for the other).  ap[7] = minMatchLength, expand,
 * useLookupTable, maxAnchorsPerPosition, advanceExactMatches, maxLCPLength, stopMappingOnceUnique.  out = (t, q, l)
 * triples in emission order; returns the count, or -3 / -4 where the reference's asserts would fire. */
long bo_map_read_to_genome(const unsigned char *genome, unsigned genomeLength, const unsigned *saIndex,
                           const unsigned *startPosTable, const unsigned *endPosTable, int lookupPrefixLength,
                           const unsigned char *read, unsigned readLength, unsigned subreadStart, unsigned subreadEnd,
                           unsigned minPrefixMatchLength, const int *ap, unsigned *out, long cap);
/* SortMatchPosList (datastructures/anchoring/MatchPos.h:120-131,166-168) on (t, q, l) triples */
void bo_sort_match_pos(unsigned *triples, long n);

/* design probe: {largest row spread of S, rows, rows with spread >= 512, >= 1024} of the
 * AffineGuidedAlign fills since the last call; enable != 0 keeps counting */
void bo_probe_row_spread(int enable, long *out4);

#ifdef __cplusplus
}
#endif
#endif
