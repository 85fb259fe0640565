// blasr_hip_dropin.h -- the code a maintainer adds to alignment/Blasr.cpp to put libblasr_hip.so behind
// `-useGPU` (INTEGRATION.md sections 2, 3, 3b).  TEST INFRASTRUCTURE in this repository: oracle/make_dropin.py
// applies the patch to a scratch copy of the reference's own Blasr.cpp / MappingParameters.h at build time and
// compiles oracle/_ref/blasrmc_gpu from it; tests/test_dropin_gpu.py runs that binary with -nproc 2 and compares its
// -m 4 lines with the unpatched binary's.  Nothing of the reference is copied here: this file only holds the new code.
//
// Included by the patched Blasr.cpp right before AlignIntervals (everything it names -- T_AlignmentCandidate,
// MappingParameters with the two new fields useGPU / gpuDevice, SMRTDistanceMatrix, SortAlignmentPointersByScore --
// is declared by then).
#ifndef BLASR_HIP_DROPIN_H_
#define BLASR_HIP_DROPIN_H_
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include <vector>

#include "blasr_hip.h"

// one context per mapping thread, like MappingBuffers (Blasr.cpp:3238): created by the holder MapReads declares
static __thread blasr_hip_ctx *tlsHipCtx = NULL;

// ---- -useGPU -gpuBatch B: MapRead for whole BLOCKS of reads ----------------------------
// A mapping thread pulls B reads from the reader (each under the reader's semaphore, as before), hands the block to
// the library in ONE call -- blasr_hip_map_from_reads_host: anchoring, interval finding, AlignIntervals, the filters
// and RefineAlignments for every read of the block -- and then walks the block read by read through the REST of
// MapReads unchanged (subread bookkeeping, StoreMapQVs, the printers): where MapReads calls MapRead it gets the
// alignments the block call left for that read.  The threads share ONE context (the index is on the device once;
// blocks of different threads queue at a mutex while the other threads read ahead and print).
struct BlasrHipBlock {
  std::vector<SMRTSequence> reads;
  std::vector<AlignmentContext> contexts;
  std::vector<blasr_hip_hit> hits;
  std::vector<int64_t> hitOff;
  // the hits' aligned sequences and op lists: grow-only buffers, NOT zero-filled (gigabytes per block otherwise, and a
  // page-fault storm when several threads do it at once)
  uint8_t *seqArena;
  uint32_t *opsArena;
  int64_t arenaCap;
  size_t n, next;
  long current;  // the read MapReads is working on
  BlasrHipBlock() : seqArena(NULL), opsArena(NULL), arenaCap(0), n(0), next(0), current(-1) {}
  ~BlasrHipBlock() {
    free(seqArena);
    free(opsArena);
  }
  void Reserve(int64_t cap) {
    if (cap <= arenaCap) return;
    free(seqArena);
    free(opsArena);
    seqArena = (uint8_t *)malloc((size_t)cap);
    opsArena = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)cap);
    arenaCap = cap;
    if (!seqArena || !opsArena) {
      cout << "ERROR, out of memory for a block's alignments." << endl;
      exit(1);
    }
  }
};
static __thread BlasrHipBlock *tlsHipBlock = NULL;
struct BlasrHipShared {
  pthread_mutex_t mu;
  blasr_hip_ctx *ctx;
  int users;
  std::vector<int32_t> countTable;
  std::vector<uint32_t> seqStarts;
};
static BlasrHipShared gHipShared = {PTHREAD_MUTEX_INITIALIZER, NULL, 0};

struct BlasrHipThreadContext {
  BlasrHipThreadContext(bool use, int device, int batch = 0) {
    if (use && batch > 0) {
      tlsHipBlock = new BlasrHipBlock;
      pthread_mutex_lock(&gHipShared.mu);
      gHipShared.users++;
      pthread_mutex_unlock(&gHipShared.mu);
      return;
    }
    if (use && blasr_hip_create(device, &tlsHipCtx) != BLASR_HIP_OK) {
      cout << "ERROR, no gfx950 device for -useGPU." << endl;
      exit(1);
    }
  }
  ~BlasrHipThreadContext() {
    if (tlsHipCtx != NULL) blasr_hip_destroy(tlsHipCtx);
    tlsHipCtx = NULL;
    if (tlsHipBlock != NULL) {
      delete tlsHipBlock;
      tlsHipBlock = NULL;
      pthread_mutex_lock(&gHipShared.mu);
      if (--gHipShared.users == 0 && gHipShared.ctx != NULL) {
        blasr_hip_destroy(gHipShared.ctx);
        gHipShared.ctx = NULL;
      }
      pthread_mutex_unlock(&gHipShared.mu);
    }
  }
};

static void BlasrHipScoreParams(MappingParameters &params, blasr_hip_score_params &sc) {
  for (int i = 0; i < 5; i++)
    for (int j = 0; j < 5; j++) sc.matrix[i * 5 + j] = SMRTDistanceMatrix[i][j];  // Blasr.cpp:807-809
  sc.ins = params.insertion;
  sc.del = params.deletion;
  sc.affine_open = params.affineOpen;
  sc.affine_extend = params.affineExtend;
}

static float BlasrHipPctSimilarity(int nMatch, int nMismatch, int nIns, int nDel) {
  int cols = nMatch + nMismatch + nIns + nDel;
  return cols ? (float)((nMatch * 2.0) / (cols + cols) * 100) : 0;  // AlignmentUtils.h:566-577
}

// Replaces SDPAlign + ComputeAlignmentStats of AlignIntervals (Blasr.cpp:1716-1724) for one candidate.
static int SDPAlignCandidateGPU(T_AlignmentCandidate &a, MappingParameters &params, int sdpTupleSize) {
  std::vector<uint8_t> seq;
  seq.insert(seq.end(), a.qAlignedSeq.seq, a.qAlignedSeq.seq + a.qAlignedSeq.length);
  seq.insert(seq.end(), a.tAlignedSeq.seq, a.tAlignedSeq.seq + a.tAlignedSeq.length);
  int64_t qOff = 0, tOff = a.qAlignedSeq.length;
  int32_t qLen = a.qAlignedSeq.length, tLen = a.tAlignedSeq.length;
  blasr_hip_sdp_batch batch = {1, seq.data(), &qOff, &qLen, &tOff, &tLen};
  blasr_hip_sdp_params sp = {sdpTupleSize, params.sdpIns, params.sdpDel, (float)(params.indelRate * 3), 0 /* Local */,
                             params.detailedSDPAlignment, params.extendFrontAlignment, params.sdpPrefix, params.recurse,
                             params.recurseOver, params.sdpMaxAnchorsPerPosition};
  blasr_hip_score_params sc;
  BlasrHipScoreParams(params, sc);
  blasr_hip_sdp_result res;
  std::vector<int32_t> blocks(3 * (size_t)(std::min(qLen, tLen) + 16));
  int64_t used = 0;
  int rc = blasr_hip_sdp_host(tlsHipCtx, &batch, &sp, &sc, &res, blocks.data(), (int64_t)blocks.size() / 3, &used);
  if (rc != BLASR_HIP_OK) {
    cout << "ERROR, GPU SDP failed: " << blasr_hip_last_error(tlsHipCtx) << endl;
    exit(1);
  }
  a.qPos = res.q_pos;
  a.tPos = res.t_pos;  // SDPAlign.h:602-621
  a.blocks.resize(res.n_blocks);
  a.gaps.clear();
  for (int b = 0; b < res.n_blocks; b++) {
    const int32_t *t3 = &blocks[3 * (res.blk_off + b)];
    a.blocks[b].qPos = t3[0];
    a.blocks[b].tPos = t3[1];
    a.blocks[b].length = t3[2];
  }
  a.score = res.stats_score;  // ComputeAlignmentStats (:1723-1724)
  a.nMatch = res.n_match;
  a.nMismatch = res.n_mismatch;
  a.nIns = res.n_ins;
  a.nDel = res.n_del;
  a.pctSimilarity = BlasrHipPctSimilarity(a.nMatch, a.nMismatch, a.nIns, a.nDel);
  return res.score;
}

// Replaces the loop of RefineAlignments (Blasr.cpp:2164-2181): all candidates of the read in one batch.
template <typename T_Sequence>
static void RefineAlignmentsGPU(T_Sequence &query, std::vector<T_AlignmentCandidate *> &alignmentPtrs,
                                MappingParameters &params) {
  std::vector<uint8_t> seq;
  std::vector<int64_t> qOff, tOff, blkOff;
  std::vector<int32_t> qLen, tLen, nBlk, blocks;
  for (UInt i = 0; i < alignmentPtrs.size(); i++) {
    T_AlignmentCandidate &c = *alignmentPtrs[i];
    int last = (int)c.blocks.size() - 1;
    DNALength qn = last >= 0 ? c.blocks[last].qPos + c.blocks[last].length : 0;  // Blasr.cpp:856-859
    DNALength tn = last >= 0 ? c.blocks[last].tPos + c.blocks[last].length : 0;  // Blasr.cpp:850-853
    qOff.push_back(seq.size());
    qLen.push_back(qn);
    seq.insert(seq.end(), &query.seq[c.qAlignedSeqPos + c.qPos], &query.seq[c.qAlignedSeqPos + c.qPos] + qn);
    tOff.push_back(seq.size());
    tLen.push_back(tn);
    seq.insert(seq.end(), &c.tAlignedSeq.seq[c.tPos], &c.tAlignedSeq.seq[c.tPos] + tn);
    blkOff.push_back(blocks.size() / 3);
    nBlk.push_back(c.blocks.size());
    for (UInt b = 0; b < c.blocks.size(); b++) {
      blocks.push_back(c.blocks[b].qPos);
      blocks.push_back(c.blocks[b].tPos);
      blocks.push_back(c.blocks[b].length);
    }
  }
  if (alignmentPtrs.size() == 0) return;
  if (blocks.empty()) blocks.push_back(0);
  if (seq.empty()) seq.push_back('N');
  blasr_hip_refine_batch batch = {(int32_t)alignmentPtrs.size(), seq.data(), qOff.data(), qLen.data(), tOff.data(),
                                  tLen.data(), blocks.data(), blkOff.data(), nBlk.data()};
  blasr_hip_score_params sc;
  BlasrHipScoreParams(params, sc);
  std::vector<blasr_hip_refine_result> res(alignmentPtrs.size());
  std::vector<uint32_t> ops(seq.size() + 2 * alignmentPtrs.size() + 16);
  int64_t used = 0;
  int rc = blasr_hip_refine_host(tlsHipCtx, &batch, params.bandSize, &sc, res.data(), ops.data(), (int64_t)ops.size(), &used);
  if (rc != BLASR_HIP_OK) {
    cout << "ERROR, GPU refine failed: " << blasr_hip_last_error(tlsHipCtx) << endl;
    exit(1);
  }
  for (UInt i = 0; i < alignmentPtrs.size(); i++) {  // copy-back, Blasr.cpp:888-914
    T_AlignmentCandidate &c = *alignmentPtrs[i];
    if (res[i].status & BLASR_HIP_TASK_EMPTY_GUIDE) continue;  // the reference leaves such candidates untouched
    if (res[i].status) {
      cout << "ERROR, this path has gone awry" << endl;  // AffineGuidedAlign.h:406
      exit(1);
    }
    c.blocks.clear();
    c.gaps.clear();
    c.gaps.push_back(GapList());
    DNALength q = 0, t = 0;
    for (int o = 0; o < res[i].n_ops; o++) {
      uint32_t op = ops[res[i].ops_off + o], len = BLASR_HIP_OP_LEN(op);
      if (BLASR_HIP_OP_TYPE(op) == BLASR_HIP_OP_M) {
        Block b;
        b.qPos = q;
        b.tPos = t;
        b.length = len;
        c.blocks.push_back(b);
        c.gaps.push_back(GapList());
        q += len;
        t += len;
      } else if (BLASR_HIP_OP_TYPE(op) == BLASR_HIP_OP_I) {
        c.gaps.back().push_back(Gap(Gap::Target, len));
        q += len;
      } else {
        c.gaps.back().push_back(Gap(Gap::Query, len));
        t += len;
      }
    }
    c.score = res[i].score;
    c.nCells = res[i].n_cells;
    c.nMatch = res[i].n_match;
    c.nMismatch = res[i].n_mismatch;
    c.nIns = res[i].n_ins;
    c.nDel = res[i].n_del;
    c.pctSimilarity = BlasrHipPctSimilarity(c.nMatch, c.nMismatch, c.nIns, c.nDel);
    // the aligned sequences become the trimmed ones, exactly as Blasr.cpp:896-914 (TakeOwnership / ReassignQSequence)
    DNASequence tSeq;
    tSeq.Copy(c.tAlignedSeq, c.tPos, tLen[i]);
    FASTQSequence qSeq;
    qSeq.ReferenceSubstring(query, c.qAlignedSeqPos + c.qPos, qLen[i]);
    c.tAlignedSeq.TakeOwnership(tSeq);
    c.ReassignQSequence(qSeq);
    c.tAlignedSeqPos += c.tPos;
    c.qAlignedSeqPos += c.qPos;
    c.tPos = res[i].t_pos;
    c.qPos = res[i].q_pos;
    c.tAlignedSeqLength = tLen[i];
    c.qAlignedSeqLength = qLen[i];
  }
  if (params.sortRefinedAlignments)  // Blasr.cpp:2178-2180
    std::sort(alignmentPtrs.begin(), alignmentPtrs.end(), SortAlignmentPointersByScore());
}
// The loop of RefineAlignments under -noUseGuidedAlign: RefineAlignment's k-band branch (Blasr.cpp:917-1046) for all
// candidates of the read in one batch.  The host does what the branch does around its one heavy call -- ComputeDrift (the
// reference's own function), the trimming (:968-981), the band rule (:990-1003), the bookkeeping of :1005-1043 -- and
// PairwiseLocalAlign's AffineKBandAlign (:695-703) runs on the device.  The statistics are the reference's own
// ComputeAlignmentStats in the non-affine form PairwiseLocalAlign uses (:740), but with the score matrix in place: the
// reference makes that call with a matrix it never initialised (:689-690), so its score there is uninitialised memory.
template <typename T_Sequence>
static void RefineAlignmentsKBandGPU(T_Sequence &query, std::vector<T_AlignmentCandidate *> &alignmentPtrs,
                                     MappingParameters &params) {
  std::vector<uint8_t> seq;
  std::vector<int64_t> qOff, tOff;
  std::vector<int32_t> qLen, tLen, kBand, owner;
  for (UInt i = 0; i < alignmentPtrs.size(); i++) {
    T_AlignmentCandidate &c = *alignmentPtrs[i];
    if (c.blocks.size() == 0) {  // :959-962
      c.score = 0;
      continue;
    }
    const int drift = ComputeDrift(c);
    const int last = (int)c.blocks.size() - 1;
    const DNALength qn = c.blocks[last].qPos + c.blocks[last].length;                       // :968-972 (qIsSubstring)
    const DNALength tn = c.blocks[last].tPos + c.blocks[last].length - c.blocks[0].tPos;   // :979-981
    int k = params.bandSize == 0 ? (int)(abs(drift) * 1.5) : params.bandSize;               // :990-996
    if (k < 10) k = 10;                                                                     // :1001-1003
    qOff.push_back(seq.size());
    qLen.push_back(qn);
    seq.insert(seq.end(), &query.seq[c.qAlignedSeqPos + c.qPos], &query.seq[c.qAlignedSeqPos + c.qPos] + qn);
    tOff.push_back(seq.size());
    tLen.push_back(tn);
    seq.insert(seq.end(), &c.tAlignedSeq.seq[c.tPos], &c.tAlignedSeq.seq[c.tPos] + tn);
    kBand.push_back((int)(k * 1.2));                                                        // :699
    owner.push_back((int32_t)i);
  }
  const int n = (int)owner.size();
  if (n > 0) {
    blasr_hip_kband_batch batch = {n, seq.data(), qOff.data(), qLen.data(), tOff.data(), tLen.data(), kBand.data()};
    blasr_hip_kband_params kp;  // :695-699
    for (int a = 0; a < 5; a++)
      for (int b = 0; b < 5; b++) kp.matrix[a * 5 + b] = SMRTDistanceMatrix[a][b];
    kp.hp_ins_open = params.indel + 2;
    kp.hp_ins_extend = params.indel - 3;
    kp.ins_open = params.indel + 2;
    kp.ins_extend = params.indel - 1;
    kp.del = params.indel;
    std::vector<blasr_hip_refine_result> res(n);
    std::vector<uint32_t> ops(seq.size() + 4 * (size_t)n + 16);
    int64_t used = 0;
    int rc = blasr_hip_kband_host(tlsHipCtx, &batch, &kp, res.data(), ops.data(), (int64_t)ops.size(), &used);
    if (rc != BLASR_HIP_OK) {
      cout << "ERROR, GPU k-band refine failed: " << blasr_hip_last_error(tlsHipCtx) << endl;
      exit(1);
    }
    DistanceMatrixScoreFunction<DNASequence, FASTQSequence> distScoreFn;
    params.InitializeScoreFunction(distScoreFn);
    distScoreFn.InitializeScoreMatrix(SMRTDistanceMatrix);  // (what :689-690 leaves out)
    for (int j = 0; j < n; j++) {
      T_AlignmentCandidate &c = *alignmentPtrs[owner[j]];
      if (res[j].status) {
        cout << "ERROR, the k-band refiner rejected a candidate (status " << res[j].status << ")" << endl;
        exit(1);
      }
      // :1005-1043: the candidate now describes the trimmed sequences
      DNASequence tSeq;
      tSeq.Copy(c.tAlignedSeq, c.tPos, tLen[j]);
      FASTQSequence qSeq;
      qSeq.ReferenceSubstring(query, c.qAlignedSeqPos + c.qPos, qLen[j]);
      const VectorIndex lastSDPBlock = c.blocks.size() - 1;
      c.tAlignedSeqPos += c.tPos;
      c.tAlignedSeqLength = c.blocks[lastSDPBlock].tPos + c.blocks[lastSDPBlock].length - c.blocks[0].tPos;
      c.tPos = 0;
      c.qAlignedSeqPos += c.qPos;
      c.qAlignedSeqLength = c.blocks[lastSDPBlock].qPos + c.blocks[lastSDPBlock].length - c.blocks[0].qPos;
      c.qPos = 0;
      c.blocks.clear();
      c.tAlignedSeq.Free();
      c.tAlignedSeq.TakeOwnership(tSeq);
      c.ReassignQSequence(qSeq);
      // PairwiseLocalAlign: the arrow path as ArrowPathToAlignment leaves it (Alignment.h:190-254: a gap list in front of
      // the first block and one behind every block, the last one empty)
      c.gaps.clear();
      c.gaps.push_back(GapList());
      DNALength q = 0, t = 0;
      for (int o = 0; o < res[j].n_ops; o++) {
        uint32_t op = ops[res[j].ops_off + o], len = BLASR_HIP_OP_LEN(op);
        if (BLASR_HIP_OP_TYPE(op) == BLASR_HIP_OP_M) {
          Block b;
          b.qPos = q;
          b.tPos = t;
          b.length = len;
          c.blocks.push_back(b);
          c.gaps.push_back(GapList());
          q += len;
          t += len;
        } else if (BLASR_HIP_OP_TYPE(op) == BLASR_HIP_OP_I) {
          c.gaps.back().push_back(Gap(Gap::Target, len));
          q += len;
        } else {
          c.gaps.back().push_back(Gap(Gap::Query, len));
          t += len;
        }
      }
      c.score = res[j].dp_score;  // :705
      ComputeAlignmentStats(c, c.qAlignedSeq.seq, c.tAlignedSeq.seq, distScoreFn);  // :740
    }
  }
  if (params.sortRefinedAlignments)  // Blasr.cpp:2178-2180
    std::sort(alignmentPtrs.begin(), alignmentPtrs.end(), SortAlignmentPointersByScore());
}

// Replaces ExtendAlignmentForward / ExtendAlignmentReverse under -extend (Blasr.cpp:1826-1838, 1885-1897): one call, one
// task.  The arrow path comes back as ops; the alignment is rebuilt with the reference's own ArrowPathToAlignment.
template <typename T_Query, typename T_Ref>
static int ExtendAlignmentGPU(T_Query &querySeq, int queryPos, T_Ref &refSeq, int refPos, T_AlignmentCandidate &alignment,
                              MappingParameters &params, bool reverse) {
  std::vector<uint8_t> seq;
  seq.insert(seq.end(), querySeq.seq, querySeq.seq + querySeq.length);
  seq.insert(seq.end(), refSeq.seq, refSeq.seq + refSeq.length);
  int64_t qOff = 0, tOff = querySeq.length;
  int32_t qLen = querySeq.length, tLen = refSeq.length, qPos = queryPos, tPos = refPos;
  uint8_t rev = reverse ? 1 : 0;
  blasr_hip_extend_batch batch = {1, seq.data(), &qOff, &qLen, &tOff, &tLen, &qPos, &tPos, &rev};
  blasr_hip_extend_params ep;
  blasr_hip_extend_default_params(&ep);
  ep.ins = params.insertion;  // distScoreFn (Blasr.cpp:3666-3668)
  ep.del = params.deletion;
  ep.band = params.extendBandSize;
  ep.max_n_drops = params.maxExtendDropoff;
  blasr_hip_refine_result res;
  std::vector<uint32_t> ops((size_t)qLen + tLen + 16);
  int64_t used = 0;
  int rc = blasr_hip_extend_host(tlsHipCtx, &batch, &ep, &res, ops.data(), (int64_t)ops.size(), &used);
  if (rc != BLASR_HIP_OK || res.status != 0) {
    cout << "ERROR, GPU extension failed: " << blasr_hip_last_error(tlsHipCtx) << endl;
    exit(1);
  }
  std::vector<Arrow> path;
  for (int o = 0; o < res.n_ops; o++) {
    uint32_t op = ops[res.ops_off + o];
    Arrow a = BLASR_HIP_OP_TYPE(op) == BLASR_HIP_OP_M ? Diagonal : BLASR_HIP_OP_TYPE(op) == BLASR_HIP_OP_I ? Up : Left;
    path.insert(path.end(), BLASR_HIP_OP_LEN(op), a);
  }
  alignment.ArrowPathToAlignment(path);  // ExtendAlign.h:374-377
  alignment.qPos = res.q_pos;
  alignment.tPos = res.t_pos;
  return res.dp_score;
}

// Replaces the two MapReadToGenome calls of MapRead (Blasr.cpp:2279-2297) for one read: both strands anchored on the
// device in the suffix array the mapper holds (uploaded once per mapping thread's context).  The lists come back in
// MapReadToGenome's own order; MapRead sorts them itself right after (:2339-2340).
static __thread bool tlsHipIndexUploaded = false;

template <typename T_GenomeSeq, typename T_SuffixArrayT, typename T_Sequence>
static void MapReadToGenomeGPU(T_GenomeSeq &genome, T_SuffixArrayT &sarray, T_Sequence &read, MappingParameters &params,
                               std::vector<ChainedMatchPos> &matchPosList, std::vector<ChainedMatchPos> &rcMatchPosList,
                               int &numKeysMatched, int &rcNumKeysMatched) {
  if (!tlsHipIndexUploaded) {
    if (blasr_hip_upload_genome(tlsHipCtx, (const uint8_t *)genome.seq, genome.length, NULL, 0) != BLASR_HIP_OK ||
        blasr_hip_upload_suffix_array(tlsHipCtx, sarray.index, sarray.length, sarray.startPosTable, sarray.endPosTable,
                                      sarray.lookupTableLength, sarray.lookupPrefixLength) != BLASR_HIP_OK) {
      cout << "ERROR, GPU index upload failed: " << blasr_hip_last_error(tlsHipCtx) << endl;
      exit(1);
    }
    tlsHipIndexUploaded = true;
  }
  blasr_hip_anchor_params ap;
  blasr_hip_anchor_default_params(&ap);
  ap.min_match_length = params.anchorParameters.minMatchLength;
  ap.min_prefix_match_length = params.lookupTableLength;
  ap.expand = params.anchorParameters.expand;
  ap.max_anchors_per_position = params.anchorParameters.maxAnchorsPerPosition;
  ap.advance_exact_matches = params.anchorParameters.advanceExactMatches;
  ap.max_lcp_length = params.anchorParameters.maxLCPLength;
  ap.use_lookup_table = params.anchorParameters.useLookupTable;
  ap.stop_mapping_once_unique = params.anchorParameters.stopMappingOnceUnique;
  ap.forward_only = params.forwardOnly;
  ap.sorted = 0;
  blasr_hip_read rd = {0, read.length, read.subreadStart, read.subreadEnd, 0};
  std::vector<blasr_hip_anchor> anchors(4 * (size_t)read.length + 1024);
  int64_t listOff[3] = {0, 0, 0}, n = 0;
  int rc = blasr_hip_anchor_reads_host(tlsHipCtx, 1, &rd, (const uint8_t *)read.seq, read.length, &ap, anchors.data(),
                                       (int64_t)anchors.size(), &n, listOff);
  if (rc == BLASR_HIP_ERR_CAPACITY) {
    anchors.resize((size_t)n + 1);
    rc = blasr_hip_anchor_reads_host(tlsHipCtx, 1, &rd, (const uint8_t *)read.seq, read.length, &ap, anchors.data(),
                                     (int64_t)anchors.size(), &n, listOff);
  }
  if (rc != BLASR_HIP_OK) {
    cout << "ERROR, GPU anchoring failed: " << blasr_hip_last_error(tlsHipCtx) << endl;
    exit(1);
  }
  for (int64_t i = listOff[0]; i < listOff[1]; i++) matchPosList.push_back(ChainedMatchPos(anchors[i].t, anchors[i].q, anchors[i].l));
  for (int64_t i = listOff[1]; i < listOff[2]; i++) rcMatchPosList.push_back(ChainedMatchPos(anchors[i].t, anchors[i].q, anchors[i].l));
  numKeysMatched = (int)matchPosList.size();
  rcNumKeysMatched = (int)rcMatchPosList.size();
}

// (defined further down in Blasr.cpp)
template <typename T_Sequence>
bool GetNextReadThroughSemaphore(ReaderAgglomerate &reader, MappingParameters &params, T_Sequence &read, AlignmentContext &context);
void AssignRefContigLocations(vector<T_AlignmentCandidate *> &alignmentPtrs, SequenceIndexDatabase<FASTQSequence> &seqdb,
                              DNASequence &genome);

// One block through the library.  Everything the library needs of the mapper's state is read here: the genome and its
// contig table (seqdb), the suffix array the mapper loaded or built, the switches of MappingParameters.
template <typename T_GenomeSeq, typename T_SuffixArrayT>
static void BlasrHipMapBlock(BlasrHipBlock &blk, T_GenomeSeq &genome, T_SuffixArrayT &sarray,
                             SequenceIndexDatabase<FASTQSequence> &seqdb, MappingParameters &params) {
  std::vector<blasr_hip_read> reads(blk.n);
  std::vector<uint8_t> pool;
  int64_t seqCap = 4096;
  for (size_t r = 0; r < blk.n; r++) {
    blasr_hip_read rd = {(int64_t)pool.size(), blk.reads[r].length, 0, blk.reads[r].length, 0};
    reads[r] = rd;
    pool.insert(pool.end(), blk.reads[r].seq, blk.reads[r].seq + blk.reads[r].length);
    seqCap += 2 * (5 * (int64_t)blk.reads[r].length / 2 + 1024);  // (a guess: about two alignments per read; corrected below)
  }
  if (pool.empty()) pool.push_back('N');
  blasr_hip_pipeline_params pp;
  blasr_hip_pipeline_default_params(&pp);
  pp.anchor.min_match_length = params.anchorParameters.minMatchLength;
  pp.anchor.min_prefix_match_length = params.lookupTableLength;
  pp.anchor.expand = params.anchorParameters.expand;
  pp.anchor.max_anchors_per_position = params.anchorParameters.maxAnchorsPerPosition;
  pp.anchor.advance_exact_matches = params.anchorParameters.advanceExactMatches;
  pp.anchor.max_lcp_length = params.anchorParameters.maxLCPLength;
  pp.anchor.use_lookup_table = params.anchorParameters.useLookupTable;
  pp.anchor.stop_mapping_once_unique = params.anchorParameters.stopMappingOnceUnique;
  pp.anchor.forward_only = params.forwardOnly;
  pp.interval.n_candidates = params.nCandidates;
  pp.interval.min_interval = params.minInterval;
  pp.interval.indel_rate = params.indelRate;
  pp.interval.remove_contained_intervals = params.removeContainedIntervals;
  pp.map.window.approximate_max_insertion_rate = params.approximateMaxInsertionRate;
  pp.map.window.extend_ends = params.extendEnds;
  pp.map.window.do_global_alignment = params.doGlobalAlignment;
  blasr_hip_sdp_params sp = {params.sdpTupleSize, params.sdpIns, params.sdpDel, (float)(params.indelRate * 3), 0 /* Local */,
                             params.detailedSDPAlignment, params.extendFrontAlignment, params.sdpPrefix, params.recurse,
                             params.recurseOver, params.sdpMaxAnchorsPerPosition};
  pp.map.sdp = sp;
  BlasrHipScoreParams(params, pp.map.score);
  pp.map.band = params.bandSize;
  pp.map.use_guided_align = params.useGuidedAlign ? 1 : 0;  // -noUseGuidedAlign: RefineAlignment's k-band branch on the device
  pp.map.n_candidates = params.nCandidates;
  pp.map.max_gap = params.maxGap;
  pp.map.max_score = params.maxScore;
  pp.map.min_pct_identity = params.minPctIdentity;
  pp.map.sort_refined = params.sortRefinedAlignments;
  pp.map.extend_alignments = params.extendAlignments;
  pp.map.extend_band_size = params.extendBandSize;
  pp.map.max_extend_dropoff = params.maxExtendDropoff;
  pp.map.refine_between_anchors_only = params.refineBetweenAnchorsOnly;  // -rbao: the chained anchors come from the library's own interval finder
  pp.map.indel = params.indel;
  pp.map.sdp_bypass_threshold = params.sdpBypassThreshold;
  blk.hits.resize(blk.n * (size_t)std::max(params.nCandidates, 1) + 1);
  blk.hitOff.assign(blk.n + 1, 0);
  blk.Reserve(seqCap);
  pthread_mutex_lock(&gHipShared.mu);
  if (gHipShared.ctx == NULL) {  // first block of the run: the index goes to the device, once for all threads
    const int nSeqPos = seqdb.nSeqPos > 0 ? seqdb.nSeqPos : 2;
    gHipShared.seqStarts.resize(nSeqPos);
    for (int i = 0; i < nSeqPos; i++) gHipShared.seqStarts[i] = seqdb.nSeqPos > 0 ? seqdb.seqStartPos[i] : (i ? genome.length : 0);
    const int nContigs = nSeqPos - 1;
    gHipShared.countTable.resize((size_t)1 << (2 * params.lookupTableLength));
    if (blasr_hip_create(params.gpuDevice, &gHipShared.ctx) != BLASR_HIP_OK ||
        // (the contig table also for a single contig: tLength is the contig's length, Blasr.cpp:1291, not the genome's with its N)
        blasr_hip_upload_genome(gHipShared.ctx, (const uint8_t *)genome.seq, genome.length,
                                params.useSeqDB ? gHipShared.seqStarts.data() : NULL, params.useSeqDB ? nContigs : 0) != BLASR_HIP_OK ||
        blasr_hip_upload_suffix_array(gHipShared.ctx, sarray.index, sarray.length, sarray.startPosTable, sarray.endPosTable,
                                      sarray.lookupTableLength, sarray.lookupPrefixLength) != BLASR_HIP_OK ||
        blasr_hip_count_tuples_device(gHipShared.ctx, params.lookupTableLength, gHipShared.countTable.data()) != BLASR_HIP_OK) {
      cout << "ERROR, GPU set-up failed: " << (gHipShared.ctx ? blasr_hip_last_error(gHipShared.ctx) : "no gfx950 device") << endl;
      exit(1);
    }
  }
  int rc = BLASR_HIP_OK;
  struct timespec ts0, ts1;
  clock_gettime(CLOCK_MONOTONIC, &ts0);
  for (int attempt = 0; attempt < 2; attempt++) {
    blk.Reserve(seqCap);
    blasr_hip_map_fetch fetch = {blk.seqArena, blk.arenaCap, 0, blk.opsArena, blk.arenaCap, 0};
    int64_t nHits = 0, nIvs = 0;
    rc = blasr_hip_map_from_reads_host(gHipShared.ctx, (const uint8_t *)genome.seq, gHipShared.seqStarts.data(),
                                       (int32_t)gHipShared.seqStarts.size(), gHipShared.countTable.data(), params.lookupTableLength,
                                       (int32_t)blk.n, reads.data(), pool.data(), (int64_t)pool.size(), &pp, blk.hits.data(),
                                       (int64_t)blk.hits.size(), &nHits, blk.hitOff.data(), &fetch, NULL, 0, &nIvs);
    if (rc != BLASR_HIP_ERR_CAPACITY) break;
    seqCap = std::max(fetch.seq_used, fetch.ops_used) + 4096;  // (rare: the arenas were a guess)
  }
  if (rc != BLASR_HIP_OK) {
    cout << "ERROR, GPU mapping of a block failed: " << blasr_hip_last_error(gHipShared.ctx) << endl;
    exit(1);
  }
  clock_gettime(CLOCK_MONOTONIC, &ts1);
  if (getenv("BLASR_HIP_DROPIN_TRACE"))
    fprintf(stderr, "[dropin] block of %zu reads mapped in %.1f ms\n", blk.n, (ts1.tv_sec - ts0.tv_sec) * 1e3 + (ts1.tv_nsec - ts0.tv_nsec) / 1e6);
  pthread_mutex_unlock(&gHipShared.mu);
}

// Replaces GetNextReadThroughSemaphore in MapReads (Blasr.cpp:3253): the next read of the thread's block; a new block is
// pulled from the reader and mapped when the last one is used up.
template <typename T_GenomeSeq, typename T_SuffixArrayT>
static bool BlasrHipNextRead(ReaderAgglomerate &reader, MappingParameters &params, SMRTSequence &read, AlignmentContext &context,
                             T_GenomeSeq &genome, T_SuffixArrayT &sarray, SequenceIndexDatabase<FASTQSequence> &seqdb) {
  if (tlsHipBlock == NULL) return GetNextReadThroughSemaphore(reader, params, read, context);
  BlasrHipBlock &blk = *tlsHipBlock;
  if (blk.next >= blk.n) {
    blk.reads.resize((size_t)params.gpuBatch);
    blk.contexts.resize((size_t)params.gpuBatch);
    blk.n = blk.next = 0;
    struct timespec tr0, tr1;
    clock_gettime(CLOCK_MONOTONIC, &tr0);
    while (blk.n < (size_t)params.gpuBatch && GetNextReadThroughSemaphore(reader, params, blk.reads[blk.n], blk.contexts[blk.n]))
      blk.n++;
    clock_gettime(CLOCK_MONOTONIC, &tr1);
    if (getenv("BLASR_HIP_DROPIN_TRACE"))
      fprintf(stderr, "[dropin] %zu reads read in %.1f ms\n", blk.n, (tr1.tv_sec - tr0.tv_sec) * 1e3 + (tr1.tv_nsec - tr0.tv_nsec) / 1e6);
    if (blk.n == 0) return false;
    BlasrHipMapBlock(blk, genome, sarray, seqdb, params);
  }
  read = blk.reads[blk.next];  // (a copy: MapReads frees its read at the end of every turn)
  blk.reads[blk.next].Free();
  context = blk.contexts[blk.next];
  blk.current = (long)blk.next++;
  return true;
}

// Replaces the MapRead call of MapReads' subread loop (Blasr.cpp:3470-3496) under -gpuBatch: the alignments the block
// call left for the current read, as T_AlignmentCandidates in the state MapRead leaves them (:2615-2644).
template <typename T_GenomeSeq>
static void BlasrHipCandidatesOfCurrentRead(SMRTSequence &read, SMRTSequence &readRC, T_GenomeSeq &genome,
                                            SequenceIndexDatabase<FASTQSequence> &seqdb, MappingParameters &params,
                                            std::vector<T_AlignmentCandidate *> &alignmentPtrs) {
  BlasrHipBlock &blk = *tlsHipBlock;
  if (read.subreadStart != 0 || read.subreadEnd != read.length) {
    cout << "ERROR, -gpuBatch maps whole reads (no region table)." << endl;
    exit(1);
  }
  for (int64_t i = blk.hitOff[blk.current]; i < blk.hitOff[blk.current + 1]; i++) {
    const blasr_hip_hit &h = blk.hits[i];
    T_AlignmentCandidate *a = new T_AlignmentCandidate;
    a->tStrand = h.t_strand;
    a->qStrand = 0;
    a->tIndex = h.t_index;
    a->tLength = h.t_length;
    a->qLength = h.q_length;
    a->tName = params.useSeqDB ? seqdb.GetSpaceDelimitedName(h.t_index) : genome.GetName();  // Blasr.cpp:1290, :1302
    DNASequence tSeq;
    tSeq.seq = new Nucleotide[h.t_aligned_seq_length + 1];
    memcpy(tSeq.seq, blk.seqArena + h.t_off, h.t_aligned_seq_length);
    tSeq.length = h.t_aligned_seq_length;
    tSeq.deleteOnExit = true;
    a->tAlignedSeq.TakeOwnership(tSeq);
    FASTQSequence qSeq;
    qSeq.ReferenceSubstring(read, h.q_aligned_seq_pos, h.q_aligned_seq_length);
    a->ReassignQSequence(qSeq);
    a->tAlignedSeqPos = h.t_aligned_seq_pos;
    a->qAlignedSeqPos = h.q_aligned_seq_pos;
    a->tAlignedSeqLength = h.t_aligned_seq_length;
    a->qAlignedSeqLength = h.q_aligned_seq_length;
    a->tPos = h.t_pos;
    a->qPos = h.q_pos;
    a->blocks.clear();
    a->gaps.clear();
    a->gaps.push_back(GapList());
    DNALength q = 0, t = 0;
    for (int o = 0; o < h.n_ops; o++) {
      const uint32_t op = blk.opsArena[h.ops_off + o], len = BLASR_HIP_OP_LEN(op);
      if (BLASR_HIP_OP_TYPE(op) == BLASR_HIP_OP_M) {
        Block b;
        b.qPos = q;
        b.tPos = t;
        b.length = len;
        a->blocks.push_back(b);
        a->gaps.push_back(GapList());
        q += len;
        t += len;
      } else if (BLASR_HIP_OP_TYPE(op) == BLASR_HIP_OP_I) {
        a->gaps.back().push_back(Gap(Gap::Target, len));
        q += len;
      } else {
        a->gaps.back().push_back(Gap(Gap::Query, len));
        t += len;
      }
    }
    a->score = h.score;
    a->nCells = h.n_cells;
    a->nMatch = h.n_match;
    a->nMismatch = h.n_mismatch;
    a->nIns = h.n_ins;
    a->nDel = h.n_del;
    a->pctSimilarity = h.pct_similarity;
    if (params.forPicard) a->OrderGapsByType();                           // Blasr.cpp:2618-2623
    a->qName = a->tStrand == 0 ? read.GetName() : readRC.GetName();       // :2629-2638
    alignmentPtrs.push_back(a);
  }
  AssignRefContigLocations(alignmentPtrs, seqdb, genome);                 // :2640
}
#endif
