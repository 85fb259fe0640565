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

#include <vector>

#include "blasr_hip.h"

// one context per mapping thread, like MappingBuffers (Blasr.cpp:3238): created by the holder MapReads declares
static __thread blasr_hip_ctx *tlsHipCtx = NULL;

struct BlasrHipThreadContext {
  BlasrHipThreadContext(bool use, int device) {
    if (use && blasr_hip_create(device, &tlsHipCtx) != BLASR_HIP_OK) {
      cout << "ERROR, no gfx950 device for -useGPU." << endl;
      exit(1);
    }
  }
  ~BlasrHipThreadContext() {
    if (tlsHipCtx != NULL) blasr_hip_destroy(tlsHipCtx);
    tlsHipCtx = NULL;
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
#endif
