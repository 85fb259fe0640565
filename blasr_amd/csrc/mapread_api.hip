// mapread_api.hip -- MapRead from the candidate intervals on (alignment/Blasr.cpp:2530-2644) for a block of reads:
// the C++ host driver above the device stages.  The per-interval work runs as three device batches over ALL
// reads of the call (windows -> SDP -> refine); the per-read ordering and pruning between them is host code in the
// reference and here (<= nCandidates records per read), in the reference's order:
//
//   AlignIntervals              windows (:1208-1336), SDPAlign + ComputeAlignmentStats (:1716-1724)
//   std::sort by score          :2547  (SortAlignmentPointersByScore :226-236)
//   RemoveLowQualityAlignments  :2594-2596 (:2022-2079)
//   RefineAlignments            :2607 (:2164-2181; RefineAlignment :850-914 incl. its trimming and copy-back), sort
//   RemoveLowQualityAlignments, RemoveOverlappingAlignments   :2615-2616 (:2081-2161)
//
// Also here: blasr_hip_refine_from_sdp_device -- RefineAlignment's trimming (:850-859) as a kernel on the SDP
// stage's own records, so that no block or sequence returns to the host between the two stages.
// No CPU algorithm on the data path: sorting / filtering of a handful of records per read is the reference's own
// host logic (blasr_hip_sort_candidates, blasr_hip_remove_low_quality, blasr_hip_remove_overlapping).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "../../include/blasr_hip.h"
#include "ctx.h"

namespace {

// slots of blasr_hip_ctx::mapBuf
enum {
  MB_SEQ = 0, MB_QOFF, MB_QLEN, MB_TOFF, MB_TLEN, MB_WIN, MB_SDPRES, MB_BLOCKS, MB_SEL, MB_REFRES, MB_OPS, MB_TRIM,
  MB_FETCH_LIST, MB_FETCH_SEQ, MB_FETCH_OPS, MB_H_READS, MB_H_POOL, MB_H_IVS, MB_X_SEQ, MB_X_BLOCKS, MB_X_TMP, MB_X_OPS,
  MB_KBAND_K
};
static_assert(MB_KBAND_K < blasr_hip_ctx::kMapBufs, "mapBuf slots");

// Candidate i of the refine call = candidate sel[i] of the SDP batch, trimmed to its SDP extent
// (tSeq.Copy(tAlignedSeq, tPos, last.tPos + last.length), qSeq.ReferenceSubstring(query, qAlignedSeqPos + qPos,
// last.qPos + last.length), Blasr.cpp:850-859).
__global__ __launch_bounds__(256) void sdp_trim_kernel(int n, const int32_t *sel, const int64_t *qOff,
                                                       const int64_t *tOff, const blasr_hip_sdp_result *res,
                                                       const int32_t *blocks, int64_t *qOff2, int32_t *qLen2,
                                                       int64_t *tOff2, int32_t *tLen2, int64_t *blkOff2,
                                                       int32_t *nBlk2, int lessFirstT) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int src = sel ? sel[i] : i;
  const blasr_hip_sdp_result r = res[src];
  const int nb = r.status == 0 ? r.n_blocks : 0;  // a candidate whose blocks did not fit the arena has no guide
  int ql = 0, tl = 0;
  if (nb > 0) {
    const int32_t *last = blocks + 3 * (r.blk_off + nb - 1);
    ql = last[0] + last[2];
    tl = last[1] + last[2];
    if (lessFirstT) tl -= blocks[3 * r.blk_off + 1];  // the k-band branch: ... - blocks[0].tPos (Blasr.cpp:979-981)
  }
  qOff2[i] = qOff[src] + r.q_pos;
  tOff2[i] = tOff[src] + r.t_pos;
  qLen2[i] = ql;
  tLen2[i] = tl;
  blkOff2[i] = r.blk_off;
  nBlk2[i] = nb;
}

// The band of RefineAlignment's k-band branch (alignment/Blasr.cpp:943, :990-1003, :699): ComputeDrift over the SDP
// blocks (AlignmentUtils.h:586-618: the largest absolute running sum of tGap - qGap between consecutive blocks), k =
// bandSize or, when that is 0, |drift| * 1.5, at least 10; AffineKBandAlign then gets k * 1.2 -- both products in double
// precision and cut to int, as the reference's conversions do.
__global__ __launch_bounds__(256) void sdp_drift_band_kernel(int n, const int64_t *blkOff, const int32_t *nBlk,
                                                             const int32_t *blocks, int bandSize, int32_t *kOut) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int nb = nBlk[i];
  const int32_t *b = blocks + 3 * blkOff[i];
  int drift = 0, maxDrift = 0;
  for (int j = 0; j + 1 < nb; j++) {
    const int tGap = (int)((uint32_t)b[3 * j + 4] - ((uint32_t)b[3 * j + 1] + (uint32_t)b[3 * j + 2]));
    const int qGap = (int)((uint32_t)b[3 * j + 3] - ((uint32_t)b[3 * j] + (uint32_t)b[3 * j + 2]));
    drift += tGap - qGap;  // (the common gap :593-597 subtracts from both cancels)
    maxDrift = max(maxDrift, abs(drift));
  }
  int k = bandSize == 0 ? (int)(maxDrift * 1.5) : bandSize;
  if (k < 10) k = 10;
  kOut[i] = (int)(k * 1.2);
}

struct FetchSeg {
  int64_t src, dst;  // in elements (bytes for sequences, dwords for ops)
  int64_t len;
};

template <typename T>
__global__ __launch_bounds__(256) void fetch_gather_kernel(int n, const FetchSeg *segs, const T *src, T *dst) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const FetchSeg s = segs[i];
  const T *p = src + s.src;
  T *q = dst + s.dst;
  for (int64_t j = threadIdx.x & 63; j < s.len; j += 64) q[j] = p[j];
}

size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

float pct_of(int nMatch, int nMismatch, int nIns, int nDel) {
  // ComputeAlignmentStats (AlignmentUtils.h:566-577): double arithmetic, stored as float
  const int cols = nMatch + nMismatch + nIns + nDel;
  if (cols == 0) return 0.0f;
  return (float)((nMatch * 2.0) / (double)(cols + cols) * 100);
}

// the trimmed batch of a refine_from_sdp call, carved from one context buffer
struct TrimArrays {
  int64_t *qOff, *tOff, *blkOff;
  int32_t *qLen, *tLen, *nBlk;
};

int carve_trim(blasr_hip_ctx *ctx, int n, TrimArrays &t) {
  const size_t a8 = up256(sizeof(int64_t) * (size_t)n), a4 = up256(sizeof(int32_t) * (size_t)n);
  const int rc = blasr_hip_ensure(ctx, ctx->mapBuf[MB_TRIM], 3 * a8 + 3 * a4);
  if (rc) return rc;
  char *p = ctx->mapBuf[MB_TRIM].p;
  t.qOff = (int64_t *)p;
  t.tOff = (int64_t *)(p + a8);
  t.blkOff = (int64_t *)(p + 2 * a8);
  t.qLen = (int32_t *)(p + 3 * a8);
  t.tLen = (int32_t *)(p + 3 * a8 + a4);
  t.nBlk = (int32_t *)(p + 3 * a8 + 2 * a4);
  return BLASR_HIP_OK;
}

}  // namespace

extern "C" {

int blasr_hip_refine_from_sdp_device(blasr_hip_ctx *ctx, const blasr_hip_sdp_batch *b,
                                     const blasr_hip_sdp_result *sdp_results, const int32_t *blocks, int32_t n_sel,
                                     const int32_t *sel, int32_t band, const blasr_hip_score_params *score,
                                     blasr_hip_refine_result *results, uint32_t *ops, int64_t ops_capacity,
                                     int64_t *ops_used) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (!b || !score || n_sel < 0 || band < 0 || ops_capacity < 0 || (!sel && n_sel != b->n))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "refine_from_sdp: bad argument");
  if (ops_used) *ops_used = 0;
  if (n_sel == 0) return BLASR_HIP_OK;
  if (!b->seq || !b->q_off || !b->t_off || !sdp_results || !blocks || !results)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "refine_from_sdp: null pointer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  TrimArrays t;
  int rc = carve_trim(ctx, n_sel, t);
  if (rc) return rc;
  hipLaunchKernelGGL(sdp_trim_kernel, dim3((unsigned)((n_sel + 255) / 256)), dim3(256), 0, ctx->stream, n_sel, sel,
                     b->q_off, b->t_off, sdp_results, blocks, t.qOff, t.qLen, t.tOff, t.tLen, t.blkOff, t.nBlk, 0);
  blasr_hip_refine_batch rb;
  rb.n = n_sel;
  rb.seq = b->seq;
  rb.q_off = t.qOff;
  rb.q_len = t.qLen;
  rb.t_off = t.tOff;
  rb.t_len = t.tLen;
  rb.blocks = blocks;
  rb.blk_off = t.blkOff;
  rb.n_blocks = t.nBlk;
  return blasr_hip_refine_device(ctx, &rb, band, score, results, ops, ops_capacity, ops_used);
} BLASR_HIP_CATCH_CTX(ctx)

int blasr_hip_kband_from_sdp_device(blasr_hip_ctx *ctx, const blasr_hip_sdp_batch *b,
                                    const blasr_hip_sdp_result *sdp_results, const int32_t *blocks, int32_t n_sel,
                                    const int32_t *sel, int32_t band, int32_t indel, const blasr_hip_score_params *score,
                                    blasr_hip_refine_result *results, uint32_t *ops, int64_t ops_capacity,
                                    int64_t *ops_used) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (!b || !score || n_sel < 0 || band < 0 || ops_capacity < 0 || (!sel && n_sel != b->n))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "kband_from_sdp: bad argument");
  if (ops_used) *ops_used = 0;
  if (n_sel == 0) return BLASR_HIP_OK;
  if (!b->seq || !b->q_off || !b->t_off || !sdp_results || !blocks || !results)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "kband_from_sdp: null pointer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  TrimArrays t;
  int rc = carve_trim(ctx, n_sel, t);
  if (rc) return rc;
  if ((rc = blasr_hip_ensure(ctx, ctx->mapBuf[MB_KBAND_K], sizeof(int32_t) * (size_t)n_sel))) return rc;
  int32_t *k = (int32_t *)ctx->mapBuf[MB_KBAND_K].p;
  const dim3 grid((unsigned)((n_sel + 255) / 256));
  hipLaunchKernelGGL(sdp_trim_kernel, grid, dim3(256), 0, ctx->stream, n_sel, sel, b->q_off, b->t_off, sdp_results, blocks,
                     t.qOff, t.qLen, t.tOff, t.tLen, t.blkOff, t.nBlk, 1);
  hipLaunchKernelGGL(sdp_drift_band_kernel, grid, dim3(256), 0, ctx->stream, n_sel, t.blkOff, t.nBlk, blocks, band, k);
  blasr_hip_kband_batch kb;
  kb.n = n_sel;
  kb.seq = b->seq;
  kb.q_off = t.qOff;
  kb.q_len = t.qLen;
  kb.t_off = t.tOff;
  kb.t_len = t.tLen;
  kb.k = k;
  blasr_hip_kband_params kp;  // PairwiseLocalAlign's call (:695-703)
  memcpy(kp.matrix, score->matrix, sizeof(kp.matrix));
  kp.hp_ins_open = indel + 2;
  kp.hp_ins_extend = indel - 3;
  kp.ins_open = indel + 2;
  kp.ins_extend = indel - 1;
  kp.del = indel;
  int64_t used = 0;
  rc = blasr_hip_kband_device(ctx, &kb, &kp, results, ops, ops_capacity, &used);
  if (ops_used) *ops_used = used;
  if (rc) return rc;
  return blasr_hip_kband_stats_device(ctx, &kb, score, 0, results, ops);  // :740
} BLASR_HIP_CATCH_CTX(ctx)

int blasr_hip_map_set_no_guided_align(blasr_hip_map_params *p, int32_t band_given) try {
  if (!p) return BLASR_HIP_ERR_INVALID;
  p->use_guided_align = 0;
  if (!band_given) p->band = 0;  // (MakeSane sets 16 only with useGuidedAlign, MappingParameters.h:533-545)
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_map_default_params(blasr_hip_map_params *p) try {
  if (!p) return BLASR_HIP_ERR_INVALID;
  memset(p, 0, sizeof(*p));
  p->window.approximate_max_insertion_rate = 1.30f;  // MappingParameters.h
  p->window.extend_ends = 1;
  p->window.do_global_alignment = 0;
  p->sdp.word_size = 11;
  p->sdp.sdp_ins = 5;
  p->sdp.sdp_del = 10;
  p->sdp.indel_rate = 0.30f * 3;  // params.indelRate * 3 (Blasr.cpp:1717)
  p->sdp.align_type = 0;          // Local
  p->sdp.detailed = 1;
  p->sdp.extend_front = 0;
  p->sdp.prefix_length = 50;
  p->sdp.recurse = 2;
  p->sdp.recurse_over = 1000;
  p->sdp.max_per_position = 0;
  static const int32_t smrt[25] = {-5, 6, 6, 6, 0, 6, -5, 6, 6, 0, 6, 6, -5, 6, 0, 6, 6, 6, -5, 0, 0, 0, 0, 0, 0};
  memcpy(p->score.matrix, smrt, sizeof(smrt));  // SMRTDistanceMatrix (ScoreMatrices.h:22-28)
  p->score.ins = 5;
  p->score.del = 5;
  p->score.affine_open = 50;
  p->score.affine_extend = 0;
  p->band = 16;
  p->extend_alignments = 0;    // MappingParameters.h:269
  p->extend_band_size = 10;    // :312
  p->max_extend_dropoff = 10;  // :270
  p->n_candidates = 10;
  p->max_gap = 0;
  p->max_score = -200;
  p->min_pct_identity = 0;
  p->sort_refined = 1;
  p->refine_between_anchors_only = 0;  // MappingParameters.h:298
  p->indel = 5;                        // :195
  p->sdp_bypass_threshold = 1000000.0f;  // :316
  p->use_guided_align = 1;               // :310
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_map_last_timing(const blasr_hip_ctx *ctx, float ms[6], int64_t counters[6]) try {
  if (!ctx || !ms || !counters) return BLASR_HIP_ERR_INVALID;
  for (int i = 0; i < 6; i++) ms[i] = ctx->mapMs[i];
  for (int i = 0; i < 6; i++) counters[i] = ctx->mapCounters[i];
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_map_device_arenas(blasr_hip_ctx *ctx, const uint8_t **seq, const uint32_t **ops) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (seq) *seq = (const uint8_t *)ctx->mapBuf[MB_SEQ].p;
  if (ops) *ops = (const uint32_t *)ctx->mapBuf[MB_OPS].p;
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CTX(ctx)

static int map_reads_impl(blasr_hip_ctx *ctx, int32_t n_reads, const blasr_hip_read *reads,
                          const uint8_t *read_pool, int32_t n_iv, const blasr_hip_interval *intervals,
                          const blasr_hip_map_params *prm, blasr_hip_hit *hits, int64_t hits_capacity,
                          int64_t *n_hits, int64_t *read_hit_off, blasr_hip_map_fetch *fetch);

int blasr_hip_map_reads_device(blasr_hip_ctx *ctx, int32_t n_reads, const blasr_hip_read *reads,
                               const uint8_t *read_pool, int32_t n_iv, const blasr_hip_interval *intervals,
                               const blasr_hip_map_params *prm, blasr_hip_hit *hits, int64_t hits_capacity,
                               int64_t *n_hits, int64_t *read_hit_off, blasr_hip_map_fetch *fetch) try {
  const double t0 = now_ms();
  const int rc = map_reads_impl(ctx, n_reads, reads, read_pool, n_iv, intervals, prm, hits, hits_capacity, n_hits, read_hit_off, fetch);
  if (ctx && getenv("BLASR_HIP_MAP_TRACE")) {
    float sum = 0;
    for (int i = 0; i < 6; i++) sum += ctx->mapMs[i];
    fprintf(stderr, "[map] the call: %.1f ms, its stages %.1f ms\n", now_ms() - t0, sum);
  }
  return rc;
} BLASR_HIP_CATCH_CTX(ctx)

static int map_reads_impl(blasr_hip_ctx *ctx, int32_t n_reads, const blasr_hip_read *reads,
                          const uint8_t *read_pool, int32_t n_iv, const blasr_hip_interval *intervals,
                          const blasr_hip_map_params *prm, blasr_hip_hit *hits, int64_t hits_capacity,
                          int64_t *n_hits, int64_t *read_hit_off, blasr_hip_map_fetch *fetch) {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (n_reads < 0 || n_iv < 0 || !prm || !n_hits || hits_capacity < 0 || (hits_capacity > 0 && !hits) ||
      (n_iv > 0 && (!reads || !read_pool || !intervals)))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "map_reads: bad argument");
  if (fetch && (fetch->seq_capacity < 0 || fetch->ops_capacity < 0 || (fetch->seq_capacity > 0 && !fetch->seq) ||
                (fetch->ops_capacity > 0 && !fetch->ops)))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "map_reads: bad fetch arenas");
  *n_hits = 0;
  if (fetch) fetch->seq_used = fetch->ops_used = 0;
  for (int i = 0; i < 6; i++) ctx->mapMs[i] = 0, ctx->mapCounters[i] = 0;
  if (read_hit_off)
    for (int r = 0; r <= n_reads; r++) read_hit_off[r] = 0;
  if (prm->window.do_global_alignment)  // -global replaces the refiner by KBandAlign over the whole window (:811-831)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_UNSUPPORTED, "map_reads: doGlobalAlignment is not built into the driver "
                          "(blasr_hip_windows_* and blasr_hip_kband_linear_* cover its pieces)");
  if (prm->refine_between_anchors_only && n_iv > 0 && !prm->iv_chain_off)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "map_reads: refine_between_anchors_only needs the intervals' chained "
                          "anchors (iv_chains / iv_chain_off, from blasr_hip_find_intervals_chains)");
  if (n_iv == 0) return BLASR_HIP_OK;
  if (!ctx->genome) return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "map_reads: no genome resident");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t n = (size_t)n_iv;
  int rc;

  // ---- the intervals' owners, on the host (grouping by read) -------------------------------------------------
  const double tEntry = now_ms();
  PinnedArray<blasr_hip_interval> iv;
  if ((rc = pinned_array(ctx, MH_IV, n, iv))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(iv.data(), intervals, sizeof(blasr_hip_interval) * n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  for (size_t k = 0; k < n; k++)
    if (iv[k].read < 0 || iv[k].read >= n_reads)
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "map_reads: interval %zu names read %d of %d", k, iv[k].read,
                            n_reads);

  // ---- a-1: windows ----------------------------------------------------------------------------------------------
  double t0 = now_ms();
  if (getenv("BLASR_HIP_MAP_TRACE")) fprintf(stderr, "[map] before the first stage: %.1f ms\n", t0 - tEntry);
  DevBuf *mb = ctx->mapBuf;
  if ((rc = blasr_hip_ensure(ctx, mb[MB_QOFF], sizeof(int64_t) * n))) return rc;
  if ((rc = blasr_hip_ensure(ctx, mb[MB_TOFF], sizeof(int64_t) * n))) return rc;
  if ((rc = blasr_hip_ensure(ctx, mb[MB_QLEN], sizeof(int32_t) * n))) return rc;
  if ((rc = blasr_hip_ensure(ctx, mb[MB_TLEN], sizeof(int32_t) * n))) return rc;
  if ((rc = blasr_hip_ensure(ctx, mb[MB_WIN], sizeof(blasr_hip_window) * n))) return rc;
  if ((rc = blasr_hip_ensure(ctx, mb[MB_SEQ], 1 << 20))) return rc;
  int64_t seqUsed = 0;
  for (int attempt = 0; attempt < 2; attempt++) {
    rc = blasr_hip_windows_device(ctx, n_iv, intervals, reads, read_pool, &prm->window, (uint8_t *)mb[MB_SEQ].p,
                                  (int64_t)mb[MB_SEQ].bytes, (int64_t *)mb[MB_QOFF].p, (int32_t *)mb[MB_QLEN].p,
                                  (int64_t *)mb[MB_TOFF].p, (int32_t *)mb[MB_TLEN].p, (blasr_hip_window *)mb[MB_WIN].p,
                                  &seqUsed);
    if (rc != BLASR_HIP_ERR_CAPACITY) break;
    const int rc2 = blasr_hip_ensure(ctx, mb[MB_SEQ], (size_t)seqUsed + 64);
    if (rc2) return rc2;
  }
  if (rc) return rc;
  PinnedArray<blasr_hip_window> win;
  PinnedArray<int32_t> qLen, tLen;
  PinnedArray<int64_t> qOff, tOff;
  if ((rc = pinned_array(ctx, MH_WIN, n, win)) || (rc = pinned_array(ctx, MH_QLEN, n, qLen)) || (rc = pinned_array(ctx, MH_TLEN, n, tLen)) ||
      (rc = pinned_array(ctx, MH_QOFF, n, qOff)) || (rc = pinned_array(ctx, MH_TOFF, n, tOff)))
    return rc;
  HIP_TRY(ctx, hipMemcpyAsync(win.data(), mb[MB_WIN].p, sizeof(blasr_hip_window) * n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(qLen.data(), mb[MB_QLEN].p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(tLen.data(), mb[MB_TLEN].p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(qOff.data(), mb[MB_QOFF].p, sizeof(int64_t) * n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(tOff.data(), mb[MB_TOFF].p, sizeof(int64_t) * n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  double t1 = now_ms();
  ctx->mapMs[0] = (float)(t1 - t0);

  // ---- a-2 .. a-5, a-9: SDPAlign + ComputeAlignmentStats of every interval ------------------------------------
  if ((rc = blasr_hip_ensure(ctx, mb[MB_SDPRES], sizeof(blasr_hip_sdp_result) * n))) return rc;
  int64_t blkCap = 0;
  for (size_t k = 0; k < n; k++) blkCap += std::min(qLen[k], tLen[k]) / 3 + 64;
  blasr_hip_sdp_batch sb;
  sb.n = n_iv;
  sb.seq = (const uint8_t *)mb[MB_SEQ].p;
  sb.q_off = (const int64_t *)mb[MB_QOFF].p;
  sb.q_len = (const int32_t *)mb[MB_QLEN].p;
  sb.t_off = (const int64_t *)mb[MB_TOFF].p;
  sb.t_len = (const int32_t *)mb[MB_TLEN].p;
  int64_t blkUsed = 0;
  for (int attempt = 0; attempt < 2; attempt++) {
    if ((rc = blasr_hip_ensure(ctx, mb[MB_BLOCKS], sizeof(int32_t) * 3 * (size_t)blkCap))) return rc;
    blkCap = (int64_t)(mb[MB_BLOCKS].bytes / (3 * sizeof(int32_t)));
    if (prm->refine_between_anchors_only) {  // -rbao (:1354-1712): alignments between the chained anchors instead
      blasr_hip_rbao_params rp;
      memset(&rp, 0, sizeof(rp));
      rp.sdp = prm->sdp;
      rp.score = prm->score;
      rp.band = prm->band;
      rp.indel = prm->indel;
      rp.sdp_bypass_threshold = prm->sdp_bypass_threshold;
      rc = blasr_hip_align_between_anchors_device(ctx, &sb, (const blasr_hip_window *)mb[MB_WIN].p, (uint32_t)ctx->genomeLen,
                                                  prm->iv_chains, prm->iv_chain_off, &rp,
                                                  (blasr_hip_sdp_result *)mb[MB_SDPRES].p, (int32_t *)mb[MB_BLOCKS].p, blkCap,
                                                  &blkUsed);
    } else {
      rc = blasr_hip_sdp_device(ctx, &sb, &prm->sdp, &prm->score, (blasr_hip_sdp_result *)mb[MB_SDPRES].p,
                                (int32_t *)mb[MB_BLOCKS].p, blkCap, &blkUsed);
    }
    if (rc != BLASR_HIP_ERR_CAPACITY) break;
    blkCap = blkUsed + 64;  // rare: the blocks of the batch did not fit -- once more with what it takes
  }
  if (rc) return rc;
  if (prm->extend_alignments) {  // -extend (:1749-1940): both ends of every SDP alignment, then its statistics again
    blasr_hip_extend_params ep;
    blasr_hip_extend_default_params(&ep);
    memcpy(ep.matrix, prm->score.matrix, sizeof(ep.matrix));
    ep.ins = prm->score.ins;
    ep.del = prm->score.del;
    ep.band = prm->extend_band_size;
    ep.max_n_drops = prm->max_extend_dropoff;
    rc = blasr_hip_extend_candidates(ctx, n_iv, intervals, reads, read_pool, &ep, &prm->score, (const uint8_t *)mb[MB_SEQ].p,
                                     (int64_t *)mb[MB_QOFF].p, (int32_t *)mb[MB_QLEN].p, (int64_t *)mb[MB_TOFF].p,
                                     (int32_t *)mb[MB_TLEN].p, (blasr_hip_window *)mb[MB_WIN].p,
                                     (blasr_hip_sdp_result *)mb[MB_SDPRES].p, (const int32_t *)mb[MB_BLOCKS].p,
                                     mb[MB_X_SEQ], mb[MB_X_BLOCKS], mb[MB_X_TMP], mb[MB_X_OPS], &blkUsed);
    if (rc) return rc;
    std::swap(mb[MB_SEQ], mb[MB_X_SEQ]);  // the candidates' sequences and blocks now live in the new arenas
    std::swap(mb[MB_BLOCKS], mb[MB_X_BLOCKS]);
    sb.seq = (const uint8_t *)mb[MB_SEQ].p;
    HIP_TRY(ctx, hipMemcpyAsync(win.data(), mb[MB_WIN].p, sizeof(blasr_hip_window) * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(qLen.data(), mb[MB_QLEN].p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(tLen.data(), mb[MB_TLEN].p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(qOff.data(), mb[MB_QOFF].p, sizeof(int64_t) * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(tOff.data(), mb[MB_TOFF].p, sizeof(int64_t) * n, hipMemcpyDeviceToHost, st));
  }
  PinnedArray<blasr_hip_sdp_result> sdp;
  if ((rc = pinned_array(ctx, MH_SDP, n, sdp))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(sdp.data(), mb[MB_SDPRES].p, sizeof(blasr_hip_sdp_result) * n, hipMemcpyDeviceToHost, st));
  std::vector<int32_t> hostBlocks;  // the maxGap rule of RemoveLowQualityAlignments reads the blocks
  if (prm->max_gap > 0 && blkUsed > 0) {
    hostBlocks.resize(3 * (size_t)blkUsed);
    HIP_TRY(ctx, hipMemcpyAsync(hostBlocks.data(), mb[MB_BLOCKS].p, sizeof(int32_t) * hostBlocks.size(),
                                hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(ctx, hipStreamSynchronize(st));
  double t2 = now_ms();
  ctx->mapMs[1] = (float)(t2 - t1);

  // ---- per read: the candidates as AlignIntervals leaves them, sort, first filter -------------------------------
  // candidates of a read in the order of its intervals (a counting sort over the owners: stable)
  // (these arrays too live in the context between calls: 20 MB of fresh pages per call otherwise, and their unmapping)
  PinnedArray<int64_t> first, at, selFirst;
  PinnedArray<int32_t> byRead, order, live, sel;
  PinnedArray<blasr_hip_candidate> cand;  // indexed like byRead (read-major)
  if ((rc = pinned_array(ctx, MH_FIRST, (size_t)n_reads + 1, first)) || (rc = pinned_array(ctx, MH_AT, (size_t)n_reads + 1, at)) ||
      (rc = pinned_array(ctx, MH_SELFIRST, (size_t)n_reads + 1, selFirst)) || (rc = pinned_array(ctx, MH_BYREAD, n, byRead)) ||
      (rc = pinned_array(ctx, MH_ORDER, n, order)) ||  // per read: its slice holds the live candidates, as local indices
      (rc = pinned_array(ctx, MH_LIVE, (size_t)std::max(n_reads, 1), live)) ||
      (rc = pinned_array(ctx, MH_SEL, n, sel)) ||      // survivors of the first filter, read-major, in rank order
      (rc = pinned_array(ctx, MH_CAND, n, cand)))
    return rc;
  memset(first.data(), 0, sizeof(int64_t) * ((size_t)n_reads + 1));
  memset(selFirst.data(), 0, sizeof(int64_t) * ((size_t)n_reads + 1));
  memset(live.data(), 0, sizeof(int32_t) * (size_t)std::max(n_reads, 1));
  memset(cand.data(), 0, sizeof(blasr_hip_candidate) * n);
  for (size_t k = 0; k < n; k++) first[(size_t)iv[k].read + 1]++;
  for (int r = 0; r < n_reads; r++) first[(size_t)r + 1] += first[r];
  {
    memcpy(at.data(), first.data(), sizeof(int64_t) * (size_t)n_reads);
    for (size_t k = 0; k < n; k++) byRead[(size_t)at[iv[k].read]++] = (int32_t)k;
  }
  size_t nSel = 0;
  // (the reads are independent: several host threads over them; the survivors' list is put together afterwards)
  std::atomic<int> badRead(-1), badCode(0);
  auto filter_reads = [&](int r0, int r1) {
    for (int r = r0; r < r1; r++) {
      const int64_t f = first[r];
      const int32_t m = (int32_t)(first[(size_t)r + 1] - f);
      for (int32_t j = 0; j < m; j++) {
        const int32_t k = byRead[(size_t)(f + j)];
        const blasr_hip_sdp_result &s = sdp[k];
        blasr_hip_candidate &c = cand[(size_t)(f + j)];
        const int nb = s.status == 0 ? s.n_blocks : 0;
        c.score = s.stats_score;
        c.t_pos = (uint32_t)s.t_pos;
        c.t_aligned_seq_pos = win[k].t_aligned_seq_pos;
        c.t_end = 0;  // TEnd() is only read after the refinement
        c.t_index = win[k].t_index;
        c.t_strand = win[k].t_strand;
        c.pct_similarity = pct_of(s.n_match, s.n_mismatch, s.n_ins, s.n_del);
        c.n_blocks = nb;
        c.blk_off = s.blk_off;
        order[(size_t)(f + j)] = j;
      }
      int32_t cnt = m;
      if (m > 0) {
        blasr_hip_sort_candidates(&cand[(size_t)f], m, &order[(size_t)f]);                        // :2547
        cnt = blasr_hip_remove_low_quality(&cand[(size_t)f], hostBlocks.empty() ? nullptr : hostBlocks.data(), m,
                                           &order[(size_t)f], prm->n_candidates, prm->max_gap, prm->max_score);  // :2594
        if (cnt < 0) {
          badRead.store(r);
          badCode.store(cnt);
          cnt = 0;
        }
      }
      live[r] = cnt;
    }
  };
  {
    const int threads = n_reads >= 8192 ? blasr_hip_host_threads((int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()))) : 1;
    const int chunk = (n_reads + threads - 1) / threads;
    const int trc = blasr_hip_run_threads(threads, [&](int t) {
      const int r0 = std::min(n_reads, t * chunk), r1 = std::min(n_reads, r0 + chunk);
      if (r0 < r1) filter_reads(r0, r1);
    });
    if (trc) return blasr_hip_fail(ctx, trc, "map_reads: a filter thread failed (out of host memory)");
  }
  if (badRead.load() >= 0) return blasr_hip_fail(ctx, badCode.load(), "map_reads: first filter failed for read %d", badRead.load());
  for (int r = 0; r < n_reads; r++) {
    const int64_t f = first[r];
    selFirst[r] = (int64_t)nSel;
    for (int32_t i = 0; i < live[r]; i++) sel[nSel++] = byRead[(size_t)(f + order[(size_t)(f + i)])];
  }
  selFirst[n_reads] = (int64_t)nSel;
  double t3 = now_ms();
  ctx->mapMs[2] = (float)(t3 - t2);

  // ---- a-6 .. a-9: RefineAlignments on the survivors ----------------------------------------------------------
  const size_t ns = nSel;
  PinnedArray<blasr_hip_refine_result> ref;
  PinnedArray<int32_t> qLen2, tLen2;
  if ((rc = pinned_array(ctx, MH_REF, ns, ref)) || (rc = pinned_array(ctx, MH_QLEN2, ns, qLen2)) || (rc = pinned_array(ctx, MH_TLEN2, ns, tLen2)))
    return rc;
  if (ns > 0) {
    if ((rc = blasr_hip_ensure(ctx, mb[MB_SEL], sizeof(int32_t) * ns))) return rc;
    if ((rc = blasr_hip_ensure(ctx, mb[MB_REFRES], sizeof(blasr_hip_refine_result) * ns))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(mb[MB_SEL].p, sel.data(), sizeof(int32_t) * ns, hipMemcpyHostToDevice, st));
    int64_t opsCap = 0;
    for (size_t i = 0; i < ns; i++) opsCap += ((int64_t)qLen[sel[i]] + tLen[sel[i]]) / 2 + 1024;
    int64_t opsUsed = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
      if ((rc = blasr_hip_ensure(ctx, mb[MB_OPS], sizeof(uint32_t) * (size_t)opsCap))) return rc;
      opsCap = (int64_t)(mb[MB_OPS].bytes / sizeof(uint32_t));
      if (prm->use_guided_align)
        rc = blasr_hip_refine_from_sdp_device(ctx, &sb, (const blasr_hip_sdp_result *)mb[MB_SDPRES].p,
                                              (const int32_t *)mb[MB_BLOCKS].p, (int32_t)ns, (const int32_t *)mb[MB_SEL].p,
                                              prm->band, &prm->score, (blasr_hip_refine_result *)mb[MB_REFRES].p,
                                              (uint32_t *)mb[MB_OPS].p, opsCap, &opsUsed);
      else  // -noUseGuidedAlign: RefineAlignment's k-band branch (:917-1046)
        rc = blasr_hip_kband_from_sdp_device(ctx, &sb, (const blasr_hip_sdp_result *)mb[MB_SDPRES].p,
                                             (const int32_t *)mb[MB_BLOCKS].p, (int32_t)ns, (const int32_t *)mb[MB_SEL].p,
                                             prm->band, prm->indel, &prm->score, (blasr_hip_refine_result *)mb[MB_REFRES].p,
                                             (uint32_t *)mb[MB_OPS].p, opsCap, &opsUsed);
      if (rc != BLASR_HIP_ERR_CAPACITY) break;
      opsCap = opsUsed + 64;
    }
    if (rc) return rc;
    TrimArrays t;
    if ((rc = carve_trim(ctx, (int)ns, t))) return rc;  // (the arrays the call above left behind)
    HIP_TRY(ctx, hipMemcpyAsync(ref.data(), mb[MB_REFRES].p, sizeof(blasr_hip_refine_result) * ns, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(qLen2.data(), t.qLen, sizeof(int32_t) * ns, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(tLen2.data(), t.tLen, sizeof(int32_t) * ns, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
  }
  std::vector<uint32_t> hostOps;  // refined blocks for the maxGap rule of the second RemoveLowQualityAlignments
  if (prm->max_gap > 0 && ns > 0) {
    int64_t total = 0;
    for (size_t i = 0; i < ns; i++) total = std::max<int64_t>(total, ref[i].ops_off + ref[i].n_ops);
    hostOps.resize((size_t)total);
    if (total > 0) HIP_TRY(ctx, hipMemcpy(hostOps.data(), mb[MB_OPS].p, sizeof(uint32_t) * (size_t)total, hipMemcpyDeviceToHost));
  }
  double t4 = now_ms();
  ctx->mapMs[3] = (float)(t4 - t3);

  // ---- per read: copy-back (:888-914), sort (:2178-2180), the two filters (:2615-2616) ---------------------------
  int64_t nHits = 0, alignedReads = 0, alignedBases = 0, cells = 0;
  std::vector<FetchSeg> seqSegs, opsSegs;
  int64_t fetchSeq = 0, fetchOps = 0;
  for (size_t i = 0; i < ns; i++)
    if ((ref[i].status & (BLASR_HIP_TASK_EMPTY_GUIDE | BLASR_HIP_TASK_BAD_GUIDE)) == 0) cells += ref[i].n_cells;
  // Per read and independent of the other reads (several host threads): the refined values copied back into its
  // candidates, the sort, the two filters; live[r] becomes the number that survives, slotOf the refine record of each.
  // The hits are then written in read order by one thread.
  PinnedArray<int32_t> slotOf;  // candidate (read-major index) -> its refine record, -1 without one
  if ((rc = pinned_array(ctx, MH_SLOT, n, slotOf))) return rc;
  std::atomic<int> badRead2(-1), badCode2(0), badCand2(-1), badStatus2(0);
  const bool tolerate = getenv("BLASR_HIP_EXP") != nullptr;  // (BLASR_HIP_EXP: timing experiments with wrong results)
  auto refine_filters = [&](int r0, int r1) {
    std::vector<int32_t> blocks2;  // per read, only under max_gap > 0
    for (int r = r0; r < r1; r++) {
      const int64_t f = first[r];
      int32_t cnt = live[r];
      const int32_t m = (int32_t)(first[(size_t)r + 1] - f);
      for (int32_t j = 0; j < m; j++) slotOf[(size_t)(f + j)] = -1;
      if (cnt == 0) continue;
      blasr_hip_candidate *cr = &cand[(size_t)f];
      int32_t *ord = &order[(size_t)f];
      blocks2.clear();
      bool failed = false;
      for (int32_t i = 0; i < cnt; i++) {
        const int32_t j = ord[i];
        const size_t s = (size_t)(selFirst[r] + i);
        const blasr_hip_refine_result &rr = ref[s];
        if (rr.status & BLASR_HIP_TASK_EMPTY_GUIDE) {  // no blocks: the guided branch leaves the candidate as it is
          if (!prm->use_guided_align) cr[j].score = 0;  // (:845), the k-band branch sets its score to 0 (:959-962)
          continue;
        }
        if (rr.status != 0 && !tolerate) {
          badRead2.store(r);
          badCand2.store(byRead[(size_t)(f + j)]);
          badStatus2.store(rr.status);
          failed = true;
          break;
        }
        slotOf[(size_t)(f + j)] = (int32_t)s;
        blasr_hip_candidate &c = cr[j];
        c.score = rr.score;
        c.t_aligned_seq_pos += c.t_pos;  // :909 (tPos of the SDP alignment)
        c.t_pos = (uint32_t)rr.t_pos;
        c.t_end = (uint32_t)(rr.n_match + rr.n_mismatch + rr.n_del);  // TEnd(): the blocks start at 0 and end with one
        c.pct_similarity = pct_of(rr.n_match, rr.n_mismatch, rr.n_ins, rr.n_del);
        if (prm->max_gap > 0) {  // the refined blocks, rebuilt from the ops
          c.blk_off = (int64_t)(blocks2.size() / 3);
          int32_t q = 0, t = 0, nb = 0;
          for (int32_t o = 0; o < rr.n_ops; o++) {
            const uint32_t op = hostOps[(size_t)(rr.ops_off + o)], len = op >> 2;
            if ((op & 3u) == BLASR_HIP_OP_M) {
              blocks2.push_back(q), blocks2.push_back(t), blocks2.push_back((int32_t)len);
              q += len, t += len, nb++;
            } else if ((op & 3u) == BLASR_HIP_OP_I) q += len;
            else t += len;
          }
          c.n_blocks = nb;
        } else {
          c.n_blocks = 1;  // (only "has blocks" is read below)
        }
      }
      if (failed) {
        live[r] = 0;
        continue;
      }
      if (prm->sort_refined) blasr_hip_sort_candidates(cr, cnt, ord);
      cnt = blasr_hip_remove_low_quality(cr, blocks2.empty() ? nullptr : blocks2.data(), cnt, ord, prm->n_candidates,
                                         prm->max_gap, prm->max_score);
      if (cnt >= 0) cnt = blasr_hip_remove_overlapping(cr, cnt, ord, prm->min_pct_identity);
      if (cnt < 0) {
        badRead2.store(r);
        badCode2.store(cnt);
        cnt = 0;
      }
      live[r] = cnt;
    }
  };
  {
    const int threads = n_reads >= 8192 ? blasr_hip_host_threads((int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()))) : 1;
    const int chunk = (n_reads + threads - 1) / threads;
    const int trc = blasr_hip_run_threads(threads, [&](int t) {
      const int r0 = std::min(n_reads, t * chunk), r1 = std::min(n_reads, r0 + chunk);
      if (r0 < r1) refine_filters(r0, r1);
    });
    if (trc) return blasr_hip_fail(ctx, trc, "map_reads: a filter thread failed (out of host memory)");
  }
  if (badStatus2.load() != 0)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_HIP, "map_reads: candidate %d of read %d: refine status %d "
                          "(the reference would exit: AffineGuidedAlign.h:394-408)", badCand2.load(), badRead2.load(), badStatus2.load());
  if (badCode2.load() != 0) return blasr_hip_fail(ctx, badCode2.load(), "map_reads: a filter after the refinement failed for read %d", badRead2.load());
  for (int r = 0; r < n_reads; r++) {
    if (read_hit_off) read_hit_off[r] = nHits;
    const int64_t f = first[r];
    const int32_t cnt = live[r];
    if (cnt == 0) continue;
    const blasr_hip_candidate *cr = &cand[(size_t)f];
    const int32_t *ord = &order[(size_t)f];
    const int32_t *slot = &slotOf[(size_t)f];
    int32_t kept = 0;
    for (int32_t i = 0; i < cnt; i++) {
      const int32_t j = ord[i];
      if (slot[j] < 0) continue;
      const int32_t k = byRead[(size_t)(f + j)];
      const size_t s = (size_t)slot[j];
      const blasr_hip_refine_result &rr = ref[s];
      const blasr_hip_candidate &c = cr[j];
      if (nHits < hits_capacity) {
        blasr_hip_hit &h = hits[nHits];
        h.read = r;
        h.interval = k;
        h.t_strand = c.t_strand;
        h.t_index = c.t_index;
        h.t_aligned_seq_pos = c.t_aligned_seq_pos;
        h.t_aligned_seq_length = (uint32_t)tLen2[s];
        h.t_length = win[k].t_length;
        h.q_aligned_seq_pos = win[k].q_aligned_seq_pos + (uint32_t)sdp[k].q_pos;  // :910
        h.q_aligned_seq_length = (uint32_t)qLen2[s];
        h.q_length = win[k].q_length;
        h.q_pos = rr.q_pos;
        h.t_pos = rr.t_pos;
        h.score = c.score;  // the refined score, or maxScore where the maxGap rule flagged it (:2041)
        h.n_match = rr.n_match;
        h.n_mismatch = rr.n_mismatch;
        h.n_ins = rr.n_ins;
        h.n_del = rr.n_del;
        h.n_cells = rr.n_cells;
        h.pct_similarity = c.pct_similarity;
        h.n_ops = rr.n_ops;
        h.ops_off = rr.ops_off;
        h.q_off = qOff[k] + sdp[k].q_pos;
        h.t_off = tOff[k] + sdp[k].t_pos;
        if (fetch) {
          FetchSeg a = {h.q_off, fetchSeq, (int64_t)qLen2[s]};
          h.q_off = fetchSeq;
          fetchSeq += a.len;
          FetchSeg b = {h.t_off, fetchSeq, (int64_t)tLen2[s]};
          h.t_off = fetchSeq;
          fetchSeq += b.len;
          FetchSeg o = {h.ops_off, fetchOps, (int64_t)rr.n_ops};
          h.ops_off = fetchOps;
          fetchOps += o.len;
          seqSegs.push_back(a), seqSegs.push_back(b), opsSegs.push_back(o);
        }
      }
      nHits++;
      kept++;
    }
    if (kept > 0) {
      alignedReads++;
      alignedBases += win[byRead[(size_t)f]].q_length;
    }
  }
  if (read_hit_off) read_hit_off[n_reads] = nHits;
  double t5 = now_ms();
  ctx->mapMs[4] = (float)(t5 - t4);
  if (getenv("BLASR_HIP_MAP_TRACE")) fprintf(stderr, "[map] stages done %.1f ms after the entry\n", t5 - tEntry);
  *n_hits = nHits;
  ctx->mapCounters[0] = n_iv;
  ctx->mapCounters[1] = (int64_t)ns;
  ctx->mapCounters[2] = alignedReads;
  ctx->mapCounters[3] = alignedBases;
  ctx->mapCounters[4] = cells;
  ctx->mapCounters[5] = nHits;
  if (nHits > hits_capacity)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_CAPACITY, "map_reads: %lld hits, room for %lld", (long long)nHits,
                          (long long)hits_capacity);

  // ---- what the epilogue reads: aligned sequences and op lists of the hits, one download each -------------------
  if (fetch) {
    fetch->seq_used = fetchSeq;
    fetch->ops_used = fetchOps;
    if (fetchSeq > fetch->seq_capacity || fetchOps > fetch->ops_capacity)
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_CAPACITY, "map_reads: fetch needs %lld sequence bytes and %lld ops",
                            (long long)fetchSeq, (long long)fetchOps);
    if (!seqSegs.empty()) {
      const size_t nSeg = seqSegs.size() + opsSegs.size();
      if ((rc = blasr_hip_ensure(ctx, mb[MB_FETCH_LIST], sizeof(FetchSeg) * nSeg))) return rc;
      if ((rc = blasr_hip_ensure(ctx, mb[MB_FETCH_SEQ], (size_t)std::max<int64_t>(fetchSeq, 1)))) return rc;
      if ((rc = blasr_hip_ensure(ctx, mb[MB_FETCH_OPS], sizeof(uint32_t) * (size_t)std::max<int64_t>(fetchOps, 1)))) return rc;
      FetchSeg *dSeg = (FetchSeg *)mb[MB_FETCH_LIST].p;
      HIP_TRY(ctx, hipMemcpyAsync(dSeg, seqSegs.data(), sizeof(FetchSeg) * seqSegs.size(), hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipMemcpyAsync(dSeg + seqSegs.size(), opsSegs.data(), sizeof(FetchSeg) * opsSegs.size(),
                                  hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(fetch_gather_kernel<uint8_t>, dim3((unsigned)((seqSegs.size() + 3) / 4)), dim3(256), 0, st,
                         (int)seqSegs.size(), dSeg, (const uint8_t *)mb[MB_SEQ].p, (uint8_t *)mb[MB_FETCH_SEQ].p);
      hipLaunchKernelGGL(fetch_gather_kernel<uint32_t>, dim3((unsigned)((opsSegs.size() + 3) / 4)), dim3(256), 0, st,
                         (int)opsSegs.size(), dSeg + seqSegs.size(), (const uint32_t *)mb[MB_OPS].p,
                         (uint32_t *)mb[MB_FETCH_OPS].p);
      if (fetchSeq > 0)
        HIP_TRY(ctx, hipMemcpyAsync(fetch->seq, mb[MB_FETCH_SEQ].p, (size_t)fetchSeq, hipMemcpyDeviceToHost, st));
      if (fetchOps > 0)
        HIP_TRY(ctx, hipMemcpyAsync(fetch->ops, mb[MB_FETCH_OPS].p, sizeof(uint32_t) * (size_t)fetchOps,
                                    hipMemcpyDeviceToHost, st));
      HIP_TRY(ctx, hipStreamSynchronize(st));
      HIP_TRY(ctx, hipGetLastError());
    }
    ctx->mapMs[5] = (float)(now_ms() - t5);
  }
  return BLASR_HIP_OK;
}

int blasr_hip_map_reads_host(blasr_hip_ctx *ctx, int32_t n_reads, const blasr_hip_read *reads,
                             const uint8_t *read_pool, int64_t read_pool_bytes, int32_t n_iv,
                             const blasr_hip_interval *intervals, const blasr_hip_map_params *prm,
                             blasr_hip_hit *hits, int64_t hits_capacity, int64_t *n_hits, int64_t *read_hit_off,
                             blasr_hip_map_fetch *fetch) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (n_reads < 0 || n_iv < 0 || read_pool_bytes < 0 || (n_reads > 0 && !reads) || (read_pool_bytes > 0 && !read_pool) ||
      (n_iv > 0 && !intervals))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "map_reads: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  DevBuf *mb = ctx->mapBuf;
  int rc;
  if ((rc = blasr_hip_ensure(ctx, mb[MB_H_READS], sizeof(blasr_hip_read) * (size_t)std::max(n_reads, 1)))) return rc;
  if ((rc = blasr_hip_ensure(ctx, mb[MB_H_POOL], (size_t)std::max<int64_t>(read_pool_bytes, 1)))) return rc;
  if ((rc = blasr_hip_ensure(ctx, mb[MB_H_IVS], sizeof(blasr_hip_interval) * (size_t)std::max(n_iv, 1)))) return rc;
  hipStream_t st = ctx->stream;
  if (n_reads > 0) HIP_TRY(ctx, hipMemcpyAsync(mb[MB_H_READS].p, reads, sizeof(blasr_hip_read) * (size_t)n_reads, hipMemcpyHostToDevice, st));
  if (read_pool_bytes > 0) HIP_TRY(ctx, hipMemcpyAsync(mb[MB_H_POOL].p, read_pool, (size_t)read_pool_bytes, hipMemcpyHostToDevice, st));
  if (n_iv > 0) HIP_TRY(ctx, hipMemcpyAsync(mb[MB_H_IVS].p, intervals, sizeof(blasr_hip_interval) * (size_t)n_iv, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return blasr_hip_map_reads_device(ctx, n_reads, (const blasr_hip_read *)mb[MB_H_READS].p,
                                    (const uint8_t *)mb[MB_H_POOL].p, n_iv, (const blasr_hip_interval *)mb[MB_H_IVS].p,
                                    prm, hits, hits_capacity, n_hits, read_hit_off, fetch);
} BLASR_HIP_CATCH_CTX(ctx)

}  // extern "C"
