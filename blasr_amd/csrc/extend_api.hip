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
}  // namespace

extern "C" {

int blasr_hip_extend_default_params(blasr_hip_extend_params *p) {
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
}

int blasr_hip_extend_device(blasr_hip_ctx *ctx, const blasr_hip_extend_batch *b, const blasr_hip_extend_params *p,
                            blasr_hip_refine_result *results, uint32_t *ops, int64_t ops_capacity, int64_t *ops_used) {
  return extend_run(ctx, b, p, results, ops, ops_capacity, ops_used);
}

int blasr_hip_extend_host(blasr_hip_ctx *ctx, const blasr_hip_extend_batch *b, const blasr_hip_extend_params *p,
                          blasr_hip_refine_result *results, uint32_t *ops, int64_t ops_capacity, int64_t *ops_used) {
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
}

}  // extern "C"
