// rbao_api.hip -- next-3: -rbao (params.refineBetweenAnchorsOnly), the branch of AlignIntervals that aligns only
// BETWEEN the chained anchors of a candidate interval (alignment/Blasr.cpp:1354-1712) instead of running one SDPAlign
// over the whole window.
//
// The anchors of an interval are a few hundred (t, q, l) triplets: making them relative to the window, the overlap
// filter and the chain of runs (:1385-1564) are host code per candidate, threaded over the candidates.  The work is in
// the gaps: one AlignSubstring (:1050-1122) per gap, i.e. per candidate a few hundred small alignments of two kinds --
//   AffineKBandAlign (Global, k = bandSize)                          -> blasr_hip_kband_device, one task per gap
//   SDPAlign (Global) + pinned end blocks + RefineAlignment           -> blasr_hip_sdp_device, then blasr_hip_refine_device
// run as device batches over the gaps of ALL candidates of the call (the sequences are the windows already on the device:
// a gap is an offset and a length into them).  The host then splices the gaps' blocks between the anchors
// (AppendSubseqAlignment :1124-1147), merges abutting blocks (:1666-1700), and rbao_stats_kernel computes the
// statistics of the block lists (ComputeAlignmentStats of an alignment without gap lists, :1709).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <stdexcept>
#include <vector>

#include "../../include/blasr_hip.h"
#include "ctx.h"
#include "wave_utils.h"

using namespace blasr_hip;

namespace {

struct Anchor {
  uint32_t t, q, l;  // DNALength arithmetic throughout, as in ChainedMatchPos
};

// Blasr.cpp:1385-1564: the interval's anchors as the gap loop sees them
void prepare_anchors(std::vector<Anchor> &m, const blasr_hip_window &w, uint32_t genomeLength) {
  // :1391-1397 relative to the window (a Reverse window is named by its place on the reverse strand)
  uint32_t targetOffset = w.t_aligned_seq_pos;
  if (w.t_strand == 1) targetOffset = genomeLength - (w.t_aligned_seq_pos + w.t_aligned_seq_length - 1) - 1;  // MakeRCCoordinate
  for (size_t i = 0; i < m.size(); i++) m[i].t -= targetOffset;
  if (w.t_strand == 1) {  // :1402-1408
    for (size_t i = 0; i < m.size(); i++) {
      m[i].q = w.q_aligned_seq_length - (m[i].q + m[i].l);
      m[i].t = w.t_aligned_seq_length - (m[i].t + m[i].l);
    }
    std::reverse(m.begin(), m.end());
  }
  // :1414-1439 overlapping anchors
  std::vector<char> toRemove(m.size(), 0);
  if (m.size() > 0) {
    size_t a = 0;
    while (a < m.size() - 1) {
      size_t n = a + 1;
      while (n < m.size() && ((m[a].q + m[a].l > m[n].q) || (m[a].t + m[a].l > m[n].t))) {
        toRemove[n] = 1;
        n++;
      }
      a = n;
    }
    size_t k = 0;
    for (size_t n = 0; n < m.size(); n++)
      if (!toRemove[n]) m[k++] = m[n];
    m.resize(k);
  }
  // :1443-1564 runs of anchors on one diagonal (within MAX_GAP), the best chain of runs, everything else dropped
  if (m.size() > 0) {
    toRemove.assign(m.size(), 0);
    const int MAX_GAP = 25;
    std::vector<int> bStart, bEnd, bSize, bScore, bPrev;
    bStart.push_back(0), bEnd.push_back(0), bSize.push_back(0), bScore.push_back(0), bPrev.push_back(0);
    size_t a = 0;
    while (a < m.size() - 1) {
      int blockSize = 0;
      const int blockStart = (int)a;
      int qNextGap = 0, tNextGap = 0;
      while (a < m.size() - 1 && abs(tNextGap - qNextGap) < MAX_GAP) {
        qNextGap = (int)(m[a + 1].q - m[a].q + m[a].l);  // (sic: + l, :1465-1466)
        tNextGap = (int)(m[a + 1].t - m[a].t + m[a].l);
        blockSize += (int)m[a].l;
        a++;
      }
      bStart.push_back(blockStart), bEnd.push_back((int)a - 1), bSize.push_back(blockSize), bScore.push_back(0), bPrev.push_back(0);
    }
    bStart.push_back((int)m.size() - 1), bEnd.push_back((int)m.size() - 1), bSize.push_back((int)m[m.size() - 1].l);
    bScore.push_back(0), bPrev.push_back(0);
    const float GAP = -0.1;
    for (size_t mi = 0; mi < bStart.size(); mi++) {
      for (size_t mj = 0; mj < mi; mj++) {
        const int p = bEnd[mj], c = bStart[mi];
        const int tGap = (int)(m[c].t - m[p].t), qGap = (int)(m[c].q - m[p].q);
        const int gap = abs(tGap - qGap);
        const int score = bScore[mj] + gap * GAP + bSize[mj];  // int + float + int, truncated (:1497)
        if (bScore[mi] < score) {
          bScore[mi] = score;
          bPrev[mi] = (int)mj;
        }
      }
    }
    int maxScore = 0, maxIndex = 0;
    for (size_t mi = 0; mi < bScore.size(); mi++)
      if (bScore[mi] > maxScore) {
        maxScore = bScore[mi];
        maxIndex = (int)mi;
      }
    int mi = maxIndex;
    while (mi > 0) {
      bScore[mi] = -1;
      mi = bPrev[mi];
    }
    for (size_t b = 1; b < bScore.size(); b++)
      if (bScore[b] != -1)
        for (int mj = bStart[b]; mj <= bEnd[b]; mj++) toRemove[mj] = 1;
    size_t k = 0;
    for (size_t n = 0; n < m.size(); n++)
      if (!toRemove[n]) m[k++] = m[n];
    m.resize(k);
  }
}

struct GapTask {
  int32_t cand;      // candidate (index into the batch)
  int32_t after;     // the gap follows this anchor of the candidate
  uint32_t qPos, tPos;
  int32_t qGap, tGap;
  int32_t kind;      // 0 AffineKBandAlign, 1 SDPAlign + RefineAlignment
  int32_t wordSize;  // adaptSDPTupleSize (:1611-1614)
  int64_t firstBlock;  // the gap's blocks (relative to the window), in `gapBlocks`
  int32_t nBlocks;
};

template <typename T>
struct Dev {
  T *p = nullptr;
  size_t n = 0;
  ~Dev() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t count) {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = count;
    return hipMalloc((void **)&p, sizeof(T) * std::max<size_t>(count, 1));
  }
};

// ComputeAlignmentStats (AlignmentUtils.h:535-584) of an alignment WITHOUT gap lists (CreateAlignmentStrings :408-446,
// :494-529; the branch clears the gaps, :1706): the gap between two blocks is the query surplus, then the target
// surplus, then the common part as aligned columns.  The first block is at (0, 0) here.  One wavefront per candidate.
__global__ __launch_bounds__(256) void rbao_stats_kernel(int n, const uint8_t *seq, const int64_t *qOff, const int64_t *tOff,
                                                         const int32_t *blocks, const int32_t *matrix, int ins, int del,
                                                         int open, int ext, blasr_hip_sdp_result *res) {
  __shared__ int m[25];
  if (threadIdx.x < 25) m[threadIdx.x] = matrix[threadIdx.x];
  __syncthreads();
  const int k = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (k >= n) return;
  const int lane = lane_id();
  blasr_hip_sdp_result r = res[k];
  if (r.n_blocks == 0) return;
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

// fn(begin, end) over [0, n) on up to 16 host threads (the per-gap loops below run over ~10^5..10^6 gaps per call)
template <typename F>
void parallel_ranges(size_t n, F fn) {
  unsigned threads = (unsigned)blasr_hip_host_threads((int)std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
  if (n < 16384) threads = 1;
  const size_t chunk = (n + threads - 1) / threads;
  const int rc = blasr_hip_run_threads((int)threads, [&](int t) {
    const size_t b = std::min(n, (size_t)t * chunk), e = std::min(n, b + chunk);
    if (b < e) fn(b, e);
  });
  if (rc == BLASR_HIP_ERR_NOMEM) throw std::bad_alloc();  // back on the caller's thread: the entry point's handler reports it
  if (rc) throw std::runtime_error("a worker thread failed");
}

// ops (M / I / D runs from (0, 0)) -> blocks, Alignment::ArrowPathToAlignment (Alignment.h:190-254)
void ops_to_blocks(const uint32_t *ops, int nOps, uint32_t q0, uint32_t t0, int32_t *out) {
  uint32_t q = q0, t = t0;
  for (int i = 0; i < nOps; i++) {
    const uint32_t len = ops[i] >> 2, ty = ops[i] & 3u;
    if (ty == BLASR_HIP_OP_M) {
      *out++ = (int32_t)q, *out++ = (int32_t)t, *out++ = (int32_t)len;
      q += len, t += len;
    } else if (ty == BLASR_HIP_OP_I) {
      q += len;
    } else {
      t += len;
    }
  }
}

// The blocks of a batch of op lists appended to gapBlocks, task j's place (firstBlock, nBlocks) noted in its gap:
// counted, placed by a prefix sum, written by several threads.  addPos: the result's own qPos / tPos shift the blocks
// (RefineAlignment's output, :888-914); returns the index of a task whose status is not 0, or -1.
int64_t append_blocks_of(const std::vector<blasr_hip_refine_result> &res, const uint32_t *ops, const size_t *taskOf,
                         std::vector<GapTask> &gaps, bool addPos, std::vector<int32_t> &gapBlocks) {
  const size_t m = res.size();
  std::vector<int64_t> at(m + 1, 0);
  std::atomic<int64_t> bad(-1);
  parallel_ranges(m, [&](size_t b, size_t e) {
    for (size_t j = b; j < e; j++) {
      const blasr_hip_refine_result &r = res[j];
      if (r.status != 0) {
        bad.store((int64_t)j);  // (any one of them will do for the message)
        continue;
      }
      int64_t c = 0;
      for (int i = 0; i < r.n_ops; i++) c += (ops[r.ops_off + i] & 3u) == BLASR_HIP_OP_M;
      at[j + 1] = c;
    }
  });
  if (bad.load() >= 0) return bad.load();
  for (size_t j = 0; j < m; j++) at[j + 1] += at[j];
  const size_t base = gapBlocks.size() / 3;
  gapBlocks.resize(3 * (base + (size_t)at[m]));
  parallel_ranges(m, [&](size_t b, size_t e) {
    for (size_t j = b; j < e; j++) {
      const blasr_hip_refine_result &r = res[j];
      GapTask &g = gaps[taskOf[j]];
      g.firstBlock = (int64_t)base + at[j];
      g.nBlocks = (int32_t)(at[j + 1] - at[j]);
      ops_to_blocks(ops + r.ops_off, r.n_ops, g.qPos + (addPos ? (uint32_t)r.q_pos : 0u), g.tPos + (addPos ? (uint32_t)r.t_pos : 0u),
                    gapBlocks.data() + 3 * g.firstBlock);
    }
  });
  return -1;
}

}  // namespace

extern "C" {

int blasr_hip_align_between_anchors_device(blasr_hip_ctx *ctx, const blasr_hip_sdp_batch *b,
                                           const blasr_hip_window *windows, uint32_t genome_length,
                                           const uint32_t *chains, const int64_t *chain_off,
                                           const blasr_hip_rbao_params *prm, blasr_hip_sdp_result *results,
                                           int32_t *blocks, int64_t blocks_capacity, int64_t *blocks_used) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (!b || b->n < 0 || !prm || blocks_capacity < 0 || (b->n > 0 && (!b->seq || !b->q_off || !b->q_len || !b->t_off ||
      !b->t_len || !windows || !chain_off || !results)) || (blocks_capacity > 0 && !blocks))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "align_between_anchors: bad argument");
  if (blocks_used) *blocks_used = 0;
  const int n = b->n;
  if (n == 0) return BLASR_HIP_OK;
  if (prm->band < 1 || 2 * prm->band + 1 > 1088)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_UNSUPPORTED, "align_between_anchors: bandSize %d outside [1, 543]", prm->band);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const bool trace = getenv("BLASR_HIP_RBAO_TRACE") != nullptr;
  auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tPhase = now();
  auto lap = [&](const char *what, size_t count) {
    if (trace) fprintf(stderr, "[rbao] %-28s %9.1f ms  (%zu)\n", what, now() - tPhase, count);
    tPhase = now();
  };

  // ---- the candidates' geometry, on the host
  std::vector<blasr_hip_window> win((size_t)n);
  std::vector<int64_t> qOff((size_t)n), tOff((size_t)n);
  std::vector<int32_t> qLen((size_t)n), tLen((size_t)n);
  HIP_TRY(ctx, hipMemcpyAsync(win.data(), windows, sizeof(blasr_hip_window) * (size_t)n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(qOff.data(), b->q_off, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(tOff.data(), b->t_off, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(qLen.data(), b->q_len, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(tLen.data(), b->t_len, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  for (int i = 0; i < n; i++)
    if (chain_off[i + 1] < chain_off[i] || (chain_off[i + 1] > chain_off[i] && !chains))
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "align_between_anchors: bad chain offsets at candidate %d", i);

  // ---- :1339-1349, :1385-1564 per candidate: the anchors the gap loop walks
  std::vector<std::vector<Anchor>> anchors((size_t)n);
  std::vector<int> bypass((size_t)n, 0);
  {
    unsigned threads = (unsigned)blasr_hip_host_threads((int)std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
    threads = (unsigned)std::max(1, std::min<int>((int)threads, n / 64 + 1));
    auto work = [&](unsigned tid) {
      for (int i = (int)tid; i < n; i += (int)threads) {
        std::vector<Anchor> &m = anchors[(size_t)i];
        const int64_t c0 = chain_off[i], c1 = chain_off[i + 1];
        m.resize((size_t)(c1 - c0));
        int intervalSize = 0;
        for (int64_t c = c0; c < c1; c++) {
          m[(size_t)(c - c0)] = Anchor{chains[3 * c], chains[3 * c + 1], chains[3 * c + 2]};
          intervalSize += (int)chains[3 * c + 2];
        }
        const int subreadLength = (int)win[(size_t)i].q_aligned_seq_length;  // subreadEnd - subreadStart (:1348)
        if (!((1.0 * intervalSize) / subreadLength < prm->sdp_bypass_threshold)) {
          bypass[(size_t)i] = 1;
          continue;
        }
        prepare_anchors(m, win[(size_t)i], genome_length);
      }
    };
    if (const int trc = blasr_hip_run_threads((int)threads, work))
      return blasr_hip_fail(ctx, trc, "align_between_anchors: a worker thread failed (out of host memory)");
  }
  for (int i = 0; i < n; i++)
    if (bypass[(size_t)i])
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_UNSUPPORTED, "align_between_anchors: candidate %d takes the -sdpbypass "
                            "branch (its anchors cover sdp_bypass_threshold of the subread), which is not offered", i);

  lap("anchors per candidate", (size_t)n);
  // ---- :1578-1634 the gaps that get an AlignSubstring
  std::vector<GapTask> gaps;
  for (int i = 0; i < n; i++) {
    const std::vector<Anchor> &m = anchors[(size_t)i];
    for (size_t a = 0; m.size() > 0 && a < m.size() - 1; a++) {
      if (m[a].t + m[a].l > (uint32_t)tLen[(size_t)i] || m[a].q + m[a].l > (uint32_t)qLen[(size_t)i])
        return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "align_between_anchors: anchor %zu of candidate %d lies outside its window", a, i);
      const int tGap = (int)(m[a + 1].t - (m[a].t + m[a].l)), qGap = (int)(m[a + 1].q - (m[a].q + m[a].l));
      if (tGap > 0 && qGap > 0) {
        const float tRatio = ((float)tGap) / qGap;
        const float qRatio = 1 / tRatio;
        const int maxGap = 100000;
        int adaptSDPTupleSize = prm->sdp.word_size;
        if (win[(size_t)i].t_aligned_seq_length < 1000 && win[(size_t)i].q_aligned_seq_length < 1000) adaptSDPTupleSize = 7;
        if (tRatio > 0.001 && qRatio > 0.001 && tGap < maxGap && qGap < maxGap) {
          GapTask g;
          g.cand = i;
          g.after = (int32_t)a;
          g.qPos = m[a].q + m[a].l;
          g.tPos = m[a].t + m[a].l;
          g.qGap = qGap;
          g.tGap = tGap;
          // AlignSubstring (:1062-1065)
          const int tLength = tGap, qLength = qGap;
          g.kind = ((tLength < prm->band || qLength < prm->band) &&
                    ((tLength >= qLength && tLength < 1.5 * qLength) || (tLength <= qLength && tLength > 0.5 * qLength))) ? 0 : 1;
          g.wordSize = adaptSDPTupleSize;
          g.firstBlock = 0;
          g.nBlocks = 0;
          if ((uint64_t)g.tPos + (uint64_t)tGap > (uint64_t)tLen[(size_t)i] || (uint64_t)g.qPos + (uint64_t)qGap > (uint64_t)qLen[(size_t)i])
            return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "align_between_anchors: a gap of candidate %d lies outside its window", i);
          gaps.push_back(g);
        }
      }
    }
    if (!m.empty()) {
      const Anchor &z = m[m.size() - 1];
      if (z.t > (uint32_t)tLen[(size_t)i] || z.q > (uint32_t)qLen[(size_t)i] || z.t + z.l > (uint32_t)tLen[(size_t)i] ||
          z.q + z.l > (uint32_t)qLen[(size_t)i])
        return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "align_between_anchors: the last anchor of candidate %d lies outside its window (the "
                              "reference asserts, :1649-1650)", i);
    }
  }

  lap("gap tasks", gaps.size());
  std::vector<int32_t> gapBlocks;  // (q, t, length) of every gap's blocks, relative to the candidate's window
  int rc;
  const size_t kChunk = 1u << 21;  // gap tasks per device batch

  // ---- AffineKBandAlign gaps
  {
    std::vector<size_t> idx;
    for (size_t g = 0; g < gaps.size(); g++)
      if (gaps[g].kind == 0) idx.push_back(g);
    blasr_hip_kband_params kp;
    memcpy(kp.matrix, prm->score.matrix, sizeof(kp.matrix));  // distScoreFn.scoreMatrix (:1067)
    kp.hp_ins_open = prm->indel + 2;
    kp.hp_ins_extend = prm->indel - 3;
    kp.ins_open = prm->indel + 2;
    kp.ins_extend = prm->indel - 1;
    kp.del = prm->indel;
    for (size_t c0 = 0; c0 < idx.size(); c0 += kChunk) {
      const size_t m = std::min(kChunk, idx.size() - c0);
      std::vector<int64_t> hq(m), ht(m);
      std::vector<int32_t> hql(m), htl(m), hk(m, prm->band);
      int64_t opsCap = 0;
      for (size_t j = 0; j < m; j++) {
        const GapTask &g = gaps[idx[c0 + j]];
        hq[j] = qOff[(size_t)g.cand] + g.qPos;
        ht[j] = tOff[(size_t)g.cand] + g.tPos;
        hql[j] = g.qGap;
        htl[j] = g.tGap;
        opsCap += (int64_t)g.qGap + g.tGap + 2 * prm->band + 8;
      }
      Dev<int64_t> dq, dt;
      Dev<int32_t> dql, dtl, dk;
      Dev<blasr_hip_refine_result> dres;
      Dev<uint32_t> dops;
      HIP_TRY(ctx, dq.alloc(m));
      HIP_TRY(ctx, dt.alloc(m));
      HIP_TRY(ctx, dql.alloc(m));
      HIP_TRY(ctx, dtl.alloc(m));
      HIP_TRY(ctx, dk.alloc(m));
      HIP_TRY(ctx, dres.alloc(m));
      HIP_TRY(ctx, dops.alloc((size_t)opsCap));
      HIP_TRY(ctx, hipMemcpyAsync(dq.p, hq.data(), sizeof(int64_t) * m, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipMemcpyAsync(dt.p, ht.data(), sizeof(int64_t) * m, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipMemcpyAsync(dql.p, hql.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipMemcpyAsync(dtl.p, htl.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipMemcpyAsync(dk.p, hk.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipStreamSynchronize(st));
      blasr_hip_kband_batch kb;
      kb.n = (int32_t)m;
      kb.seq = b->seq;
      kb.q_off = dq.p;
      kb.q_len = dql.p;
      kb.t_off = dt.p;
      kb.t_len = dtl.p;
      kb.k = dk.p;
      int64_t used = 0;
      if ((rc = blasr_hip_kband_device(ctx, &kb, &kp, dres.p, dops.p, opsCap, &used))) return rc;
      std::vector<blasr_hip_refine_result> hres(m);
      std::vector<uint32_t> hops((size_t)std::max<int64_t>(used, 1));
      HIP_TRY(ctx, hipMemcpy(hres.data(), dres.p, sizeof(blasr_hip_refine_result) * m, hipMemcpyDeviceToHost));
      if (used > 0) HIP_TRY(ctx, hipMemcpy(hops.data(), dops.p, sizeof(uint32_t) * (size_t)used, hipMemcpyDeviceToHost));
      // (a fresh T_AlignmentCandidate: qPos = tPos = 0, :1126-1136 add the gap's place)
      const int64_t badK = append_blocks_of(hres, hops.data(), idx.data() + c0, gaps, false, gapBlocks);
      if (badK >= 0) {
        const GapTask &g = gaps[idx[c0 + (size_t)badK]];
        return blasr_hip_fail(ctx, BLASR_HIP_ERR_HIP, "align_between_anchors: AffineKBandAlign of a %d x %d gap of candidate %d: "
                              "status %d", g.qGap, g.tGap, g.cand, hres[(size_t)badK].status);
      }
    }
  }

  lap("AffineKBandAlign gaps", gapBlocks.size() / 3);
  // ---- SDPAlign gaps, by word size: SDPAlign (Global), the pinned end blocks, RefineAlignment on the gap
  for (int pass = 0; pass < 2; pass++) {
    const int wordSize = pass == 0 ? prm->sdp.word_size : 7;
    if (pass == 1 && wordSize == prm->sdp.word_size) break;
    std::vector<size_t> idx;
    for (size_t g = 0; g < gaps.size(); g++)
      if (gaps[g].kind == 1 && gaps[g].wordSize == wordSize) idx.push_back(g);
    blasr_hip_sdp_params sp = prm->sdp;
    sp.word_size = wordSize;
    sp.indel_rate = 0.25f;
    sp.align_type = 1;     // Global
    sp.extend_front = 1;
    sp.sw_align_type = 1;  // Global
    for (size_t c0 = 0; c0 < idx.size(); c0 += kChunk) {
      const size_t m = std::min(kChunk, idx.size() - c0);
      std::vector<int64_t> hq(m), ht(m);
      std::vector<int32_t> hql(m), htl(m);
      int64_t blkCap = 0;
      for (size_t j = 0; j < m; j++) {
        const GapTask &g = gaps[idx[c0 + j]];
        hq[j] = qOff[(size_t)g.cand] + g.qPos;
        ht[j] = tOff[(size_t)g.cand] + g.tPos;
        hql[j] = g.qGap;
        htl[j] = g.tGap;
        blkCap += std::min(g.qGap, g.tGap) / 2 + 8;
      }
      Dev<int64_t> dq, dt;
      Dev<int32_t> dql, dtl;
      Dev<blasr_hip_sdp_result> dres;
      Dev<int32_t> dblk;
      HIP_TRY(ctx, dq.alloc(m));
      HIP_TRY(ctx, dt.alloc(m));
      HIP_TRY(ctx, dql.alloc(m));
      HIP_TRY(ctx, dtl.alloc(m));
      HIP_TRY(ctx, dres.alloc(m));
      HIP_TRY(ctx, hipMemcpyAsync(dq.p, hq.data(), sizeof(int64_t) * m, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipMemcpyAsync(dt.p, ht.data(), sizeof(int64_t) * m, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipMemcpyAsync(dql.p, hql.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipMemcpyAsync(dtl.p, htl.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipStreamSynchronize(st));
      blasr_hip_sdp_batch sb;
      sb.n = (int32_t)m;
      sb.seq = b->seq;
      sb.q_off = dq.p;
      sb.q_len = dql.p;
      sb.t_off = dt.p;
      sb.t_len = dtl.p;
      int64_t blkUsed = 0;
      for (int attempt = 0; attempt < 2; attempt++) {
        HIP_TRY(ctx, dblk.alloc(3 * (size_t)blkCap));
        rc = blasr_hip_sdp_device(ctx, &sb, &sp, &prm->score, dres.p, dblk.p, blkCap, &blkUsed);
        if (rc != BLASR_HIP_ERR_CAPACITY) break;
        blkCap = blkUsed + 64;
      }
      if (rc) return rc;
      lap("  SDPAlign batch", m);
      std::vector<blasr_hip_sdp_result> hres(m);
      std::vector<int32_t> hblk(3 * (size_t)std::max<int64_t>(blkUsed, 1));
      HIP_TRY(ctx, hipMemcpy(hres.data(), dres.p, sizeof(blasr_hip_sdp_result) * m, hipMemcpyDeviceToHost));
      if (blkUsed > 0) HIP_TRY(ctx, hipMemcpy(hblk.data(), dblk.p, sizeof(int32_t) * 3 * (size_t)blkUsed, hipMemcpyDeviceToHost));
      // :1094-1117 the alignment pinned to both ends of the gap, then RefineAlignment's guided branch (:831-915):
      // the sequences up to the end of the last block, the blocks as the guide
      // (counted per gap, placed by prefix sums, written by several threads: ~10^5..10^6 gaps per call)
      std::vector<int64_t> blkAt(m + 1, 0), taskAt(m + 1, 0);
      int64_t badS = -1;
      for (size_t j = 0; j < m; j++) {
        const GapTask &g = gaps[idx[c0 + j]];
        const blasr_hip_sdp_result &r = hres[j];
        if (r.status != 0) {
          badS = (int64_t)j;
          break;
        }
        int64_t nb = 0;
        if (r.n_blocks > 0) {  // (else nothing seeded: the gap stays open, :1092, :842-845, :1128)
          const int32_t *src = hblk.data() + 3 * r.blk_off, *last = src + 3 * (r.n_blocks - 1);
          // (a Global SDPAlign leaves qPos = tPos = 0 and absolute block positions)
          nb = r.n_blocks + ((src[1] > 0 && src[0] > 0) ? 1 : 0) + ((last[1] + last[2] < g.tGap && last[0] + last[2] < g.qGap) ? 1 : 0);
        }
        blkAt[j + 1] = blkAt[j] + nb;
        taskAt[j + 1] = taskAt[j] + (nb > 0 ? 1 : 0);
      }
      if (badS >= 0)
        return blasr_hip_fail(ctx, BLASR_HIP_ERR_HIP, "align_between_anchors: SDPAlign of a gap of candidate %d: status %d",
                              gaps[idx[c0 + (size_t)badS]].cand, hres[(size_t)badS].status);
      const size_t nr = (size_t)taskAt[m];
      std::vector<int32_t> guide(3 * (size_t)blkAt[m]);
      std::vector<int64_t> rq(nr), rt(nr), rblkOff(nr);
      std::vector<int32_t> rql(nr), rtl(nr), rnb(nr);
      std::vector<size_t> rTask(nr);
      parallel_ranges(m, [&](size_t jb, size_t je) {
        for (size_t j = jb; j < je; j++) {
          const int64_t nb = blkAt[j + 1] - blkAt[j];
          if (nb == 0) continue;
          const GapTask &g = gaps[idx[c0 + j]];
          const blasr_hip_sdp_result &r = hres[j];
          const int32_t *src = hblk.data() + 3 * r.blk_off, *last = src + 3 * (r.n_blocks - 1);
          int32_t *dst = guide.data() + 3 * blkAt[j], *w = dst;
          if (src[1] > 0 && src[0] > 0) *w++ = 0, *w++ = 0, *w++ = 1;  // :1107-1116 pinned to the beginning
          memcpy(w, src, sizeof(int32_t) * 3 * (size_t)r.n_blocks);
          w += 3 * r.n_blocks;
          if (last[1] + last[2] < g.tGap && last[0] + last[2] < g.qGap) *w++ = g.qGap - 1, *w++ = g.tGap - 1, *w++ = 1;  // :1095-1104
          const int32_t *z = dst + 3 * (nb - 1);
          const size_t k = (size_t)taskAt[j];
          rq[k] = hq[j];
          rt[k] = ht[j];
          rql[k] = z[0] + z[2];
          rtl[k] = z[1] + z[2];
          rblkOff[k] = blkAt[j];
          rnb[k] = (int32_t)nb;
          rTask[k] = idx[c0 + j];
        }
      });
      int64_t opsCap = 0;
      for (size_t k = 0; k < nr; k++) opsCap += (int64_t)rql[k] + rtl[k] + 16;
      lap("  pins and guides", nr);
      if (nr == 0) continue;
      Dev<int64_t> gq, gt, gbo;
      Dev<int32_t> gql, gtl, gnb, gblk;
      Dev<blasr_hip_refine_result> gres;
      Dev<uint32_t> gops;
      HIP_TRY(ctx, gq.alloc(nr));
      HIP_TRY(ctx, gt.alloc(nr));
      HIP_TRY(ctx, gbo.alloc(nr));
      HIP_TRY(ctx, gql.alloc(nr));
      HIP_TRY(ctx, gtl.alloc(nr));
      HIP_TRY(ctx, gnb.alloc(nr));
      HIP_TRY(ctx, gblk.alloc(guide.size()));
      HIP_TRY(ctx, gres.alloc(nr));
      HIP_TRY(ctx, hipMemcpyAsync(gq.p, rq.data(), sizeof(int64_t) * nr, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipMemcpyAsync(gt.p, rt.data(), sizeof(int64_t) * nr, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipMemcpyAsync(gbo.p, rblkOff.data(), sizeof(int64_t) * nr, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipMemcpyAsync(gql.p, rql.data(), sizeof(int32_t) * nr, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipMemcpyAsync(gtl.p, rtl.data(), sizeof(int32_t) * nr, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipMemcpyAsync(gnb.p, rnb.data(), sizeof(int32_t) * nr, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipMemcpyAsync(gblk.p, guide.data(), sizeof(int32_t) * guide.size(), hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipStreamSynchronize(st));
      blasr_hip_refine_batch rb;
      rb.n = (int32_t)nr;
      rb.seq = b->seq;
      rb.q_off = gq.p;
      rb.q_len = gql.p;
      rb.t_off = gt.p;
      rb.t_len = gtl.p;
      rb.blocks = gblk.p;
      rb.blk_off = gbo.p;
      rb.n_blocks = gnb.p;
      int64_t opsUsed = 0;
      for (int attempt = 0; attempt < 2; attempt++) {
        HIP_TRY(ctx, gops.alloc((size_t)opsCap));
        rc = blasr_hip_refine_device(ctx, &rb, prm->band, &prm->score, gres.p, gops.p, opsCap, &opsUsed);
        if (rc != BLASR_HIP_ERR_CAPACITY) break;
        opsCap = opsUsed + 64;
      }
      if (rc) return rc;
      lap("  RefineAlignment batch", nr);
      std::vector<blasr_hip_refine_result> hr(nr);
      std::vector<uint32_t> hops((size_t)std::max<int64_t>(opsUsed, 1));
      HIP_TRY(ctx, hipMemcpy(hr.data(), gres.p, sizeof(blasr_hip_refine_result) * nr, hipMemcpyDeviceToHost));
      if (opsUsed > 0) HIP_TRY(ctx, hipMemcpy(hops.data(), gops.p, sizeof(uint32_t) * (size_t)opsUsed, hipMemcpyDeviceToHost));
      // :888-914 then :1134-1135: the refined blocks start at (0, 0) of refinedAlignment.qPos / tPos
      const int64_t badR = append_blocks_of(hr, hops.data(), rTask.data(), gaps, true, gapBlocks);
      if (badR >= 0)
        return blasr_hip_fail(ctx, BLASR_HIP_ERR_HIP, "align_between_anchors: RefineAlignment of a gap of candidate %d: status %d "
                              "(the reference would exit: AffineGuidedAlign.h:394-408)", gaps[rTask[(size_t)badR]].cand, hr[(size_t)badR].status);
    }
  }

  lap("SDPAlign + refine gaps", gapBlocks.size() / 3);
  // ---- :1578-1705 per candidate: anchors and gap alignments in order, qPos / tPos taken out, abutting blocks merged
  std::vector<blasr_hip_sdp_result> res((size_t)n);
  std::vector<size_t> gapFirst((size_t)n + 1, gaps.size());  // the candidates' first gaps (the gaps are in candidate order)
  for (size_t g = gaps.size(); g-- > 0;) gapFirst[(size_t)gaps[g].cand] = g;
  for (int i = n - 1; i >= 0; i--)
    if (gapFirst[(size_t)i] == gaps.size()) gapFirst[(size_t)i] = gapFirst[(size_t)i + 1];
  std::vector<std::vector<int32_t>> perCand((size_t)n);
  std::atomic<int> badCand(-1);
  parallel_ranges((size_t)n, [&](size_t ib, size_t ie) {
    std::vector<int32_t> cur;
    for (size_t ii = ib; ii < ie; ii++) {
      const int i = (int)ii;
      const std::vector<Anchor> &m = anchors[ii];
      size_t gi = gapFirst[ii];
      cur.clear();
      for (size_t a = 0; m.size() > 0 && a < m.size() - 1; a++) {
        cur.push_back((int32_t)m[a].q), cur.push_back((int32_t)m[a].t), cur.push_back((int32_t)m[a].l);
        if (gi < gaps.size() && gaps[gi].cand == i && gaps[gi].after == (int32_t)a) {
          const GapTask &g = gaps[gi++];
          cur.insert(cur.end(), gapBlocks.begin() + 3 * g.firstBlock, gapBlocks.begin() + 3 * (g.firstBlock + g.nBlocks));
        }
      }
      if (!m.empty()) {
        const Anchor &z = m[m.size() - 1];
        cur.push_back((int32_t)z.q), cur.push_back((int32_t)z.t), cur.push_back((int32_t)z.l);
      }
      blasr_hip_sdp_result &r = res[ii];
      memset(&r, 0, sizeof(r));
      const size_t nb = cur.size() / 3;
      std::vector<int32_t> &mine = perCand[ii];
      if (nb > 0) {
        r.q_pos = cur[0];
        r.t_pos = cur[1];
        bool ok = true;
        for (size_t k = 0; k < nb; k++) {
          if (cur[3 * k] < r.q_pos || cur[3 * k + 1] < r.t_pos) ok = false;  // (the reference asserts, :1672-1673)
          cur[3 * k] -= r.q_pos;
          cur[3 * k + 1] -= r.t_pos;
        }
        if (!ok) {
          badCand.store(i);
          continue;
        }
        size_t c = 0, next = 1;  // :1686-1697
        while (next < nb) {
          while (next < nb && cur[3 * c + 1] + cur[3 * c + 2] == cur[3 * next + 1] && cur[3 * c] + cur[3 * c + 2] == cur[3 * next]) {
            cur[3 * c + 2] += cur[3 * next + 2];
            cur[3 * next + 2] = 0;
            next += 1;
          }
          c = next;
          next += 1;
        }
        mine.reserve(cur.size());
        for (size_t k = 0; k < nb; k++)  // :1698-1710
          if (cur[3 * k + 2] != 0) mine.push_back(cur[3 * k]), mine.push_back(cur[3 * k + 1]), mine.push_back(cur[3 * k + 2]);
      }
      r.n_blocks = (int32_t)(mine.size() / 3);
    }
  });
  if (badCand.load() >= 0)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_HIP, "align_between_anchors: a block of candidate %d starts before its first block "
                          "(the reference asserts, :1672-1673)", badCand.load());
  std::vector<int32_t> out;
  {
    int64_t at = 0;
    for (int i = 0; i < n; i++) {
      res[(size_t)i].blk_off = at;
      at += res[(size_t)i].n_blocks;
    }
    out.resize(3 * (size_t)at);
    parallel_ranges((size_t)n, [&](size_t ib, size_t ie) {
      for (size_t ii = ib; ii < ie; ii++)
        if (!perCand[ii].empty()) memcpy(out.data() + 3 * res[ii].blk_off, perCand[ii].data(), sizeof(int32_t) * perCand[ii].size());
    });
  }
  const int64_t total = (int64_t)(out.size() / 3);
  if (blocks_used) *blocks_used = total;
  if (total > blocks_capacity)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_CAPACITY, "align_between_anchors: %lld blocks, room for %lld", (long long)total,
                          (long long)blocks_capacity);
  HIP_TRY(ctx, hipMemcpyAsync(results, res.data(), sizeof(blasr_hip_sdp_result) * (size_t)n, hipMemcpyHostToDevice, st));
  if (total > 0) HIP_TRY(ctx, hipMemcpyAsync(blocks, out.data(), sizeof(int32_t) * out.size(), hipMemcpyHostToDevice, st));
  Dev<int32_t> dMatrix;
  HIP_TRY(ctx, dMatrix.alloc(25));
  HIP_TRY(ctx, hipMemcpyAsync(dMatrix.p, prm->score.matrix, sizeof(int32_t) * 25, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(rbao_stats_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, n, b->seq, b->q_off, b->t_off, blocks,
                     dMatrix.p, prm->score.ins, prm->score.del, prm->score.affine_open, prm->score.affine_extend, results);
  HIP_TRY(ctx, hipStreamSynchronize(st));
  HIP_TRY(ctx, hipGetLastError());
  lap("assembly + statistics", (size_t)total);
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CTX(ctx)

int blasr_hip_align_between_anchors_host(blasr_hip_ctx *ctx, const blasr_hip_sdp_batch *b, int64_t seq_bytes,
                                         const blasr_hip_window *windows, uint32_t genome_length,
                                         const uint32_t *chains, const int64_t *chain_off,
                                         const blasr_hip_rbao_params *prm, blasr_hip_sdp_result *results,
                                         int32_t *blocks, int64_t blocks_capacity, int64_t *blocks_used) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (!b || b->n < 0 || seq_bytes < 0 || blocks_capacity < 0 || (b->n > 0 && (!b->seq || !b->q_off || !b->q_len || !b->t_off ||
      !b->t_len || !windows || !results)) || (blocks_capacity > 0 && !blocks))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "align_between_anchors: bad argument");
  if (blocks_used) *blocks_used = 0;
  const size_t n = (size_t)b->n;
  if (n == 0) return BLASR_HIP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  Dev<uint8_t> dSeq;
  Dev<int64_t> dq, dt;
  Dev<int32_t> dql, dtl, dblk;
  Dev<blasr_hip_window> dWin;
  Dev<blasr_hip_sdp_result> dRes;
  HIP_TRY(ctx, dSeq.alloc((size_t)seq_bytes + 64));
  HIP_TRY(ctx, dq.alloc(n));
  HIP_TRY(ctx, dt.alloc(n));
  HIP_TRY(ctx, dql.alloc(n));
  HIP_TRY(ctx, dtl.alloc(n));
  HIP_TRY(ctx, dWin.alloc(n));
  HIP_TRY(ctx, dRes.alloc(n));
  HIP_TRY(ctx, dblk.alloc(3 * (size_t)blocks_capacity));
  HIP_TRY(ctx, hipMemcpy(dSeq.p, b->seq, (size_t)seq_bytes, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(dq.p, b->q_off, sizeof(int64_t) * n, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(dt.p, b->t_off, sizeof(int64_t) * n, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(dql.p, b->q_len, sizeof(int32_t) * n, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(dtl.p, b->t_len, sizeof(int32_t) * n, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(dWin.p, windows, sizeof(blasr_hip_window) * n, hipMemcpyHostToDevice));
  blasr_hip_sdp_batch db = *b;
  db.seq = dSeq.p;
  db.q_off = dq.p;
  db.t_off = dt.p;
  db.q_len = dql.p;
  db.t_len = dtl.p;
  int64_t used = 0;
  const int rc = blasr_hip_align_between_anchors_device(ctx, &db, dWin.p, genome_length, chains, chain_off, prm, dRes.p, dblk.p,
                                                        blocks_capacity, &used);
  if (blocks_used) *blocks_used = used;
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpy(results, dRes.p, sizeof(blasr_hip_sdp_result) * n, hipMemcpyDeviceToHost));
  if (used > 0) HIP_TRY(ctx, hipMemcpy(blocks, dblk.p, sizeof(int32_t) * 3 * (size_t)used, hipMemcpyDeviceToHost));
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CTX(ctx)

}  // extern "C"
