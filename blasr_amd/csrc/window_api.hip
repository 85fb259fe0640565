// window_api.hip -- a-1: the candidate windows of AlignIntervals (alignment/Blasr.cpp:1208-1336) as a
// gather from the device-resident genome, and a-10: the host-side candidate ordering / pruning.
//
//   window_plan_kernel    one thread per interval: the window arithmetic of :1243-1310 (unsigned and
//                         single-precision float exactly as the reference computes it), contig clamp
//   (device exclusive scan of the byte counts)
//   window_gather_kernel  one wavefront per interval: copies the forward read's subread and the genome
//                         window, reverse-complemented for Reverse-strand intervals (DNASequence.h:219-233)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "../../include/blasr_hip.h"
#include "ctx.h"
#include "wave_utils.h"

namespace blasr_hip {

struct WindowPlan {  // per interval, between the two kernels
  uint32_t fwdStart;  // first base of the window on the forward genome
  uint32_t tLen, qLen, qStart;
};

__global__ __launch_bounds__(256) void window_plan_kernel(int n, const blasr_hip_interval *intervals,
                                                          const blasr_hip_read *reads, uint32_t genomeLen,
                                                          const uint32_t *contigStarts, int nContigs,
                                                          blasr_hip_window_params prm, blasr_hip_window *windows,
                                                          WindowPlan *plan, int64_t *bytes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const blasr_hip_interval iv = intervals[i];
  const blasr_hip_read rd = reads[iv.read];
  uint32_t matchIntervalStart = iv.start, matchIntervalEnd = iv.end;  // :1208-1210
  // :1243-1248 the subread on the strand of the interval (MakeRCCoordinate, DNASequence.h:215-217)
  uint32_t sStart = rd.subread_start, sEnd = rd.subread_end;
  if (iv.strand == 1) {
    sEnd = (rd.length - rd.subread_start - 1u) + 1u;
    sStart = rd.length - (rd.subread_end - 1u) - 1u;
  }
  if (prm.extend_ends) {  // :1250-1266; unsigned * float in single precision, truncated
    const uint32_t before = (uint32_t)__fmul_rn((float)(iv.q_start - sStart), prm.approximate_max_insertion_rate);
    const uint32_t after = (uint32_t)__fmul_rn((float)(sEnd - iv.q_end), prm.approximate_max_insertion_rate);
    if (matchIntervalStart < before || prm.do_global_alignment) matchIntervalStart = 0;
    else matchIntervalStart -= before;
    if (genomeLen < matchIntervalEnd + after || prm.do_global_alignment) matchIntervalEnd = genomeLen;
    else matchIntervalEnd += after;
  }
  int tIndex = 0;
  uint32_t tLength = genomeLen;  // :1300-1304
  if (contigStarts != nullptr) {  // :1272-1298
    int idx = 0;
    if (nContigs + 1 != 1) {  // SequenceIndexDatabase.h:108-118
      int lo = 1, hi = nContigs + 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (!(iv.start < contigStarts[mid])) lo = mid + 1; else hi = mid;
      }
      idx = lo - 1;
    }
    const uint32_t contigStartPos = contigStarts[idx];
    if (contigStartPos > matchIntervalStart) matchIntervalStart = contigStartPos;
    const uint32_t contigEndPos = contigStarts[idx + 1] - 1u;
    if (contigEndPos < matchIntervalEnd) matchIntervalEnd = contigEndPos;
    tLength = contigEndPos - contigStartPos;
    tIndex = idx;
  }
  blasr_hip_window w;
  w.t_aligned_seq_length = matchIntervalEnd - matchIntervalStart;  // :1310
  w.t_aligned_seq_pos = matchIntervalStart;                         // :1309
  if (iv.strand == 1)  // :1316-1323 the window re-addressed on the reverse strand
    w.t_aligned_seq_pos = genomeLen - (matchIntervalStart + w.t_aligned_seq_length - 1u) - 1u;
  w.t_strand = iv.strand == 1 ? 1 : 0;
  w.t_index = tIndex;
  w.t_length = tLength;
  w.q_aligned_seq_pos = rd.subread_start;  // :1332 always the forward read
  w.q_aligned_seq_length = sEnd - sStart;
  w.q_length = rd.length;
  windows[i] = w;
  WindowPlan p;
  p.fwdStart = matchIntervalStart;
  p.tLen = w.t_aligned_seq_length;
  p.qLen = w.q_aligned_seq_length;
  p.qStart = rd.subread_start;
  plan[i] = p;
  bytes[i] = (int64_t)p.tLen + (int64_t)p.qLen;
}

// ReverseComplementNuc (NucConversion.h:337-370): codes 0..3, ACGTN in both cases, everything else 127
__device__ __forceinline__ uint8_t rc_nuc(uint8_t c) {
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

__global__ __launch_bounds__(256) void window_gather_kernel(int n, const blasr_hip_interval *intervals,
                                                            const blasr_hip_read *reads, const uint8_t *readPool,
                                                            const uint8_t *genome, const WindowPlan *plan,
                                                            const int64_t *offs, int64_t capacity, uint8_t *seqOut,
                                                            int64_t *qOff, int32_t *qLen, int64_t *tOff,
                                                            int32_t *tLen) {
  __shared__ uint8_t rcOf[256];  // ReverseComplementNuc by character
  rcOf[threadIdx.x] = rc_nuc((uint8_t)threadIdx.x);
  __syncthreads();
  const int i = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (i >= n) return;
  const int lane = lane_id();
  const WindowPlan p = plan[i];
  const int64_t at = offs[i];
  if (lane == 0) {
    qOff[i] = at;
    qLen[i] = (int32_t)p.qLen;
    tOff[i] = at + p.qLen;
    tLen[i] = (int32_t)p.tLen;
  }
  if (at + (int64_t)p.qLen + (int64_t)p.tLen > capacity) return;  // reported by the host
  const blasr_hip_interval iv = intervals[i];
  const uint8_t *q = readPool + reads[iv.read].off + p.qStart;
  uint8_t *dq = seqOut + at, *dt = dq + p.qLen;
  // sixteen bytes per lane and trip (unaligned loads and stores: the pools have no alignment), the tail byte by byte
  auto copy = [&](uint8_t *dst, const uint8_t *src, uint32_t len) {
    uint32_t j = 16u * lane;
    for (; j + 16u <= len; j += 16u * 64u) {
      uint4 v;
      __builtin_memcpy(&v, src + j, 16);
      __builtin_memcpy(dst + j, &v, 16);
    }
    const uint32_t done = len & ~15u;
    if (done + lane < len) dst[done + lane] = src[done + lane];
  };
  copy(dq, q, p.qLen);
  const uint8_t *g = genome + p.fwdStart;
  if (iv.strand == 1) {
    // CopyAsRC: output bytes j .. j+15 are the complements of source bytes tLen-1-j .. tLen-16-j
    uint32_t j = 16u * lane;
    for (; j + 16u <= p.tLen; j += 16u * 64u) {
      uint32_t w[4], o[4];
      __builtin_memcpy(w, g + (p.tLen - 16u - j), 16);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t x = w[3 - k];  // its highest byte comes first
        o[k] = (uint32_t)rcOf[x >> 24] | (uint32_t)rcOf[(x >> 16) & 0xffu] << 8 | (uint32_t)rcOf[(x >> 8) & 0xffu] << 16 |
               (uint32_t)rcOf[x & 0xffu] << 24;
      }
      __builtin_memcpy(dt + j, o, 16);
    }
    const uint32_t done = p.tLen & ~15u;
    if (done + lane < p.tLen) dt[done + lane] = rcOf[g[p.tLen - 1u - (done + lane)]];
  } else {
    copy(dt, g, p.tLen);
  }
}

}  // namespace blasr_hip

using namespace blasr_hip;

namespace {
void drop_genome(blasr_hip_ctx *ctx) {
  if (ctx->genome && ctx->genomeOwned) (void)hipFree(ctx->genome);
  if (ctx->contigStarts) (void)hipFree(ctx->contigStarts);
  ctx->genome = nullptr;
  ctx->saExtGenome = nullptr;  // (anchor_api.hip's companion array was made from this genome)
  ctx->indexEpoch++;           // (what a borrower shares has changed: blasr_hip_borrow_index)
  ctx->genomeOwned = true;
  ctx->contigStarts = nullptr;
  ctx->genomeLen = 0;
  ctx->nContigs = 0;
}
}  // namespace

extern "C" {

int blasr_hip_adopt_genome_device(blasr_hip_ctx *ctx, const uint8_t *ascii, int64_t length,
                                  const uint32_t *contig_starts_host, int32_t n_contigs) try {
  if (!ctx || !ascii || length <= 0 || length >= (1ll << 32) || n_contigs < 0 || (contig_starts_host && n_contigs < 1))
    return ctx ? blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "adopt_genome: bad arguments") : BLASR_HIP_ERR_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  drop_genome(ctx);
  if (contig_starts_host) {
    HIP_TRY(ctx, hipMalloc(&ctx->contigStarts, sizeof(uint32_t) * (size_t)(n_contigs + 1)));
    HIP_TRY(ctx, hipMemcpy(ctx->contigStarts, contig_starts_host, sizeof(uint32_t) * (size_t)(n_contigs + 1),
                           hipMemcpyHostToDevice));
    ctx->nContigs = n_contigs;
  }
  ctx->genome = const_cast<uint8_t *>(ascii);
  ctx->genomeOwned = false;
  ctx->genomeLen = length;
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CTX(ctx)

int blasr_hip_upload_genome(blasr_hip_ctx *ctx, const uint8_t *ascii, int64_t length, const uint32_t *contig_starts,
                            int32_t n_contigs) try {
  if (!ctx || !ascii || length <= 0 || length >= (1ll << 32) || n_contigs < 0 || (contig_starts && n_contigs < 1))
    return ctx ? blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "upload_genome: bad arguments") : BLASR_HIP_ERR_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  drop_genome(ctx);
  HIP_TRY(ctx, hipMalloc(&ctx->genome, (size_t)length));
  ctx->genomeOwned = true;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->genome, ascii, (size_t)length, hipMemcpyHostToDevice, ctx->stream));
  if (contig_starts) {
    HIP_TRY(ctx, hipMalloc(&ctx->contigStarts, sizeof(uint32_t) * (size_t)(n_contigs + 1)));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->contigStarts, contig_starts, sizeof(uint32_t) * (size_t)(n_contigs + 1),
                                hipMemcpyHostToDevice, ctx->stream));
    ctx->nContigs = n_contigs;
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->genomeLen = length;
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CTX(ctx)

int blasr_hip_windows_device(blasr_hip_ctx *ctx, int32_t n, const blasr_hip_interval *intervals,
                             const blasr_hip_read *reads, const uint8_t *read_pool,
                             const blasr_hip_window_params *params, uint8_t *seq_out, int64_t seq_capacity,
                             int64_t *q_off, int32_t *q_len, int64_t *t_off, int32_t *t_len,
                             blasr_hip_window *windows, int64_t *seq_used) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (n < 0 || !params || !seq_used || (n > 0 && (!intervals || !reads || !read_pool || !seq_out || !q_off || !q_len ||
                                                  !t_off || !t_len || !windows)))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "windows: bad arguments");
  if (!ctx->genome) return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "windows: no genome uploaded");
  *seq_used = 0;
  if (n == 0) return BLASR_HIP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  // temporaries: plan, byte counts, offsets, scan scratch
  size_t scanBytes = 0;
  int64_t *nullI64 = nullptr;
  HIP_TRY(ctx, rocprim::exclusive_scan(nullptr, scanBytes, nullI64, nullI64, (int64_t)0, (size_t)n + 1,
                                       rocprim::plus<int64_t>(), st));
  const size_t need = sizeof(WindowPlan) * (size_t)n + 2 * sizeof(int64_t) * ((size_t)n + 1) + scanBytes + 1024;
  if (ctx->workspaceBytes < need) {
    if (ctx->workspace) (void)hipFree(ctx->workspace);
    ctx->workspace = nullptr;
    ctx->workspaceBytes = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->workspace, need));
    ctx->workspaceBytes = need;
  }
  char *at = ctx->workspace;
  auto take = [&](size_t bytes) {
    char *p = at;
    at += (bytes + 255) & ~(size_t)255;
    return p;
  };
  int64_t *bytes = reinterpret_cast<int64_t *>(take(sizeof(int64_t) * ((size_t)n + 1)));
  int64_t *offs = reinterpret_cast<int64_t *>(take(sizeof(int64_t) * ((size_t)n + 1)));
  WindowPlan *plan = reinterpret_cast<WindowPlan *>(take(sizeof(WindowPlan) * (size_t)n));
  void *scanTmp = take(scanBytes);
  HIP_TRY(ctx, hipMemsetAsync(bytes + n, 0, sizeof(int64_t), st));
  hipLaunchKernelGGL(window_plan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, intervals, reads,
                     (uint32_t)ctx->genomeLen, ctx->contigStarts, ctx->nContigs, *params, windows, plan, bytes);
  HIP_TRY(ctx, rocprim::exclusive_scan(scanTmp, scanBytes, bytes, offs, (int64_t)0, (size_t)n + 1,
                                       rocprim::plus<int64_t>(), st));
  hipLaunchKernelGGL(window_gather_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, n, intervals, reads,
                     read_pool, ctx->genome, plan, offs, seq_capacity, seq_out, q_off, q_len, t_off, t_len);
  int64_t total = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&total, offs + n, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  HIP_TRY(ctx, hipGetLastError());
  *seq_used = total;
  if (total > seq_capacity)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_CAPACITY, "windows: %lld bytes needed, %lld provided", (long long)total,
                          (long long)seq_capacity);
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CTX(ctx)

int blasr_hip_windows_host(blasr_hip_ctx *ctx, int32_t n, const blasr_hip_interval *intervals, int32_t n_reads,
                           const blasr_hip_read *reads, const uint8_t *read_pool, int64_t read_pool_bytes,
                           const blasr_hip_window_params *params, uint8_t *seq_out, int64_t seq_capacity,
                           int64_t *q_off, int32_t *q_len, int64_t *t_off, int32_t *t_len,
                           blasr_hip_window *windows, int64_t *seq_used) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (n < 0 || n_reads < 0 || read_pool_bytes < 0 || seq_capacity < 0 || !params || !seq_used)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "windows: bad arguments");
  for (int i = 0; i < n; i++)
    if (intervals[i].read < 0 || intervals[i].read >= n_reads)
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "windows: interval %d names read %d of %d", i,
                            intervals[i].read, n_reads);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  struct Dev {
    void *p = nullptr;
    ~Dev() {
      if (p) (void)hipFree(p);
    }
  };
  Dev dIv, dRd, dPool, dSeq, dQOff, dQLen, dTOff, dTLen, dWin;
  const size_t n1 = (size_t)std::max(n, 1), r1 = (size_t)std::max(n_reads, 1);
  HIP_TRY(ctx, hipMalloc(&dIv.p, sizeof(blasr_hip_interval) * n1));
  HIP_TRY(ctx, hipMalloc(&dRd.p, sizeof(blasr_hip_read) * r1));
  HIP_TRY(ctx, hipMalloc(&dPool.p, (size_t)std::max<int64_t>(read_pool_bytes, 1)));
  HIP_TRY(ctx, hipMalloc(&dSeq.p, (size_t)std::max<int64_t>(seq_capacity, 1)));
  HIP_TRY(ctx, hipMalloc(&dQOff.p, sizeof(int64_t) * n1));
  HIP_TRY(ctx, hipMalloc(&dTOff.p, sizeof(int64_t) * n1));
  HIP_TRY(ctx, hipMalloc(&dQLen.p, sizeof(int32_t) * n1));
  HIP_TRY(ctx, hipMalloc(&dTLen.p, sizeof(int32_t) * n1));
  HIP_TRY(ctx, hipMalloc(&dWin.p, sizeof(blasr_hip_window) * n1));
  if (n > 0) HIP_TRY(ctx, hipMemcpyAsync(dIv.p, intervals, sizeof(blasr_hip_interval) * (size_t)n, hipMemcpyHostToDevice, st));
  if (n_reads > 0) HIP_TRY(ctx, hipMemcpyAsync(dRd.p, reads, sizeof(blasr_hip_read) * (size_t)n_reads, hipMemcpyHostToDevice, st));
  if (read_pool_bytes > 0) HIP_TRY(ctx, hipMemcpyAsync(dPool.p, read_pool, (size_t)read_pool_bytes, hipMemcpyHostToDevice, st));
  const int rc = blasr_hip_windows_device(
      ctx, n, static_cast<blasr_hip_interval *>(dIv.p), static_cast<blasr_hip_read *>(dRd.p),
      static_cast<uint8_t *>(dPool.p), params, static_cast<uint8_t *>(dSeq.p), seq_capacity,
      static_cast<int64_t *>(dQOff.p), static_cast<int32_t *>(dQLen.p), static_cast<int64_t *>(dTOff.p),
      static_cast<int32_t *>(dTLen.p), static_cast<blasr_hip_window *>(dWin.p), seq_used);
  if (rc != BLASR_HIP_OK && rc != BLASR_HIP_ERR_CAPACITY) return rc;
  if (n > 0) {
    HIP_TRY(ctx, hipMemcpyAsync(q_off, dQOff.p, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(t_off, dTOff.p, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(q_len, dQLen.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(t_len, dTLen.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(windows, dWin.p, sizeof(blasr_hip_window) * (size_t)n, hipMemcpyDeviceToHost, st));
    if (rc == BLASR_HIP_OK && *seq_used > 0)
      HIP_TRY(ctx, hipMemcpyAsync(seq_out, dSeq.p, (size_t)*seq_used, hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return rc;
} BLASR_HIP_CATCH_CTX(ctx)

// ---------------------------------------------------------------------------------------------
// a-10 (host).  The reference sorts a vector of candidate pointers with std::sort; sorting the index
// list with the same comparator through the same libstdc++ algorithm performs the same swaps, so ties
// (equal score and equal tPos + tAlignedSeqPos) fall the same way.
// ---------------------------------------------------------------------------------------------
int blasr_hip_sort_candidates(const blasr_hip_candidate *cands, int32_t n, int32_t *order) try {
  if (n < 0 || (n > 0 && (!cands || !order))) return BLASR_HIP_ERR_INVALID;
  std::sort(order, order + n, [cands](int32_t l, int32_t r) {  // SortAlignmentPointersByScore, Blasr.cpp:226-236
    const blasr_hip_candidate &lhs = cands[l], &rhs = cands[r];
    if (lhs.score == rhs.score) return lhs.t_pos + lhs.t_aligned_seq_pos < rhs.t_pos + rhs.t_aligned_seq_pos;
    return lhs.score < rhs.score;
  });
  return n;
} BLASR_HIP_CATCH_CODE

int blasr_hip_remove_low_quality(blasr_hip_candidate *cands, const int32_t *blocks, int32_t n, int32_t *order,
                                 int32_t n_candidates, int32_t max_gap, int32_t max_score) try {
  if (n < 0 || (n > 0 && (!cands || !order))) return BLASR_HIP_ERR_INVALID;
  const int32_t lim = std::min(n_candidates, n);  // MIN(params.nCandidates, alignmentPtrs.size()), :2027
  for (int32_t i = 0; i < lim; i++) {
    blasr_hip_candidate &c = cands[order[i]];
    if (max_gap > 0) {  // :2031-2046
      if (!blocks && c.n_blocks > 1) return BLASR_HIP_ERR_INVALID;
      for (int32_t j = 1; j < c.n_blocks; j++) {
        const int32_t *b = blocks + 3 * (c.blk_off + j), *p = b - 3;
        const int gq = (int)((uint32_t)b[0] - (uint32_t)p[0] - (uint32_t)b[2]);
        const int gt = (int)((uint32_t)b[1] - (uint32_t)p[1] - (uint32_t)b[2]);
        if ((gq >= max_gap && gt < max_gap) || (gt > max_gap && gq < max_gap)) {
          c.score = max_score;
          break;
        }
      }
    }
    if (c.n_blocks == 0 || c.score > max_score) return i;  // this one and all after it are removed (:2048-2068)
  }
  return n;
} BLASR_HIP_CATCH_CODE

int blasr_hip_remove_overlapping(const blasr_hip_candidate *cands, int32_t n, int32_t *order,
                                 float min_pct_identity) try {
  if (n < 0 || (n > 0 && (!cands || !order))) return BLASR_HIP_ERR_INVALID;
  std::vector<unsigned char> contained((size_t)std::max(n, 1), 0);
  for (int32_t i = 0; i + 1 < n; i++) {
    const blasr_hip_candidate &a = cands[order[i]];
    if (a.pct_similarity < min_pct_identity) continue;  // :2094-2096
    const uint32_t aB = a.t_aligned_seq_pos + a.t_pos, aE = aB + a.t_end;  // GenomicTBegin / GenomicTEnd
    for (int32_t j = i + 1; j < n; j++) {
      if (contained[j]) continue;
      const blasr_hip_candidate &b = cands[order[j]];
      if (a.t_index != b.t_index) continue;
      const uint32_t bB = b.t_aligned_seq_pos + b.t_pos, bE = bB + b.t_end;
      if (aB <= bB && aE >= bE && a.t_strand == b.t_strand) {
        if (a.score <= b.score) contained[j] = 1;
      } else if (bB <= aB && bE >= aE && a.t_strand == b.t_strand) {
        if (b.score <= a.score) contained[i] = 1;
      }
    }
  }
  int32_t kept = 0;
  for (int32_t i = 0; i < n; i++)
    if (!contained[i]) order[kept++] = order[i];
  return kept;
} BLASR_HIP_CATCH_CODE

}  // extern "C"
