// The context behind blasr_hip_ctx (shared by the refine and SDP host layers).
#ifndef BLASR_AMD_CTX_H_
#define BLASR_AMD_CTX_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <exception>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/blasr_hip.h"
#include "abi_guard.h"

// A grow-only device buffer owned by the context (freed in blasr_hip_destroy).
struct DevBuf {
  char *p = nullptr;
  size_t bytes = 0;
};

struct blasr_hip_ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  hipStream_t sideStream = nullptr;  // long thin kernels that can run underneath the main ones
  // the sweep's kernels of one recursion level work on disjoint invocations: they run side by side (sdp_api.hip)
  enum { kAuxStreams = 4 };
  hipStream_t auxStream[kAuxStreams] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t auxEvent[kAuxStreams + 1] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // [0] fork, [1..] joins
  hipEvent_t events[16] = {};        // timing / cross-stream events of the refine call, created with the context
  char *workspace = nullptr;
  size_t workspaceBytes = 0;
  int64_t workspaceLimit = 48ll << 30;
  std::string lastError;
  float ms[4] = {0, 0, 0, 0};
  int64_t counters[3] = {0, 0, 0};
  // the dominant kernel of the refine stage on its own: fillp_kernel launches of the last call
  float fill4Ms = 0;
  int64_t fill4Launches = 0, fill4Bytes = 0, fill4Cells = 0;
  // every heavy kernel of the refine stage on its own (blasr_hip_refine_last_kernels), indexed by BLASR_HIP_KERNEL_*
  blasr_hip_kernel_record kernelRec[BLASR_HIP_KERNEL_KINDS] = {};
  float sdpMs[4] = {0, 0, 0, 0};
  int64_t sdpCounters[3] = {0, 0, 0};
  int64_t sdpBytes[4] = {0, 0, 0, 0};
  // every heavy SDP kernel on its own (blasr_hip_sdp_last_kernels); the events its launches are bracketed with are
  // created on demand and kept (destroyed with the context)
  blasr_hip_sdp_kernel_record sdpKernelRec[BLASR_HIP_SDP_KERNEL_KINDS] = {};
  std::vector<hipEvent_t> sdpEventPool;
  double sdpBytesPerBase = 48.0;  // level arrays per base of query + target, as the last pieces needed them
  // the resident genome (ASCII) and its contig table, for the window gather
  uint8_t *genome = nullptr;
  bool genomeOwned = true;  // false: adopted from the caller (blasr_hip_adopt_genome_device)
  int64_t genomeLen = 0;
  uint32_t *contigStarts = nullptr;
  int nContigs = 0;
  // the mapper's device-side state between its stages (blasr_hip_map_reads_*): window pool, SDP batch arrays,
  // result records, block and op arenas, the trimmed refine batch
  enum { kMapBufs = 24 };
  DevBuf mapBuf[kMapBufs];
  // the mapper's host-side arrays that receive device data (windows, lengths, SDP and refine records): pinned, grow-only
  // and kept between calls -- as pageable std::vectors they cost a staged copy each and ~40 MB of fresh pages per call
  enum { kMapHostBufs = 24 };
  struct HostBuf {
    void *p = nullptr;
    size_t bytes = 0;
  } mapHost[kMapHostBufs];
  float mapMs[6] = {0, 0, 0, 0, 0, 0};
  int64_t mapCounters[6] = {0, 0, 0, 0, 0, 0};
  // the resident suffix array of the genome (blasr_hip_upload_suffix_array): index, the lookup table's ranges
  uint32_t *saIndex = nullptr, *saStartPos = nullptr, *saEndPos = nullptr;
  bool saOwned = true;  // false: index, lookup table and companion array belong to another context (blasr_hip_borrow_index)
  // Counts every change of what a borrower would share -- genome, index array, lookup table, companion arrays (drop_genome,
  // blasr_hip_drop_index, blasr_hip_ensure_index_ext).  A borrower keeps the owner and the count it borrowed at: an address
  // that was freed and handed out again compares equal, the count does not.
  uint64_t indexEpoch = 0;
  const blasr_hip_ctx *saBorrowedFrom = nullptr;
  uint64_t saBorrowedEpoch = 0;
  int64_t saLen = 0, saTableLen = 0;
  int saPrefixLen = 0;
  // per array position, the bases that follow the lookup prefix (anchor_api.hip: the search reads them instead of
  // index[] + genome[]); built at the first anchor call for the (genome, index) pair it was made from
  uint16_t *saExt = nullptr;
  uint2 *saRanges = nullptr;  // made with it: {saStartPos[w], saEndPos[w]} side by side
  const void *saExtGenome = nullptr, *saExtIndex = nullptr;
  // anchor stage: what the host needs back after each piece (the lists' first anchors) is written by the device
  // straight into pinned host memory -- no copy queued behind other transfers; the probe count and the assert flag
  // stay on the device until the call ends
  int64_t *anchorPinned = nullptr;
  size_t anchorPinnedBytes = 0;
  unsigned long long *anchorFlags = nullptr;  // device: {probes, error}
  // the whole-mapper driver (pipeline_api.hip): all anchors of a block on the device and mirrored in pinned host
  // memory, the intervals found
  DevBuf pipeAnchors, pipeIntervals;
  void *pipePinned = nullptr;
  size_t pipePinnedBytes = 0;
  // the overlapped form of that driver: a second context on the same device (own streams, workspace and anchor
  // buffers; genome and index borrowed) on which a host thread anchors sub-block b + 1 while this one runs the path on b
  blasr_hip_ctx *pipeHelper = nullptr;
  float pipeMs[4] = {0, 0, 0, 0};
  int64_t pipeCounters[2] = {0, 0};
  enum { kAnchorBufs = 4 };
  DevBuf anchorBuf[kAnchorBufs];  // sort keys / values of the anchor stage (double-buffered)
  float anchorMs[4] = {0, 0, 0, 0};
  int64_t anchorCounters[4] = {0, 0, 0, 0};
};

// what the workspace may grow to: the caller's limit, and the context's own allocation plus the free device memory
// less a reserve
int64_t blasr_hip_workspace_reach(const blasr_hip_ctx *ctx);
// grows (never shrinks) the workspace; the old contents are NOT kept.  BLASR_HIP_ERR_NOMEM beyond the reach.
int blasr_hip_grow_workspace(blasr_hip_ctx *ctx, size_t bytes);
// grows (never shrinks) a context buffer; the old contents are NOT kept
int blasr_hip_ensure(blasr_hip_ctx *ctx, DevBuf &b, size_t bytes);

// the -extend step of AlignIntervals on the SDP stage's device records (extend_api.hip)
int blasr_hip_extend_candidates(blasr_hip_ctx *ctx, int n, const blasr_hip_interval *intervals,
                                const blasr_hip_read *reads, const uint8_t *read_pool,
                                const blasr_hip_extend_params *ep, const blasr_hip_score_params *score,
                                const uint8_t *seq, int64_t *qOff, int32_t *qLen, int64_t *tOff, int32_t *tLen,
                                blasr_hip_window *win, blasr_hip_sdp_result *res, const int32_t *blocks,
                                DevBuf &newSeq, DevBuf &newBlocks, DevBuf &scratch, DevBuf &opsBuf,
                                int64_t *blocks_used);

// what the device's interval finder hands to the host (intervals_device.hip -> intervals_api.hip): the chain of one
// cluster of anchors -- its first and last anchor, the sum of its anchors' lengths, their number
struct blasr_hip_iv_cluster {
  uint32_t first_t, first_q, first_l, last_t, last_q, last_l;
  int32_t sum_l, count;
};
int blasr_hip_intervals_from_clusters(const uint8_t *genome, int64_t genome_length, const uint32_t *seq_start_pos,
                                      int32_t n_seq_pos, const int32_t *count_table, int32_t count_tuple_size,
                                      int32_t n_reads, const blasr_hip_read *reads, const blasr_hip_iv_cluster *clusters,
                                      const int64_t *cluster_off, const blasr_hip_interval_params *params,
                                      blasr_hip_interval *intervals, float *p_values, int64_t capacity,
                                      int64_t *n_intervals, int64_t *read_interval_off, int64_t *cluster_of_interval = nullptr);

int blasr_hip_fail(blasr_hip_ctx *ctx, int code, const char *fmt, ...);

// index array, lookup table and companion array of the context: freed when they are its own, forgotten either way
void blasr_hip_drop_index(blasr_hip_ctx *ctx);
// the companion array of the resident index (anchor_api.hip), made once per (genome, index) pair
int blasr_hip_ensure_index_ext(blasr_hip_ctx *ctx, hipStream_t st);
// false when the index was borrowed and its owner has replaced or rebuilt any of it since (ctx.h: indexEpoch)
bool blasr_hip_borrowed_index_is_current(const blasr_hip_ctx *ctx);

// A host array of the mapper that receives device data: a view of one of the context's pinned, grow-only buffers
// (ctx.h: mapHost).  Not initialised: every element is written by the copy that fills it.
template <typename T>
struct PinnedArray {
  T *p = nullptr;
  size_t n = 0;
  T *data() { return p; }
  const T *data() const { return p; }
  size_t size() const { return n; }
  T &operator[](size_t i) { return p[i]; }
  const T &operator[](size_t i) const { return p[i]; }
};
enum { MH_IV, MH_WIN, MH_QLEN, MH_TLEN, MH_QOFF, MH_TOFF, MH_SDP, MH_REF, MH_QLEN2, MH_TLEN2, MH_GEO, MH_SDP_TOP,
       MH_FIRST, MH_BYREAD, MH_AT, MH_CAND, MH_ORDER, MH_LIVE, MH_SEL, MH_SELFIRST, MH_SLOT };
static_assert(MH_SLOT < blasr_hip_ctx::kMapHostBufs, "mapHost slots");
template <typename T>
inline int pinned_array(blasr_hip_ctx *ctx, int slot, size_t count, PinnedArray<T> &out) {
  blasr_hip_ctx::HostBuf &b = ctx->mapHost[slot];
  const size_t bytes = sizeof(T) * std::max<size_t>(count, 1);
  if (b.bytes < bytes) {
    if (b.p) (void)hipHostFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
    const size_t want = bytes + bytes / 4;
    if (hipHostMalloc(&b.p, want, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      b.p = nullptr;
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_NOMEM, "map_reads: %zu bytes of pinned host memory", want);
    }
    b.bytes = want;
  }
  out.p = (T *)b.p;
  out.n = count;
  return BLASR_HIP_OK;
}


#define HIP_TRY(ctx, expr)                                                                            \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess)                                                                             \
      return blasr_hip_fail(ctx, e_ == hipErrorOutOfMemory ? BLASR_HIP_ERR_NOMEM : BLASR_HIP_ERR_HIP, \
                            "%s failed: %s", #expr, hipGetErrorString(e_));                           \
  } while (0)

#endif
