// sdp_api.hip -- host side of blasr_hip_sdp_*: the level-by-level orchestration of the SDP
// kernels (sdp_kernels.hip).  SDPAlign's recursion (SDPAlign.h:445-456, :509-520) is run as
// successive waves of kernels; all variable-size intermediates live in stream-ordered device
// allocations sized from device-side scans.  No CPU algorithm here: only planning and launches.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "../../include/blasr_hip.h"
#include "ctx.h"
#include "sdp_kernels.h"

using namespace blasr_hip;

static_assert(sizeof(SdpResult) == sizeof(blasr_hip_sdp_result), "result record layout");

#ifdef BLASR_IDX_CLOCKS
namespace blasr_hip { void sdp_print_idx_clocks(); }
#endif
#ifdef BLASR_ROWS_CLOCKS
namespace blasr_hip { void sdp_print_rows_clocks(); }
#endif
namespace {

// Two-ended bump allocator over the context's ONE grow-only workspace (the region the refine stage uses after this
// stage is done with it: the analogue of MappingBuffers' "grow to the high-water mark and keep", Blasr.cpp:313-369).
// get() takes from the bottom: the arrays of ONE recursion level (keys, fragments, the sweep's arrays, counts, sort
// temporaries), given back as a whole by release_level() once the level's kernels are done; keep() takes from the
// top: what outlives its level (the block pool entries and SWAlign jobs a level emitted, the next level's tasks) --
// so the stage's footprint is the largest level plus what the levels handed on, not the sum of all levels.
// A request that does not fit what is left between the two ends gets an allocation of its own (a level's are freed
// with the level, the others when the call ends); the workspace is then grown to the call's high-water mark, so that
// the next call of that size is served from the one region again.  Nothing is ever left fragmented or unaccounted:
// what the context holds is the workspace, bounded by blasr_hip_set_workspace_limit.
struct Arena {
  blasr_hip_ctx *ctx;
  hipStream_t st;
  size_t bottom = 0, top = 0;            // bytes taken from either end of the workspace
  size_t ownLevel = 0, ownKept = 0;      // bytes in allocations of their own
  size_t high = 0;                       // the most that was in use at once
  std::vector<void *> overflowLevel, overflowKept;
  hipError_t err = hipSuccess;
  int64_t *highOut = nullptr;  // receives the high-water mark
  const bool trace = getenv("BLASR_HIP_SDP_TRACE") != nullptr && atoi(getenv("BLASR_HIP_SDP_TRACE")) > 1;
  Arena(blasr_hip_ctx *c, hipStream_t s) : ctx(c), st(s) {}
  Arena(const Arena &) = delete;
  Arena &operator=(const Arena &) = delete;
  size_t capacity() const { return ctx->workspaceBytes / 256 * 256; }
  static size_t rounded(size_t bytes) { return (std::max<size_t>(bytes, 256) + 255) / 256 * 256; }
  void note() { high = std::max(high, bottom + top + ownLevel + ownKept); }
  void *own(size_t bytes, std::vector<void *> &list, size_t &counter) {
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      err = e;
      return nullptr;
    }
    list.push_back(p);
    counter += bytes;
    note();
    return p;
  }
  template <typename T>
  T *get(size_t n) {  // lives until release_level()
    const size_t bytes = rounded(n * sizeof(T));
    if (trace && bytes >= (512u << 20)) fprintf(stderr, "[sdp arena] get %.2f GB (elements of %zu bytes) at %.2f GB\n", bytes / 1e9, sizeof(T), bottom / 1e9);
    if (bottom + bytes + top <= capacity()) {
      char *p = ctx->workspace + bottom;
      bottom += bytes;
      note();
      return reinterpret_cast<T *>(p);
    }
    return reinterpret_cast<T *>(own(bytes, overflowLevel, ownLevel));
  }
  template <typename T>
  T *keep(size_t n) {  // lives until the call ends
    const size_t bytes = rounded(n * sizeof(T));
    if (trace && bytes >= (512u << 20)) fprintf(stderr, "[sdp arena] keep %.2f GB (elements of %zu bytes), kept so far %.2f GB\n", bytes / 1e9, sizeof(T), top / 1e9);
    if (bottom + bytes + top <= capacity()) {
      top += bytes;
      note();
      return reinterpret_cast<T *>(ctx->workspace + capacity() - top);
    }
    return reinterpret_cast<T *>(own(bytes, overflowKept, ownKept));
  }
  // the level's kernels are done (the caller has synchronised the stream): its arrays go back
  void release_level(size_t mark) {
    bottom = mark;
    for (void *p : overflowLevel) (void)hipFree(p);
    overflowLevel.clear();
    ownLevel = 0;
  }
  // end of the call (every path): the stream must be idle before memory it may still use goes away
  ~Arena() {
    if (highOut) *highOut = (int64_t)high;
    if (overflowLevel.empty() && overflowKept.empty()) return;
    (void)hipStreamSynchronize(st);
    for (void *p : overflowLevel) (void)hipFree(p);
    for (void *p : overflowKept) (void)hipFree(p);
    const std::string keepMsg = ctx->lastError;  // best effort: failing to grow is not an error, and not the call's message
    (void)blasr_hip_grow_workspace(ctx, high + high / 8);
    ctx->lastError = keepMsg;
  }
};

struct Events {
  hipEvent_t a = nullptr, b = nullptr;
  ~Events() {
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
  }
};

struct Counters {
  unsigned long long pool, sw, child;
  int32_t overflow, pad;
};

int bits_for(uint64_t v) {
  int b = 1;
  while (b < 64 && (v >> b) != 0) b++;
  return b;
}

}  // namespace

extern "C" {

int blasr_hip_sdp_last_timing(const blasr_hip_ctx *ctx, float ms[4], int64_t counters[3]) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  for (int i = 0; i < 4; i++) ms[i] = ctx->sdpMs[i];
  for (int i = 0; i < 3; i++) counters[i] = ctx->sdpCounters[i];
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_sdp_last_bytes(const blasr_hip_ctx *ctx, int64_t bytes[4]) try {
  if (!ctx || !bytes) return BLASR_HIP_ERR_INVALID;
  for (int i = 0; i < 4; i++) bytes[i] = ctx->sdpBytes[i];
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_sdp_last_kernels(const blasr_hip_ctx *ctx, blasr_hip_sdp_kernel_record *out, int capacity) try {
  if (!ctx || (!out && capacity > 0) || capacity < 0) return BLASR_HIP_ERR_INVALID;
  int n = 0;
  for (int k = 0; k < BLASR_HIP_SDP_KERNEL_KINDS; k++) {
    if (ctx->sdpKernelRec[k].launches <= 0) continue;
    if (n < capacity) out[n] = ctx->sdpKernelRec[k];
    n++;
  }
  return n;
} BLASR_HIP_CATCH_CODE

const char *blasr_hip_sdp_kernel_name(int kind) {
  switch (kind) {  // (as rocprofv3 prints the kernel that dominates the record)
    case BLASR_HIP_SDP_KERNEL_MATCH_INDEX: return "blasr_hip::sdp_match_index_kernel";
    case BLASR_HIP_SDP_KERNEL_MATCH_SHORT: return "blasr_hip::sdp_match_kernel";
    case BLASR_HIP_SDP_KERNEL_SMALL: return "blasr_hip::sdp_small4_kernel + sdp_small_kernel";
    case BLASR_HIP_SDP_KERNEL_CHAIN4: return "blasr_hip::sdp_chain4_kernel";
    case BLASR_HIP_SDP_KERNEL_CHAIN_BITS_SMALL: return "blasr_hip::sdp_chain_bits_kernel<512>";
    case BLASR_HIP_SDP_KERNEL_CHAIN_BITS_MID: return "blasr_hip::sdp_chain_bits_kernel<2048>";
    case BLASR_HIP_SDP_KERNEL_CHAIN_BITS_LARGE: return "blasr_hip::sdp_chain_bits_kernel<4096>";
    case BLASR_HIP_SDP_KERNEL_CHAIN_BITS4: return "blasr_hip::sdp_chain_bits4_kernel<false>";
    case BLASR_HIP_SDP_KERNEL_CHAIN_BITS4_WIDE: return "blasr_hip::sdp_chain_bits4_kernel<true>";
    case BLASR_HIP_SDP_KERNEL_CHAIN_ROWS: return "blasr_hip::sdp_chain_rows_kernel";
    case BLASR_HIP_SDP_KERNEL_FILLS: return "blasr_hip::sw_small_kernel + sw_kernel";
  }
  return "";
}

}  // extern "C"

namespace {

// One chunk of a batch: arguments validated by blasr_hip_sdp_device; blk_off of the results counts from blkBase
// (the blocks of the chunks before), `blocks` / `blocksCap` are what is left of the caller's arena.
int sdp_device_chunk(blasr_hip_ctx *ctx, const blasr_hip_sdp_batch *b, const blasr_hip_sdp_params *sp,
                     const blasr_hip_score_params *score, blasr_hip_sdp_result *results, int32_t *blocks,
                     int64_t blocksCap, int64_t *blocksUsed, int64_t blkBase, int64_t *arenaHigh) {
  const int n = b->n;
  hipStream_t st = ctx->stream;
  Arena arena(ctx, st);
  arena.highOut = arenaHigh;
  Events ev;
  HIP_TRY(ctx, hipEventCreate(&ev.a));
  HIP_TRY(ctx, hipEventCreate(&ev.b));
  hipEvent_t evA = ev.a, evB = ev.b;
  float phaseMs[4] = {0, 0, 0, 0}, levelMs[4] = {0, 0, 0, 0};  // (levelMs: the level at hand, for BLASR_HIP_SDP_TRACE)
  int64_t phaseBytes[4] = {0, 0, 0, 0};  // algorithmic bytes by SURVEY 8(d)'s model, see blasr_hip_sdp_last_bytes
  auto tick = [&]() { (void)hipEventRecord(evA, st); };
  auto tock = [&](int phase) {
    (void)hipEventRecord(evB, st);
    (void)hipEventSynchronize(evB);
    float t = 0;
    (void)hipEventElapsedTime(&t, evA, evB);
    phaseMs[phase] += t;
    levelMs[phase] += t;
  };
#define ARENA_CHECK()                                                                                        \
  do {                                                                                                       \
    if (arena.err != hipSuccess)                                                                             \
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_NOMEM, "device allocation failed: %s", hipGetErrorString(arena.err)); \
  } while (0)

  // ---- level 0 tasks from the batch (lengths are needed on the host to plan) ----
  std::vector<int64_t> qOff(n), tOff(n);
  std::vector<int32_t> qLen(n), tLen(n);
  HIP_TRY(ctx, hipMemcpyAsync(qOff.data(), b->q_off, sizeof(int64_t) * n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(tOff.data(), b->t_off, sizeof(int64_t) * n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(qLen.data(), b->q_len, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(tLen.data(), b->t_len, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  PinnedArray<SdpTask> top;  // (pinned and kept between calls, ctx.h: 14 MB at configs[2])
  if (int rcTop = pinned_array(ctx, MH_SDP_TOP, (size_t)n, top)) return rcTop;
  int32_t maxQLen = 0;
  for (int i = 0; i < n; i++) {
    if (qLen[i] < 0 || tLen[i] < 0 || qLen[i] >= (1 << 24) || tLen[i] >= (1 << 24)) {
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_UNSUPPORTED, "candidate %d: sequence length outside [0, 2^24)", i);
    }
    maxQLen = std::max(maxQLen, qLen[i]);
    SdpTask &t = top[i];
    t.qOff = qOff[i];
    t.tOff = tOff[i];
    t.qLen = qLen[i];
    t.tLen = tLen[i];
    t.k = sp->word_size;
    t.type = sp->align_type;
    t.prefix = sp->prefix_length;
    t.recurse = sp->recurse;
    t.maxPerPos = sp->max_per_position;
    t.top = i;
    t.qOrg = 0;
    t.tOrg = 0;
    t.flags = (sp->extend_front ? kSdpExtendFront : 0) | (sp->detailed ? kSdpDetailed : 0) | ((sp->sw_align_type & 0xff) << kSdpSwTypeShift);
    t.pad = 0;
  }
  SdpTask *dTop = arena.keep<SdpTask>(n);  // (needed until the call ends, like everything the levels hand on)
  int32_t *dMatrix = arena.keep<int32_t>(32);
  Counters *dCounters = arena.keep<Counters>(1);
  unsigned long long *dKernelSums = arena.keep<unsigned long long>(3 * BLASR_HIP_SDP_KERNEL_KINDS);
  ARENA_CHECK();
  HIP_TRY(ctx, hipMemsetAsync(dKernelSums, 0, sizeof(unsigned long long) * 3 * BLASR_HIP_SDP_KERNEL_KINDS, st));
  SdpSpans spans;
  spans.pool = &ctx->sdpEventPool;
  float kMs[BLASR_HIP_SDP_KERNEL_KINDS] = {};
  int32_t kLaunches[BLASR_HIP_SDP_KERNEL_KINDS] = {}, kConcurrent[BLASR_HIP_SDP_KERNEL_KINDS] = {};
  HIP_TRY(ctx, hipMemcpyAsync(dTop, top.data(), sizeof(SdpTask) * n, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dMatrix, score->matrix, sizeof(int32_t) * 25, hipMemcpyHostToDevice, st));

  SdpParams prm;
  prm.sdpIns = sp->sdp_ins;
  prm.sdpDel = sp->sdp_del;
  prm.indelRate = sp->indel_rate;
  prm.noRecurseUnder = sp->recurse_over;
  prm.match = score->matrix[0];
  prm.ins = score->ins;
  prm.del = score->del;

  struct LevelOut {
    uint64_t *poolKeys, *poolVals;
    int64_t poolCount;
    SwTask *sw;
    int64_t swCount;
  };
  std::vector<LevelOut> outs;
  const SdpTask *dTasks = dTop;
  int nTasks = n;
  int k = sp->word_size;
  int64_t totalInvocations = 0, totalFrags = 0;

  const int idxMaxT = getenv("BLASR_HIP_NO_INDEX") ? 0 : sdp_match_index_max_target(ctx->device);
  const size_t levelMark = arena.bottom;  // (what was taken so far lives for the whole call)
  for (int level = 0; nTasks > 0; level++) {
    totalInvocations += nTasks;
    SdpLevelArrays lv;
    memset(&lv, 0, sizeof(lv));
    lv.n = nTasks;
    lv.tasks = dTasks;
    lv.kernelSums = dKernelSums;
    lv.spans = &spans;
    // ---- k-mer keys of the targets, sorted per task ----
    tick();
    lv.idxMaxT = idxMaxT;
    lv.noIdxFilter = getenv("BLASR_HIP_NO_IDX_FILTER") ? 1 : 0;
    lv.idxFilterX2 = getenv("BLASR_HIP_IDX_FILTER_X2") ? std::max(1, atoi(getenv("BLASR_HIP_IDX_FILTER_X2"))) : 2;
    lv.bigList = arena.get<int32_t>(nTasks + 1);
    lv.bigCount = lv.bigList + nTasks;
    lv.midList = arena.get<int32_t>(nTasks + 1);
    lv.midCount = lv.midList + nTasks;
    lv.keyList = arena.get<int32_t>(nTasks + 1);
    lv.keyCount = lv.keyList + nTasks;
    lv.idxFlag = arena.get<int32_t>(nTasks);
    lv.idxWave = arena.get<int32_t>((size_t)4 * nTasks);
    lv.fused = arena.get<int32_t>(nTasks);
    lv.chainList = nullptr;
    if (!getenv("BLASR_HIP_NO_CHAIN4")) {
      lv.chainList = arena.get<int32_t>(nTasks + kChainCats);
      lv.chainCount = lv.chainList + nTasks;
      if (const char *e = getenv("BLASR_HIP_CHAIN_SCATTER")) lv.scatterMode = atoi(e);
      // a row per step for the scattered invocations with >= 4 fragments per query row on average and >= 1024 fragments
      // (BLASR_HIP_CHAIN_ROWS="minFrags,densityX2"; "0" switches the kernel off; BLASR_HIP_ROWS_PRIO raises its wavefronts'
      // priority, which is NOT the default).  configs[4], three runs each on one box: sweep 528-530 ms with it, 544-646 without (profiles/NOTES.md)
      lv.bits4Wide = getenv("BLASR_HIP_BITS4_WIDE") != nullptr ? atoi(getenv("BLASR_HIP_BITS4_WIDE")) : 4096;
      lv.rowsPrio = getenv("BLASR_HIP_ROWS_PRIO") != nullptr ? 1 : 0;
      lv.rowsMinFrags = 1024;
      lv.rowsDensityX2 = 8;
      if (const char *e = getenv("BLASR_HIP_CHAIN_ROWS")) {
        lv.rowsMinFrags = atoi(e);
        if (const char *c = strchr(e, ',')) lv.rowsDensityX2 = atoi(c + 1);
      }
    }
    ARENA_CHECK();
    launch_sdp_classify(lv, st);
    int64_t *sizes = arena.get<int64_t>(nTasks + 1);
    int64_t *keyOff = arena.get<int64_t>(nTasks + 1);
    ARENA_CHECK();
    HIP_TRY(ctx, hipMemsetAsync(sizes + nTasks, 0, sizeof(int64_t), st));
    launch_sdp_gather_sizes(lv, 0, sizes, st);
    size_t tmpBytes = 0;
    HIP_TRY(ctx, rocprim::exclusive_scan(nullptr, tmpBytes, sizes, keyOff, (int64_t)0, (size_t)nTasks + 1,
                                         rocprim::plus<int64_t>(), st));
    void *tmp = arena.get<char>(tmpBytes);
    ARENA_CHECK();
    HIP_TRY(ctx, rocprim::exclusive_scan(tmp, tmpBytes, sizes, keyOff, (int64_t)0, (size_t)nTasks + 1,
                                         rocprim::plus<int64_t>(), st));
    int64_t totalKeys = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&totalKeys, keyOff + nTasks, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (totalKeys >= (1ll << 32)) {
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_UNSUPPORTED, "batch too large: %lld target bases in one level",
                            (long long)totalKeys);
    }
    phaseBytes[0] += totalKeys * 8 * (1 + 2 * ((2 * k + 7) / 8));  // key + position records, P radix passes read + write
    uint64_t *keysIn = arena.get<uint64_t>((size_t)totalKeys);
    uint64_t *keysOut = arena.get<uint64_t>((size_t)totalKeys);
    ARENA_CHECK();
    lv.keyOff = keyOff;
    lv.keys = keysIn;
    launch_sdp_keys(lv, b->seq, st);
    launch_sdp_sort_keys(lv, keysOut, st);
    tock(0);

    // ---- fragments ----
    tick();
    lv.nRaw = arena.get<int32_t>((size_t)3 * nTasks);
    lv.nFrag = arena.get<int32_t>(nTasks);
    lv.chainEnd = arena.get<int32_t>(nTasks);
    int64_t *fragOff = arena.get<int64_t>(nTasks + 1);
    ARENA_CHECK();
    {  // per-query-position search results, written by the count pass and reused by the write pass
      int64_t *qPosOff = arena.get<int64_t>(nTasks + 1);
      ARENA_CHECK();
      launch_sdp_gather_sizes(lv, 3, sizes, st);
      HIP_TRY(ctx, rocprim::exclusive_scan(tmp, tmpBytes, sizes, qPosOff, (int64_t)0, (size_t)nTasks + 1,
                                           rocprim::plus<int64_t>(), st));
      int64_t totalQ = 0;
      HIP_TRY(ctx, hipMemcpyAsync(&totalQ, qPosOff + nTasks, sizeof(int64_t), hipMemcpyDeviceToHost, st));
      HIP_TRY(ctx, hipStreamSynchronize(st));
      phaseBytes[1] += totalQ * 8;
      lv.qPosOff = qPosOff;
      lv.hits = arena.get<uint64_t>((size_t)std::max<int64_t>(totalQ, 1));
      if (k >= 16) {  // (k: the level's largest word) the query's tuples as StoreMatchingPositions makes them (sdp_kernels.hip: kKeyBases)
        uint64_t *qWords = arena.get<uint64_t>((size_t)std::max<int64_t>(totalQ, 1));
        lv.qWords = qWords;
        phaseBytes[1] += totalQ * 8;
      }
      ARENA_CHECK();
      launch_sdp_query_words(lv, b->seq, st);
    }
    const bool matchSerial = getenv("BLASR_HIP_MATCH_SERIAL") != nullptr;
    launch_sdp_match_all(lv, b->seq, false, st, matchSerial ? nullptr : ctx->auxStream, matchSerial ? nullptr : ctx->auxEvent);
    launch_sdp_gather_sizes(lv, 1, sizes, st);
    HIP_TRY(ctx, rocprim::exclusive_scan(tmp, tmpBytes, sizes, fragOff, (int64_t)0, (size_t)nTasks + 1,
                                         rocprim::plus<int64_t>(), st));
    int64_t nFragTotal = 0, nColTotal = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&nFragTotal, fragOff + nTasks, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    // column slots of the invocations whose fragments may scatter (sdp_chain_bits_kernel): one per target base
    int64_t *colOff = nullptr;
    if (lv.chainList != nullptr && !getenv("BLASR_HIP_NO_CHAIN_BITS")) {
      colOff = arena.get<int64_t>(nTasks + 1);
      ARENA_CHECK();
      launch_sdp_gather_sizes(lv, 4, sizes, st);
      HIP_TRY(ctx, rocprim::exclusive_scan(tmp, tmpBytes, sizes, colOff, (int64_t)0, (size_t)nTasks + 1,
                                           rocprim::plus<int64_t>(), st));
      HIP_TRY(ctx, hipMemcpyAsync(&nColTotal, colOff + nTasks, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (nColTotal > 0) {
      lv.colOff = colOff;
      lv.colEnt = arena.get<int4>((size_t)nColTotal);
      ARENA_CHECK();
    }
    if (nFragTotal >= (1ll << 31)) {
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_UNSUPPORTED, "batch too large: %lld fragments in one level",
                            (long long)nFragTotal);
    }
    totalFrags += nFragTotal;
    phaseBytes[1] += nFragTotal * 16;  // fragments written
    phaseBytes[2] += nFragTotal * 16;  // and read by the sweep
    const size_t nf = (size_t)std::max<int64_t>(nFragTotal, 1);
    lv.fragOff = fragOff;
    lv.raw = arena.get<Frag>(nf);
    lv.frags = arena.get<Frag>(nf);
    lv.yxKeys = arena.get<uint64_t>(nf);
    lv.yxIdx = arena.get<uint32_t>(nf);
    uint64_t *yxKeysOut = arena.get<uint64_t>(nf);
    uint32_t *yxIdxOut = arena.get<uint32_t>(nf);
    lv.cost = arena.get<int32_t>(nf);
    lv.chainPrev = arena.get<int32_t>(nf);
    lv.chainLen = arena.get<int32_t>(nf);
    lv.predCl = arena.get<int32_t>(nf);
    lv.above = arena.get<int32_t>(nf);
    lv.colSet = arena.get<int4>(nf + 64);
    lv.tieList = arena.get<int32_t>(lv.n + 1);
    lv.tieCount = lv.tieList + lv.n;
    lv.cleanList = nullptr;
    lv.cleanCount = nullptr;
    // the general kernels beside the small ones (general_task in sdp_kernels.hip), unless the small ones are switched off
    // (not at level 0, where nearly every invocation is a general one: a wavefront each balances better than a list)
    const bool orderBeside = level > 0 && !getenv("BLASR_HIP_NO_SMALL4") && !getenv("BLASR_HIP_ORDER_SERIAL");
    if (orderBeside) {
      lv.leftList = arena.get<int32_t>(lv.n + 1);
      lv.leftCount = lv.leftList + lv.n;
      lv.bigGenList = arena.get<int32_t>(lv.n + 1);
      lv.bigGenCount = lv.bigGenList + lv.n;
    }
    lv.scratchA = arena.get<int32_t>(nf);
    lv.scratchB = arena.get<int32_t>(3 * nf);
    ARENA_CHECK();
    launch_sdp_match_all(lv, b->seq, true, st, matchSerial ? nullptr : ctx->auxStream, matchSerial ? nullptr : ctx->auxEvent);
    {  // invocations with <= 64 fragments: order .. chain in one kernel; those with <= 16 four per wavefront, by size
      int32_t *smallOrder = nullptr;
      if (!getenv("BLASR_HIP_NO_SMALL4")) {
        uint32_t *sk = arena.get<uint32_t>(nTasks), *skOut = arena.get<uint32_t>(nTasks);
        int32_t *sv = arena.get<int32_t>(nTasks);
        smallOrder = arena.get<int32_t>(nTasks);
        ARENA_CHECK();
        if (orderBeside) HIP_TRY(ctx, hipMemsetAsync(lv.bigGenCount, 0, sizeof(int32_t), st));
        launch_sdp_small4_keys(lv, sk, sv, st);
        size_t sortBytes = 0;
        HIP_TRY(ctx, rocprim::radix_sort_pairs_desc(nullptr, sortBytes, sk, skOut, sv, smallOrder, (size_t)nTasks, 0u, 5u, st));
        void *sortTmp = arena.get<char>(sortBytes);
        ARENA_CHECK();
        HIP_TRY(ctx, rocprim::radix_sort_pairs_desc(sortTmp, sortBytes, sk, skOut, sv, smallOrder, (size_t)nTasks, 0u, 5u, st));
      }
      if (orderBeside) {
        // the invocations with more than 64 raw fragments are never fused: their order .. above_predcl run on a side stream
        // while the small kernels work through the rest
        HIP_TRY(ctx, hipMemsetAsync(lv.leftCount, 0, sizeof(int32_t), st));
        (void)hipEventRecord(ctx->auxEvent[0], st);
        (void)hipStreamWaitEvent(ctx->auxStream[0], ctx->auxEvent[0], 0);
        SdpLevelArrays big = lv;
        big.generalMode = 2;
        big.leftList = lv.bigGenList;
        big.leftCount = lv.bigGenCount;
        launch_sdp_order(big, ctx->auxStream[0]);
        launch_sdp_yx_keys(big, ctx->auxStream[0]);
        launch_sdp_sort_yx(big, yxKeysOut, yxIdxOut, ctx->auxStream[0]);
        launch_sdp_above_predcl(big, ctx->auxStream[0]);
        (void)hipEventRecord(ctx->auxEvent[1], ctx->auxStream[0]);
      }
      launch_sdp_small(lv, prm, smallOrder, st);
      if (orderBeside) {
        (void)hipStreamWaitEvent(st, ctx->auxEvent[1], 0);
        lv.generalMode = 2;  // what the small kernels left over (a tie among 17 .. 64 fragments: rare)
      }
    }
    if (lv.generalMode == 0 && lv.n >= 4096 && !getenv("BLASR_HIP_TIES_SERIAL") && ctx->auxStream[1] && ctx->auxEvent[0] && ctx->auxEvent[2]) {
      // A level whose invocations are all general ones (the top level: windows of kilobases): half of configs[2]'s have a
      // fragment in two of their three lists -- an alignment that starts in the window's first bases -- and the replay of
      // std::sort's schedule for them (one wavefront each, mostly one lane of it: 8 ms at a twentieth of the issue slots) ran
      // alone on the device, before the (y, x) sort and the static predecessors of ALL invocations.  The invocations
      // sdp_order_kernel orders itself go on at once (its list), the others follow the replay on another stream.
      lv.cleanList = arena.get<int32_t>(lv.n + 1);
      ARENA_CHECK();
      lv.cleanCount = lv.cleanList + lv.n;
      launch_sdp_order(lv, st, false);
      hipStream_t sTies = ctx->auxStream[1];
      (void)hipEventRecord(ctx->auxEvent[0], st);
      (void)hipStreamWaitEvent(sTies, ctx->auxEvent[0], 0);
      SdpLevelArrays part = lv;
      part.generalMode = 2;
      part.leftList = lv.tieList;
      part.leftCount = lv.tieCount;
      launch_sdp_order_ties(lv, sTies);
      launch_sdp_yx_keys(part, sTies);
      launch_sdp_sort_yx(part, yxKeysOut, yxIdxOut, sTies);
      launch_sdp_above_predcl(part, sTies);
      (void)hipEventRecord(ctx->auxEvent[2], sTies);
      part.leftList = lv.cleanList;
      part.leftCount = lv.cleanCount;
      launch_sdp_yx_keys(part, st);
      launch_sdp_sort_yx(part, yxKeysOut, yxIdxOut, st);
      launch_sdp_above_predcl(part, st);
      (void)hipStreamWaitEvent(st, ctx->auxEvent[2], 0);
      lv.cleanList = nullptr;
      lv.cleanCount = nullptr;
    } else {
      launch_sdp_order(lv, st);
      launch_sdp_yx_keys(lv, st);
      launch_sdp_sort_yx(lv, yxKeysOut, yxIdxOut, st);
      launch_sdp_above_predcl(lv, st);
    }
    lv.generalMode = 0;
    launch_sdp_kernel_sums(lv, st);
    tock(1);

    // ---- chain ----
    tick();
    if (lv.chainList != nullptr) {
      // longest invocations first, the four of a wavefront of about the same size
      uint32_t *ck = arena.get<uint32_t>(nTasks), *ckOut = arena.get<uint32_t>(nTasks);
      int32_t *cv = arena.get<int32_t>(nTasks);
      ARENA_CHECK();
      launch_sdp_chain_list(lv, ck, cv, st);
      size_t sortBytes = 0;
      HIP_TRY(ctx, rocprim::radix_sort_pairs_desc(nullptr, sortBytes, ck, ckOut, cv, lv.chainList, (size_t)nTasks, 0u, 32u, st));
      void *sortTmp = arena.get<char>(sortBytes);
      ARENA_CHECK();
      HIP_TRY(ctx, rocprim::radix_sort_pairs_desc(sortTmp, sortBytes, ck, ckOut, cv, lv.chainList, (size_t)nTasks, 0u, 32u, st));
    }
    int32_t chainCounts[kChainCats] = {};
    const int32_t *countsForGrids = nullptr;
    if (lv.chainList != nullptr && !getenv("BLASR_HIP_CHAIN_FULL_GRIDS")) {  // (one small copy and a wait: the kernels' grids)
      HIP_TRY(ctx, hipMemcpyAsync(chainCounts, lv.chainCount, sizeof(chainCounts), hipMemcpyDeviceToHost, st));
      HIP_TRY(ctx, hipStreamSynchronize(st));
      countsForGrids = chainCounts;
    }
    if (getenv("BLASR_HIP_CHAIN_SERIAL")) launch_sdp_chain(lv, prm, st, nullptr, nullptr, countsForGrids);
    else launch_sdp_chain(lv, prm, st, ctx->auxStream, ctx->auxEvent, countsForGrids);
    tock(2);
    if (getenv("BLASR_HIP_SDP_TRACE") && lv.chainList != nullptr) {  // the sweep's categories and its largest invocations
      int32_t cnt[kChainCats] = {};
      HIP_TRY(ctx, hipMemcpy(cnt, lv.chainCount, sizeof(cnt), hipMemcpyDeviceToHost));
      fprintf(stderr, "[sdp] level %d sweep: chain4 %d, wide %d, bits4 %d + %d with the larger map, bits %d, bits-mid %d, bits-large %d, rows %d;",
              level, cnt[0], cnt[1], cnt[2], cnt[7], cnt[3], cnt[4], cnt[5], cnt[6]);
      int at = 0;
      for (int c = kChainCats - 1; c >= 0; c--) {
        if (cnt[c] > 0) {
          int32_t first = 0, nf = 0;
          HIP_TRY(ctx, hipMemcpy(&first, lv.chainList + at, sizeof(int32_t), hipMemcpyDeviceToHost));
          HIP_TRY(ctx, hipMemcpy(&nf, lv.nFrag + first, sizeof(int32_t), hipMemcpyDeviceToHost));
          fprintf(stderr, " largest of category %d: %d fragments;", c, nf);
        }
        at += cnt[c];
      }
      fprintf(stderr, "\n");
      if (atoi(getenv("BLASR_HIP_SDP_TRACE")) >= 1) {  // where the fragments are: per category, invocations and fragments by size
        int total = 0;
        for (int c = 0; c < kChainCats; c++) total += cnt[c];
        std::vector<int32_t> list((size_t)std::max(total, 1)), nf((size_t)nTasks);
        HIP_TRY(ctx, hipMemcpy(list.data(), lv.chainList, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(nf.data(), lv.nFrag, sizeof(int32_t) * (size_t)nTasks, hipMemcpyDeviceToHost));
        std::vector<SdpTask> hostTasks((size_t)nTasks);
        HIP_TRY(ctx, hipMemcpy(hostTasks.data(), dTasks, sizeof(SdpTask) * (size_t)nTasks, hipMemcpyDeviceToHost));
        int pos = 0;
        for (int c = kChainCats - 1; c >= 0; c--) {
          long long inv[10] = {0}, fr[10] = {0};
          for (int i = 0; i < cnt[c]; i++) {
            const int f = nf[(size_t)list[(size_t)(pos + i)]];
            int b = 0;
            while (b < 9 && f > (64 << b)) b++;
            inv[b]++;
            fr[b] += f;
          }
          pos += cnt[c];
          if (cnt[c] == 0) continue;
          fprintf(stderr, "[sdp]   category %d by fragments (<=64,128,...,16384,more): ", c);
          for (int b = 0; b < 10; b++) fprintf(stderr, "%lld/%lldk ", inv[b], fr[b] / 1000);
          fprintf(stderr, "\n");
          // fragments per query row (the sweep scores a row's fragments against one state of the column set): fragments of
          // the category by their invocation's nFrag / qLen
          long long dens[7] = {0};
          for (int i = 0; i < cnt[c]; i++) {
            const int ti = list[(size_t)(pos - cnt[c] + i)];
            const double d = (double)nf[(size_t)ti] / std::max(hostTasks[(size_t)ti].qLen, 1);
            int b = 0;
            while (b < 6 && d >= (0.5 * (1 << b))) b++;
            dens[b] += nf[(size_t)ti];
          }
          fprintf(stderr, "[sdp]   category %d fragments by fragments per row (<0.5,<1,<2,<4,<8,<16,more): ", c);
          for (int b = 0; b < 7; b++) fprintf(stderr, "%lldk ", dens[b] / 1000);
          fprintf(stderr, "\n");
        }
      }
    }

    // ---- blocks, gap-fill jobs, next level ----
    tick();
    // what every task will emit, then exclusive scans -> contention-free slots
    int32_t *counts = arena.get<int32_t>((size_t)4 * nTasks + 4);
    int64_t *offPool = arena.get<int64_t>(nTasks + 1), *offSw = arena.get<int64_t>(nTasks + 1),
            *offChild = arena.get<int64_t>(nTasks + 1);
    ARENA_CHECK();
    HIP_TRY(ctx, hipMemsetAsync(counts + (size_t)4 * nTasks, 0, 4 * sizeof(int32_t), st));
    launch_sdp_blocks_build(lv, prm, counts, st);
    {
      int64_t *offs[3] = {offPool, offSw, offChild};
      for (int c = 0; c < 3; c++) {
        auto in = rocprim::make_transform_iterator(
            rocprim::make_counting_iterator<size_t>(0),
            [counts, c] __device__(size_t i) { return (int64_t)counts[4 * i + c]; });
        size_t scanBytes = 0;
        HIP_TRY(ctx, rocprim::exclusive_scan(nullptr, scanBytes, in, offs[c], (int64_t)0, (size_t)nTasks + 1,
                                             rocprim::plus<int64_t>(), st));
        void *scanTmp = arena.get<char>(scanBytes);
        ARENA_CHECK();
        HIP_TRY(ctx, rocprim::exclusive_scan(scanTmp, scanBytes, in, offs[c], (int64_t)0, (size_t)nTasks + 1,
                                             rocprim::plus<int64_t>(), st));
      }
    }
    int64_t emitTotals[3] = {0, 0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(&emitTotals[0], offPool + nTasks, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(&emitTotals[1], offSw + nTasks, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(&emitTotals[2], offChild + nTasks, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    SdpEmit e;
    memset(&e, 0, sizeof(e));
    e.poolCap = emitTotals[0];
    e.swCap = emitTotals[1];
    e.childCap = emitTotals[2];
    // (what the level hands on outlives its arrays)
    e.poolKeys = arena.keep<uint64_t>((size_t)std::max<int64_t>(emitTotals[0], 1));
    e.poolVals = arena.keep<uint64_t>((size_t)std::max<int64_t>(emitTotals[0], 1));
    e.sw = arena.keep<SwTask>((size_t)std::max<int64_t>(emitTotals[1], 1));
    e.children = arena.keep<SdpTask>((size_t)std::max<int64_t>(emitTotals[2], 1));
    ARENA_CHECK();
    launch_sdp_blocks_emit(lv, prm, e, counts, offPool, offSw, offChild, st);
    HIP_TRY(ctx, hipStreamSynchronize(st));
    HIP_TRY(ctx, hipGetLastError());
    tock(3);
    spans.collect(kMs, kLaunches, kConcurrent, BLASR_HIP_SDP_KERNEL_KINDS);  // (every stream of the level was joined and waited for)
    Counters hc;
    memset(&hc, 0, sizeof(hc));
    hc.pool = (unsigned long long)emitTotals[0];
    hc.sw = (unsigned long long)emitTotals[1];
    hc.child = (unsigned long long)emitTotals[2];
    if (getenv("BLASR_HIP_SDP_TRACE")) {
      int32_t nTies = 0;  // invocations whose three fragment lists share an (x, y): sdp_order_ties_kernel's
      (void)hipMemcpy(&nTies, lv.tieCount, sizeof(int32_t), hipMemcpyDeviceToHost);
      fprintf(stderr, "[sdp] level %d: tasks %d keys %lld frags %lld -> blocks %llu sw %llu children %llu; %d invocations with ties\n", level,
              nTasks, (long long)totalKeys, (long long)nFragTotal, hc.pool, hc.sw, hc.child, nTies);
      fprintf(stderr, "[sdp] level %d: keys + sort %.1f ms, match + order %.1f, sweep %.1f, blocks %.1f\n", level, levelMs[0], levelMs[1],
              levelMs[2], levelMs[3]);
      for (float &m : levelMs) m = 0;
    }
    LevelOut lo;
    lo.poolKeys = e.poolKeys;
    lo.poolVals = e.poolVals;
    lo.poolCount = (int64_t)hc.pool;
    lo.sw = e.sw;
    lo.swCount = (int64_t)hc.sw;
    outs.push_back(lo);
    dTasks = e.children;
    nTasks = (int)hc.child;
    k = k - 4 > 5 ? k - 4 : 5;
    arena.release_level(levelMark);  // (the stream was synchronised above: nothing of the level is in flight)
  }

  // ---- gap fills (SWAlign) and assembly ----
  tick();
  int64_t poolTotal = 0, swTotal = 0;
  for (auto &lo : outs) {
    poolTotal += lo.poolCount;
    swTotal += lo.swCount;
  }
  // q*t < noRecurseUnder bounds the blocks of one fill by min(q,t) <= sqrt(noRecurseUnder)
  int64_t perFill = 1;
  while (perFill * perFill < std::max(sp->recurse_over, 1)) perFill++;
  // the pool's capacity: a fill aligns at most min(q, t) columns, summed over the fills on the device (the bound
  // swTotal * perFill is five times that on real batches, and these four arrays were the stage's high-water mark)
  int64_t fillBlocks = 0;
  {
    int64_t *dSum = arena.get<int64_t>(outs.size() + 1);
    ARENA_CHECK();
    size_t redBytes = 0;
    std::vector<int64_t> sums(outs.size(), 0);
    for (size_t i = 0; i < outs.size(); i++) {
      if (outs[i].swCount <= 0) continue;
      const SwTask *sw = outs[i].sw;
      auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<size_t>(0), [sw] __device__(size_t j) {
        return (int64_t)(sw[j].qLen < sw[j].tLen ? sw[j].qLen : sw[j].tLen);
      });
      size_t need = 0;
      HIP_TRY(ctx, rocprim::reduce(nullptr, need, in, dSum + i, (int64_t)0, (size_t)outs[i].swCount, rocprim::plus<int64_t>(), st));
      if (need > redBytes) redBytes = need;
    }
    void *redTmp = arena.get<char>(std::max<size_t>(redBytes, 256));
    ARENA_CHECK();
    for (size_t i = 0; i < outs.size(); i++) {
      if (outs[i].swCount <= 0) continue;
      const SwTask *sw = outs[i].sw;
      auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<size_t>(0), [sw] __device__(size_t j) {
        return (int64_t)(sw[j].qLen < sw[j].tLen ? sw[j].qLen : sw[j].tLen);
      });
      size_t need = redBytes;
      HIP_TRY(ctx, rocprim::reduce(redTmp, need, in, dSum + i, (int64_t)0, (size_t)outs[i].swCount, rocprim::plus<int64_t>(), st));
      HIP_TRY(ctx, hipMemcpyAsync(&sums[i], dSum + i, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(ctx, hipStreamSynchronize(st));
    for (int64_t v : sums) fillBlocks += v;
  }
  const int64_t poolCap = poolTotal + std::min(fillBlocks, swTotal * perFill) + 16;
  uint64_t *poolKeys = arena.get<uint64_t>((size_t)poolCap), *poolVals = arena.get<uint64_t>((size_t)poolCap);
  uint64_t *poolKeysSorted = arena.get<uint64_t>((size_t)poolCap), *poolValsSorted = arena.get<uint64_t>((size_t)poolCap);
  ARENA_CHECK();
  int64_t at = 0;
  for (auto &lo : outs) {
    if (lo.poolCount > 0) {
      HIP_TRY(ctx, hipMemcpyAsync(poolKeys + at, lo.poolKeys, sizeof(uint64_t) * lo.poolCount, hipMemcpyDeviceToDevice, st));
      HIP_TRY(ctx, hipMemcpyAsync(poolVals + at, lo.poolVals, sizeof(uint64_t) * lo.poolCount, hipMemcpyDeviceToDevice, st));
      at += lo.poolCount;
    }
  }
  Counters init;
  memset(&init, 0, sizeof(init));
  init.pool = (unsigned long long)at;
  HIP_TRY(ctx, hipMemcpyAsync(dCounters, &init, sizeof(Counters), hipMemcpyHostToDevice, st));
  if (swTotal > 0) {
    // scratch per lane: two score rows of (t+1) ints + (q+1)(t+1) arrow bytes, q*t < noRecurseUnder
    const int64_t nru = std::max(sp->recurse_over, 1);
    const int64_t parkOff = ((2 * (nru + 1) * 4 + (2 * nru + 2) + 255) / 256) * 256;
    const int64_t scratchPerLane = parkOff + ((perFill * 16 + 255) / 256) * 256;  // + parked Diagonal runs
    int64_t swMax = 0;
    for (auto &lo : outs) swMax = std::max<int64_t>(swMax, lo.swCount);
    // fills with both sides <= 63 run with their score row in registers, sorted by shape; the cell keys
    // carry the arrow in their low bits, so the scores must leave room for that
    int64_t magnitude = std::max<int64_t>(std::abs((int64_t)score->ins), std::abs((int64_t)score->del));
    for (int i = 0; i < 25; i++) magnitude = std::max<int64_t>(magnitude, std::abs((int64_t)score->matrix[i]));
    const bool smallOk = magnitude < (1 << 20);
    // (workgroups of sw_kernel, a lane per fill with its matrix in global scratch: 128 left half of the CUs without one;
    //  at configs[2] the stage's fills + assembly take 78.7 / 73.8 / 72.3 / 72.9 ms with 128 / 256 / 512 / 1024)
    const int bigBlocks = getenv("BLASR_HIP_SW_BLOCKS") ? std::max(1, atoi(getenv("BLASR_HIP_SW_BLOCKS"))) : 512;
    const int64_t lanes = (int64_t)bigBlocks * 256;
    uint8_t *scratch = arena.get<uint8_t>((size_t)(scratchPerLane * lanes));
    uint32_t *smallScratch = arena.get<uint32_t>((size_t)sw_small_scratch_words(sw_small_blocks(swMax)));
    uint32_t *swKeys = arena.get<uint32_t>((size_t)swMax), *swKeysSorted = arena.get<uint32_t>((size_t)swMax);
    uint32_t *swIdx = arena.get<uint32_t>((size_t)swMax), *swOrder = arena.get<uint32_t>((size_t)swMax);
    int32_t *nBig = arena.get<int32_t>(64);
    size_t swSortBytes = 0;
    HIP_TRY(ctx, rocprim::radix_sort_pairs(nullptr, swSortBytes, swKeys, swKeysSorted, swIdx, swOrder, (size_t)swMax, 0u,
                                           13u, st));
    void *swSortTmp = arena.get<char>(swSortBytes);
    ARENA_CHECK();
    SdpEmit e;
    memset(&e, 0, sizeof(e));
    e.poolKeys = poolKeys;
    e.poolVals = poolVals;
    e.poolCount = &dCounters->pool;
    e.poolCap = poolCap;
    e.overflow = &dCounters->overflow;
    int level = 0;
    for (auto &lo : outs) {
      if (lo.swCount <= 0) continue;
      if (lo.swCount >= (1ll << 32)) return blasr_hip_fail(ctx, BLASR_HIP_ERR_UNSUPPORTED, "batch too large: %lld fills", (long long)lo.swCount);
      int32_t *nBigL = nBig + (level++ & 63);
      HIP_TRY(ctx, hipMemsetAsync(nBigL, 0, sizeof(int32_t), st));
      launch_sw_sort_keys(lo.sw, lo.swCount, smallOk, swKeys, swIdx, nBigL, st);
      size_t bytes = swSortBytes;
      HIP_TRY(ctx, rocprim::radix_sort_pairs(swSortTmp, bytes, swKeys, swKeysSorted, swIdx, swOrder, (size_t)lo.swCount,
                                             0u, 13u, st));
      spans.begin(BLASR_HIP_SDP_KERNEL_FILLS, st);
      launch_sw_small(lo.sw, swOrder, lo.swCount, nBigL, b->seq, dMatrix, prm, smallScratch, e, st);
      launch_sw(lo.sw, swOrder, lo.swCount, nBigL, bigBlocks, b->seq, dMatrix, prm, scratch, scratchPerLane, parkOff, e, st);
      spans.end(st);
    }
  }
  Counters hc;
  HIP_TRY(ctx, hipMemcpyAsync(&hc, dCounters, sizeof(Counters), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  HIP_TRY(ctx, hipGetLastError());
  const int64_t nPool = (int64_t)hc.pool;
  if (getenv("BLASR_HIP_SDP_TRACE"))
    fprintf(stderr, "[sdp] assembled: chain blocks %lld, sw fills %lld, pool %lld\n", (long long)poolTotal,
            (long long)swTotal, (long long)nPool);
  if (hc.overflow || nPool > poolCap) {
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_HIP, "internal: block pool overflow");
  }
  int64_t *blkOff = arena.get<int64_t>(n + 1);
  ARENA_CHECK();
  const uint64_t *sortedKeys = poolKeys, *sortedVals = poolVals;
  if (nPool > 0) {
    if (nPool >= (1ll << 32)) {
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_UNSUPPORTED, "batch too large: %lld blocks", (long long)nPool);
    }
    // key = candidate << 32 | q: the bits between the longest query's and bit 32 are zero everywhere, so two stable
    // sorts (by q, then by candidate) pass over the pool fewer times than one over bits 0 .. 32 + candidate bits
    const unsigned qBits = (unsigned)bits_for((uint64_t)maxQLen), endBit = (unsigned)(32 + bits_for((uint64_t)n));
    const bool twoSorts = (qBits + 7) / 8 + (endBit - 32 + 7) / 8 < (endBit + 7) / 8 && !getenv("BLASR_HIP_POOL_ONE_SORT");
    size_t sortBytes = 0;
    HIP_TRY(ctx, rocprim::radix_sort_pairs(nullptr, sortBytes, poolKeys, poolKeysSorted, poolVals, poolValsSorted,
                                           (size_t)nPool, 0u, endBit, st));
    void *sortTmp = arena.get<char>(sortBytes);
    ARENA_CHECK();
    if (twoSorts) {
      size_t bytes = sortBytes;
      HIP_TRY(ctx, rocprim::radix_sort_pairs(sortTmp, bytes, poolKeys, poolKeysSorted, poolVals, poolValsSorted,
                                             (size_t)nPool, 0u, qBits, st));
      bytes = sortBytes;
      HIP_TRY(ctx, rocprim::radix_sort_pairs(sortTmp, bytes, poolKeysSorted, poolKeys, poolValsSorted, poolVals,
                                             (size_t)nPool, 32u, endBit, st));
    } else {
      HIP_TRY(ctx, rocprim::radix_sort_pairs(sortTmp, sortBytes, poolKeys, poolKeysSorted, poolVals, poolValsSorted,
                                             (size_t)nPool, 0u, endBit, st));
      sortedKeys = poolKeysSorted;
      sortedVals = poolValsSorted;
    }
  }
  // blkOff[c] = first block of candidate c in the sorted pool (the pool is ordered by candidate)
  launch_pool_offsets(sortedKeys, nPool, n, blkOff, st);
  launch_sdp_finish(n, dTop, b->seq, sortedKeys, sortedVals, blkOff, dMatrix, score->ins, score->del,
                    score->affine_open, score->affine_extend, reinterpret_cast<SdpResult *>(results), blocks, blocksCap,
                    blkBase, st);
  HIP_TRY(ctx, hipStreamSynchronize(st));
  HIP_TRY(ctx, hipGetLastError());
  tock(3);
  phaseBytes[3] += nPool * 12;
  {  // the heavy kernels one by one: times from the brackets, what they processed from the device's sums
    spans.collect(kMs, kLaunches, kConcurrent, BLASR_HIP_SDP_KERNEL_KINDS);
    unsigned long long sums[3 * BLASR_HIP_SDP_KERNEL_KINDS];
    HIP_TRY(ctx, hipMemcpy(sums, dKernelSums, sizeof(sums), hipMemcpyDeviceToHost));
    for (int kd = 0; kd < BLASR_HIP_SDP_KERNEL_KINDS; kd++) {
      blasr_hip_sdp_kernel_record &r = ctx->sdpKernelRec[kd];
      r.kind = kd;
      r.launches += kLaunches[kd];
      r.ms += kMs[kd];
      r.concurrent = std::max(r.concurrent, kConcurrent[kd]);
      int64_t fr = (int64_t)sums[3 * kd], inv = (int64_t)sums[3 * kd + 1], qp = (int64_t)sums[3 * kd + 2], bytes = 0;
      if (kd == BLASR_HIP_SDP_KERNEL_FILLS) {  // a fill reads its two sides and writes at most min(q, t) blocks of 12 B
        fr = inv = swTotal;
        bytes = fillBlocks * (2 + 12);
      } else if (kd == BLASR_HIP_SDP_KERNEL_MATCH_INDEX || kd == BLASR_HIP_SDP_KERNEL_MATCH_SHORT) {
        bytes = qp * 8 + fr * 16;  // (8 B per query position, 16 B per fragment written)
      } else if (kd == BLASR_HIP_SDP_KERNEL_SMALL) {
        bytes = fr * 16 * 2;       // fragments ordered and swept in one kernel
      } else {
        bytes = fr * 16;           // 16 B per fragment read by the sweep
      }
      r.fragments += fr;
      r.invocations += inv;
      r.algorithmic_bytes += bytes;
    }
  }
#ifdef BLASR_IDX_CLOCKS
  sdp_print_idx_clocks();
#endif
#ifdef BLASR_ROWS_CLOCKS
  sdp_print_rows_clocks();
#endif
  for (int i = 0; i < 4; i++) ctx->sdpMs[i] = phaseMs[i];
  for (int i = 0; i < 4; i++) ctx->sdpBytes[i] = phaseBytes[i];
  ctx->sdpCounters[0] = totalInvocations;
  ctx->sdpCounters[1] = totalFrags;
  ctx->sdpCounters[2] = swTotal;
  if (blocksUsed) *blocksUsed = nPool;
  if (nPool > blocksCap)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_CAPACITY, "blocks arena too small: need %lld more, have %lld",
                          (long long)nPool, (long long)blocksCap);
  return BLASR_HIP_OK;
}

}  // namespace

extern "C" {

int blasr_hip_sdp_device(blasr_hip_ctx *ctx, const blasr_hip_sdp_batch *b, const blasr_hip_sdp_params *sp,
                         const blasr_hip_score_params *score, blasr_hip_sdp_result *results, int32_t *blocks,
                         int64_t blocksCap, int64_t *blocksUsed) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (!b || !sp || !score || b->n < 0 || blocksCap < 0) return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "bad argument");
  if (blocksUsed) *blocksUsed = 0;
  for (int i = 0; i < 4; i++) ctx->sdpMs[i] = 0;
  for (int i = 0; i < 3; i++) ctx->sdpCounters[i] = 0;
  for (int i = 0; i < 4; i++) ctx->sdpBytes[i] = 0;
  for (auto &r : ctx->sdpKernelRec) r = blasr_hip_sdp_kernel_record();
  const int n = b->n;
  if (n == 0) return BLASR_HIP_OK;
  if (!b->seq || !b->q_off || !b->q_len || !b->t_off || !b->t_len || !results || (!blocks && blocksCap > 0))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "null pointer in batch");
  if ((sp->align_type != 0 && sp->align_type != 1) || sp->word_size < 1 || sp->word_size > 32 || sp->recurse < 0 || sp->prefix_length < 0 || sp->sdp_ins <= -64 ||
      sp->sdp_ins >= 64 || sp->sdp_del <= -64 || sp->sdp_del >= 64)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_UNSUPPORTED,
                          "SDP parameters outside the device domain (align_type Local/Global, 1 <= word_size <= 32 -- a tuple is 64 bits, "
                          "tuples/DNATuple.h:55-94 --, "
                          "|sdp_ins|, |sdp_del| < 64)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // ---- chunks.  The level arrays take ~35 bytes per base of query + target (fragments, keys, the sweep's arrays,
  // the block pool); a batch that would not fit what the device can give this context is taken in pieces.  Tasks
  // are independent, so the pieces only share the caller's block arena.
  std::vector<int32_t> qLen((size_t)n), tLen((size_t)n);
  HIP_TRY(ctx, hipMemcpyAsync(qLen.data(), b->q_len, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(tLen.data(), b->t_len, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  // Pieces: as many candidates as the memory the context may use (the caller's limit, what the device can still
  // give) holds at the bytes per base the level arrays took so far -- ~48 for alignments, ten times that where
  // chance clusters make the recursion flood (configs[4]); the figure follows what the pieces actually used.  A piece
  // that does not fit after all (the device refuses an allocation) is taken again as two halves.
  const char *forced = getenv("BLASR_HIP_SDP_CHUNK_BASES");
  float ms[4] = {0, 0, 0, 0};
  int64_t counters[3] = {0, 0, 0}, used = 0, bytes[4] = {0, 0, 0, 0};
  bool overflow = false;
  int first = 0;
  int limitCount = n;  // a piece that failed caps the next attempt at half its candidates
  int64_t largestAcc = 0;
  double largestPerBase = 0;
  while (first < n) {
    int64_t budgetBases = std::max<int64_t>((int64_t)(blasr_hip_workspace_reach(ctx) / std::max(ctx->sdpBytesPerBase, 16.0)), 1 << 16);
    if (forced) budgetBases = std::max<int64_t>(atoll(forced), 1);
    else {  // what is left, in pieces of about the same size: every piece pays the levels' tails once
      int64_t remaining = 0;
      for (int i = first; i < n; i++) remaining += (int64_t)std::max(qLen[i], 0) + std::max(tLen[i], 0);
      const int64_t piecesLeft = std::max<int64_t>((remaining + budgetBases - 1) / budgetBases, 1);
      if (piecesLeft > 1) budgetBases = std::min(budgetBases, remaining / piecesLeft + remaining / (piecesLeft * 32) + 1);
    }
    int cnt = 0;
    int64_t acc = 0;
    while (first + cnt < n && cnt < limitCount) {
      const int64_t bases = (int64_t)std::max(qLen[first + cnt], 0) + std::max(tLen[first + cnt], 0);
      if (cnt > 0 && acc + bases > budgetBases) break;
      acc += bases;
      cnt++;
    }
    blasr_hip_sdp_batch sub = *b;
    sub.n = cnt;
    sub.q_off = b->q_off + first;
    sub.q_len = b->q_len + first;
    sub.t_off = b->t_off + first;
    sub.t_len = b->t_len + first;
    const int64_t left = std::max<int64_t>(blocksCap - used, 0);
    int64_t usedHere = 0, arenaHigh = 0;
    const int rc = sdp_device_chunk(ctx, &sub, sp, score, results + first, blocks ? blocks + 3 * std::min(used, blocksCap) : nullptr,
                                    left, &usedHere, used, &arenaHigh);
    if (getenv("BLASR_HIP_SDP_TRACE"))
      fprintf(stderr, "[sdp] piece of %d candidates, %lld bases: arena high-water %.2f GB (%.1f B per base), planned with %.1f B per base, rc %d\n", cnt,
              (long long)acc, (double)arenaHigh / 1e9, acc > 0 ? (double)arenaHigh / (double)acc : 0.0, std::max(ctx->sdpBytesPerBase, 16.0), rc);
    if (acc > 0 && arenaHigh > 0 && !forced) {  // (an aborted piece reports what it had asked for so far: a lower bound)
      ctx->sdpBytesPerBase = std::max(ctx->sdpBytesPerBase * (rc == BLASR_HIP_OK ? 0.5 : 1.0), 1.25 * (double)arenaHigh / (double)acc);
      // what the NEXT call plans with is the figure of this call's largest piece: a small last piece carries the
      // stage's fixed costs on few bases and would make the next call cut a batch in two that fits in one
      if (rc == BLASR_HIP_OK && acc >= largestAcc) {
        largestAcc = acc;
        largestPerBase = ctx->sdpBytesPerBase;
      }
    }
    if (rc == BLASR_HIP_ERR_NOMEM && cnt > 1) {
      limitCount = cnt / 2;
      ctx->sdpBytesPerBase *= 1.5;
      if (getenv("BLASR_HIP_SDP_TRACE"))
        fprintf(stderr, "[sdp] a piece of %d candidates did not fit the device memory: again in halves\n", cnt);
      continue;
    }
    if (rc == BLASR_HIP_ERR_CAPACITY) overflow = true;
    else if (rc != BLASR_HIP_OK) return rc;
    limitCount = n;
    first += cnt;
    used += usedHere;
    for (int i = 0; i < 4; i++) ms[i] += ctx->sdpMs[i];
    for (int i = 0; i < 3; i++) counters[i] += ctx->sdpCounters[i];
    for (int i = 0; i < 4; i++) bytes[i] += ctx->sdpBytes[i];
  }
  if (largestPerBase > 0) ctx->sdpBytesPerBase = largestPerBase;
  for (int i = 0; i < 4; i++) ctx->sdpBytes[i] = bytes[i];
  for (int i = 0; i < 4; i++) ctx->sdpMs[i] = ms[i];
  for (int i = 0; i < 3; i++) ctx->sdpCounters[i] = counters[i];
  if (blocksUsed) *blocksUsed = used;
  if (overflow)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_CAPACITY, "blocks arena too small: need %lld, have %lld", (long long)used,
                          (long long)blocksCap);
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CTX(ctx)

int blasr_hip_sdp_host(blasr_hip_ctx *ctx, const blasr_hip_sdp_batch *b, const blasr_hip_sdp_params *sp,
                       const blasr_hip_score_params *score, blasr_hip_sdp_result *results, int32_t *blocks,
                       int64_t blocksCap, int64_t *blocksUsed) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (!b || !sp || !score || b->n < 0) return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "bad argument");
  if (blocksUsed) *blocksUsed = 0;
  const int n = b->n;
  if (n == 0) return BLASR_HIP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int64_t seqBytes = 0;
  for (int i = 0; i < n; i++) {
    if (b->q_len[i] < 0 || b->t_len[i] < 0 || b->q_off[i] < 0 || b->t_off[i] < 0)
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "negative length or offset in candidate %d", i);
    seqBytes = std::max<int64_t>(seqBytes, std::max(b->q_off[i] + b->q_len[i], b->t_off[i] + b->t_len[i]));
  }
  struct Dev {
    void *p = nullptr;
    ~Dev() {
      if (p) (void)hipFree(p);
    }
  };
  Dev dSeq, dQOff, dQLen, dTOff, dTLen, dRes, dBlocks;
  auto up = [&](Dev &d, const void *src, size_t bytes) -> hipError_t {
    hipError_t e = hipMalloc(&d.p, std::max<size_t>(bytes, 16));
    if (e != hipSuccess) return e;
    if (bytes) e = hipMemcpy(d.p, src, bytes, hipMemcpyHostToDevice);
    return e;
  };
  HIP_TRY(ctx, up(dSeq, b->seq, (size_t)seqBytes));
  HIP_TRY(ctx, up(dQOff, b->q_off, sizeof(int64_t) * n));
  HIP_TRY(ctx, up(dQLen, b->q_len, sizeof(int32_t) * n));
  HIP_TRY(ctx, up(dTOff, b->t_off, sizeof(int64_t) * n));
  HIP_TRY(ctx, up(dTLen, b->t_len, sizeof(int32_t) * n));
  HIP_TRY(ctx, hipMalloc(&dRes.p, sizeof(blasr_hip_sdp_result) * n));
  HIP_TRY(ctx, hipMalloc(&dBlocks.p, std::max<size_t>(sizeof(int32_t) * 3 * (size_t)blocksCap, 16)));
  blasr_hip_sdp_batch db;
  db.n = n;
  db.seq = (const uint8_t *)dSeq.p;
  db.q_off = (const int64_t *)dQOff.p;
  db.q_len = (const int32_t *)dQLen.p;
  db.t_off = (const int64_t *)dTOff.p;
  db.t_len = (const int32_t *)dTLen.p;
  int64_t used = 0;
  int rc = blasr_hip_sdp_device(ctx, &db, sp, score, (blasr_hip_sdp_result *)dRes.p, (int32_t *)dBlocks.p, blocksCap,
                                &used);
  if (blocksUsed) *blocksUsed = used;
  if (rc != BLASR_HIP_OK && rc != BLASR_HIP_ERR_CAPACITY) return rc;
  HIP_TRY(ctx, hipMemcpy(results, dRes.p, sizeof(blasr_hip_sdp_result) * n, hipMemcpyDeviceToHost));
  const int64_t copyBlocks = std::min(used, blocksCap);
  if (copyBlocks > 0)
    HIP_TRY(ctx, hipMemcpy(blocks, dBlocks.p, sizeof(int32_t) * 3 * (size_t)copyBlocks, hipMemcpyDeviceToHost));
  return rc;
} BLASR_HIP_CATCH_CTX(ctx)

}  // extern "C"
