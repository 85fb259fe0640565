// The whole of MapRead for a block of reads, from the reads alone (alignment/Blasr.cpp:2247-2644): the anchoring
// (anchor_api.hip), the interval finding (intervals_api.hip) and the path (mapread_api.hip) strung together in C++.
//
// The reads go through the anchoring in sub-blocks, and since round 3 a sub-block's anchors never leave the device:
// blasr_hip_find_intervals_device (intervals_device.hip) runs RemoveOverlappingAnchors, the window clustering and
// GlobalChain as kernels right behind the anchor search, one 32-byte record per cluster of anchors comes back and the
// host ends with the p-values and the interval sets.  The path then runs once on all the intervals found.
// (BLASR_HIP_INTERVALS_HOST=1 keeps round 2's route for comparison: while the device searches sub-block b + 1 the anchors
// of sub-block b cross PCIe on the side stream into pinned memory and a host thread, which fans out over the reads, turns
// them into candidate intervals with blasr_hip_find_intervals.)
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/blasr_hip.h"
#include "ctx.h"

namespace {

struct Job {
  int32_t first, count;     // reads [first, first + count)
  int64_t anchorOff;        // first anchor of the sub-block in the pinned buffer
  const blasr_hip_anchor *pinned;  // the pinned buffer as it was when the job was queued
  std::vector<int64_t> listOff;
  hipEvent_t copied;
  int copyError;            // a copy of the sub-block's anchors failed
};

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// what every sub-block of a call shares
struct PipeCall {
  const uint8_t *genomeHost;
  const uint32_t *seqStartPos;
  int32_t nSeqPos;
  const int32_t *countTable;
  int32_t countTupleSize;
  const blasr_hip_read *readsHost;
  const uint8_t *readPoolDevice;
  const blasr_hip_anchor_params *anchor;
  const blasr_hip_interval_params *interval;
  bool wantChains;
};

// a sub-block's candidate intervals (read indices relative to the sub-block) and, for -rbao, their chains
struct BlockOut {
  std::vector<blasr_hip_interval> ivs;
  std::vector<uint32_t> chains;    // (t, q, l) triplets, back to back
  std::vector<int64_t> chainOff;   // one entry per interval, + 1
  std::vector<int64_t> listOff, ivOff;
  int64_t anchors = 0;
  double ivMs = 0;                 // the interval finding (kernels + the host's end)
};

// MapReadToGenome x 2 + SortMatchPosList, then the interval finder, for reads [first, first + count) on context c: the
// anchors stay in c's buffer, the intervals come back
int anchor_block(blasr_hip_ctx *c, const PipeCall &k, int32_t first, int32_t count, BlockOut &o) {
  o.listOff.assign(2 * (size_t)count + 1, 0);
  int64_t blockBases = 0;
  for (int32_t r = first; r < first + count; r++) blockBases += k.readsHost[r].length;
  int64_t expect = 2 * blockBases + 4096, n = 0;
  int rcA = BLASR_HIP_OK;
  for (int attempt = 0; attempt < 2; attempt++) {
    if ((rcA = blasr_hip_ensure(c, c->pipeAnchors, sizeof(blasr_hip_anchor) * (size_t)expect)) != BLASR_HIP_OK) return rcA;
    rcA = blasr_hip_anchor_reads_device(c, count, k.readsHost + first, k.readPoolDevice, k.anchor,
                                        (blasr_hip_anchor *)c->pipeAnchors.p,
                                        (int64_t)(c->pipeAnchors.bytes / sizeof(blasr_hip_anchor)), &n, o.listOff.data());
    if (rcA != BLASR_HIP_ERR_CAPACITY) break;
    expect = n + n / 8 + 1;  // (the count a short call returns)
  }
  if (rcA != BLASR_HIP_OK) return rcA;
  o.anchors = n;
  const double h0 = now_ms();
  o.ivs.resize((size_t)count * (size_t)std::max(k.interval->n_candidates, 1) + 1);
  o.ivOff.assign((size_t)count + 1, 0);
  o.chains.clear();
  o.chainOff.clear();
  int64_t nIv = 0;
  int rcI;
  if (k.wantChains) {
    int64_t chainsUsed = 0;
    o.chains.resize(3 * (size_t)std::max<int64_t>(n, 1));  // (grown below when ten intervals of a read share anchors)
    o.chainOff.assign(o.ivs.size() + 1, 0);
    for (int attempt = 0; attempt < 2; attempt++) {
      rcI = blasr_hip_find_intervals_device_chains(c, k.genomeHost, c->genomeLen, k.seqStartPos, k.nSeqPos, k.countTable,
                                                   k.countTupleSize, count, k.readsHost + first,
                                                   (const blasr_hip_anchor *)c->pipeAnchors.p, o.listOff.data(), k.interval,
                                                   o.ivs.data(), nullptr, (int64_t)o.ivs.size(), &nIv, o.ivOff.data(),
                                                   o.chains.data(), (int64_t)(o.chains.size() / 3), &chainsUsed,
                                                   o.chainOff.data());
      if (rcI != BLASR_HIP_ERR_CAPACITY || nIv > (int64_t)o.ivs.size()) break;
      o.chains.resize(3 * (size_t)chainsUsed);
    }
    if (rcI == BLASR_HIP_OK) {
      o.chains.resize(3 * (size_t)chainsUsed);
      o.chainOff.resize((size_t)nIv + 1);
    }
  } else {
    rcI = blasr_hip_find_intervals_device(c, k.genomeHost, c->genomeLen, k.seqStartPos, k.nSeqPos, k.countTable,
                                          k.countTupleSize, count, k.readsHost + first,
                                          (const blasr_hip_anchor *)c->pipeAnchors.p, o.listOff.data(), k.interval,
                                          o.ivs.data(), nullptr, (int64_t)o.ivs.size(), &nIv, o.ivOff.data());
  }
  if (rcI != BLASR_HIP_OK) return rcI;
  o.ivs.resize((size_t)nIv);
  o.ivMs = now_ms() - h0;
  return BLASR_HIP_OK;
}

int map_from_reads_overlapped(blasr_hip_ctx *ctx, const PipeCall &call, int32_t perBlock, int32_t n_reads,
                              const blasr_hip_read *reads_device, const blasr_hip_map_params &map, blasr_hip_hit *hits,
                              int64_t hits_capacity, int64_t *n_hits, int64_t *read_hit_off, blasr_hip_map_fetch *fetch,
                              blasr_hip_interval *intervals_out, int64_t intervals_capacity, int64_t *n_intervals);

}  // namespace

extern "C" {

int blasr_hip_pipeline_default_params(blasr_hip_pipeline_params *p) try {
  if (!p) return BLASR_HIP_ERR_INVALID;
  blasr_hip_anchor_default_params(&p->anchor);
  blasr_hip_interval_default_params(&p->interval);
  blasr_hip_map_default_params(&p->map);
  p->reads_per_block = 100000;  // (round 2: 6250, sized for the host's interval finding to keep up with the device)
  p->overlap = 0;
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_pipeline_last_timing(const blasr_hip_ctx *ctx, float ms[4], int64_t counters[2]) try {
  if (!ctx || !ms || !counters) return BLASR_HIP_ERR_INVALID;
  for (int i = 0; i < 4; i++) ms[i] = ctx->pipeMs[i];
  counters[0] = ctx->pipeCounters[0];
  counters[1] = ctx->pipeCounters[1];
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_map_from_reads_device(blasr_hip_ctx *ctx, const uint8_t *genome_host, const uint32_t *seq_start_pos,
                                    int32_t n_seq_pos, const int32_t *count_table, int32_t count_tuple_size,
                                    int32_t n_reads, const blasr_hip_read *reads_host, const blasr_hip_read *reads_device,
                                    const uint8_t *read_pool_device, const blasr_hip_pipeline_params *params,
                                    blasr_hip_hit *hits, int64_t hits_capacity, int64_t *n_hits, int64_t *read_hit_off,
                                    blasr_hip_map_fetch *fetch, blasr_hip_interval *intervals_out,
                                    int64_t intervals_capacity, int64_t *n_intervals) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (!genome_host || !seq_start_pos || n_seq_pos < 1 || !count_table || n_reads < 0 || !params || !n_hits ||
      (n_reads > 0 && (!reads_host || !reads_device || !read_pool_device)) || (hits_capacity > 0 && !hits) ||
      (intervals_capacity > 0 && !intervals_out))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "map_from_reads: bad arguments");
  if (!ctx->genome || !ctx->saIndex)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "map_from_reads: no genome / suffix array on the device");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  blasr_hip_anchor_params ap = params->anchor;
  ap.sorted = 1;
  const int32_t perBlock = params->reads_per_block > 0 ? params->reads_per_block : 100000;
  const double t0 = now_ms();

  // -rbao reads the intervals' chained anchors: both interval finders hand them out on request
  const bool wantChains = params->map.refine_between_anchors_only != 0;
  if (!getenv("BLASR_HIP_INTERVALS_HOST")) {
    PipeCall call{genome_host, seq_start_pos, n_seq_pos, count_table, count_tuple_size, reads_host, read_pool_device, &ap,
                  &params->interval, wantChains};
    if (params->overlap != 0 && n_reads > perBlock && !getenv("BLASR_HIP_PIPE_SERIAL"))
      return map_from_reads_overlapped(ctx, call, perBlock, n_reads, reads_device, params->map, hits, hits_capacity, n_hits,
                                       read_hit_off, fetch, intervals_out, intervals_capacity, n_intervals);
    // ---- anchors and intervals of every sub-block on the device; the anchor buffer is the same for all of them
    std::vector<blasr_hip_interval> ivs;
    std::vector<uint32_t> allChains;     // (t, q, l) triplets of the intervals' chains, back to back
    std::vector<int64_t> allChainOff;    // one entry per interval, + 1
    BlockOut blk;
    int64_t anchorsInAll = 0;
    double ivMs = 0;
    for (int32_t first = 0; first < n_reads; first += perBlock) {
      const int32_t count = std::min(perBlock, n_reads - first);
      const int rcB = anchor_block(ctx, call, first, count, blk);
      if (rcB != BLASR_HIP_OK) return rcB;
      anchorsInAll += blk.anchors;
      ivMs += blk.ivMs;
      if (wantChains) {
        const int64_t shift = (int64_t)(allChains.size() / 3);
        for (size_t i = 0; i < blk.ivs.size(); i++) allChainOff.push_back(shift + blk.chainOff[i]);
        allChains.insert(allChains.end(), blk.chains.begin(), blk.chains.end());
      }
      for (auto iv : blk.ivs) {
        iv.read += first;
        ivs.push_back(iv);
      }
    }
    const double t1 = now_ms();
    const int64_t total = (int64_t)ivs.size();
    if (n_intervals) *n_intervals = total;
    if (intervals_out) memcpy(intervals_out, ivs.data(), sizeof(blasr_hip_interval) * (size_t)std::min(total, intervals_capacity));
    int rc = blasr_hip_ensure(ctx, ctx->pipeIntervals, sizeof(blasr_hip_interval) * (size_t)(total + 1));
    if (rc != BLASR_HIP_OK) return rc;
    HIP_TRY(ctx, hipMemcpy(ctx->pipeIntervals.p, ivs.data(), sizeof(blasr_hip_interval) * (size_t)total, hipMemcpyHostToDevice));
    blasr_hip_map_params mapParams = params->map;
    if (wantChains) {
      allChainOff.push_back((int64_t)(allChains.size() / 3));
      mapParams.iv_chains = allChains.data();
      mapParams.iv_chain_off = allChainOff.data();
    }
    rc = blasr_hip_map_reads_device(ctx, n_reads, reads_device, read_pool_device, (int32_t)total,
                                    (const blasr_hip_interval *)ctx->pipeIntervals.p, &mapParams, hits, hits_capacity, n_hits,
                                    read_hit_off, fetch);
    const double t3 = now_ms();
    ctx->pipeMs[0] = (float)(t1 - t0);  // anchoring + interval finding of all sub-blocks
    ctx->pipeMs[1] = 0;
    ctx->pipeMs[2] = (float)(t3 - t1);  // the path
    ctx->pipeMs[3] = (float)ivMs;       // of [0]: the interval finding (kernels + the host's end)
    ctx->pipeCounters[0] = anchorsInAll;
    ctx->pipeCounters[1] = total;
    if (rc == BLASR_HIP_OK && intervals_out && total > intervals_capacity)
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_CAPACITY, "map_from_reads: %lld intervals, capacity %lld", (long long)total,
                            (long long)intervals_capacity);
    return rc;
  }

  // ---- the host stage: one thread takes the sub-blocks in order
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Job *> queue;
  bool done = false;
  int hostError = BLASR_HIP_OK;
  std::vector<std::vector<blasr_hip_interval>> blockIvs;
  std::vector<std::vector<int64_t>> blockIvOff;
  std::vector<std::vector<uint32_t>> blockChains;     // (t, q, l) triplets of a sub-block's intervals, back to back
  std::vector<std::vector<int64_t>> blockChainOff;    // per interval of the sub-block, + 1
  double hostMs = 0;
  std::thread worker([&]() {
    for (;;) {
      Job *job = nullptr;
      {
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&]() { return !queue.empty() || done; });
        if (queue.empty()) return;
        job = queue.front();
        queue.pop_front();
      }
      int rc = BLASR_HIP_OK;
      std::vector<blasr_hip_interval> ivs;
      std::vector<int64_t> off, chainOff;
      std::vector<uint32_t> chains;
      try {
        // (a failed copy would leave stale anchors in the pinned buffer: the sub-block is an error then)
        if (hipSetDevice(ctx->device) != hipSuccess || hipEventSynchronize(job->copied) != hipSuccess || job->copyError)
          rc = BLASR_HIP_ERR_HIP;
        const double h0 = now_ms();
        if (rc == BLASR_HIP_OK) {
          ivs.resize((size_t)job->count * (size_t)std::max(params->interval.n_candidates, 1) + 1);
          off.resize((size_t)job->count + 1);
          int64_t n = 0;
          if (wantChains) {
            const int64_t nAnchors = job->listOff[2 * (size_t)job->count] - job->listOff[0];
            int64_t chainsUsed = 0;
            chains.resize(3 * (size_t)std::max<int64_t>(nAnchors, 1));  // (a chain is a subset of one list's anchors)
            chainOff.assign(ivs.size() + 1, 0);
            rc = blasr_hip_find_intervals_chains(genome_host, ctx->genomeLen, seq_start_pos, n_seq_pos, count_table, count_tuple_size,
                                                 job->count, reads_host + job->first, job->pinned + job->anchorOff,
                                                 job->listOff.data(), nullptr, &params->interval, ivs.data(), nullptr,
                                                 (int64_t)ivs.size(), &n, off.data(), chains.data(), (int64_t)(chains.size() / 3),
                                                 &chainsUsed, chainOff.data());
            if (rc == BLASR_HIP_ERR_CAPACITY && n <= (int64_t)ivs.size()) {  // (ten intervals of a read may share anchors)
              chains.resize(3 * (size_t)chainsUsed);
              rc = blasr_hip_find_intervals_chains(genome_host, ctx->genomeLen, seq_start_pos, n_seq_pos, count_table, count_tuple_size,
                                                   job->count, reads_host + job->first, job->pinned + job->anchorOff,
                                                   job->listOff.data(), nullptr, &params->interval, ivs.data(), nullptr,
                                                   (int64_t)ivs.size(), &n, off.data(), chains.data(), (int64_t)(chains.size() / 3),
                                                   &chainsUsed, chainOff.data());
            }
            chains.resize(3 * (size_t)std::max<int64_t>(chainsUsed, 0));
            chainOff.resize((size_t)std::max<int64_t>(n, 0) + 1);
          } else {
            rc = blasr_hip_find_intervals(genome_host, ctx->genomeLen, seq_start_pos, n_seq_pos, count_table, count_tuple_size,
                                          job->count, reads_host + job->first, job->pinned + job->anchorOff, job->listOff.data(),
                                          nullptr, &params->interval, ivs.data(), nullptr, (int64_t)ivs.size(), &n, off.data());
          }
          ivs.resize((size_t)std::max<int64_t>(n, 0));
          for (auto &iv : ivs) iv.read += job->first;
        }
        hostMs += now_ms() - h0;
      } catch (const std::bad_alloc &) {
        rc = BLASR_HIP_ERR_NOMEM;
        ivs.clear();
        chains.clear();
        chainOff.assign(1, 0);
      }
      std::lock_guard<std::mutex> lock(mu);
      if (rc != BLASR_HIP_OK && hostError == BLASR_HIP_OK) hostError = rc;
      try {  // (the lists of lists grow here: nothing may leave this thread)
        blockIvs.push_back(std::move(ivs));
        blockIvOff.push_back(std::move(off));
        blockChains.push_back(std::move(chains));
        blockChainOff.push_back(std::move(chainOff));
      } catch (...) {
        if (hostError == BLASR_HIP_OK) hostError = BLASR_HIP_ERR_NOMEM;
      }
    }
  });
  auto finish_worker = [&]() {
    {
      std::lock_guard<std::mutex> lock(mu);
      done = true;
    }
    cv.notify_all();
    if (worker.joinable()) worker.join();
  };
  struct JoinOnExit {  // an exception on this thread must not find the worker still running (std::terminate)
    decltype(finish_worker) &f;
    ~JoinOnExit() { f(); }
  } joinOnExit{finish_worker};

  // ---- the device stage
  // (an exception in this thread after the worker has started must not reach std::terminate through its joinable
  //  handle, nor cross the C boundary: the body below runs inside try / catch)
  int rc = BLASR_HIP_OK;
  int64_t used = 0;
  std::vector<Job> jobs;
  struct Cleanup {
    std::vector<Job> &jobs;
    ~Cleanup() {
      for (auto &j : jobs)
        if (j.copied) (void)hipEventDestroy(j.copied);
    }
  } cleanup{jobs};
  try {
  jobs.resize((size_t)((n_reads + perBlock - 1) / perBlock));
  for (auto &j : jobs) {
    j.copied = nullptr;
    j.pinned = nullptr;
    j.copyError = 0;
  }
  for (size_t b = 0; b < jobs.size() && rc == BLASR_HIP_OK; b++) {
    Job &j = jobs[b];
    j.first = (int32_t)(b * (size_t)perBlock);
    j.count = std::min(perBlock, n_reads - j.first);
    j.listOff.assign(2 * (size_t)j.count + 1, 0);
    int64_t blockBases = 0;
    for (int32_t r = j.first; r < j.first + j.count; r++) blockBases += reads_host[r].length;
    int64_t expect = 2 * blockBases + 4096;  // anchors of this sub-block: a guess, corrected by the count a short call returns
    for (int attempt = 0; attempt < 2 && rc == BLASR_HIP_OK; attempt++) {
      // ---- room on the device (grow-only; what earlier sub-blocks wrote moves along)
      const size_t wantDev = sizeof(blasr_hip_anchor) * (size_t)(used + expect);
      if (wantDev > ctx->pipeAnchors.bytes) {
        DevBuf bigger;
        if ((rc = blasr_hip_ensure(ctx, bigger, wantDev + wantDev / 4)) != BLASR_HIP_OK) break;
        (void)hipStreamSynchronize(ctx->sideStream);  // copies out of the old buffer
        if (used > 0)
          (void)hipMemcpy(bigger.p, ctx->pipeAnchors.p, sizeof(blasr_hip_anchor) * (size_t)used, hipMemcpyDeviceToDevice);
        if (ctx->pipeAnchors.p) (void)hipFree(ctx->pipeAnchors.p);
        ctx->pipeAnchors = bigger;
      }
      const int64_t cap = (int64_t)(ctx->pipeAnchors.bytes / sizeof(blasr_hip_anchor)) - used;
      int64_t n = 0;
      rc = blasr_hip_anchor_reads_device(ctx, j.count, reads_host + j.first, read_pool_device, &ap,
                                         (blasr_hip_anchor *)ctx->pipeAnchors.p + used, cap, &n, j.listOff.data());
      if (rc == BLASR_HIP_ERR_CAPACITY && attempt == 0) {
        expect = n + 1;
        rc = BLASR_HIP_OK;
        continue;
      }
      if (rc != BLASR_HIP_OK) break;
      // ---- the pinned host mirror, same offsets.  Growing it: the host stage must have finished with the old one
      // (what it has already turned into intervals is not needed again)
      const size_t wantPinned = sizeof(blasr_hip_anchor) * (size_t)(used + n + 1);
      if (wantPinned > ctx->pipePinnedBytes) {
        for (;;) {
          {
            std::lock_guard<std::mutex> lock(mu);
            if (queue.empty() && blockIvs.size() == b) break;
          }
          std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        (void)hipStreamSynchronize(ctx->sideStream);
        if (ctx->pipePinned) (void)hipHostFree(ctx->pipePinned);
        ctx->pipePinned = nullptr;
        ctx->pipePinnedBytes = 0;
        const int64_t blocksLeft = (int64_t)(jobs.size() - b);
        const size_t bytes = sizeof(blasr_hip_anchor) * (size_t)(used + (n + n / 8) * blocksLeft + 1024);
        void *p = nullptr;
        if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
          rc = blasr_hip_fail(ctx, BLASR_HIP_ERR_NOMEM, "map_from_reads: %zu bytes of pinned host memory", bytes);
          break;
        }
        ctx->pipePinned = p;
        ctx->pipePinnedBytes = bytes;
      }
      j.pinned = (const blasr_hip_anchor *)ctx->pipePinned;
      j.anchorOff = used;
      if (hipEventCreateWithFlags(&j.copied, hipEventDisableTiming) != hipSuccess) {
        rc = blasr_hip_fail(ctx, BLASR_HIP_ERR_HIP, "map_from_reads: event");
        break;
      }
      // the anchor call has synchronised the main stream: the copy can start at once, under the next sub-block
      // (in 64 MB pieces: what the next sub-block's pieces report through pinned memory then waits behind 2 ms of
      //  transfer, not behind the whole sub-block's 30)
      for (int64_t at = 0; at < n; at += (64ll << 20) / (int64_t)sizeof(blasr_hip_anchor)) {
        const int64_t m = std::min<int64_t>(n - at, (64ll << 20) / (int64_t)sizeof(blasr_hip_anchor));
        if (hipMemcpyAsync((blasr_hip_anchor *)ctx->pipePinned + used + at, (blasr_hip_anchor *)ctx->pipeAnchors.p + used + at,
                           sizeof(blasr_hip_anchor) * (size_t)m, hipMemcpyDeviceToHost, ctx->sideStream) != hipSuccess)
          j.copyError = 1;
      }
      (void)hipEventRecord(j.copied, ctx->sideStream);
      {
        std::lock_guard<std::mutex> lock(mu);
        queue.push_back(&j);
      }
      cv.notify_all();
      used += n;
      break;
    }
  }
  } catch (const std::bad_alloc &) {
    finish_worker();
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_NOMEM, "map_from_reads: out of host memory");
  }
  const double t1 = now_ms();
  finish_worker();
  const double t2 = now_ms();
  if (rc == BLASR_HIP_OK && hostError != BLASR_HIP_OK) rc = blasr_hip_fail(ctx, hostError, "map_from_reads: find_intervals failed");
  if (rc != BLASR_HIP_OK) return rc;

  // ---- all intervals, in read order; then the path
  int64_t total = 0;
  for (auto &v : blockIvs) total += (int64_t)v.size();
  std::vector<blasr_hip_interval> ivs;
  ivs.reserve((size_t)total);
  for (auto &v : blockIvs) ivs.insert(ivs.end(), v.begin(), v.end());
  if (n_intervals) *n_intervals = total;
  if (intervals_out) memcpy(intervals_out, ivs.data(), sizeof(blasr_hip_interval) * (size_t)std::min(total, intervals_capacity));
  if ((rc = blasr_hip_ensure(ctx, ctx->pipeIntervals, sizeof(blasr_hip_interval) * (size_t)(total + 1))) != BLASR_HIP_OK) return rc;
  HIP_TRY(ctx, hipMemcpy(ctx->pipeIntervals.p, ivs.data(), sizeof(blasr_hip_interval) * (size_t)total, hipMemcpyHostToDevice));
  blasr_hip_map_params mapParams = params->map;
  std::vector<uint32_t> allChains;
  std::vector<int64_t> allChainOff;
  if (wantChains) {  // the sub-blocks' chains back to back, like their intervals
    allChainOff.reserve((size_t)total + 1);
    for (size_t b = 0; b < blockIvs.size(); b++) {
      const int64_t shift = (int64_t)(allChains.size() / 3);
      for (size_t i = 0; i < blockIvs[b].size(); i++) allChainOff.push_back(shift + blockChainOff[b][i]);
      allChains.insert(allChains.end(), blockChains[b].begin(), blockChains[b].end());
    }
    allChainOff.push_back((int64_t)(allChains.size() / 3));
    mapParams.iv_chains = allChains.data();
    mapParams.iv_chain_off = allChainOff.data();
  }
  rc = blasr_hip_map_reads_device(ctx, n_reads, reads_device, read_pool_device, (int32_t)total,
                                  (const blasr_hip_interval *)ctx->pipeIntervals.p, &mapParams, hits, hits_capacity, n_hits,
                                  read_hit_off, fetch);
  const double t3 = now_ms();
  ctx->pipeMs[0] = (float)(t1 - t0);  // anchoring of all sub-blocks (the host stage runs underneath)
  ctx->pipeMs[1] = (float)(t2 - t1);  // waiting for the host stage's last sub-block
  ctx->pipeMs[2] = (float)(t3 - t2);  // the path
  ctx->pipeMs[3] = (float)hostMs;     // host time spent finding intervals (hidden under [0] except [1])
  ctx->pipeCounters[0] = used;
  ctx->pipeCounters[1] = total;
  if (rc == BLASR_HIP_OK && intervals_out && total > intervals_capacity)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_CAPACITY, "map_from_reads: %lld intervals, capacity %lld", (long long)total,
                          (long long)intervals_capacity);
  return rc;
} BLASR_HIP_CATCH_CTX(ctx)

int blasr_hip_map_from_reads_host(blasr_hip_ctx *ctx, const uint8_t *genome_host, const uint32_t *seq_start_pos,
                                  int32_t n_seq_pos, const int32_t *count_table, int32_t count_tuple_size, int32_t n_reads,
                                  const blasr_hip_read *reads, const uint8_t *read_pool, int64_t read_pool_bytes,
                                  const blasr_hip_pipeline_params *params, blasr_hip_hit *hits, int64_t hits_capacity,
                                  int64_t *n_hits, int64_t *read_hit_off, blasr_hip_map_fetch *fetch,
                                  blasr_hip_interval *intervals_out, int64_t intervals_capacity, int64_t *n_intervals) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (n_reads < 0 || read_pool_bytes < 0 || (n_reads > 0 && (!reads || !read_pool)))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "map_from_reads_host: bad arguments");
  for (int32_t r = 0; r < n_reads; r++)
    if (reads[r].off < 0 || reads[r].off + (int64_t)reads[r].length > read_pool_bytes)
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "map_from_reads_host: read %d outside the pool", r);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  struct Free {
    void *p = nullptr;
    ~Free() {
      if (p) (void)hipFree(p);
    }
  } dReads, dPool;
  HIP_TRY(ctx, hipMalloc(&dReads.p, sizeof(blasr_hip_read) * (size_t)std::max(n_reads, 1)));
  HIP_TRY(ctx, hipMalloc(&dPool.p, (size_t)std::max<int64_t>(read_pool_bytes, 1)));
  if (n_reads > 0) HIP_TRY(ctx, hipMemcpy(dReads.p, reads, sizeof(blasr_hip_read) * (size_t)n_reads, hipMemcpyHostToDevice));
  if (read_pool_bytes > 0) HIP_TRY(ctx, hipMemcpy(dPool.p, read_pool, (size_t)read_pool_bytes, hipMemcpyHostToDevice));
  return blasr_hip_map_from_reads_device(ctx, genome_host, seq_start_pos, n_seq_pos, count_table, count_tuple_size, n_reads,
                                         reads, (const blasr_hip_read *)dReads.p, (const uint8_t *)dPool.p, params, hits,
                                         hits_capacity, n_hits, read_hit_off, fetch, intervals_out, intervals_capacity,
                                         n_intervals);
} BLASR_HIP_CATCH_CTX(ctx)

}  // extern "C"

namespace {

// The overlapped form (blasr_hip_pipeline_params.overlap): a host thread anchors the sub-blocks one after the other on
// the helper context -- its own streams, workspace and anchor buffers; the genome and the index are the caller's -- and
// this thread runs the path on sub-block b as soon as its intervals are there, while the helper is already searching for
// sub-block b + 1.  The search waits for random sectors of HBM most of the time and the path's kernels for their own
// instructions, so the device takes less for the two side by side than for one after the other.  MapRead is a function
// of one read (alignment/Blasr.cpp:2247-2644): the hits are those of the serial form, in its order.
int map_from_reads_overlapped(blasr_hip_ctx *ctx, const PipeCall &call, int32_t perBlock, int32_t n_reads,
                              const blasr_hip_read *reads_device, const blasr_hip_map_params &map, blasr_hip_hit *hits,
                              int64_t hits_capacity, int64_t *n_hits, int64_t *read_hit_off, blasr_hip_map_fetch *fetch,
                              blasr_hip_interval *intervals_out, int64_t intervals_capacity, int64_t *n_intervals) {
  const double t0 = now_ms();
  // ---- the helper context
  blasr_hip_ctx *h = ctx->pipeHelper;
  if (!h) {
    const int rc = blasr_hip_create(ctx->device, &h);
    if (rc != BLASR_HIP_OK) return blasr_hip_fail(ctx, rc, "map_from_reads: no helper context");
    ctx->pipeHelper = h;
    // (full-size pieces of the search need 12 GB of workspace: anchor_api.hip's pieceLimit)
  }
  // The caller's limit is what the TWO contexts may hold together, and it may have changed since the helper was made: both
  // limits are set again on every call -- the helper a quarter of it, at most the 16 GB its largest pieces ask for, this
  // context the rest for the call's duration.
  struct LimitScope {
    blasr_hip_ctx *c;
    int64_t saved;
    ~LimitScope() { c->workspaceLimit = saved; }
  } limitScope{ctx, ctx->workspaceLimit};
  h->workspaceLimit = std::min<int64_t>(ctx->workspaceLimit / 4, 16ll << 30);
  ctx->workspaceLimit -= h->workspaceLimit;
  const bool wantExt = !getenv("BLASR_HIP_ANCHOR_NO_EXT");
  // what the helper borrowed is compared by the owner's epoch, not by addresses (one freed and handed out again would pass)
  if (h->saBorrowedFrom != ctx || h->saBorrowedEpoch != ctx->indexEpoch || h->genome != ctx->genome ||
      (wantExt && !ctx->saExt)) {
    const int rc = blasr_hip_borrow_index(h, ctx);
    if (rc != BLASR_HIP_OK) return blasr_hip_fail(ctx, rc, "map_from_reads: %s", h->lastError.c_str());
  }

  const int nBlocks = (n_reads + perBlock - 1) / perBlock;
  std::vector<BlockOut> blocks((size_t)nBlocks);
  std::vector<int> state((size_t)nBlocks, 0);  // 0 not yet, 1 ready, < 0 the anchoring's error code
  std::mutex mu;
  std::condition_variable cv;
  bool stop = false;
  double anchorBusyMs = 0, ivMs = 0;
  float anchorMs[4] = {0, 0, 0, 0};
  int64_t anchorCounters[4] = {0, 0, 0, 0};
  std::thread helper([&]() {
    int rc = hipSetDevice(h->device) == hipSuccess ? BLASR_HIP_OK : BLASR_HIP_ERR_HIP;
    for (int b = 0; b < nBlocks; b++) {
      {
        std::lock_guard<std::mutex> lock(mu);
        if (stop) rc = BLASR_HIP_ERR_INTERNAL;
      }
      if (rc == BLASR_HIP_OK) {
        const double a0 = now_ms();
        const int32_t first = b * perBlock;
        try {
          rc = anchor_block(h, call, first, std::min(perBlock, n_reads - first), blocks[(size_t)b]);
        } catch (const std::bad_alloc &) {
          rc = BLASR_HIP_ERR_NOMEM;
        } catch (...) {
          rc = BLASR_HIP_ERR_INTERNAL;
        }
        anchorBusyMs += now_ms() - a0;
        ivMs += blocks[(size_t)b].ivMs;
        for (int i = 0; i < 4; i++) {
          anchorMs[i] += h->anchorMs[i];
          anchorCounters[i] += h->anchorCounters[i];
        }
      }
      {
        std::lock_guard<std::mutex> lock(mu);
        state[(size_t)b] = rc == BLASR_HIP_OK ? 1 : rc;
      }
      cv.notify_all();
    }
  });
  struct JoinOnExit {  // nothing may leave this function with the helper still running
    std::thread &t;
    std::mutex &mu;
    bool &stop;
    ~JoinOnExit() {
      {
        std::lock_guard<std::mutex> lock(mu);
        stop = true;
      }
      if (t.joinable()) t.join();
    }
  } joinOnExit{helper, mu, stop};

  // ---- the path, sub-block by sub-block
  int rc = BLASR_HIP_OK;
  bool hitsShort = false, fetchShort = false;
  int64_t hitsUsed = 0, ivsUsed = 0, seqUsed = 0, opsUsed = 0;
  double waitMs = 0, pathMs = 0;
  float mapMs[6] = {0, 0, 0, 0, 0, 0};
  int64_t mapCounters[6] = {0, 0, 0, 0, 0, 0};
  for (int b = 0; b < nBlocks && rc == BLASR_HIP_OK; b++) {
    const double w0 = now_ms();
    int st;
    {
      std::unique_lock<std::mutex> lock(mu);
      cv.wait(lock, [&]() { return state[(size_t)b] != 0; });
      st = state[(size_t)b];
    }
    const double w1 = now_ms();
    waitMs += w1 - w0;
    if (st < 0) {
      rc = blasr_hip_fail(ctx, st, "map_from_reads: sub-block %d: %s", b, h->lastError.c_str());
      break;
    }
    BlockOut &o = blocks[(size_t)b];
    const int32_t first = b * perBlock, count = std::min(perBlock, n_reads - first);
    const int64_t nIv = (int64_t)o.ivs.size();
    if ((rc = blasr_hip_ensure(ctx, ctx->pipeIntervals, sizeof(blasr_hip_interval) * (size_t)(nIv + 1))) != BLASR_HIP_OK) break;
    // (on the path's own stream: a copy on the null stream would wait for the helper's search as well)
    HIP_TRY(ctx, hipMemcpyAsync(ctx->pipeIntervals.p, o.ivs.data(), sizeof(blasr_hip_interval) * (size_t)nIv, hipMemcpyHostToDevice,
                                ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // (the mapper's side streams read the intervals too)
    blasr_hip_map_params mapParams = map;
    if (call.wantChains) {
      mapParams.iv_chains = o.chains.data();
      mapParams.iv_chain_off = o.chainOff.data();
    }
    blasr_hip_map_fetch sub = {nullptr, 0, 0, nullptr, 0, 0};
    if (fetch) {
      sub.seq = fetch->seq ? fetch->seq + std::min(seqUsed, fetch->seq_capacity) : nullptr;
      sub.seq_capacity = std::max<int64_t>(fetch->seq_capacity - seqUsed, 0);
      sub.ops = fetch->ops ? fetch->ops + std::min(opsUsed, fetch->ops_capacity) : nullptr;
      sub.ops_capacity = std::max<int64_t>(fetch->ops_capacity - opsUsed, 0);
    }
    const int64_t room = std::max<int64_t>(hits_capacity - hitsUsed, 0);
    int64_t nh = 0;
    int rcM = blasr_hip_map_reads_device(ctx, count, reads_device + first, call.readPoolDevice, (int32_t)nIv,
                                         (const blasr_hip_interval *)ctx->pipeIntervals.p, &mapParams,
                                         hits ? hits + std::min(hitsUsed, hits_capacity) : nullptr, room, &nh,
                                         read_hit_off ? read_hit_off + first : nullptr, fetch ? &sub : nullptr);
    if (rcM == BLASR_HIP_ERR_CAPACITY) {  // the counts are valid: go on, so that the caller learns what the whole call needs
      if (nh > room) hitsShort = true;
      else fetchShort = true;
      rcM = BLASR_HIP_OK;
    }
    if (rcM != BLASR_HIP_OK) {
      rc = rcM;
      break;
    }
    const int64_t got = std::min(nh, room);
    for (int64_t i = 0; i < got; i++) {
      blasr_hip_hit &x = hits[hitsUsed + i];
      x.read += first;
      x.interval += (int32_t)ivsUsed;
      if (fetch) {
        x.ops_off += opsUsed;
        x.q_off += seqUsed;
        x.t_off += seqUsed;
      }
    }
    if (read_hit_off)
      for (int32_t r = 0; r <= count; r++) read_hit_off[first + r] += hitsUsed;
    if (intervals_out)
      for (int64_t i = 0; i < nIv && ivsUsed + i < intervals_capacity; i++) {
        intervals_out[ivsUsed + i] = o.ivs[(size_t)i];
        intervals_out[ivsUsed + i].read += first;
      }
    hitsUsed += nh;
    ivsUsed += nIv;
    if (fetch) {
      seqUsed += sub.seq_used;
      opsUsed += sub.ops_used;
    }
    for (int i = 0; i < 6; i++) {
      mapMs[i] += ctx->mapMs[i];
      mapCounters[i] += ctx->mapCounters[i];
    }
    BlockOut().ivs.swap(o.ivs);  // (a sub-block's lists are not needed again)
    std::vector<uint32_t>().swap(o.chains);
    pathMs += now_ms() - w1;
  }
  if (rc != BLASR_HIP_OK) return rc;  // (joinOnExit stops the helper)
  helper.join();
  *n_hits = hitsUsed;
  if (n_intervals) *n_intervals = ivsUsed;
  if (fetch) {
    fetch->seq_used = seqUsed;
    fetch->ops_used = opsUsed;
  }
  int64_t anchorsInAll = 0;
  for (auto &o : blocks) anchorsInAll += o.anchors;
  for (int i = 0; i < 6; i++) {
    ctx->mapMs[i] = mapMs[i];
    ctx->mapCounters[i] = mapCounters[i];
  }
  for (int i = 0; i < 4; i++) {
    ctx->anchorMs[i] = anchorMs[i];
    ctx->anchorCounters[i] = anchorCounters[i];
  }
  ctx->pipeMs[0] = (float)anchorBusyMs;  // the helper's time in anchoring + interval finding (under the path, except [1])
  ctx->pipeMs[1] = (float)waitMs;        // this thread waiting for a sub-block's intervals
  ctx->pipeMs[2] = (float)pathMs;        // the path's calls
  ctx->pipeMs[3] = (float)ivMs;          // of [0]: the interval finding
  ctx->pipeCounters[0] = anchorsInAll;
  ctx->pipeCounters[1] = ivsUsed;
  (void)t0;
  if (hitsShort)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_CAPACITY, "map_from_reads: %lld hits, capacity %lld", (long long)hitsUsed,
                          (long long)hits_capacity);
  if (fetchShort)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_CAPACITY, "map_from_reads: the fetch arenas need %lld bytes and %lld ops",
                          (long long)seqUsed, (long long)opsUsed);
  if (intervals_out && ivsUsed > intervals_capacity)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_CAPACITY, "map_from_reads: %lld intervals, capacity %lld", (long long)ivsUsed,
                          (long long)intervals_capacity);
  return BLASR_HIP_OK;
}

}  // namespace
