// Internal interface between the C-ABI host layer (sdp_api.hip) and the gfx950
// kernels of the SDP stage (sdp_kernels.hip).  Public surface: include/blasr_hip.h.
#ifndef BLASR_AMD_SDP_KERNELS_H_
#define BLASR_AMD_SDP_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace blasr_hip {

// Host side: HIP-event brackets around the stage's heavy kernels, each on the stream it is launched on
// (blasr_hip_sdp_last_kernels).  begin() / end() enclose the launches of ONE kind on ONE stream; collect() -- after the
// streams were joined and synchronised -- adds the elapsed times to the records.
struct SdpSpans {
  struct Span {
    int kind, concurrent;
    hipEvent_t a, b;
  };
  std::vector<hipEvent_t> *pool = nullptr;  // the context's events, grow-only
  size_t used = 0;
  std::vector<Span> spans;
  Span cur = {-1, 0, nullptr, nullptr};
  hipEvent_t take();
  void begin(int kind, hipStream_t s, int concurrent = 1);
  void end(hipStream_t s);
  // ms[kind] += elapsed, launches[kind]++, concurrent[kind] = max; forgets the spans (their events are reused)
  void collect(float *ms, int32_t *launches, int32_t *concurrent, int kinds);
};

// One SDPAlign invocation (SDPAlign.h:95-626): a top-level candidate or a
// recursive call on a gap rectangle.
struct SdpTask {
  int64_t qOff, tOff;  // first base of the query / target rectangle in the sequence pool
  int32_t qLen, tLen;
  int32_t k;           // wordSize
  int32_t type;        // 0 = Local, 1 = Global (AlignmentUtils.h:14-15)
  int32_t prefix;      // sdpPrefixLength
  int32_t recurse;     // recursion budget left
  int32_t maxPerPos;   // maxMatchesPerPosition
  int32_t top;         // index of the top-level candidate this call belongs to
  int32_t qOrg, tOrg;  // origin of the rectangle in the top-level candidate's coordinates
  int32_t flags;       // kSdpExtendFront: extendFrontByLocalAlignment; kSdpDetailed: detailedAlignment; bits 8..15: sdp_sw_type
  int32_t pad;
};
enum { kSdpExtendFront = 1, kSdpDetailed = 2, kSdpSwTypeShift = 8 };
// bits 8..15 of flags: smithWatermanAlignType of the call when it is not the default (EndAnchored = 6, SDPAlign.h:116)
__host__ __device__ inline int sdp_sw_type(int flags) { return ((flags >> kSdpSwTypeShift) & 0xff) ? ((flags >> kSdpSwTypeShift) & 0xff) : 6; }

// SWAlign(Global) on a gap rectangle with q*t < noRecurseUnder (SDPAlign.h:502-504, :438-441)
struct SwTask {
  int64_t qOff, tOff;
  int32_t qLen, tLen;
  int32_t top, qOrg, tOrg, pad;
};

struct Frag {  // sdp/SDPFragment.h:5-17, the fields that are data
  uint32_t x, y, len, pad;
};

// Finished blocks, in top-level coordinates, as two parallel arrays (keys sort them):
//   key = candidate << 32 | q        val = t << 32 | length

struct SdpParams {
  int sdpIns, sdpDel;
  float indelRate;
  int noRecurseUnder;
  int match;       // scoreMatrix[0][0]
  int ins, del;    // score function gap costs (SWAlign)
};

constexpr int kChainCats = 8;  // categories of the sweep's sorted list (SdpLevelArrays::chainCount)
struct SdpLevelArrays {
  int n;
  const SdpTask *tasks;
  const int64_t *keyOff;   // n+1, exclusive scan of tLen
  uint64_t *keys;          // target k-mer keys, sorted per task
  const int64_t *qPosOff;  // n+1, exclusive scan of qLen
  uint64_t *hits;          // per query position: index of the first matching key | match count << 32
  uint64_t *qWords;        // per query position (same offsets): the reference's query tuple, where words have >= 16 bases; else nullptr
  int32_t *nRaw;           // 3n: prefix / middle / suffix fragment counts
  const int64_t *fragOff;  // n+1, exclusive scan of the raw fragment counts
  Frag *raw;               // prefix | middle | suffix fragments as StoreMatchingPositions emits them
  Frag *frags;             // sorted by (x,y), duplicates removed (SDPAlign.h:256-269)
  int32_t *nFrag;          // n
  uint64_t *yxKeys;        // (y << 32 | x) for StoreAbove
  uint32_t *yxIdx;
  int32_t *cost, *chainPrev, *chainLen, *predCl, *above;  // per fragment
  int idxMaxT;                  // targets up to this length (and >= 1536) are matched by sdp_match_index_kernel
  int noIdxFilter;              // tests / experiments (BLASR_HIP_NO_IDX_FILTER): the index matcher's count pass without its filter
  int idxFilterX2;              // the filter is used where the target's k-mers x idxFilterX2 / 2 < the filter's bits (default 2: fewer k-mers than bits; configs[4], 30 kb windows: match + order 347 / 332 ms with 4, 321 / 319 with 2)
  int32_t *bigList, *bigCount;  // those tasks
  int32_t *keyList, *keyCount;  // the tasks whose targets get sorted k-mer keys (sdp_keys_kernel, sdp_sort_keys_kernel)
  int32_t *midList, *midCount;  // the tasks of sdp_match_kernel's bucket instantiation (targets >= 256 bases, not indexed)
  // The sweep's invocations (not fused, >= 1 fragment), sorted by category (descending) and fragment count, and the
  // kChainCats category counts: chainCount[0] sdp_chain4_kernel's; the others are the invocations whose fragments scatter
  // over the rectangle (the recursion's Global calls: unrelated sequence gives chance matches anywhere), which come FIRST
  // in the list: [6] sdp_chain_rows_kernel, [5] sdp_chain_bits_kernel with a target beyond 65536 columns, [4] beyond
  // 16384, [3] a smaller one, [2] sdp_chain_bits4_kernel, [7] the same with the larger map, [1] sdp_chain_wide_kernel.  chainList null: everything through
  // sdp_chain_kernel.
  int32_t *chainList, *chainCount;
  int scatterMode;              // tests: 1 the scattered invocations take sdp_chain_wide_kernel; 2 the bitmap sweep one per wavefront;
                                // 3 sdp_chain_rows_kernel, all of them
  int bits4Wide;                // the scattered invocations of at most this many fragments on targets of 4097 .. 16384 columns go four per
                                // wavefront too (category 7; BLASR_HIP_BITS4_WIDE, 0: they keep a wavefront each, category 3)
  int rowsPrio;                     // experiments (BLASR_HIP_ROWS_PRIO): sdp_chain_rows_kernel's wavefronts at a raised priority (default: not raised)
  int rowsMinFrags, rowsDensityX2;  // sdp_chain_rows_kernel takes the scattered invocations with at least rowsMinFrags fragments and
                                    // rowsDensityX2 / 2 fragments per query row on average (rowsMinFrags 0: none)
  const int64_t *colOff;        // n+1: first column slot of a task in colEnt (tLen slots for a task that may scatter)
  int4 *colEnt;                 // per target column: {column, fragment, cost, x+y} of the column set's entry there
  int32_t *fused;               // per task: sdp_small_kernel did order .. chain (<= 64 raw fragments)
  int32_t *idxFlag;             // per task: some query position has more than three matches
  int32_t *idxWave;             // 4 per task: the middle fragments of each wavefront's share of the query (index kernel)
  int32_t *tieList, *tieCount;  // tasks whose fragment lists share an (x,y), see sdp_order_ties_kernel
  int32_t *cleanList, *cleanCount;  // (or null) the tasks sdp_order_kernel ordered itself: what sdp_yx_keys .. sdp_above_predcl take
                                    // while sdp_order_ties_kernel replays the others' sort on another stream (sdp_api.hip)
  int generalMode;              // which invocations sdp_order .. sdp_above_predcl take (general_task in sdp_kernels.hip)
  int32_t *leftList, *leftCount;  // mode 2's list: what sdp_small_kernel could not fuse although it has <= 64 fragments
  int32_t *bigGenList, *bigGenCount;  // the invocations with more than 64 raw fragments (sdp_small4_keys_kernel lists them)
  int4 *colSet;  // column set storage {column, fragment, cost, x+y}, one slot per fragment
  int32_t *chainEnd;                                      // n: last fragment of the reported chain
  int32_t *scratchA;                                      // per fragment: chain indices
  int32_t *scratchB;                                      // 3 per fragment: candidate blocks
  // per kernel kind (BLASR_HIP_SDP_KERNEL_*) three sums over the invocations it takes: {fragments, invocations, query
  // positions}, accumulated on the device (launch_sdp_kernel_sums, launch_sdp_chain_list); null: not wanted
  unsigned long long *kernelSums;
  SdpSpans *spans;  // host only: the launch functions bracket their heavy kernels with it (null: no timing)
};

struct SdpEmit {
  uint64_t *poolKeys, *poolVals;
  unsigned long long *poolCount;
  int64_t poolCap;
  SwTask *sw;
  unsigned long long *swCount;
  int64_t swCap;
  SdpTask *children;
  unsigned long long *childCount;
  int64_t childCap;
  int32_t *overflow;
};

void launch_sdp_keys(const SdpLevelArrays &lv, const uint8_t *seq, hipStream_t st);
// the query's tuples as StoreMatchingPositions makes them, for the invocations with words of sixteen bases and more (lv.qWords)
void launch_sdp_query_words(const SdpLevelArrays &lv, const uint8_t *seq, hipStream_t st);
void launch_sdp_sort_keys(const SdpLevelArrays &lv, uint64_t *tmp, hipStream_t st);
void launch_sdp_sort_yx(const SdpLevelArrays &lv, uint64_t *tmpKeys, uint32_t *tmpIdx, hipStream_t st);
void launch_sdp_match(const SdpLevelArrays &lv, const uint8_t *seq, bool write, hipStream_t st, hipStream_t side = nullptr);
void launch_sdp_match_all(const SdpLevelArrays &lv, const uint8_t *seq, bool write, hipStream_t st, hipStream_t *aux,
                          hipEvent_t *ev);
void launch_sdp_classify(const SdpLevelArrays &lv, hipStream_t st);
// order: the level's invocations sorted by launch_sdp_small4_keys' keys, descending (nullptr: no sdp_small4_kernel)
void launch_sdp_small4_keys(const SdpLevelArrays &lv, uint32_t *keys, int32_t *vals, hipStream_t st);
void launch_sdp_small(const SdpLevelArrays &lv, const SdpParams &p, const int32_t *order, hipStream_t st);
int sdp_match_index_max_target(int device);
void launch_sdp_match_index(const SdpLevelArrays &lv, const uint8_t *seq, bool write, hipStream_t st, hipStream_t side = nullptr);
void launch_sdp_order(const SdpLevelArrays &lv, hipStream_t st, bool tiesToo = true);
void launch_sdp_order_ties(const SdpLevelArrays &lv, hipStream_t st);
void launch_sdp_yx_keys(const SdpLevelArrays &lv, hipStream_t st);
void launch_sdp_above_predcl(const SdpLevelArrays &lv, hipStream_t st);
// the (fragment count, index) pairs of the invocations sdp_chain4_kernel takes; sorted descending by the caller into
// lv.chainList; lv.chainCount receives their number
void launch_sdp_chain_list(const SdpLevelArrays &lv, uint32_t *keys, int32_t *vals, hipStream_t st);
// aux / ev: three more streams and four events (fork, three joins), or nullptr: everything on st, one after the other
void launch_sdp_chain(const SdpLevelArrays &lv, const SdpParams &p, hipStream_t st, hipStream_t *aux = nullptr,
                      hipEvent_t *ev = nullptr, const int32_t *counts = nullptr);
// counts: 4 ints per task {blocks, SWAlign fills, recursive jobs, 0}
void launch_sdp_blocks_build(const SdpLevelArrays &lv, const SdpParams &p, int32_t *counts, hipStream_t st);
void launch_sdp_blocks_emit(const SdpLevelArrays &lv, const SdpParams &p, const SdpEmit &e, const int32_t *counts,
                            const int64_t *offPool, const int64_t *offSw, const int64_t *offChild, hipStream_t st);
void launch_sw_sort_keys(const SwTask *tasks, int64_t n, bool smallOk, uint32_t *keys, uint32_t *vals, int32_t *nBig,
                         hipStream_t st);
int64_t sw_small_scratch_words(int blocks);
int sw_small_blocks(int64_t n);
void launch_sw_small(const SwTask *tasks, const uint32_t *order, int64_t n, const int32_t *nBig, const uint8_t *seq,
                     const int32_t *matrix, const SdpParams &p, uint32_t *scratch, const SdpEmit &e, hipStream_t st);
void launch_sw(const SwTask *tasks, const uint32_t *order, int64_t n, const int32_t *nBig, int blocks,
               const uint8_t *seq, const int32_t *matrix, const SdpParams &p, uint8_t *scratch, int64_t scratchPerLane,
               int64_t parkOff, const SdpEmit &e, hipStream_t st);

// top-level results
struct SdpResult {  // layout-identical to blasr_hip_sdp_result
  int32_t status, score, q_pos, t_pos, n_blocks, stats_score, n_match, n_mismatch, n_ins, n_del;
  int64_t blk_off;
};
void launch_sdp_finish(int n, const SdpTask *tasks, const uint8_t *seq, const uint64_t *poolKeys,
                       const uint64_t *poolVals, const int64_t *blkOff,
                       const int32_t *matrix, int ins, int del, int open, int ext, SdpResult *results, int32_t *blocks,
                       int64_t blocksCap, int64_t blkBase, hipStream_t st);
void launch_pool_offsets(const uint64_t *poolKeys, int64_t nPool, int n, int64_t *off, hipStream_t st);
// the matchers' and the fused small kernels' sums for lv.kernelSums (after the match write pass and launch_sdp_small)
void launch_sdp_kernel_sums(const SdpLevelArrays &lv, hipStream_t st);
// widest target sdp_chain_bits_kernel takes (its bitmap lives in LDS)
int sdp_chain_bits_max_target();
// tLen (or the three raw fragment counts summed, or nFrag; 4: the column slots of lv.colEnt) of every task as int64,
// for the offset scans
void launch_sdp_gather_sizes(const SdpLevelArrays &lv, int what, int64_t *out, hipStream_t st);
void launch_sdp_segment_ends(const SdpLevelArrays &lv, int64_t *ends, hipStream_t st);

}  // namespace blasr_hip
#endif
