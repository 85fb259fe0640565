// sdp_kernels.hip -- gfx950 kernels for BLASR's SDP stage.
//
// What the reference does per candidate interval (one CPU thread, one call):
//   SDPAlign                         SDPAlign.h:95-626
//     SequenceToTupleList            tuples/DNATuple.h:308-359
//     TupleList::Sort / FindAll      tuples/TupleList.h:98-126
//     StoreMatchingPositions         tuples/TupleMatching.h:37-61
//     std::sort + de-duplication     SDPAlign.h:256-269
//     SDPLongestCommonSubsequence    sdp/SparseDynamicProgramming.h:70-322
//     block merge / trim / filter    SDPAlign.h:308-413
//     SWAlign(Global) gap fills      SWAlign.h:19-389
//     recursive SDPAlign on big gaps SDPAlign.h:445-456, :509-520
//
// How it is laid out here: the recursion becomes successive kernel WAVES (level 0 = the candidates, level 1 =
// their large gap rectangles, level 2 = theirs); within a level every SDPAlign invocation runs through
//   sdp_classify       long targets (>= 1536 bases) are listed for the LDS-index matcher
//   sdp_match_index    long targets: one workgroup per invocation, the target's k-mers in an LDS hash index built
//                      with a stable counting scatter; count pass + copy pass (index rebuilt only if needed)
//   sdp_keys / sdp_sort_keys / sdp_match   mid-size targets: 2-bit k-mer keys, a wave-level stable LSD radix sort,
//                      binary search; targets <= 128 bases: k-mers in registers, all-pairs by v_readlane;
//                      prefix/suffix 5-mer fragments by direct comparison
//   sdp_small          invocations with <= 64 fragments: sort + keep-first, StoreAbove, sweep-set predecessor and
//                      the sweep itself in one kernel, fragments one per lane
//   sdp_order (+ sdp_order_ties)   merge of the prefix/middle/suffix lists; when duplicates exist the libstdc++
//                      introsort schedule is replayed in LDS so that the same duplicate survives
//   sdp_yx_keys / sdp_sort_yx / sdp_above_predcl   the two STATIC predecessors of every fragment ("above", and the
//                      predecessor in the sweep set, which depends only on geometry), all fragments in parallel
//   sdp_chain          the row sweep proper: sequential over fragments, the fragment window and the tail of the
//                      column set in registers
//   sdp_blocks_build / sdp_blocks_emit   chain -> blocks -> trim/filter; finished blocks, SWAlign jobs and the next
//                      level's SDPAlign jobs through count -> scan -> emit
//   sw_sort_keys / sw_small / sw   SWAlign(Global) gap fills, sorted by shape, one per lane
//   (device radix sort of all finished blocks by (candidate, query position): gap fills slot in
//    between their parents' blocks, so sorting by query position IS the in-order assembly)
//   pool_offsets / sdp_finish   per-candidate block ranges, Local re-basing, SDPAlign's return score,
//                      ComputeAlignmentStats
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/blasr_hip.h"
#include "introsort_replay.h"
#include "sdp_kernels.h"
#include "wave_utils.h"

namespace blasr_hip {

constexpr int kPosBits = 24;  // target positions < 2^24 (checked on the host)

// long targets are matched by sdp_match_index_kernel (further down) and need no sorted keys
constexpr int kIdxBuckets = 4096;
constexpr int kIdxMinT = 1536;   // shorter targets keep the wavefront-per-task kernels
constexpr int kIdxMaxK = 13;     // 2k bits of k-mer + 2*15 bits of shift must fit a 64-bit window
constexpr int kIdxSmallCap = 12288;  // targets up to this size: 52 KB of LDS per workgroup with the filter and its list, three
                                     // workgroups per CU (measured by padding the allocation: two cost + 9 ms of match + order at
                                     // configs[2], one + 30; a fourth would take the filter, half the buckets and the list to fit)
constexpr int kIdxMidCap = 20480;    // ... and up to this: two

__host__ __device__ inline bool task_uses_index(int tLen, int k, int idxMaxT) {
  return tLen >= kIdxMinT && tLen <= idxMaxT && k <= kIdxMaxK;
}
// ... and so do very short ones: their k-mers sit in registers, two per lane (sdp_match_kernel)
constexpr int kTinyT = 128;
// An invocation that may take sdp_chain_bits_kernel (and so gets a column slot per target base): a Global call --
// the recursion's; a top-level Local alignment is near-diagonal -- that the small kernels cannot have (they take at
// most 64 raw fragments, and only without ties above 16), on a target the LDS bitmap spans.
constexpr int kBitsMaxTarget = 131072;
constexpr int kBitsSmallTarget = 16384;  // sdp_chain_bits_kernel's small instantiation (2 KB of LDS per wavefront)
constexpr int kBitsMidTarget = 65536;    // ... and its middle one (8 KB: five workgroups of four wavefronts per CU; the large one, 16 KB, two)
#ifndef BLASR_BITS4_COLS
#define BLASR_BITS4_COLS 4096
#endif
constexpr int kBits4Target = BLASR_BITS4_COLS;   // sdp_chain_bits4_kernel: columns (128 words of LDS per invocation) ...
#ifndef BLASR_BITS4_FRAGS
#define BLASR_BITS4_FRAGS 4096
#endif
constexpr int kBits4MaxFrags = BLASR_BITS4_FRAGS; // ... and fragments of the invocations it takes
constexpr int kWideSets = 8;         // register sets of 64 column-set entries in sdp_chain_wide_kernel
__host__ __device__ inline bool task_may_scatter(const SdpTask &t, int rawFragments) {
  return t.type == 1 && rawFragments > 16 && t.tLen <= kBitsMaxTarget;
}
// Words of kKeyBases bases and more (-sdpTupleSize 16 .. 32; a tuple is 64 bits, tuples/DNATuple.h:55-94).  Two things change
// there.  (i) The sort key of a target position holds the tuple's LOW 32 bits -- the word's first sixteen bases; with kPosBits
// of position and the "no tuple here" bit that is the 64 bits of a key -- and the tuple's upper half is compared on the target's
// bases when the keys agree (target_tuple_high).  The list is sorted by (tuple, position) (DNATuple.h:147-154) and positions
// only leave it, so matches, their order, counts and maxMatchesPerPosition stay the reference's.  (ii) The QUERY's tuples are
// not its words: StoreMatchingPositions (TupleMatching.h:37-61) makes the first tuple of a clean stretch with FromStringRL
// (64-bit, exact) and every later one with ShiftAddRL (DNATuple.h:85-94), `tuple += (TwoBit[nuc] << ((tupleSize-1)*2))` with
// `static int TwoBit[]` (NucConversion.h:7): a 32-bit shift by 30 and more.  Undefined in C++; what the reference's builds
// compute is the count mod 32 and the int sign-extended into the sum, a recurrence over the whole stretch
// (sdp_query_words_kernel restates it, one lane per invocation; pinned to the reference built from its own sources by
// tests/golden/long_words_golden.npz).
constexpr int kKeyBases = 16;
__host__ __device__ inline int key_bases(int k) { return k < kKeyBases ? k : kKeyBases; }
// (the tiny targets' register path keeps a word in 32 bits with two values set aside for "no word": words of fewer than 16 bases)
__host__ __device__ inline bool task_is_tiny(int tLen, int k) { return tLen <= kTinyT && k < kKeyBases; }
__host__ __device__ inline bool task_needs_keys(int tLen, int k, int idxMaxT) {
  return !task_is_tiny(tLen, k) && !task_uses_index(tLen, k, idxMaxT);
}
// bits 32 .. 2k - 1 of the tuple of the k-base word at t (all ACGT): its bases 16 .. k - 1
__device__ __forceinline__ uint32_t target_tuple_high(const uint8_t *t, int k) {
  uint32_t v = 0;
  for (int j = kKeyBases; j < k; j++) v |= (uint32_t)(base_code(t[j]) & 3) << (2 * (j - kKeyBases));
  return v;
}

// The k-base words at positions p0 + lane of a sequence of `len` bases (2 bits per base, the first base lowest: FromStringRL,
// DNATuple.h:55-83); ok = the window lies inside the sequence and is ACGT throughout.  ONE byte load per lane (two for the
// k - 1 lanes whose windows reach into the next 64 positions): lane p reads base p0 + p and takes the bases behind it off the
// lanes above, a DPP wave_shl:1 per base.  The loops this replaces -- for (j < k) code(seq[p + j]), k known only at run time
// -- compile to a byte load AND ITS WAIT per base: up to eleven dependent round trips to memory per word, which is what the
// matchers of the short targets spent their time on (phase clocks, round 6: 22 of the 26 us of a tiny invocation).
// (the same from codes already in registers: c0 / c1 = base_code of positions p0 + lane / p0 + 64 + lane, 4 beyond the sequence)
__device__ __forceinline__ uint64_t words_from_codes(int c0, int c1, int k, bool &ok) {
  uint64_t v = 0;
  int any = 0;
  for (int j = 0; j < k; j++) {
    v |= (uint64_t)(c0 & 3) << (2 * j);
    any |= c0;
    const int edge = __builtin_amdgcn_readlane(c1, 0);
    c0 = dpp<0x130, 0xf>(edge, c0);  // lane i takes lane i + 1's, lane 63 the first of the next 64
    c1 = dpp<0x130, 0xf>(4, c1);
  }
  ok = any < 4;
  return v;
}
// the code of base p of a sequence, 4 beyond its end: an UNCONDITIONAL load (index clamped) so that several of them are in
// flight together -- behind a branch each the compiler waits for every one in turn
__device__ __forceinline__ int code_or_none(const uint8_t *s, int len, int p) {
  const int c = base_code(s[min(p, max(len - 1, 0))]);
  return p < len ? c : 4;
}
__device__ __forceinline__ uint64_t words_at(const uint8_t *s, int len, int p0, int k, bool &ok) {
  const int lane = lane_id();
  const int c0 = code_or_none(s, len, p0 + lane), c1 = code_or_none(s, len, p0 + 64 + lane);
  return words_from_codes(c0, c1, k, ok);
}

__device__ __forceinline__ int wave_task_index() {
  return __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
}

// Which invocation a wavefront of the GENERAL kernels between the match and the sweep takes (sdp_order .. above_predcl:
// those sdp_small4_kernel / sdp_small_kernel do not fuse), -1: none (any more).  lv.generalMode 0: a wavefront per
// invocation of the level, after the small kernels, by lv.fused (level 0, where nearly all are general ones); 2: a fixed
// grid striding over a list, lv.leftList -- first, BESIDE the small kernels, the invocations with more than 64 raw
// fragments, which are never fused (sdp_small4_keys_kernel lists them), then what the small kernels left over of the rest
// (<= 64 fragments, yet a tie that needs std::sort's schedule: rare).  At the recursion levels nine invocations in ten are
// fused, and a wavefront each that finds nothing to do cost every one of these kernels a millisecond.
__device__ __forceinline__ int general_task(const SdpLevelArrays &lv, int it) {
  const int slot = wave_task_index() + it * (int)(gridDim.x * 4);
  if (lv.generalMode == 2) return slot < *lv.leftCount ? lv.leftList[slot] : -1;
  if (it > 0 || slot >= lv.n) return -1;
  return lv.fused[slot] ? -1 : slot;
}

// ---------------------------------------------------------------------------
// sdp_keys: every k-mer of the target whose k bases are all ACGT (DNATuple.h:308-359),
// first base in the least-significant bits (FromStringRL :55-83)
// ---------------------------------------------------------------------------
// (a fixed grid over sdp_classify_kernel's keyList: at the recursion levels most targets are tiny and have no keys)
__global__ __launch_bounds__(256) void sdp_keys_kernel(SdpLevelArrays lv, const uint8_t *seq) {
  const int nKeyed = *lv.keyCount;
  for (int e = wave_task_index(); e < nKeyed; e += (int)gridDim.x * 4) {
  const int idx = lv.keyList[e];
  const int lane = lane_id();
  const SdpTask t = lv.tasks[idx];
  uint64_t *keys = lv.keys + lv.keyOff[idx];
  const uint8_t *tSeq = seq + t.tOff;
  const int kk = key_bases(t.k);
  const uint64_t invalid = 1ull << (kPosBits + 2 * kk);
  const uint64_t keyMask = kk < 32 ? (1ull << (2 * kk)) - 1ull : ~0ull;
  for (int p0 = 0; p0 < t.tLen; p0 += 64) {
    const int p = p0 + lane;
    bool ok;
    const uint64_t word = words_at(tSeq, t.tLen, p0, t.k, ok);
    if (p < t.tLen) keys[p] = ok ? (((word & keyMask) << kPosBits) | (uint64_t)p) : (invalid | (uint64_t)p);
  }
  }
}

// The query's tuples of the invocations with words of sixteen bases and more, as the reference's StoreMatchingPositions makes
// them (see kKeyBases above): one LANE per invocation walks its query.  Position s has a tuple iff its window is ACGT
// throughout (the matchers test that themselves); the first tuple of such a stretch is the word, the later ones follow
// ShiftAddRL's 32-bit shift as compiled.  qWords[s] = the tuple, for every window of the query.
__global__ __launch_bounds__(256) void sdp_query_words_kernel(SdpLevelArrays lv, const uint8_t *seq) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= lv.n) return;
  const SdpTask t = lv.tasks[i];
  const int k = t.k;
  if (k < kKeyBases) return;
  const uint8_t *q = seq + t.qOff;
  uint64_t *w = lv.qWords + lv.qPosOff[i];
  const int sh = (2 * (k - 1)) & 31;
  bool res = false;  // the position before had a tuple (TupleMatching.h:44-47)
  uint64_t tup = 0;
  int lastBad = -1;  // the last base that is not ACGT among q[0 .. s + k - 1]
  for (int j = 0; j < k - 1 && j < t.qLen; j++)
    if (base_code(q[j]) >= 4) lastBad = j;
  for (int s = 0; s + k <= t.qLen; s++) {
    const int c = base_code(q[s + k - 1]);
    if (c >= 4) lastBad = s + k - 1;
    if (res) {  // ShiftAddRL (DNATuple.h:85-94)
      if (c >= 4) {
        res = false;
      } else {
        tup >>= 2;
        tup += (uint64_t)(int64_t)(int32_t)((uint32_t)c << sh);
      }
    }
    if (!res && lastBad < s) {  // FromStringRL (:55-83): the last base in the highest bits
      tup = 0;
      for (int j = k - 1; j > 0; j--) {
        tup += (uint64_t)(base_code(q[s + j]) & 3);
        tup <<= 2;
      }
      tup += (uint64_t)(base_code(q[s]) & 3);
      res = true;
    }
    w[s] = tup;
  }
}

// ---------------------------------------------------------------------------
// Stable LSD radix sort of one task's records by bits [beginBit, endBit) of the key, done by one
// wavefront (8-bit digits, histogram in LDS, ranks inside a 64-record chunk from ballots).
// Stability is what makes it usable here: the k-mer keys enter ordered by position and are sorted by
// k-mer only; the fragments enter ordered by (x,y) and are sorted by y only.
// ---------------------------------------------------------------------------
template <bool kHasVals>
__device__ void wave_radix_sort(uint64_t *keysA, uint64_t *keysB, uint32_t *valsA, uint32_t *valsB, int n, int beginBit,
                                int endBit, unsigned *hist /* 512 LDS words of this wave */, bool &resultInB) {
  constexpr int kAhead = 8;  // 64-record chunks loaded ahead of the serial rank/scatter step
  const int lane = lane_id();
  const unsigned long long ltMask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  resultInB = false;
  if (n <= 0 || endBit <= beginBit) return;
  // histogram of the first digit; the later ones are counted while the previous pass scatters
  for (int b = lane; b < 256; b += 64) hist[b] = 0;
  __builtin_amdgcn_wave_barrier();
  for (int i0 = 0; i0 < n; i0 += 64 * kAhead) {
    uint64_t k[kAhead];
#pragma unroll
    for (int j = 0; j < kAhead; j++) {
      const int i = i0 + 64 * j + lane;
      k[j] = i < n ? keysA[i] : 0;
    }
#pragma unroll
    for (int j = 0; j < kAhead; j++)
      if (i0 + 64 * j + lane < n) atomicAdd(&hist[(unsigned)(k[j] >> beginBit) & 255u], 1u);
  }
  int pass = 0;
  for (int shift = beginBit; shift < endBit; shift += 8, pass++) {
    const uint64_t *kin = resultInB ? keysB : keysA;
    uint64_t *kout = resultInB ? keysA : keysB;
    const uint32_t *vin = resultInB ? valsB : valsA;
    uint32_t *vout = resultInB ? valsA : valsB;
    unsigned *cur = hist + (pass & 1) * 256, *nxt = hist + ((pass & 1) ^ 1) * 256;
    const bool more = shift + 8 < endBit;
    __builtin_amdgcn_wave_barrier();
    {  // exclusive prefix over the 256 bins: lane owns bins 4*lane .. 4*lane+3
      unsigned c0 = cur[4 * lane], c1 = cur[4 * lane + 1], c2 = cur[4 * lane + 2], c3 = cur[4 * lane + 3];
      const unsigned sum = c0 + c1 + c2 + c3;
      const unsigned excl = wave_prefix_sum(sum) - sum;
      __builtin_amdgcn_wave_barrier();
      cur[4 * lane] = excl;
      cur[4 * lane + 1] = excl + c0;
      cur[4 * lane + 2] = excl + c0 + c1;
      cur[4 * lane + 3] = excl + c0 + c1 + c2;
      for (int b = lane; b < 256; b += 64) nxt[b] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    uint64_t kn[kAhead];
    uint32_t vn[kAhead];
#pragma unroll
    for (int j = 0; j < kAhead; j++) {
      const int i = 64 * j + lane;
      kn[j] = i < n ? kin[i] : 0;
      vn[j] = 0;
      if (kHasVals && i < n) vn[j] = vin[i];
    }
    for (int i0 = 0; i0 < n; i0 += 64 * kAhead) {
      uint64_t k[kAhead];
      uint32_t v[kAhead];
#pragma unroll
      for (int j = 0; j < kAhead; j++) {
        k[j] = kn[j];
        v[j] = vn[j];
      }
      if (i0 + 64 * kAhead < n) {
#pragma unroll
        for (int j = 0; j < kAhead; j++) {
          const int i = i0 + 64 * (kAhead + j) + lane;
          kn[j] = i < n ? kin[i] : 0;
          if (kHasVals && i < n) vn[j] = vin[i];
        }
      }
#pragma unroll
      for (int j = 0; j < kAhead; j++) {
        if (i0 + 64 * j >= n) break;
        const bool on = i0 + 64 * j + lane < n;
        const uint64_t key = k[j];
        const unsigned d = (unsigned)(key >> shift) & 255u;
        unsigned long long same = __ballot(on);
        for (int b = 0; b < 8; b++) {
          const unsigned long long m = __ballot((d >> b) & 1u);
          same &= ((d >> b) & 1u) ? m : ~m;
        }
        const unsigned rank = (unsigned)__popcll(same & ltMask);
        unsigned base = 0;
        if (on) base = cur[d];
        __builtin_amdgcn_wave_barrier();
        if (on) {
          kout[base + rank] = key;
          if (kHasVals) vout[base + rank] = v[j];
          if (rank == 0) cur[d] = base + (unsigned)__popcll(same);
          if (more) atomicAdd(&nxt[(unsigned)(key >> (shift + 8)) & 255u], 1u);
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    resultInB = !resultInB;
  }
}

__global__ __launch_bounds__(256) void sdp_sort_keys_kernel(SdpLevelArrays lv, uint64_t *tmp) {
  __shared__ unsigned hist[4][512];
  const int nKeyed = *lv.keyCount;
  for (int e = wave_task_index(); e < nKeyed; e += (int)gridDim.x * 4) {
    const int idx = lv.keyList[e];
    const int lane = lane_id();
    const SdpTask t = lv.tasks[idx];
    uint64_t *a = lv.keys + lv.keyOff[idx], *b = tmp + lv.keyOff[idx];
    bool inB;
    wave_radix_sort<false>(a, b, nullptr, nullptr, t.tLen, kPosBits, kPosBits + 2 * key_bases(t.k) + 1, hist[threadIdx.x >> 6], inB);
    if (inB)
      for (int i = lane; i < t.tLen; i += 64) a[i] = b[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(256) void sdp_sort_yx_kernel(SdpLevelArrays lv, uint64_t *tmpKeys, uint32_t *tmpIdx) {
  __shared__ unsigned hist[4][512];
  for (int it = 0;; it++) {
    const int idx = general_task(lv, it);
    if (idx < 0) return;
    const int lane = lane_id();
    const int F = lv.nFrag[idx];
    const int64_t base = lv.fragOff[idx];
    bool inB;
    // keys are y << 32 | x, already ordered by (x,y): a stable sort on the 24 bits of y gives (y,x) order
    const int yBits = 32 - __builtin_clz((unsigned)max(lv.tasks[idx].tLen, 1));  // y < tLen
    wave_radix_sort<true>(lv.yxKeys + base, tmpKeys + base, lv.yxIdx + base, tmpIdx + base, F, 32, 32 + yBits,
                          hist[threadIdx.x >> 6], inB);
    if (inB)
      for (int i = lane; i < F; i += 64) lv.yxIdx[base + i] = tmpIdx[base + i];
    __builtin_amdgcn_wave_barrier();
  }
}

// small-word (<= 5) k-mer of s[p .. p+ks), or -1 when it holds a non-ACGT base
__device__ __forceinline__ int small_kmer(const uint8_t *s, int p, int ks) {
  int v = 0;
  for (int j = 0; j < ks; j++) {
    const int c = base_code(s[p + j]);
    if (c >= 4) return -1;
    v |= c << (2 * j);
  }
  return v;
}

__device__ __forceinline__ int lower_bound_u64(const uint64_t *a, int n, uint64_t key) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// ---------------------------------------------------------------------------
// sdp_match: StoreMatchingPositions for the prefix, the whole sequence and the suffix
// (SDPAlign.h:148-242).  kWrite = false counts, kWrite = true writes the fragments.
// ---------------------------------------------------------------------------
// prefix / suffix part of StoreMatchingPositions for one wavefront: every query position's small k-mer
// against every target position's.  With both parts <= 64 bases (sdpPrefixLength is 50 by default) the
// target k-mers sit one per lane and are read with v_readlane.
// small_part's count for parts of at most 64 bases whose codes are in registers already (lane p: the code of base p of either
// part, 4 beyond it): the number of (query position, target position) pairs with equal words of ks bases
__device__ __forceinline__ int small_part_count(int tc, int qc, int tn, int qn, int ks, int maxPerPos) {
  const int lane = lane_id();
  const int ny = tn - ks + 1;
  if (ks < 1 || ny <= 0 || qn < ks) return 0;
  int tv = 0, qv = 0, tAny = 0, qAny = 0;
  for (int j = 0; j < ks; j++) {
    tv |= (tc & 3) << (2 * j);
    qv |= (qc & 3) << (2 * j);
    tAny |= tc;
    qAny |= qc;
    tc = dpp<0x130, 0xf>(4, tc);
    qc = dpp<0x130, 0xf>(4, qc);
  }
  const int tk = (lane < ny && tAny < 4) ? tv : -2, qk = (lane + ks <= qn && qAny < 4) ? qv : -1;
  int cnt = 0;
  for (int y = 0; y < ny; y++) cnt += __builtin_amdgcn_readlane(tk, y) == qk;
  if (maxPerPos != 0 && cnt > maxPerPos) cnt = 0;
  return __builtin_amdgcn_readlane((int)wave_prefix_sum((unsigned)cnt), 63);
}
template <bool kWrite>
__device__ int small_part(const uint8_t *qs, int qn, const uint8_t *ts, int tn, int xOff, int yOff, int ks,
                          int maxPerPos, Frag *out, int written) {
  const int lane = lane_id();
  if (ks < 1) return 0;
  if (qn <= 64 && tn <= 64) {
    const int ny = tn - ks + 1;
    if (ny <= 0 || qn < ks) return 0;
    // The words of both parts from ONE byte load per lane and sequence: lane p reads base p, the bases behind it come off the
    // lanes above (DPP wave_shl:1, one step per base of the word).  small_kmer's loop -- a byte load per base with an exit at
    // the first base that is not ACGT -- is a chain of up to five DEPENDENT round trips to memory, and this function ran four
    // such chains per invocation (both sequences, prefix and suffix part): at the recursion levels, where a wavefront's whole
    // life is one tiny invocation, that chain was most of the 27 us a wavefront of the short-target matcher took.
    int tk = -2, qk = -1;
    {
      int tc = 4, qc = 4;  // (4: not ACGT, or beyond the part)
      if (lane < tn) tc = base_code(ts[lane]);
      if (lane < qn) qc = base_code(qs[lane]);
      int tv = 0, qv = 0, tAny = 0, qAny = 0;
      for (int j = 0; j < ks; j++) {
        tv |= (tc & 3) << (2 * j);
        qv |= (qc & 3) << (2 * j);
        tAny |= tc;
        qAny |= qc;
        tc = dpp<0x130, 0xf>(4, tc);  // lane i takes lane i + 1's, lane 63 "not ACGT"
        qc = dpp<0x130, 0xf>(4, qc);
      }
      if (lane < ny && tAny < 4) tk = tv;
      if (lane + ks <= qn && qAny < 4) qk = qv;
    }
    int cnt = 0;
    for (int y = 0; y < ny; y++) cnt += __builtin_amdgcn_readlane(tk, y) == qk;
    if (maxPerPos != 0 && cnt > maxPerPos) cnt = 0;
    const unsigned incl = wave_prefix_sum((unsigned)cnt);
    if (kWrite && cnt > 0) {
      int w = written + (int)incl - cnt;
      for (int y = 0; y < ny; y++)
        if (__builtin_amdgcn_readlane(tk, y) == qk) {
          Frag f;
          f.x = (uint32_t)(lane + xOff);
          f.y = (uint32_t)(y + yOff);
          f.len = (uint32_t)ks;
          f.pad = 0;
          out[w++] = f;
        }
    }
    return __builtin_amdgcn_readlane((int)incl, 63);
  }
  int total = 0;
  for (int x0 = 0; x0 + ks <= qn; x0 += 64) {
    const int x = x0 + lane;
    int cnt = 0, qk = -1;
    if (x + ks <= qn) {
      qk = small_kmer(qs, x, ks);
      if (qk >= 0)
        for (int y = 0; y + ks <= tn; y++) cnt += small_kmer(ts, y, ks) == qk;
      if (maxPerPos != 0 && cnt > maxPerPos) cnt = 0;
    }
    const unsigned incl = wave_prefix_sum((unsigned)cnt);
    if (kWrite && cnt > 0) {
      int w = written + total + (int)incl - cnt;
      for (int y = 0; y + ks <= tn; y++)
        if (small_kmer(ts, y, ks) == qk) {
          Frag f;
          f.x = (uint32_t)(x + xOff);
          f.y = (uint32_t)(y + yOff);
          f.len = (uint32_t)ks;
          f.pad = 0;
          out[w++] = f;
        }
    }
    total += __builtin_amdgcn_readlane((int)incl, 63);
  }
  return total;
}

// ---------------------------------------------------------------------------
// sdp_match_index: the same StoreMatchingPositions for LONG targets, one 256-thread workgroup per task
// and no k-mer sort at all.  The target's bases are packed into LDS (2 bits + a "not ACGT" bit) and the
// positions of its k-mers bucketed by a hash of the k-mer: 4096 buckets, a position is two bytes, the
// bucket bounds two bytes each -- 39 KB of LDS for a 12 kb target, four workgroups per CU.  A bucket is filled in
// whatever order the atomics fall (no ticket loop, one set of counters); every query k-mer walks its
// bucket -- three entries on average -- and compares the k-mer AT each position in the packed target with its own.
// Matches must come out ordered by (query position, target position), as the sorted tuple list of the
// reference yields them (TupleList.h:98-126, TupleMatching.h:37-61): up to three hits of a position are
// ordered in registers, the rare longer lists are sorted where they are written.
// ---------------------------------------------------------------------------
struct IdxLds {
  uint32_t *tCodes, *tBad, *cnt, *qStage, *part, *filt;
  uint8_t *lut;
  uint16_t *ent;   // bucket by bucket: target positions
  uint16_t *list;  // per wavefront: the positions of its chunk of the query whose k-mer passed the filter
};
// The count pass's FILTER: one bit per hash of a k-mer (kIdxFilterBits of them), set for every k-mer of the target.  A query
// position whose bit is clear matches nothing and never walks a bucket; the positions that pass (a seventh of the ones
// without a match for a 10 kb target and an 11-base word) are listed and walked 64 to a trip -- the walk runs as long as the LONGEST bucket among its
// 64 lanes, and with every position in it a wavefront spent five sixths of its walk steps on lanes with nothing left to do.
constexpr int kIdxFilterBits = 65536;
constexpr int kIdxSortCap = 16;  // matches of one query position the write pass orders in LDS (in the filter's words: 256 x 32 bytes)
static_assert(256 * kIdxSortCap * 2 <= kIdxFilterBits / 8, "the write pass sorts in the filter's words");
__host__ __device__ inline int idx_lds_words(int tCap, int *offs /* 9 */) {
  int at = 0;
  offs[0] = at, at += tCap / 16 + 4;          // tCodes
  offs[1] = at, at += tCap / 32 + 4;          // tBad
  offs[2] = at, at += kIdxBuckets / 2 + 4;    // cnt: two buckets per word {count / fill pointer / end : 16} x 2
  offs[3] = at, at += 4 * 104;                // qStage: per wavefront 66 code words + 34 bad words
  offs[4] = at, at += 16;                     // part
  offs[5] = at, at += 64;                     // lut
  offs[7] = at, at += kIdxFilterBits / 32;    // filt
  offs[8] = at, at += 4 * 256;                // list: per wavefront 512 positions of two bytes
  at = (at + 3) & ~3;
  offs[6] = at, at += tCap / 2 + 8;           // ent (16-byte aligned)
  return at;
}
int64_t idx_lds_bytes(int tCap) {
  int offs[9];
  return (int64_t)idx_lds_words(tCap, offs) * 4;
}
__device__ __forceinline__ int idx_filter(uint32_t kmer) {  // (exact for words of up to seven bases)
  return (int)((kmer ^ (kmer >> 16) ^ (kmer >> 9)) & (uint32_t)(kIdxFilterBits - 1));
}
// 16 consecutive bases starting at p0 -> 16 x 2-bit codes and 16 "not ACGT / outside" bits
__device__ __forceinline__ void stage16(const uint8_t *s, int len, int p0, const uint8_t *lut, uint32_t &codes,
                                        uint32_t &bad) {
  uint32_t w[4] = {0x4e4e4e4eu, 0x4e4e4e4eu, 0x4e4e4e4eu, 0x4e4e4e4eu};  // 'N'
  if (p0 + 16 <= len) {
    __builtin_memcpy(w, s + p0, 16);
  } else if (p0 < len) {
    // the last, partial chunk: sixteen UNCONDITIONAL byte loads (index clamped to the last base) and one wait -- as
    // "if (p0 + i < len) load" every load sat behind a branch and the compiler waited for each in turn, sixteen round
    // trips with the rest of the workgroup at the barrier behind it, once per packed sequence
    uint32_t b[16];
#pragma unroll
    for (int i = 0; i < 16; i++) b[i] = (uint32_t)s[min(p0 + i, len - 1)];
#pragma unroll
    for (int i = 0; i < 16; i++)
      if (p0 + i < len) w[i >> 2] = (w[i >> 2] & ~(0xffu << (8 * (i & 3)))) | (b[i] << (8 * (i & 3)));
  }
  codes = 0;
  bad = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const uint32_t c = lut[(w[i >> 2] >> (8 * (i & 3))) & 0xffu];
    codes |= (c & 3u) << (2 * i);
    bad |= (c >> 2) << i;
  }
}
__device__ __forceinline__ uint32_t code_window(const uint32_t *codes, int p, uint32_t kmerMask) {
  const int w = p >> 4;
  return __builtin_amdgcn_alignbit(codes[w + 1], codes[w], (uint32_t)(p & 15) * 2u) & kmerMask;
}
__device__ __forceinline__ uint32_t bad_window(const uint32_t *bad, int p, uint32_t kMask) {
  const int w = p >> 5;
  return __builtin_amdgcn_alignbit(bad[w + 1], bad[w], (uint32_t)(p & 31)) & kMask;
}
// bucket = low twelve bits of the k-mer mixed with the bits above them
__device__ __forceinline__ int idx_bucket(uint32_t kmer) {
  const uint32_t h = kmer >> 12;
  return (int)((kmer ^ h ^ (h >> 12)) & (kIdxBuckets - 1));
}
// the bounds of bucket b once the index is filled: cnt holds every bucket's END (two per word)
__device__ __forceinline__ int idx_end(const uint32_t *cnt, int b) { return (int)((cnt[b >> 1] >> ((b & 1) * 16)) & 0xffffu); }

__global__ __launch_bounds__(256) void sdp_classify_kernel(SdpLevelArrays lv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool big = false, mid = false;
  if (i < lv.n) {
    big = task_uses_index(lv.tasks[i].tLen, lv.tasks[i].k, lv.idxMaxT);
    mid = !big && lv.tasks[i].tLen >= 256;  // sdp_match_kernel<false, true>'s
  }
  const unsigned long long mask = __ballot(big);
  int base = 0;
  if (mask != 0ull && lane_id() == 0) base = atomicAdd(lv.bigCount, (int)__popcll(mask));
  base = __builtin_amdgcn_readfirstlane(base);
  if (big) lv.bigList[base + __popcll(mask & ((1ull << lane_id()) - 1ull))] = i;
  const bool keyed = i < lv.n && task_needs_keys(lv.tasks[i].tLen, lv.tasks[i].k, lv.idxMaxT);
  const unsigned long long maskK = __ballot(keyed);
  int baseK = 0;
  if (maskK != 0ull && lane_id() == 0) baseK = atomicAdd(lv.keyCount, (int)__popcll(maskK));
  baseK = __builtin_amdgcn_readfirstlane(baseK);
  if (keyed) lv.keyList[baseK + __popcll(maskK & ((1ull << lane_id()) - 1ull))] = i;
  const unsigned long long maskM = __ballot(mid);
  int baseM = 0;
  if (maskM != 0ull && lane_id() == 0) baseM = atomicAdd(lv.midCount, (int)__popcll(maskM));
  baseM = __builtin_amdgcn_readfirstlane(baseM);
  if (mid) lv.midList[baseM + __popcll(maskM & ((1ull << lane_id()) - 1ull))] = i;
}

// The count pass leaves, per query position, {match count : 16, the first three target positions : 3 x 16} in
// lv.hits; the write pass turns that into fragments and rebuilds the index only for a task in which some
// position has more than three matches (lv.idxFlag).
#ifdef BLASR_IDX_CLOCKS  // experiment: where a workgroup of the index matcher spends its cycles (make EXTRA=-DBLASR_IDX_CLOCKS)
__device__ unsigned long long g_idxClocks[16];
#define IDX_CLOCK(slot)                                                            \
  do {                                                                             \
    const unsigned long long now_ = clock64();                                     \
    if (threadIdx.x == 0) atomicAdd(&g_idxClocks[(kWrite ? 8 : 0) + (slot)], now_ - clk_); \
    clk_ = now_;                                                                   \
  } while (0)
__device__ unsigned long long g_tinyClocks[8];
#define TINY_CLOCK(slot)                                                                 \
  do {                                                                                   \
    const unsigned long long now_ = clock64();                                           \
    tph_[(slot)] += now_ - tclk_;  /* (in registers: the atomics wait for the end) */    \
    tclk_ = now_;                                                                        \
  } while (0)
void sdp_print_idx_clocks() {
  {
    unsigned long long t[8];
    if (hipMemcpyFromSymbol(t, HIP_SYMBOL(g_tinyClocks), sizeof(t)) == hipSuccess)
      fprintf(stderr, "[tiny clocks] count pass of the tiny-target matcher, per sampled invocation: task+setup %llu prefix %llu target words %llu query+walk %llu suffix %llu cycles; %llu sampled\n",
              t[0] / (t[6] + 1), t[1] / (t[6] + 1), t[2] / (t[6] + 1), t[3] / (t[6] + 1), t[5] / (t[6] + 1), t[6]);
  }
  unsigned long long h[16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_idxClocks), sizeof(h)) != hipSuccess) return;
  fprintf(stderr, "[idx clocks] count pass: pack %llu count %llu scan %llu scatter %llu query %llu tail %llu tasks %llu | write pass: pack %llu count %llu "
                  "scan %llu scatter %llu query %llu tail %llu tasks %llu (Mcycles of thread 0, summed over workgroups)\n",
          h[0] >> 20, h[1] >> 20, h[2] >> 20, h[3] >> 20, h[4] >> 20, h[5] >> 20, h[6], h[8] >> 20, h[9] >> 20, h[10] >> 20, h[11] >> 20, h[12] >> 20,
          h[13] >> 20, h[14]);
}
#else
#define IDX_CLOCK(slot) do { } while (0)
#define TINY_CLOCK(slot) do { } while (0)
#endif
template <bool kWrite>
__global__ __launch_bounds__(256) void sdp_match_index_kernel(SdpLevelArrays lv, const uint8_t *seq, int tCap, int minT) {
  extern __shared__ uint4 idxLdsRaw4[];  // 16-byte aligned
  uint32_t *idxLdsRaw = reinterpret_cast<uint32_t *>(idxLdsRaw4);
  IdxLds L;
  {
    int offs[9];
    idx_lds_words(tCap, offs);
    L.filt = idxLdsRaw + offs[7];
    L.list = reinterpret_cast<uint16_t *>(idxLdsRaw + offs[8]);
    L.tCodes = idxLdsRaw + offs[0];
    L.tBad = idxLdsRaw + offs[1];
    L.cnt = idxLdsRaw + offs[2];
    L.qStage = idxLdsRaw + offs[3];
    L.part = idxLdsRaw + offs[4];
    L.lut = reinterpret_cast<uint8_t *>(idxLdsRaw + offs[5]);
    L.ent = reinterpret_cast<uint16_t *>(idxLdsRaw + offs[6]);
  }
  const int tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
  L.lut[tid] = (uint8_t)base_code((unsigned)tid);  // 0..3, or 4 = bad bit
  const int nBig = *lv.bigCount;
  uint32_t *qCodes = L.qStage + wv * 104, *qBad = qCodes + 68;
  for (int e = blockIdx.x; e < nBig; e += gridDim.x) {
    const int idx = lv.bigList[e];
    const SdpTask t = lv.tasks[idx];
    if (t.tLen > tCap || t.tLen <= minT) continue;  // another size class takes it
    __syncthreads();
    const uint8_t *qSeq = seq + t.qOff, *tSeq = seq + t.tOff;
    const int k = t.k, ks = k < 5 ? k : 5;
    const uint32_t kmerMask = (1u << (2 * k)) - 1u, kMask = (1u << k) - 1u;
    if (kWrite && lv.idxFlag[idx] == 0) continue;  // sdp_match_index_copy_kernel's
    const bool build = true;
    // the filter pays where not every position of the query passes it: a target with fewer k-mers than the filter has bits
    // (and than there are k-mers of that length at all: a five-base word on a long target passes everywhere)
    const int kmerSpace = 2 * k >= 16 ? kIdxFilterBits : (1 << (2 * k));
    const bool useFilter = !kWrite && (int64_t)(t.tLen - k + 1) * lv.idxFilterX2 < (int64_t)2 * min(kmerSpace, kIdxFilterBits) && !lv.noIdxFilter;
#ifdef BLASR_IDX_CLOCKS
    unsigned long long clk_ = clock64();
    if (threadIdx.x == 0) atomicAdd(&g_idxClocks[(kWrite ? 8 : 0) + 6], 1ull);
#endif
    if (build) {
      // ---- the target, packed
      for (int wi = tid; wi < (t.tLen + 15) / 16 + 2; wi += 256) {
        uint32_t c, bd;
        stage16(tSeq, t.tLen, wi * 16, L.lut, c, bd);
        L.tCodes[wi] = c;
        reinterpret_cast<uint16_t *>(L.tBad)[wi] = (uint16_t)bd;
      }
      for (int i = tid; i < kIdxBuckets / 2 + 4; i += 256) L.cnt[i] = 0;
      if (useFilter)
        for (int i = tid; i < kIdxFilterBits / 32; i += 256) L.filt[i] = 0;
      __syncthreads();
      IDX_CLOCK(0);
      // ---- bucket sizes (two sixteen-bit counters to a word: a target has < 65536 positions, nothing carries over)
      const int nKmerT = t.tLen - k + 1;
      // (four positions per thread and trip: their LDS round trips overlap)
      constexpr int kBU = 4;
      for (int p = tid; p < nKmerT; p += 256 * kBU) {
        bool ok[kBU];
        uint32_t km[kBU];
#pragma unroll
        for (int u = 0; u < kBU; u++) {
          const int pu = p + 256 * u, pc = min(pu, nKmerT - 1);
          ok[u] = pu < nKmerT && bad_window(L.tBad, pc, kMask) == 0u;
          km[u] = code_window(L.tCodes, pc, kmerMask);
        }
#pragma unroll
        for (int u = 0; u < kBU; u++) {
          const int d = idx_bucket(km[u]);
          if (ok[u]) atomicAdd(&L.cnt[d >> 1], 1u << ((d & 1) * 16));
        }
        if (useFilter) {
#pragma unroll
          for (int u = 0; u < kBU; u++) {
            const int f = idx_filter(km[u]);
            if (ok[u]) atomicOr(&L.filt[f >> 5], 1u << (f & 31));
          }
        }
      }
      __syncthreads();
      IDX_CLOCK(1);
      {  // exclusive scan over the buckets, sixteen per thread: every bucket's counter becomes its first slot
        uint32_t w[8], sum = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          w[j] = L.cnt[8 * tid + j];
          sum += (w[j] & 0xffffu) + (w[j] >> 16);
        }
        const unsigned incl = wave_prefix_sum(sum);
        if (lane == 63) L.part[wv] = incl;
        __syncthreads();
        uint32_t at = incl - sum;
        for (int w0 = 0; w0 < wv; w0++) at += L.part[w0];
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const uint32_t lo = w[j] & 0xffffu, hi = w[j] >> 16;
          L.cnt[8 * tid + j] = at | ((at + lo) << 16);
          at += lo + hi;
        }
      }
      __syncthreads();
      IDX_CLOCK(2);
      // ---- scatter: a position takes the next slot of its bucket, whichever lane comes first
      for (int p = tid; p < nKmerT; p += 256 * kBU) {
        bool ok[kBU];
        int d[kBU];
        uint32_t old[kBU];
#pragma unroll
        for (int u = 0; u < kBU; u++) {
          const int pu = p + 256 * u, pc = min(pu, nKmerT - 1);
          ok[u] = pu < nKmerT && bad_window(L.tBad, pc, kMask) == 0u;
          d[u] = idx_bucket(code_window(L.tCodes, pc, kmerMask));
        }
#pragma unroll
        for (int u = 0; u < kBU; u++) {
          old[u] = 0;
          if (ok[u]) old[u] = atomicAdd(&L.cnt[d[u] >> 1], 1u << ((d[u] & 1) * 16));
        }
#pragma unroll
        for (int u = 0; u < kBU; u++)
          if (ok[u]) L.ent[(old[u] >> ((d[u] & 1) * 16)) & 0xffffu] = (uint16_t)(p + 256 * u);
      }
      __syncthreads();
      IDX_CLOCK(3);
    }
    const uint32_t *ends = L.cnt;  // the fill pointers have run to their buckets' ends
    // ---- the query against the index
    const int nKmerQ = max(t.qLen - k + 1, 0);
    const int perQ = (((nKmerQ + 3) / 4) + 63) & ~63;
    const int qlo = min(wv * perQ, nKmerQ), qhi = min(nKmerQ, qlo + perQ);
    uint64_t *hits = lv.hits + lv.qPosOff[idx];
    Frag *out = nullptr;
    int nP = 0, nMall = 0, wBase = 0;
    if (kWrite) {
      out = lv.raw + lv.fragOff[idx];
      nP = lv.nRaw[3 * idx];
      nMall = lv.nRaw[3 * idx + 1];
      wBase = nP;
      for (int w = 0; w < wv; w++) wBase += lv.idxWave[4 * idx + w];
    }
    int nM = 0;
    bool many = false;  // some position with more than three matches
    for (int c0 = qlo; c0 < qhi; c0 += 1024) {
      if (build) {
        __builtin_amdgcn_wave_barrier();
        for (int wi = lane; wi < 66; wi += 64) {
          uint32_t c, bd;
          stage16(qSeq, t.qLen, c0 + wi * 16, L.lut, c, bd);
          qCodes[wi] = c;
          reinterpret_cast<uint16_t *>(qBad)[wi] = (uint16_t)bd;
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (!kWrite && useFilter) {
        // Count pass with the filter.  Pass 1: every position of the chunk tests its bit; the ones that fail are done (no
        // match), the others are listed.  Pass 2: the listed positions, two blocks of 64 per trip, walk their buckets.
        uint16_t *list = L.list + wv * 512;
        for (int half = 0; half < 2; half++) {  // (the list holds 512 positions: the staged chunk in two halves)
        int nSurv = 0;
        // (four blocks of 64 positions per trip: their chains of LDS round trips -- the window of codes, the bad bits, the
        //  filter word -- are in flight together; one after the other a wavefront waited out each of them, at four per SIMD)
        for (int j = 8 * half; j < 8 * half + 8; j += 4) {
          if (c0 + 64 * j >= qhi) break;
          int rel4[4];
          bool on4[4];
          uint32_t km4[4], bad4[4], word4[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int x = c0 + 64 * (j + u) + lane;
            on4[u] = x < qhi;
            rel4[u] = min(x - c0, 1023);
            km4[u] = code_window(qCodes, rel4[u], kmerMask);
            bad4[u] = bad_window(qBad, rel4[u], kMask);
          }
#pragma unroll
          for (int u = 0; u < 4; u++) word4[u] = L.filt[idx_filter(km4[u]) >> 5];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int f = idx_filter(km4[u]);
            const bool surv = on4[u] && bad4[u] == 0u && ((word4[u] >> (f & 31)) & 1u) != 0u;
            const unsigned long long m = __ballot(surv);
            if (surv) list[nSurv + (int)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)rel4[u];
            if (on4[u] && !surv) hits[c0 + rel4[u]] = 0ull;
            nSurv += (int)__popcll(m);
          }
        }
        __builtin_amdgcn_wave_barrier();
        constexpr int kU = 2;
        for (int base = 0; base < nSurv; base += 64 * kU) {
          int xs[kU], bs[kU], be[kU], g[kU], cnt[kU];
          uint32_t km[kU], p0[kU], p1[kU], p2[kU];
          bool on[kU];
#pragma unroll
          for (int u = 0; u < kU; u++) {
            const int sIdx = base + 64 * u + lane;
            on[u] = sIdx < nSurv;
            const int rel = on[u] ? (int)list[sIdx] : 0;
            xs[u] = c0 + rel;
            km[u] = code_window(qCodes, rel, kmerMask);
            const int b = idx_bucket(km[u]);
            const int lo = idx_end(ends, max(b - 1, 0)), hi = idx_end(ends, b);
            bs[u] = on[u] ? (b ? lo : 0) : 0;
            be[u] = on[u] ? hi : 0;
            g[u] = bs[u] & ~3;
            cnt[u] = 0;
            p0[u] = p1[u] = p2[u] = 0xffffffffu;
          }
          while (g[0] < be[0] || g[1] < be[1]) {
#pragma unroll
            for (int u = 0; u < kU; u++) {
              const uint2 v = *reinterpret_cast<const uint2 *>(L.ent + (g[u] < be[u] ? g[u] : 0));  // (a finished walk: masked below)
              const uint32_t pos4[4] = {v.x & 0xffffu, v.x >> 16, v.y & 0xffffu, v.y >> 16};
#pragma unroll
              for (int e = 0; e < 4; e++) {
                const int i = g[u] + e;
                const bool in = i >= bs[u] && i < be[u];
                const uint32_t tp = in ? pos4[e] : 0u;
                const bool hit = in && code_window(L.tCodes, (int)tp, kmerMask) == km[u];
                const uint32_t v0 = hit ? tp : 0xffffffffu;
                cnt[u] += hit ? 1 : 0;
                const uint32_t a = min(p0[u], v0), r1 = max(p0[u], v0);
                const uint32_t m = min(p1[u], r1), r2 = max(p1[u], r1);
                p0[u] = a;
                p1[u] = m;
                p2[u] = min(p2[u], r2);
              }
              g[u] += 4;
            }
          }
#pragma unroll
          for (int u = 0; u < kU; u++) {
            int c = cnt[u];
            const uint64_t pos3 = ((uint64_t)(c > 0 ? p0[u] & 0xffffu : 0u) << 16) | ((uint64_t)(c > 1 ? p1[u] & 0xffffu : 0u) << 32) |
                                  ((uint64_t)(c > 2 ? p2[u] & 0xffffu : 0u) << 48);
            if (t.maxPerPos != 0 && c > t.maxPerPos) c = 0;
            if (c > 0xffff) c = 0xffff;  // cannot happen: a bucket holds < 65536 entries
            if (on[u]) hits[xs[u]] = c ? (pos3 | (uint64_t)c) : 0ull;
            many = many || c > 3;
            nM += c;
          }
        }
        __builtin_amdgcn_wave_barrier();
        }
        continue;
      }
      if (!kWrite) {
        // Count pass: TWO blocks of 64 positions per trip (the chains of LDS round trips -- bucket bounds, entries, the
        // k-mers at the entries -- of the two overlap), and the walk without a branch: a candidate that is outside the
        // bucket or holds another k-mer becomes "no position" (all ones) and goes through the same three-element insertion
        // network as a match -- which keeps the three SMALLEST matching positions in order; with more than three matches
        // only the count is used (the write pass walks the bucket again, lv.idxFlag).
        constexpr int kU = 2;
        for (int j = 0; j < 16; j += kU) {
          if (c0 + 64 * j >= qhi) break;
          int xs[kU], bs[kU], be[kU], g[kU], cnt[kU];
          uint32_t km[kU], p0[kU], p1[kU], p2[kU];
          bool on[kU];
#pragma unroll
          for (int u = 0; u < kU; u++) {
            xs[u] = c0 + 64 * (j + u) + lane;
            on[u] = xs[u] < qhi;
            const int rel = min(xs[u] - c0, 1023);
            const bool ok = on[u] && bad_window(qBad, rel, kMask) == 0u;
            km[u] = code_window(qCodes, rel, kmerMask);
            const int b = idx_bucket(km[u]);
            const int lo = idx_end(ends, max(b - 1, 0)), hi = idx_end(ends, b);
            bs[u] = ok ? (b ? lo : 0) : 0;
            be[u] = ok ? hi : 0;
            g[u] = bs[u] & ~3;
            cnt[u] = 0;
            p0[u] = p1[u] = p2[u] = 0xffffffffu;
          }
          while (g[0] < be[0] || g[1] < be[1]) {
#pragma unroll
            for (int u = 0; u < kU; u++) {
              const uint2 v = *reinterpret_cast<const uint2 *>(L.ent + (g[u] < be[u] ? g[u] : 0));  // (a finished walk: masked below)
              const uint32_t pos4[4] = {v.x & 0xffffu, v.x >> 16, v.y & 0xffffu, v.y >> 16};
#pragma unroll
              for (int e = 0; e < 4; e++) {
                const int i = g[u] + e;
                const bool in = i >= bs[u] && i < be[u];
                const uint32_t tp = in ? pos4[e] : 0u;
                const bool hit = in && code_window(L.tCodes, (int)tp, kmerMask) == km[u];
                const uint32_t v0 = hit ? tp : 0xffffffffu;
                cnt[u] += hit ? 1 : 0;
                const uint32_t a = min(p0[u], v0), r1 = max(p0[u], v0);
                const uint32_t m = min(p1[u], r1), r2 = max(p1[u], r1);
                p0[u] = a;
                p1[u] = m;
                p2[u] = min(p2[u], r2);
              }
              g[u] += 4;
            }
          }
#pragma unroll
          for (int u = 0; u < kU; u++) {
            int c = cnt[u];
            const uint64_t pos3 = ((uint64_t)(c > 0 ? p0[u] & 0xffffu : 0u) << 16) | ((uint64_t)(c > 1 ? p1[u] & 0xffffu : 0u) << 32) |
                                  ((uint64_t)(c > 2 ? p2[u] & 0xffffu : 0u) << 48);
            if (t.maxPerPos != 0 && c > t.maxPerPos) c = 0;
            if (c > 0xffff) c = 0xffff;  // cannot happen: a bucket holds < 65536 entries
            if (on[u]) hits[xs[u]] = c ? (pos3 | (uint64_t)c) : 0ull;
            many = many || c > 3;
            nM += c;
          }
        }
        continue;
      }
      for (int j = 0; j < 16; j++) {
        const int x0 = c0 + 64 * j;
        if (x0 >= qhi) break;
        const int x = x0 + lane;
        const bool on = x < qhi;
        const int rel = x - c0;
        uint64_t h = 0;
        int cnt = 0, bs = 0, be = 0;
        uint32_t km = 0;
        if (kWrite) {
          if (on) h = hits[x];
          cnt = (int)(h & 0xffffu);
        }
        if (build) {
          const bool ok = on && bad_window(qBad, rel, kMask) == 0u;
          km = code_window(qCodes, rel, kmerMask);
          if (ok && (!kWrite || cnt > 3)) {
            const int b = idx_bucket(km);
            bs = b ? idx_end(ends, b - 1) : 0;
            be = idx_end(ends, b);
          }
        }
        if (!kWrite) {
          // walk the bucket four positions at a time: a position matches iff the k-mer there is the query's.  The
          // first three matches are kept in increasing order (the bucket is not).
          uint32_t p0 = 0, p1 = 0, p2 = 0;
          for (int g = bs & ~3; g < be; g += 4) {
            const uint2 v = *reinterpret_cast<const uint2 *>(L.ent + g);
            const uint32_t pos4[4] = {v.x & 0xffffu, v.x >> 16, v.y & 0xffffu, v.y >> 16};
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const int i = g + u;
              const bool in = i >= bs && i < be;
              const uint32_t tp = in ? pos4[u] : 0u;
              if (in && code_window(L.tCodes, (int)tp, kmerMask) == km) {
                // (sorted insert into p0 <= p1 <= p2; beyond three only the count matters)
                if (cnt == 0) p0 = tp;
                else if (cnt == 1) { p1 = max(p0, tp); p0 = min(p0, tp); }
                else if (cnt == 2) {
                  const uint32_t lo = min(p0, tp), mid = min(max(p0, tp), p1), hi = max(p1, tp);
                  p0 = lo, p1 = mid, p2 = hi;
                }
                cnt++;
              }
            }
          }
          const uint64_t pos3 = ((uint64_t)p0 << 16) | ((uint64_t)p1 << 32) | ((uint64_t)p2 << 48);
          if (t.maxPerPos != 0 && cnt > t.maxPerPos) cnt = 0;
          if (cnt > 0xffff) cnt = 0xffff;  // cannot happen: a bucket holds < 65536 entries
          if (on) hits[x] = cnt ? (pos3 | (uint64_t)cnt) : 0ull;
          many = many || cnt > 3;
          nM += cnt;
        } else {
          const unsigned incl = wave_prefix_sum((unsigned)cnt);
          int w = wBase + (int)incl - cnt;
          Frag f;
          f.x = (uint32_t)x;
          f.len = (uint32_t)k;
          f.pad = 0;
          if (cnt > 0 && cnt <= 3) {
            for (int u = 0; u < cnt; u++) {
              f.y = (uint32_t)((h >> (16 * (u + 1))) & 0xffffu);
              out[w++] = f;
            }
          } else if (cnt > 3 && cnt <= kIdxSortCap) {
            // Up to sixteen matches of a position are ordered in LDS (the filter's words are free in the write pass: 32 bytes
            // per thread) and go out in one run.  With a seven-base word on a 16-64 kb target nearly every position of every
            // invocation comes this way, and ordering them in the output -- a dependent load of the fragment before per step
            // of the insertion -- made the write pass of configs[4]'s level 1 longer than its count pass.
            uint16_t *mine = reinterpret_cast<uint16_t *>(L.filt) + tid * kIdxSortCap;
            int n = 0;
            for (int i = bs; i < be; i++) {
              const uint32_t tp = L.ent[i];
              if (code_window(L.tCodes, (int)tp, kmerMask) == km && n < kIdxSortCap) {
                int at = n++;
                while (at > 0 && mine[at - 1] > tp) {
                  mine[at] = mine[at - 1];
                  at--;
                }
                mine[at] = (uint16_t)tp;
              }
            }
            for (int u = 0; u < n; u++) {
              f.y = (uint32_t)mine[u];
              out[w++] = f;
            }
          } else if (cnt > 3) {  // only in a task whose index was rebuilt
            const int w0 = w;
            for (int i = bs; i < be; i++) {
              const uint32_t tp = L.ent[i];
              if (code_window(L.tCodes, (int)tp, kmerMask) == km) {
                // insertion into the part already written: the fragments of a position go out by target position
                int at = w++;
                while (at > w0 && out[at - 1].y > tp) {
                  out[at] = out[at - 1];
                  at--;
                }
                f.y = tp;
                out[at] = f;
              }
            }
          }
          wBase += __builtin_amdgcn_readlane((int)incl, 63);
        }
      }
    }
    IDX_CLOCK(4);
    // ---- prefix and suffix parts (SDPAlign.h:148-182), wavefront 0 and 1
    const int prefixLength = min(t.tLen, t.prefix);
    const int suffixLength = min(t.tLen - prefixLength, t.prefix);
    const int suffixPos = t.tLen - suffixLength;
    const int qPrefixLength = min(t.qLen, t.prefix);
    const int qSuffixLength = min(t.qLen - qPrefixLength, t.prefix);
    const int qSuffixPos = t.qLen - qSuffixLength;
    if (!kWrite) {
      for (int off = 32; off > 0; off >>= 1) nM += __shfl_xor(nM, off);
      const bool anyMany = __ballot(many) != 0ull;
      __syncthreads();
      if (lane == 0) {
        L.part[8 + wv] = (uint32_t)nM;
        L.part[12 + wv] = anyMany ? 1u : 0u;
      }
      __syncthreads();
    }
    if (wv == 0) {
      const int n0 = small_part<kWrite>(qSeq, qPrefixLength, tSeq, prefixLength, 0, 0, ks, t.maxPerPos, out, 0);
      if (!kWrite && lane == 0) {
        lv.nRaw[3 * idx] = n0;
        lv.nRaw[3 * idx + 1] = (int)(L.part[8] + L.part[9] + L.part[10] + L.part[11]);
        lv.idxFlag[idx] = (int)(L.part[12] | L.part[13] | L.part[14] | L.part[15]);
      }
      // (what each wavefront's share of the query matched: the write pass starts from these instead of summing `hits`)
      if (!kWrite && lane < 4) lv.idxWave[4 * idx + lane] = (int)L.part[8 + lane];
    } else if (wv == 1) {
      const int n2 = small_part<kWrite>(qSeq + qSuffixPos, qSuffixLength, tSeq + suffixPos, suffixLength, qSuffixPos,
                                        suffixPos, ks, t.maxPerPos, out, nP + nMall);
      if (!kWrite && lane == 0) lv.nRaw[3 * idx + 2] = n2;
    }
    IDX_CLOCK(5);
  }
}

// Write pass for the invocations whose count pass found at most three matches per query position (lv.idxFlag
// = 0, nearly all of them): every fragment is already spelled out in lv.hits, so this is a copy that needs no
// index and no LDS -- a separate kernel so that its occupancy is not limited by the index's LDS footprint.
__global__ __launch_bounds__(256) void sdp_match_index_copy_kernel(SdpLevelArrays lv, const uint8_t *seq) {
  const int tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
  const int nBig = *lv.bigCount;
  for (int e = blockIdx.x; e < nBig; e += gridDim.x) {
    const int idx = lv.bigList[e];
    if (lv.idxFlag[idx] != 0) continue;  // sdp_match_index_kernel<true> rebuilds the index for it
    const SdpTask t = lv.tasks[idx];
    const uint8_t *qSeq = seq + t.qOff, *tSeq = seq + t.tOff;
    const int k = t.k, ks = k < 5 ? k : 5;
    const int nKmerQ = max(t.qLen - k + 1, 0);
    const int perQ = (((nKmerQ + 3) / 4) + 63) & ~63;
    const int qlo = min(wv * perQ, nKmerQ), qhi = min(nKmerQ, qlo + perQ);
    const uint64_t *hits = lv.hits + lv.qPosOff[idx];
    Frag *out = lv.raw + lv.fragOff[idx];
    const int nP = lv.nRaw[3 * idx], nMall = lv.nRaw[3 * idx + 1];
    int wBase = nP;
    for (int w = 0; w < wv; w++) wBase += lv.idxWave[4 * idx + w];
    for (int x0 = qlo; x0 < qhi; x0 += 64) {
      const int x = x0 + lane;
      uint64_t h = 0;
      if (x < qhi) h = hits[x];
      const int cnt = (int)(h & 0xffffu);
      const unsigned incl = wave_prefix_sum((unsigned)cnt);
      int w = wBase + (int)incl - cnt;
      Frag f;
      f.x = (uint32_t)x;
      f.len = (uint32_t)k;
      f.pad = 0;
      for (int u = 0; u < cnt; u++) {
        f.y = (uint32_t)((h >> (16 * (u + 1))) & 0xffffu);
        out[w++] = f;
      }
      wBase += __builtin_amdgcn_readlane((int)incl, 63);
    }
    // prefix and suffix parts (SDPAlign.h:148-182), wavefront 0 and 1
    const int prefixLength = min(t.tLen, t.prefix);
    const int suffixLength = min(t.tLen - prefixLength, t.prefix);
    const int suffixPos = t.tLen - suffixLength;
    const int qPrefixLength = min(t.qLen, t.prefix);
    const int qSuffixLength = min(t.qLen - qPrefixLength, t.prefix);
    const int qSuffixPos = t.qLen - qSuffixLength;
    if (wv == 0) small_part<true>(qSeq, qPrefixLength, tSeq, prefixLength, 0, 0, ks, t.maxPerPos, out, 0);
    else if (wv == 1)
      small_part<true>(qSeq + qSuffixPos, qSuffixLength, tSeq + suffixPos, suffixLength, qSuffixPos, suffixPos, ks,
                       t.maxPerPos, out, nP + nMall);
  }
}

constexpr int kBucketBits = 11;  // LDS bucket index over the leading bits of the k-mer (per wavefront: 8 KB)

// kBuckets: the count pass of targets >= 256 bases narrows its binary searches with a bucket index in LDS (32 KB
// per workgroup); everything else runs in the instantiation without it, at full occupancy.
// The bucket instantiation strides over ITS list (sdp_classify_kernel's midList) with a fixed grid: at the recursion levels it
// is one invocation in a hundred, and a workgroup with 32 KB of LDS per four invocations of the level, nearly all of which
// returned at once, took as long to get through the dispatcher (20 ms) as the other matchers take to do their work.
template <bool kWrite, bool kBuckets>
__device__ __forceinline__ void match_task(const SdpLevelArrays &lv, const uint8_t *seq, const int idx, uint32_t *bucketLds,
                                           const uint4 *tinyQuadsLds);
template <bool kWrite, bool kBuckets>
__global__ __launch_bounds__(256) void sdp_match_kernel(SdpLevelArrays lv, const uint8_t *seq) {
  __shared__ uint32_t bucketStart[kBuckets ? 4 : 1][kBuckets ? (1 << kBucketBits) + 1 : 1];
  __shared__ uint4 tinyT4[kBuckets ? 1 : 4][kBuckets ? 1 : kTinyT / 4];  // a tiny target's k-mers, by position
  uint32_t *bucketLds = bucketStart[kBuckets ? threadIdx.x >> 6 : 0];
  const uint4 *tinyLds = tinyT4[kBuckets ? 0 : threadIdx.x >> 6];
  if (kBuckets) {
    const int nMid = *lv.midCount;
    for (int e = wave_task_index(); e < nMid; e += (int)gridDim.x * 4) {
      match_task<kWrite, kBuckets>(lv, seq, lv.midList[e], bucketLds, tinyLds);
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  const int idx = wave_task_index();
  if (idx >= lv.n) return;
  match_task<kWrite, kBuckets>(lv, seq, idx, bucketLds, tinyLds);
}
template <bool kWrite, bool kBuckets>
__device__ __forceinline__ void match_task(const SdpLevelArrays &lv, const uint8_t *seq, const int idx, uint32_t *bucketLds,
                                           const uint4 *tinyQuadsLds) {
  const int lane = lane_id();
#ifdef BLASR_IDX_CLOCKS
  unsigned long long tclk_ = clock64(), tph_[6] = {0, 0, 0, 0, 0, 0};
#endif
  const SdpTask t = lv.tasks[idx];
  if (task_uses_index(t.tLen, t.k, lv.idxMaxT)) return;  // sdp_match_index_kernel's
  if ((!kWrite && t.tLen >= 256) != kBuckets) return;    // the other instantiation's
#ifdef BLASR_IDX_CLOCKS
  const bool tinyClk_ = !kWrite && !kBuckets && task_is_tiny(t.tLen, t.k);
#endif
  const uint8_t *qSeq = seq + t.qOff, *tSeq = seq + t.tOff;
  const uint64_t *keys = lv.keys + lv.keyOff[idx];
  const int k = t.k, ks = k < 5 ? k : 5;
  // SDPAlign.h:148-182
  const int prefixLength = min(t.tLen, t.prefix);
  const int suffixLength = min(t.tLen - prefixLength, t.prefix);
  const int suffixPos = t.tLen - suffixLength;
  const int qPrefixLength = min(t.qLen, t.prefix);
  const int qSuffixLength = min(t.qLen - qPrefixLength, t.prefix);
  const int qSuffixPos = t.qLen - qSuffixLength;
  Frag *out = kWrite ? lv.raw + lv.fragOff[idx] : nullptr;
  int written = 0;  // uniform running count

  TINY_CLOCK(0);
  if (!kWrite && !kBuckets && task_is_tiny(t.tLen, k) && t.prefix <= 64 && t.tLen > 0 && t.qLen > 0) {
    // ---- The count pass of a TINY invocation (target of at most 128 bases: nine invocations in ten at the recursion levels,
    // a wavefront each, and what those levels' match phase waits for).  Everything it reads of the two sequences -- the
    // target's 128 bases, the query's first 128, the two suffix parts -- is six byte loads per lane, ISSUED TOGETHER: one
    // round trip, where the general path below has one per part (prefix, target words, query words, suffix), each behind the
    // arithmetic of the part before.  The prefix parts are the sequences' first bases: no loads of their own.
    const int t0 = code_or_none(tSeq, t.tLen, lane), t1 = code_or_none(tSeq, t.tLen, 64 + lane);
    int q0 = code_or_none(qSeq, t.qLen, lane), q1 = code_or_none(qSeq, t.qLen, 64 + lane);
    // (an empty suffix part reads a base of the sequence itself: its code is not used)
    const int tS = code_or_none(suffixLength > 0 ? tSeq + suffixPos : tSeq, suffixLength, lane);
    const int qS = code_or_none(qSuffixLength > 0 ? qSeq + qSuffixPos : qSeq, qSuffixLength, lane);
    uint64_t *hitsT = lv.hits + lv.qPosOff[idx];
    uint32_t *tinyT = reinterpret_cast<uint32_t *>(const_cast<uint4 *>(tinyQuadsLds));
    const int nPre = small_part_count(lane < prefixLength ? t0 : 4, lane < qPrefixLength ? q0 : 4, prefixLength, qPrefixLength, ks,
                                      t.maxPerPos);
    TINY_CLOCK(1);
    {
      bool ok0, ok1;
      const uint32_t tk0 = (uint32_t)words_from_codes(t0, t1, k, ok0), tk1 = (uint32_t)words_from_codes(t1, 4, k, ok1);
      tinyT[lane] = ok0 ? tk0 : 0xffffffffu;
      tinyT[lane + 64] = ok1 ? tk1 : 0xffffffffu;
      __builtin_amdgcn_wave_barrier();
    }
    TINY_CLOCK(2);
    const int nT = max(t.tLen - k + 1, 0);
    int nMid = 0;
    for (int x0 = 0; x0 + k <= t.qLen; x0 += 64) {
      const int x = x0 + lane;
      bool okq;
      uint32_t qk = (uint32_t)words_from_codes(q0, q1, k, okq);
      q0 = q1;  // (the next 64 positions: one more load per trip, rare -- a tiny target's query is seldom longer than 128 bases)
      q1 = x0 + 128 < t.qLen ? code_or_none(qSeq, t.qLen, x0 + 128 + lane) : 4;
      if (!okq) qk = 0xfffffffeu;  // (matches nothing)
      int cnt = 0;
      uint64_t pos3 = 0;
      for (int y0 = 0; y0 < nT; y0 += 4) {
        const uint4 tv = tinyQuadsLds[y0 >> 2];
        const bool e0 = tv.x == qk, e1 = tv.y == qk, e2 = tv.z == qk, e3 = tv.w == qk;
        if (__ballot(e0 || e1 || e2 || e3) != 0ull) {
          if (e0) {
            if (cnt < 3) pos3 |= (uint64_t)y0 << (16 * (cnt + 1));
            cnt++;
          }
          if (e1) {
            if (cnt < 3) pos3 |= (uint64_t)(y0 + 1) << (16 * (cnt + 1));
            cnt++;
          }
          if (e2) {
            if (cnt < 3) pos3 |= (uint64_t)(y0 + 2) << (16 * (cnt + 1));
            cnt++;
          }
          if (e3) {
            if (cnt < 3) pos3 |= (uint64_t)(y0 + 3) << (16 * (cnt + 1));
            cnt++;
          }
        }
      }
      if (!okq || (t.maxPerPos != 0 && cnt > t.maxPerPos)) cnt = 0;
      if (x + k <= t.qLen) hitsT[x] = cnt ? (pos3 | (uint64_t)cnt) : 0ull;
      nMid += __builtin_amdgcn_readlane((int)wave_prefix_sum((unsigned)cnt), 63);
    }
    TINY_CLOCK(3);
    const int nSuf = small_part_count(tS, qS, suffixLength, qSuffixLength, ks, t.maxPerPos);
    if (lane == 0) {
      lv.nRaw[3 * idx] = nPre;
      lv.nRaw[3 * idx + 1] = nMid;
      lv.nRaw[3 * idx + 2] = nSuf;
    }
    TINY_CLOCK(5);
#ifdef BLASR_IDX_CLOCKS
    if (tinyClk_ && lane == 0 && (idx & 63) == 0) {
      for (int c_ = 0; c_ < 6; c_++) atomicAdd(&g_tinyClocks[c_], tph_[c_]);
      atomicAdd(&g_tinyClocks[6], 1ull);
    }
#endif
    return;
  }
  const int nP = small_part<kWrite>(qSeq, qPrefixLength, tSeq, prefixLength, 0, 0, ks, t.maxPerPos, out, written);
  written += nP;
  TINY_CLOCK(1);

  // middle (part 1): the whole query against the sorted k-mer keys of the whole target.
  // For long targets the binary search is narrowed by a bucket index over the leading k-mer bits that the
  // wavefront builds in LDS from the sorted keys (one pass), instead of ~2 x log2(tLen) dependent HBM reads.
  const int kk = key_bases(k);  // (the bases of a word the keys hold; a longer word's other bases are compared below)
  const int bBits = 2 * kk < kBucketBits ? 2 * kk : kBucketBits;
  const int bShift = kPosBits + 2 * kk - bBits;  // key >> bShift = bucket (invalid keys land beyond the last bucket)
  const bool useBuckets = kBuckets;  // (t.tLen <= kTinyT takes the register path below)
  uint32_t *bucket = bucketLds;
  if (useBuckets) {
    const int nB = 1 << bBits;
    for (int b = lane; b <= nB; b += 64) bucket[b] = (uint32_t)t.tLen;
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < t.tLen; i += 64) {
      const int cur = (int)min((uint64_t)nB, keys[i] >> bShift);
      const int prev = i == 0 ? -1 : (int)min((uint64_t)nB, keys[i - 1] >> bShift);
      for (int b = prev + 1; b <= cur; b++) bucket[b] = (uint32_t)i;  // first key of every bucket in (prev, cur]
    }
    __builtin_amdgcn_wave_barrier();
  }
  int nM = 0;
  uint64_t *hits = lv.hits + lv.qPosOff[idx];  // per query position: first matching key | count << 32
  const uint64_t *qWords = k >= kKeyBases ? lv.qWords + lv.qPosOff[idx] : nullptr;
  const bool tiny = !kBuckets && task_is_tiny(t.tLen, k);  // (the instantiation with the bucket index only sees targets >= 256)
  if (tiny) {
    // the target's k-mers in registers (positions lane and lane + 64); every query k-mer is compared with
    // all of them in position order, which is the order of the sorted tuple list
    // (the words of 64 consecutive positions at a time: words_at, one load per lane)
    // count pass: per query position {count : 16, the first three target positions : 3 x 16} go to `hits`, so
    // the write pass is a copy unless some position has more than three matches
    // The target's k-mers go to LDS once (0xffffffff where there is none: a k-mer has at most 22 bits) and every query
    // lane walks them four to a read, in position order; lanes meet a branch only where some lane has a match.  (With the
    // k-mers in registers and a v_readlane per target position the loop was a chain of twelve dependent scalar
    // instructions per position.)
    bool haveT = false;
    uint32_t *tinyT = reinterpret_cast<uint32_t *>(const_cast<uint4 *>(tinyQuadsLds));
    const uint4 *tinyQuads = tinyQuadsLds;
    const int nT = max(t.tLen - k + 1, 0);
    for (int x0 = 0; x0 + k <= t.qLen; x0 += 64) {
      const int x = x0 + lane;
      const bool inQ = x + k <= t.qLen;
      uint64_t h = 0;
      int cnt = 0;
      bool replay = !kWrite;
      if (kWrite) {
        if (inQ) h = hits[x];
        cnt = (int)(h & 0xffffu);
        replay = __ballot(cnt > 3) != 0ull;
      }
      uint32_t qk = 0;
      if (replay) {
        if (!haveT) {
          bool ok0, ok1;
          const uint32_t tk0 = (uint32_t)words_at(tSeq, t.tLen, 0, k, ok0);
          const uint32_t tk1 = (uint32_t)words_at(tSeq, t.tLen, 64, k, ok1);
          tinyT[lane] = ok0 ? tk0 : 0xffffffffu;
          tinyT[lane + 64] = ok1 ? tk1 : 0xffffffffu;
          __builtin_amdgcn_wave_barrier();
          haveT = true;
          TINY_CLOCK(2);
        }
        bool okq;
        qk = (uint32_t)words_at(qSeq, t.qLen, x0, k, okq);
        if (!okq) qk = 0xfffffffeu;  // (matches nothing)
        if (!kWrite) {
          uint64_t pos3 = 0;
          for (int y0 = 0; y0 < nT; y0 += 4) {
            const uint4 tv = tinyQuads[y0 >> 2];
            const bool e0 = tv.x == qk, e1 = tv.y == qk, e2 = tv.z == qk, e3 = tv.w == qk;
            if (__ballot(e0 || e1 || e2 || e3) != 0ull) {
              if (e0) {
                if (cnt < 3) pos3 |= (uint64_t)y0 << (16 * (cnt + 1));
                cnt++;
              }
              if (e1) {
                if (cnt < 3) pos3 |= (uint64_t)(y0 + 1) << (16 * (cnt + 1));
                cnt++;
              }
              if (e2) {
                if (cnt < 3) pos3 |= (uint64_t)(y0 + 2) << (16 * (cnt + 1));
                cnt++;
              }
              if (e3) {
                if (cnt < 3) pos3 |= (uint64_t)(y0 + 3) << (16 * (cnt + 1));
                cnt++;
              }
            }
          }
          if (!okq || (t.maxPerPos != 0 && cnt > t.maxPerPos)) cnt = 0;
          if (inQ) hits[x] = cnt ? (pos3 | (uint64_t)cnt) : 0ull;
        }
      }
      const unsigned incl = wave_prefix_sum((unsigned)cnt);
      if (kWrite && cnt > 0) {
        int w = written + nM + (int)incl - cnt;
        Frag f;
        f.x = (uint32_t)x;
        f.len = (uint32_t)k;
        f.pad = 0;
        if (cnt <= 3) {
          for (int u = 0; u < cnt; u++) {
            f.y = (uint32_t)((h >> (16 * (u + 1))) & 0xffffu);
            out[w++] = f;
          }
        } else {
          for (int y = 0; y < nT; y++)
            if (tinyT[y] == qk) {
              f.y = (uint32_t)y;
              out[w++] = f;
            }
        }
      }
      nM += __builtin_amdgcn_readlane((int)incl, 63);
    }
  }
  // The count pass over the sorted keys, TWO query positions per lane and trip and their four binary searches (first key
  // >= k-mer, first key > k-mer, for either position) in lockstep: every step of a search is a dependent read of `keys`,
  // and one position's two searches one after the other were sixteen round trips per trip where the range is a whole
  // 255-base target -- which is what these kernels' time was (VALU issue 0.02 - 0.26).  Counting does not care about the
  // order of the positions; the write pass below keeps its own.
  for (int x0 = 0; !kWrite && !tiny && x0 + k <= t.qLen; x0 += 128) {
    int sLo[4], sHi[4];
    uint64_t sKey[4];
    bool okP[2];
    uint32_t qHigh[2] = {0u, 0u};
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int x = x0 + 64 * u + lane;
      bool ok;
      uint64_t kmer = words_at(qSeq, t.qLen, x0 + 64 * u, k, ok);
      if (kk < 32) kmer &= (1ull << (2 * kk)) - 1ull;
      if (k >= kKeyBases) {  // the reference's tuple of this position, not the word (sdp_query_words_kernel)
        const uint64_t word = ok ? qWords[x] : 0ull;
        kmer = word & 0xffffffffull;
        qHigh[u] = (uint32_t)(word >> 32);
      }
      int rLo = 0, rHi = ok ? t.tLen : 0;
      if (useBuckets && ok) {
        const int b = (int)(kmer >> (2 * kk - bBits));
        rLo = (int)bucket[b];
        rHi = (int)bucket[b + 1];
      }
      okP[u] = ok;
      sLo[2 * u] = sLo[2 * u + 1] = rLo;
      sHi[2 * u] = sHi[2 * u + 1] = rHi;
      sKey[2 * u] = kmer << kPosBits;
      sKey[2 * u + 1] = (kmer + 1) << kPosBits;
    }
    for (;;) {
      bool any = false;
      uint64_t v[4];
      int mid[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++) {
        mid[s4] = (sLo[s4] + sHi[s4]) >> 1;
        const bool open = sLo[s4] < sHi[s4];
        any = any || open;
        v[s4] = open ? keys[mid[s4]] : 0ull;  // (four reads in flight)
      }
      if (__ballot(any) == 0ull) break;
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++)
        if (sLo[s4] < sHi[s4]) {
          if (v[s4] < sKey[s4]) sLo[s4] = mid[s4] + 1;
          else sHi[s4] = mid[s4];
        }
    }
    int cntBoth = 0;
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int x = x0 + 64 * u + lane;
      int cnt = okP[u] ? sLo[2 * u + 1] - sLo[2 * u] : 0;
      if (k >= kKeyBases && cnt > 0) {  // equal keys: the tuples agree in their low halves -- count those that agree throughout
        int pass = 0;
        for (int j = 0; j < cnt; j++) {
          const int y = (int)(keys[sLo[2 * u] + j] & ((1ull << kPosBits) - 1));
          pass += target_tuple_high(tSeq + y, k) == qHigh[u] ? 1 : 0;
        }
        cnt = pass;
      }
      if (t.maxPerPos != 0 && cnt > t.maxPerPos) cnt = 0;
      if (x + k <= t.qLen) hits[x] = (uint64_t)(uint32_t)(okP[u] ? sLo[2 * u] : 0) | ((uint64_t)(uint32_t)cnt << 32);
      cntBoth += cnt;
    }
    nM += (int)__builtin_amdgcn_readlane((int)wave_prefix_sum((unsigned)cntBoth), 63);
  }
  for (int x0 = 0; kWrite && !tiny && x0 + k <= t.qLen; x0 += 64) {
    const int x = x0 + lane;
    int lo = 0, cnt = 0;
    if (x + k <= t.qLen) {
      if (kWrite) {  // the count pass already searched
        const uint64_t h = hits[x];
        lo = (int)(uint32_t)h;
        cnt = (int)(h >> 32);
      } else {
        uint64_t kmer = 0;
        bool ok = true;
        for (int j = 0; j < k; j++) {
          const int c = base_code(qSeq[x + j]);
          ok = ok && c < 4;
          if (j < kk) kmer |= (uint64_t)(c & 3) << (2 * j);
        }
        if (ok) {
          int rLo = 0, rHi = t.tLen;
          if (useBuckets) {
            const int b = (int)(kmer >> (2 * kk - bBits));
            rLo = (int)bucket[b];
            rHi = (int)bucket[b + 1];
          }
          lo = rLo + lower_bound_u64(keys + rLo, rHi - rLo, kmer << kPosBits);
          const int hi = lo + lower_bound_u64(keys + lo, rHi - lo, (kmer + 1) << kPosBits);
          cnt = hi - lo;
          if (t.maxPerPos != 0 && cnt > t.maxPerPos) cnt = 0;
        }
        hits[x] = (uint64_t)(uint32_t)lo | ((uint64_t)(uint32_t)cnt << 32);
      }
    }
    const unsigned incl = wave_prefix_sum((unsigned)cnt);
    if (kWrite && cnt > 0) {
      int w = written + nM + (int)incl - cnt;
      // (a word of sixteen bases and more: `lo` is the first equal KEY and `cnt` the keys behind it whose tuples agree
      //  throughout -- the count pass found that many, so the walk ends)
      const uint32_t high = k >= kKeyBases ? (uint32_t)(qWords[x] >> 32) : 0u;
      for (int j = 0, done = 0; done < cnt; j++) {
        Frag f;
        f.x = (uint32_t)x;
        f.y = (uint32_t)(keys[lo + j] & ((1ull << kPosBits) - 1));
        f.len = (uint32_t)k;
        f.pad = 0;
        if (k >= kKeyBases && target_tuple_high(tSeq + f.y, k) != high) continue;
        out[w++] = f;
        done++;
      }
    }
    nM += __builtin_amdgcn_readlane((int)incl, 63);
  }
  written += nM;
  TINY_CLOCK(3);
  const int nS = small_part<kWrite>(qSeq + qSuffixPos, qSuffixLength, tSeq + suffixPos, suffixLength, qSuffixPos,
                                    suffixPos, ks, t.maxPerPos, out, written);
  TINY_CLOCK(5);
#ifdef BLASR_IDX_CLOCKS
  if (tinyClk_ && lane == 0 && (idx & 63) == 0) {  // (a sample: one invocation in 64)
    for (int c_ = 0; c_ < 6; c_++) atomicAdd(&g_tinyClocks[c_], tph_[c_]);
    atomicAdd(&g_tinyClocks[6], 1ull);
  }
#endif
  if (!kWrite && lane == 0) {
    lv.nRaw[3 * idx] = nP;
    lv.nRaw[3 * idx + 1] = nM;
    lv.nRaw[3 * idx + 2] = nS;
  }
}

// ---------------------------------------------------------------------------
// sdp_order: std::sort by (x,y) + keep the first of each (x,y) (SDPAlign.h:256-269)
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t frag_key(const Frag &f) { return ((uint64_t)f.x << 32) | f.y; }
__device__ __forceinline__ bool frag_less(const Frag &a, const Frag &b) { return frag_key(a) < frag_key(b); }

__device__ int lower_bound_frag(const Frag *a, int n, uint64_t key) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (frag_key(a[mid]) < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// libstdc++ (GCC 11.4) std::sort schedule, replayed by one lane.  Only reached when the prefix /
// middle / suffix lists share an (x,y): which duplicate survives depends on this exact schedule.
// T is either a fragment in global memory or, for the usual sizes, a packed (rank << 16 | index) word
// in LDS, where the rank orders exactly like (x,y) and equal keys have equal ranks.
constexpr int kTieLdsSmall = 2560, kTieLdsCap = 12288;  // fragments per task replayed in LDS (two size classes)
struct FragLess {
  __device__ __forceinline__ bool operator()(const Frag &a, const Frag &b) const { return frag_less(a, b); }
};
struct RankLess {
  __device__ __forceinline__ bool operator()(uint32_t a, uint32_t b) const { return (a >> 16) < (b >> 16); }
};
__device__ __forceinline__ void order_task(const SdpLevelArrays &lv, const int idx) {
  const int lane = lane_id();
  const int nP = lv.nRaw[3 * idx], nM = lv.nRaw[3 * idx + 1], nS = lv.nRaw[3 * idx + 2];
  const Frag *P = lv.raw + lv.fragOff[idx], *M = P + nP, *S = M + nM;
  Frag *out = lv.frags + lv.fragOff[idx];
  const int nAll = nP + nM + nS;
  // duplicates can only occur ACROSS the three lists (each list has unique (x,y))
  bool tie = false;
  for (int i = lane; i < nP; i += 64) {
    const uint64_t key = frag_key(P[i]);
    int j = lower_bound_frag(M, nM, key);
    if (j < nM && frag_key(M[j]) == key) tie = true;
    j = lower_bound_frag(S, nS, key);
    if (j < nS && frag_key(S[j]) == key) tie = true;
  }
  for (int i = lane; i < nS; i += 64) {
    const uint64_t key = frag_key(S[i]);
    const int j = lower_bound_frag(M, nM, key);
    if (j < nM && frag_key(M[j]) == key) tie = true;
  }
  if (__ballot(tie) == 0ull) {
    // all keys distinct: the sorted order is unique, place every fragment at its rank
    for (int i = lane; i < nP; i += 64) {
      const uint64_t key = frag_key(P[i]);
      out[i + lower_bound_frag(M, nM, key) + lower_bound_frag(S, nS, key)] = P[i];
    }
    for (int i = lane; i < nM; i += 64) {
      const uint64_t key = frag_key(M[i]);
      const int shift = (nP ? lower_bound_frag(P, nP, key) : 0) + (nS ? lower_bound_frag(S, nS, key) : 0);
      out[i + shift] = M[i];
    }
    for (int i = lane; i < nS; i += 64) {
      const uint64_t key = frag_key(S[i]);
      out[i + lower_bound_frag(P, nP, key) + lower_bound_frag(M, nM, key)] = S[i];
    }
    if (lane == 0) {
      lv.nFrag[idx] = nAll;
      if (lv.cleanList != nullptr) lv.cleanList[atomicAdd(lv.cleanCount, 1)] = idx;
    }
  } else if (lane == 0) {
    lv.tieList[atomicAdd(lv.tieCount, 1)] = idx;  // resolved by sdp_order_ties_kernel
  }
}
__global__ __launch_bounds__(256) void sdp_order_kernel(SdpLevelArrays lv) {
  for (int it = 0;; it++) {
    const int idx = general_task(lv, it);  // (not sdp_small_kernel's)
    if (idx < 0) return;
    order_task(lv, idx);
  }
}

// The tasks whose lists share an (x,y): replay std::sort's schedule to learn which duplicate comes first.
// One wavefront per workgroup, the fragments as packed (rank << 16 | index) words in LDS.  The replay is
// parallel over the recursion tree: std::__introsort_loop's two sub-ranges after a partition are disjoint
// and never touched from outside again, so every lane replays one range per round; a range of <= 16
// elements is final for the loop and receives its share of __final_insertion_sort at once (elements
// left of a cut are <= those right of it, so the linear inserts never cross a cut).
__device__ __forceinline__ bool rank_less(uint32_t a, uint32_t b) { return (a >> 16) < (b >> 16); }
__device__ __forceinline__ void lds_insertion_sort(uint32_t *v, int f, int l) {
  for (int i = f + 1; i < l; i++) {
    const uint32_t val = v[i];
    int j = i;
    while (j > f && rank_less(val, v[j - 1])) {
      v[j] = v[j - 1];
      j--;
    }
    v[j] = val;
  }
}
__device__ __forceinline__ int lds_partition(uint32_t *v, int f, int l) {  // returns the cut
  const int mid = f + (l - f) / 2;
  // __move_median_to_first(f, f+1, mid, l-1)
  const uint32_t a = v[f + 1], b = v[mid], c = v[l - 1];
  int m;
  if (rank_less(a, b)) m = rank_less(b, c) ? mid : (rank_less(a, c) ? l - 1 : f + 1);
  else m = rank_less(a, c) ? f + 1 : (rank_less(b, c) ? l - 1 : mid);
  const uint32_t pivot = v[m];
  v[m] = v[f];
  v[f] = pivot;
  // __unguarded_partition(f+1, l, f)
  int lo = f + 1, hi = l;
  for (;;) {
    uint32_t x = v[lo];
    while (rank_less(x, pivot)) x = v[++lo];
    uint32_t y = v[--hi];
    while (rank_less(pivot, y)) y = v[--hi];
    if (!(lo < hi)) return lo;
    v[lo] = y;
    v[hi] = x;
    ++lo;
  }
}
// __unguarded_partition of a LONG range by the whole wavefront.  The serial two-pointer loop swaps the k-th
// element from the left that is not below the pivot (position L[k]) with the k-th from the right that is not
// above it (R[k]) for as long as L[k] < R[k]; both pointers only ever look at elements no swap has touched, so L
// and R can be listed from the original array in parallel, and the loop returns min(L[K], R[K-1]) at the first K
// with L[K] >= R[K].  `pos` holds the two position lists.
__device__ int coop_partition(uint32_t *v, int f, int l, unsigned short *pos, int cap) {
  const int lane = lane_id();
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int mid = f + (l - f) / 2;
  // __move_median_to_first(f, f+1, mid, l-1): every lane evaluates it, lane 0 swaps
  const uint32_t a = v[f + 1], b = v[mid], c = v[l - 1];
  int m;
  if (rank_less(a, b)) m = rank_less(b, c) ? mid : (rank_less(a, c) ? l - 1 : f + 1);
  else m = rank_less(a, c) ? f + 1 : (rank_less(b, c) ? l - 1 : mid);
  const uint32_t pivot = v[m];
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    v[m] = v[f];
    v[f] = pivot;
  }
  __builtin_amdgcn_wave_barrier();
  const uint32_t p = pivot >> 16;
  unsigned short *L = pos, *R = pos + cap;
  int nL = 0, nR = 0;
  for (int base = f + 1; base < l; base += 64) {
    const int i = base + lane;
    const bool flag = i < l && (v[i] >> 16) >= p;
    const unsigned long long mask = __ballot(flag);
    if (flag) L[nL + __popcll(mask & lt)] = (unsigned short)i;
    nL += (int)__popcll(mask);
  }
  for (int top = l - 1; top > f; top -= 64) {
    const int i = top - lane;
    const bool flag = i > f && (v[i] >> 16) <= p;
    const unsigned long long mask = __ballot(flag);
    if (flag) R[nR + __popcll(mask & lt)] = (unsigned short)i;
    nR += (int)__popcll(mask);
  }
  __builtin_amdgcn_wave_barrier();
  const int nMin = min(nL, nR);
  int K = 0;
  for (int k0 = 0; k0 < nMin; k0 += 64) {
    const int k = k0 + lane;
    const unsigned long long mask = __ballot(k < nMin && L[k] < R[k]);
    K += (int)__popcll(mask);
    if (mask != ~0ull) break;  // L ascends and R descends: the condition is a prefix property
  }
  for (int k = lane; k < K; k += 64) {
    const int ia = L[k], ib = R[k];
    const uint32_t x = v[ia], y = v[ib];
    v[ia] = y;
    v[ib] = x;
  }
  __builtin_amdgcn_wave_barrier();
  int cut = INT32_MAX;
  if (K < nL) cut = L[K];
  if (K > 0) cut = min(cut, (int)R[K - 1]);
  return cut;
}

template <int kCap>
__global__ __launch_bounds__(64) void sdp_order_ties_kernel(SdpLevelArrays lv, int minAll) {
  constexpr int kQueue = kCap / 16;
  constexpr bool kCoop = kCap <= 4096;  // long ranges are partitioned by the whole wavefront (LDS for the lists)
  constexpr int kCoopMin = 32;
  __shared__ uint32_t packed[kCap];
  __shared__ uint2 queue[2][kQueue];  // {f | l << 16, depth limit}
  __shared__ __attribute__((aligned(4))) unsigned short posLists[kCoop ? 2 * kCap : 2];
  constexpr int kPsCap = 512;
  __shared__ uint64_t psKeys[kPsCap];  // keys of the prefix and suffix lists
  const int lane = lane_id();
  const int nTies = *lv.tieCount;
  for (int e = blockIdx.x; e < nTies; e += gridDim.x) {
    const int idx = lv.tieList[e];
    const int nP = lv.nRaw[3 * idx], nM = lv.nRaw[3 * idx + 1], nS = lv.nRaw[3 * idx + 2];
    const int nAll = nP + nM + nS;
    if (nAll < minAll || (nAll > kCap && kCap < kTieLdsCap)) continue;  // another size class takes it
    const Frag *P = lv.raw + lv.fragOff[idx], *M = P + nP, *S = M + nM;
    Frag *out = lv.frags + lv.fragOff[idx];
    if (nAll <= kCap) {
      // rank = number of fragments with a smaller (x,y): equal keys get equal ranks.  The prefix and suffix
      // lists are short: with their keys in LDS a middle fragment's rank needs no dependent HBM reads
      if (nP + nS <= kPsCap) {
        for (int i = lane; i < nP; i += 64) psKeys[i] = frag_key(P[i]);
        for (int i = lane; i < nS; i += 64) psKeys[nP + i] = frag_key(S[i]);
        __syncthreads();
        auto lb = [&](const uint64_t *a, int n, uint64_t key) {
          int lo = 0, hi = n;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (a[mid] < key) lo = mid + 1; else hi = mid;
          }
          return lo;
        };
        for (int i = lane; i < nAll; i += 64) {
          const bool inM = i >= nP && i < nP + nM;
          const uint64_t key = inM ? frag_key(P[i]) : psKeys[i < nP ? i : i - nM];
          const int rp = lb(psKeys, nP, key), rs = lb(psKeys + nP, nS, key);
          const int rm = inM ? i - nP : lower_bound_frag(M, nM, key);
          // bit 15: the key occurs in more than one of the three lists, i.e. this element is part of a tie
          const int lists = (rp < nP && psKeys[rp] == key) + (rs < nS && psKeys[nP + rs] == key) +
                            (inM || (rm < nM && frag_key(M[rm]) == key));
          packed[i] = ((uint32_t)(rp + rs + rm) << 16) | (lists >= 2 ? 0x8000u : 0u) | (uint32_t)i;
        }
      } else {
        for (int i = lane; i < nAll; i += 64) {
          const uint64_t key = frag_key(P[i]);
          const int rp = lower_bound_frag(P, nP, key), rm = lower_bound_frag(M, nM, key), rs = lower_bound_frag(S, nS, key);
          const int lists = (rp < nP && frag_key(P[rp]) == key) + (rm < nM && frag_key(M[rm]) == key) +
                            (rs < nS && frag_key(S[rs]) == key);
          packed[i] = ((uint32_t)(rp + rs + rm) << 16) | (lists >= 2 ? 0x8000u : 0u) | (uint32_t)i;
        }
      }
      __syncthreads();
      int lg = 0;
      for (int m = nAll; m > 1; m >>= 1) lg++;
      int nCur = 0, which = 0;
      if (nAll > 16) {
        if (lane == 0) queue[0][0] = make_uint2((uint32_t)nAll << 16, (uint32_t)(2 * lg));
        nCur = 1;
      } else if (lane == 0)
        lds_insertion_sort(packed, 0, nAll);
      __syncthreads();
      while (nCur > 0) {
        int nNext = 0;
        for (int base = 0; base < nCur; base += 64) {
          bool has = base + lane < nCur;
          int cf[2] = {0, 0}, cl[2] = {0, 0};
          uint32_t depth = 0;
          uint2 r = make_uint2(0, 0);
          if (has) r = queue[which][base + lane];
          if (kCoop) {
            const bool big = has && r.y != 0 && (int)(r.x >> 16) - (int)(r.x & 0xffffu) >= kCoopMin;
            for (unsigned long long bm = __ballot(big); bm != 0ull; bm &= bm - 1ull) {
              const int src = __builtin_ctzll(bm);
              const uint32_t rx = (uint32_t)__builtin_amdgcn_readlane((int)r.x, src);
              const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)r.y, src) - 1u;
              const int f = (int)(rx & 0xffffu), l = (int)(rx >> 16);
              const int cut = coop_partition(packed, f, l, posLists, kCap);
              const int bf[2] = {f, cut}, bl[2] = {cut, l};
              for (int c = 0; c < 2; c++) {
                const int size = bl[c] - bf[c];
                // A range without two elements of a tie has distinct keys: whatever introsort does with it, it
                // ends up sorted.  With consecutive ranks (the rule) that is a scatter, no replay.
                if (size >= kCoopMin) {
                  int nTie = 0, rMin = INT32_MAX, rMax = -1;
                  for (int i0 = bf[c]; i0 < bl[c]; i0 += 64) {
                    const int i = i0 + lane;
                    const uint32_t w = i < bl[c] ? packed[i] : 0u;
                    nTie += (int)__popcll(__ballot(i < bl[c] && (w & 0x8000u) != 0u));
                    if (i < bl[c]) {
                      rMin = min(rMin, (int)(w >> 16));
                      rMax = max(rMax, (int)(w >> 16));
                    }
                  }
                  rMax = wave_max(rMax);
                  rMin = -wave_max(-rMin);
                  if (nTie <= 1 && rMax - rMin + 1 == size) {
                    uint32_t *tmp = reinterpret_cast<uint32_t *>(posLists);
                    for (int i = bf[c] + lane; i < bl[c]; i += 64) tmp[i - bf[c]] = packed[i];
                    __builtin_amdgcn_wave_barrier();
                    for (int i = lane; i < size; i += 64) {
                      const uint32_t w = tmp[i];
                      packed[bf[c] + (int)(w >> 16) - rMin] = w;
                    }
                    __builtin_amdgcn_wave_barrier();
                    continue;
                  }
                }
                if (size > 16) {
                  if (lane == 0) queue[which ^ 1][nNext] = make_uint2((uint32_t)bf[c] | ((uint32_t)bl[c] << 16), d);
                  nNext++;
                } else if (size > 1 && lane == 0)
                  lds_insertion_sort(packed, bf[c], bl[c]);
              }
              __builtin_amdgcn_wave_barrier();
            }
            has = has && !big;
          }
          if (has) {
            const int f = (int)(r.x & 0xffffu), l = (int)(r.x >> 16);
            if (r.y == 0) {
              is_heapsort(packed + f, packed + l, RankLess());
            } else {
              depth = r.y - 1;
              const int cut = lds_partition(packed, f, l);
              cf[0] = f, cl[0] = cut, cf[1] = cut, cl[1] = l;
            }
          }
          for (int c = 0; c < 2; c++) {
            const int size = cl[c] - cf[c];
            const bool push = size > 16;
            if (has && !push && size > 1) lds_insertion_sort(packed, cf[c], cl[c]);
            const uint64_t mask = __ballot(push);
            if (push)
              queue[which ^ 1][nNext + __popcll(mask & ((1ull << lane) - 1ull))] =
                  make_uint2((uint32_t)cf[c] | ((uint32_t)cl[c] << 16), depth);
            nNext += __popcll(mask);
          }
        }
        __syncthreads();
        which ^= 1;
        nCur = nNext;
      }
      // unique(): the first of every run of equal keys survives
      int kept = 0;
      for (int i0 = 0; i0 < nAll; i0 += 64) {
        const int i = i0 + lane;
        bool keep = false;
        uint32_t w = 0;
        if (i < nAll) {
          w = packed[i];
          keep = i == 0 || (packed[i - 1] >> 16) != (w >> 16);
        }
        const uint64_t mask = __ballot(keep);
        if (keep) out[kept + __popcll(mask & ((1ull << lane) - 1ull))] = P[w & 0x7fffu];
        kept += __popcll(mask);
      }
      if (lane == 0) lv.nFrag[idx] = kept;
      __syncthreads();
    } else {
      for (int i = lane; i < nAll; i += 64) out[i] = P[i];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (lane == 0) {
        introsort_replay(out, (long)nAll, FragLess());
        int f = 0, cur = 0;
        while (f + 1 <= nAll) {
          out[cur] = out[f];
          while (f < nAll && out[cur].x == out[f].x && out[cur].y == out[f].y) f++;
          cur++;
        }
        lv.nFrag[idx] = cur;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// StoreAbove (SparseDynamicProgramming.h:55-68) and the sweep-set predecessor
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sdp_yx_keys_kernel(SdpLevelArrays lv) {
  for (int it = 0;; it++) {
    const int idx = general_task(lv, it);
    if (idx < 0) return;
    const int lane = lane_id();
    const int F = lv.nFrag[idx];
    const int64_t base = lv.fragOff[idx];
    for (int i = lane; i < F; i += 64) {
      const Frag f = lv.frags[base + i];
      lv.yxKeys[base + i] = ((uint64_t)f.y << 32) | f.x;
      lv.yxIdx[base + i] = (uint32_t)i;
      lv.above[base + i] = -1;
    }
  }
}

__device__ __forceinline__ void above_predcl_task(const SdpLevelArrays &lv, const int idx) {
  const int lane = lane_id();
  const int F = lv.nFrag[idx];
  const int64_t base = lv.fragOff[idx];
  const Frag *fr = lv.frags + base;
  const int k = lv.tasks[idx].k;
  // above: neighbours in (y,x) order (the sorted index list is in yxIdx)
  for (int j = 1 + lane; j < F; j += 64) {
    const int a = (int)lv.yxIdx[base + j - 1], b = (int)lv.yxIdx[base + j];
    const Frag fa = fr[a], fb = fr[b];
    if (fa.x <= fb.x && fa.x + fa.len > fb.x && fa.y < fb.y) lv.above[base + b] = a;
  }
  // sweepSet.Predecessor (SDPSet.h:94-119 with Fragment::operator< = (diag, x), SDPFragment.h:78-91): when
  // fragment f is scored the set holds exactly the fragments of the k+1 previous rows, so the answer is a
  // function of the geometry alone: the largest (diag, x) with diag <= f.diag among x in [f.x-k-1, f.x-1].
  for (int i = lane; i < F; i += 64) {
    const Frag f = fr[i];
    const int fdiag = (int)(f.y - f.x);
    const uint32_t xLow = f.x > (uint32_t)(k + 1) ? f.x - (uint32_t)(k + 1) : 0u;
    // the candidates are the few fragments just before i (sorted by x): walk back until x < xLow.  (diag, x)
    // pairs are unique, so the order in which they are looked at does not matter
    // (eight at a time: the walk is a chain of loads otherwise -- a scattered invocation with a short word has thirty
    //  candidates per fragment)
    int best = -1, bestDiag = 0;
    uint32_t bestX = 0;
    bool done = false;
    constexpr int kBack = 8;
    for (int j0 = i - 1; j0 >= 0 && !done; j0 -= kBack) {
      uint2 p[kBack];
#pragma unroll
      for (int u = 0; u < kBack; u++) p[u] = *reinterpret_cast<const uint2 *>(fr + max(j0 - u, 0));  // {x, y}
#pragma unroll
      for (int u = 0; u < kBack; u++) {
        const int j = j0 - u;
        if (j < 0 || p[u].x < xLow) done = true;
        if (done || p[u].x >= f.x) continue;  // (>= f.x: the same row)
        const int d = (int)(p[u].y - p[u].x);
        if (d <= fdiag && (best < 0 || d > bestDiag || (d == bestDiag && p[u].x > bestX))) {
          best = j;
          bestDiag = d;
          bestX = p[u].x;
        }
      }
    }
    lv.predCl[base + i] = best;
  }
}
__global__ __launch_bounds__(256) void sdp_above_predcl_kernel(SdpLevelArrays lv) {
  for (int it = 0;; it++) {
    const int idx = general_task(lv, it);
    if (idx < 0) return;
    above_predcl_task(lv, idx);
  }
}

// ---------------------------------------------------------------------------
// sdp_chain: SDPLongestCommonSubsequence (SparseDynamicProgramming.h:70-322)
// ---------------------------------------------------------------------------
// The column set (SDPSet<SDPColumn>, sorted by column).  An entry carries what a later query needs
// from its fragment: {column, fragment index, fragment cost, x + y}.  The spilled part of the set lives in
// global memory; the wavefront keeps the LAST <= 64 entries in registers (lane j = entry nBase + j),
// because fragments retire in (x,y) order and the alignment is near-diagonal: almost every insert,
// replace, erase and predecessor query lands in that tail and needs no memory round trip.
// kS: register sets of 64 entries.  One is enough for an alignment (near-diagonal); sdp_chain_wide_kernel keeps four
// -- 256 entries -- for invocations whose fragments scatter: there the set has a few hundred entries and is touched
// anywhere, and with the whole set in registers an insert in its middle is a handful of DPP moves instead of a shifted
// array in memory.
template <int kS>
struct ColSetT {
  int4 *g;        // global storage, entries [0, nBase) are valid there
  int n, nBase;   // total entries; first entry held in registers
  int4 r[kS];     // r[j], lane l = entry nBase + 64 j + l (valid while that index < n)
};
typedef ColSetT<1> ColSet;
__device__ int cs_global_lower_bound(const int4 *g, int n, int key) {  // wave-cooperative 64-ary search
  const int lane = lane_id();
  int lo = 0, hi = n;
  while (hi - lo > 64) {
    const int stride = (hi - lo + 63) / 64;
    const int p = lo + lane * stride;
    const int v = p < hi ? g[p].x : INT32_MAX;
    const int c = __popcll(__ballot(p < hi && v < key));
    const int nlo = c > 0 ? lo + (c - 1) * stride + 1 : lo;
    const int nhi = min(hi, lo + c * stride);
    lo = nlo;
    hi = nhi;
  }
  const int p = lo + lane;
  const int v = p < hi ? g[p].x : INT32_MAX;
  return lo + __popcll(__ballot(p < hi && v < key));
}
__device__ __forceinline__ int4 lane_bcast(const int4 &v, int srcLane) {
  return make_int4(__builtin_amdgcn_readlane(v.x, srcLane), __builtin_amdgcn_readlane(v.y, srcLane),
                   __builtin_amdgcn_readlane(v.z, srcLane), __builtin_amdgcn_readlane(v.w, srcLane));
}
__device__ __forceinline__ int4 lane_shift_from(const int4 &v, int delta) {  // lane i <- lane i + delta
  const int src = (lane_id() + delta) * 4;
  return make_int4(__builtin_amdgcn_ds_bpermute(src, v.x), __builtin_amdgcn_ds_bpermute(src, v.y),
                   __builtin_amdgcn_ds_bpermute(src, v.z), __builtin_amdgcn_ds_bpermute(src, v.w));
}
__device__ __forceinline__ int4 lane_shr1(const int4 &v, const int4 &edge) {  // lane i <- lane i - 1, lane 0 <- edge
  return make_int4(wave_shr1(v.x, edge.x), wave_shr1(v.y, edge.y), wave_shr1(v.z, edge.z), wave_shr1(v.w, edge.w));
}
template <int kS>
__device__ void cs_flush(ColSetT<kS> &s) {  // registers -> global
  const int lane = lane_id();
#pragma unroll
  for (int j = 0; j < kS; j++)
    if (s.nBase + 64 * j + lane < s.n) s.g[s.nBase + 64 * j + lane] = s.r[j];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
template <int kS>
__device__ void cs_reload(ColSetT<kS> &s) {  // global -> registers: the last <= 64 kS entries
  const int lane = lane_id();
  s.nBase = s.n > 64 * kS ? s.n - 64 * kS : 0;
#pragma unroll
  for (int j = 0; j < kS; j++)
    if (s.nBase + 64 * j + lane < s.n) s.r[j] = s.g[s.nBase + 64 * j + lane];
}
template <int kS>
__device__ int cs_lower_bound(const ColSetT<kS> &s, int key) {
  if (s.n == 0) return 0;
  const int lane = lane_id();
  if (s.nBase == 0 || key > __builtin_amdgcn_readlane(s.r[0].x, 0)) {
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < kS; j++) cnt += __popcll(__ballot(s.nBase + 64 * j + lane < s.n && s.r[j].x < key));
    return s.nBase + cnt;
  }
  return cs_global_lower_bound(s.g, s.nBase, key);
}
template <int kS>
__device__ int4 cs_get(const ColSetT<kS> &s, int pos) {
  if (pos >= s.nBase) {
    const int rel = pos - s.nBase;
    int4 v = lane_bcast(s.r[0], rel & 63);
#pragma unroll
    for (int j = 1; j < kS; j++)
      if ((rel >> 6) == j) v = lane_bcast(s.r[j], rel & 63);
    return v;
  }
  return s.g[pos];
}
template <int kS>
__device__ int cs_get_cost(const ColSetT<kS> &s, int pos) {
  if (pos >= s.nBase) {
    const int rel = pos - s.nBase;
    int v = __builtin_amdgcn_readlane(s.r[0].z, rel & 63);
#pragma unroll
    for (int j = 1; j < kS; j++)
      if ((rel >> 6) == j) v = __builtin_amdgcn_readlane(s.r[j].z, rel & 63);
    return v;
  }
  return s.g[pos].z;
}
template <int kS>
__device__ void cs_set(ColSetT<kS> &s, int pos, int4 e) {
  const int lane = lane_id();
  if (pos >= s.nBase) {
    const int rel = pos - s.nBase;
#pragma unroll
    for (int j = 0; j < kS; j++)
      if ((rel >> 6) == j && lane == (rel & 63)) s.r[j] = e;
  } else {
    if (lane == 0) s.g[pos] = e;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}
template <int kS>
__device__ void cs_insert_at(ColSetT<kS> &s, int pos, int4 e) {
  const int lane = lane_id();
  if (pos >= s.nBase && s.n - s.nBase == 64 * kS) {
    // registers full: their lowest entries move to global memory (half a set when there is only one)
    constexpr int kSpill = kS == 1 ? 32 : 64;
    if (lane < kSpill) s.g[s.nBase + lane] = s.r[0];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (kS == 1) {
      s.r[0] = lane_shift_from(s.r[0], 32);
    } else {
#pragma unroll
      for (int j = 0; j + 1 < kS; j++) s.r[j] = s.r[j + 1];
    }
    s.nBase += kSpill;
  }
  if (pos >= s.nBase) {
    const int rel = pos - s.nBase, jp = rel >> 6, lp = rel & 63;
    const int4 zero = make_int4(0, 0, 0, 0);
    // every entry at or above the position moves up by one; from the top set down, so that a set still holds its
    // old last entry when the set above takes it over
#pragma unroll
    for (int j = kS - 1; j >= 0; j--) {
      if (j > jp) {
        s.r[j] = lane_shr1(s.r[j], lane_bcast(s.r[j > 0 ? j - 1 : 0], 63));
      } else if (j == jp) {
        const int4 up = lane_shr1(s.r[j], zero);
        if (lane > lp) s.r[j] = up;
        if (lane == lp) s.r[j] = e;
      }
    }
    s.n++;
    return;
  }
  // rare: the insertion point is in the spilled part -> shift [pos, nBase) up by one in global memory
  for (int base = pos + ((s.nBase - 1 - pos) / 64) * 64; base >= pos; base -= 64) {
    const int i = base + lane;
    int4 v = make_int4(0, 0, 0, 0);
    if (i < s.nBase) v = s.g[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (i < s.nBase) s.g[i + 1] = v;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  if (lane == 0) s.g[pos] = e;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  s.nBase++;
  s.n++;
}
template <int kS>
__device__ void cs_erase_range(ColSetT<kS> &s, int first, int count) {  // removes [first, first+count)
  const int lane = lane_id();
  if (first >= s.nBase) {
    if (kS == 1) {
      const int4 down = lane_shift_from(s.r[0], count);
      if (lane >= first - s.nBase) s.r[0] = down;
    } else {
      // register index i (>= first - nBase) takes the entry of index i + count: a whole number of sets plus a lane shift
      const int f = first - s.nBase, q = count >> 6, sh = count & 63;
      const int src = lane + sh;
      int4 nw[kS];
#pragma unroll
      for (int j = 0; j < kS; j++) {
        int4 a = make_int4(0, 0, 0, 0), b = make_int4(0, 0, 0, 0);
#pragma unroll
        for (int jq = 0; jq < kS; jq++) {
          if (j + q == jq) a = s.r[jq];
          if (j + q + 1 == jq) b = s.r[jq];
        }
        const int4 va = lane_shift_from(a, sh), vb = lane_shift_from(b, sh - 64);
        nw[j] = src < 64 ? va : vb;
      }
#pragma unroll
      for (int j = 0; j < kS; j++)
        if (64 * j + lane >= f) s.r[j] = nw[j];
    }
    s.n -= count;
    return;
  }
  // rare: the run starts in the spilled part
  cs_flush(s);
  for (int base = first + count; base < s.n; base += 64) {
    const int i = base + lane;
    int4 v = make_int4(0, 0, 0, 0);
    if (i < s.n) v = s.g[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (i < s.n) s.g[i - count] = v;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  s.n -= count;
  cs_reload(s);
}

// IndelPenalty (SparseDynamicProgramming.h:27-53)
// The reference forms the product in double and truncates; with |drift| < 2^25 (sequences < 2^24, checked on the
// host) and penalties below 64 the product is an exact integer below 2^31, so integer arithmetic gives the same.
__device__ __forceinline__ int indel_penalty(int x1, int y1, int x2, int y2, int insertion, int deletion) {
  const int drift = (x1 - y1) - (x2 - y2);
  if (drift > 0) return drift * insertion;
  if (drift < 0) return -drift * deletion;
  return 0;
}
// (int)sqrt((double)d) for 0 <= d < 2^31, without the double-precision pipeline
__device__ __forceinline__ int isqrt_floor(int d) {
  // the single-precision root is off by one at most (d < 2^31 rounds to 24 bits: 2^-25 of the root, which is < 46341), so one
  // step down and one step up settle it; s, s + 1 < 2^24 and their squares < 2^32: the full-rate 24-bit multiply is exact
  int s = (int)sqrtf((float)d);
  s -= (__umul24((unsigned)s, (unsigned)s) > (unsigned)d) ? 1 : 0;
  s += (__umul24((unsigned)(s + 1), (unsigned)(s + 1)) <= (unsigned)d) ? 1 : 0;
  return s;
}

// A window of 64 consecutive fragments held one per lane, so that the serial sweep reads the fragment it
// is scoring, and the recently scored ones it refers to, with v_readlane instead of memory loads.
struct FragWindow {
  int x, y, cost, len;  // cost / chain length are filled in as the sweep advances
};
struct FragView {
  int x, y, cost, len;
};

// The sweep of one invocation.  kPreloaded: the (<= 64) fragments and their static predecessors are handed
// over in registers (lane = fragment) by sdp_small_kernel instead of being read from memory.
template <bool kPreloaded, int kS = 1>
__device__ __forceinline__ void chain_sweep(const SdpLevelArrays &lv, const SdpParams &prm, int idx, const SdpTask &t,
                                            int F, int preX, int preY, int preLen, int prePred, int preAbove) {
  const int lane = lane_id();
  if (F == 0) {
    if (lane == 0) lv.chainEnd[idx] = -1;
    return;
  }
  const int64_t base = lv.fragOff[idx];
  const Frag *fr = lv.frags + base;
  int *cost = lv.cost + base, *chainPrev = lv.chainPrev + base, *chainLen = lv.chainLen + base;
  const int *predCl = lv.predCl + base, *above = lv.above + base;
  ColSetT<kS> cols;
  cols.g = lv.colSet + base;
  cols.n = cols.nBase = 0;
#pragma unroll
  for (int j = 0; j < kS; j++) cols.r[j] = make_int4(0, 0, 0, 0);
  const int k = t.k;
  const uint32_t fragmentLength = (uint32_t)k;
  const uint32_t weight = (uint32_t)k;
  int fTrail = 0;
  int minCost = INT32_MAX, minIdx = -1, maxLenIdx = -1;
  uint32_t maxLen = 0;
  FragWindow cur, prev;
  int curPrev = -1;
  cur.x = cur.y = cur.cost = cur.len = 0;
  prev = cur;
  for (int w0 = 0; w0 < F; w0 += 64) {
    prev = cur;
    int fxV = 0, fyV = 0, flV = 0, fpV = -1, faV = -1;
    if (kPreloaded) {
      fxV = preX, fyV = preY, flV = preLen, fpV = prePred, faV = preAbove;
    } else if (w0 + lane < F) {
      const Frag f = fr[w0 + lane];
      fxV = (int)f.x;
      fyV = (int)f.y;
      flV = (int)f.len;
      fpV = predCl[w0 + lane];
      faV = above[w0 + lane];
    }
    cur.x = fxV;
    cur.y = fyV;
    // fragment p as seen by the sweep: from the two register windows, else from memory
    // kLen = false leaves out the chain length (only the chosen predecessor's is ever needed)
    auto view = [&](int p, bool wantLen) {
      FragView v;
      v.len = 0;
      if (p >= w0) {
        v.x = __builtin_amdgcn_readlane(cur.x, p - w0);
        v.y = __builtin_amdgcn_readlane(cur.y, p - w0);
        v.cost = __builtin_amdgcn_readlane(cur.cost, p - w0);
        if (wantLen) v.len = __builtin_amdgcn_readlane(cur.len, p - w0);
      } else if (p >= w0 - 64) {
        v.x = __builtin_amdgcn_readlane(prev.x, p - w0 + 64);
        v.y = __builtin_amdgcn_readlane(prev.y, p - w0 + 64);
        v.cost = __builtin_amdgcn_readlane(prev.cost, p - w0 + 64);
        if (wantLen) v.len = __builtin_amdgcn_readlane(prev.len, p - w0 + 64);
      } else {
        const Frag f = fr[p];
        v.x = (int)f.x;
        v.y = (int)f.y;
        v.cost = cost[p];
        if (wantLen) v.len = chainLen[p];
      }
      return v;
    };
    auto view_len = [&](int p) {
      if (p >= w0) return __builtin_amdgcn_readlane(cur.len, p - w0);
      if (p >= w0 - 64) return __builtin_amdgcn_readlane(prev.len, p - w0 + 64);
      return chainLen[p];
    };
    const int nHere = min(64, F - w0);
    for (int j = 0; j < nHere; j++) {
      const int i = w0 + j;
      const int fx = __builtin_amdgcn_readlane(fxV, j), fy = __builtin_amdgcn_readlane(fyV, j);
      const int fl = __builtin_amdgcn_readlane(flV, j);
      // rows leaving the window: fragments with x <= f.x - k - 2 are retired before row f.x is scored (:248-309)
      while (fTrail < i) {
        const FragView ft = view(fTrail, false);
        if ((int64_t)ft.x + k + 2 > (int64_t)fx) break;
        const int pos = cs_lower_bound(cols, ft.y);
        int4 at = make_int4(0, 0, 0, 0);
        bool member = false;
        if (pos < cols.n) {
          at = cs_get(cols, pos);
          member = at.x == ft.y;
        }
        bool store = true;
        if (member) store = at.z < ft.cost;  // sic (:260-267): replaced only when the existing entry is CHEAPER
        if (store) {
          const int4 e = make_int4(ft.y, fTrail, ft.cost, ft.x + ft.y);
          if (member) cs_set(cols, pos, e);
          else cs_insert_at(cols, pos, e);
          // delete the costlier successors; Successor() needs at least two elements (SDPSet.h:72-87)
          int del = 0;
          while (cols.n - del >= 2 && pos + 1 + del < cols.n && cs_get_cost(cols, pos + 1 + del) > ft.cost) del++;
          if (del) cs_erase_range(cols, pos + 1, del);
        }
        fTrail++;
      }
      int cp = INT32_MAX, cl = INT32_MAX, ca = INT32_MAX, cpFrag = -1;
      bool foundPrev = false;
      if (cols.n > 0) {
        int pos = cs_lower_bound(cols, fy);
        bool have = false;
        int4 at = make_int4(0, 0, 0, 0);
        if (pos < cols.n) {
          at = cs_get(cols, pos);
          have = at.x == fy;
        }
        if (!have && pos > 0) {
          at = cs_get(cols, pos - 1);
          have = true;
        }
        if (have) {
          cpFrag = at.y;
          cp = at.z + isqrt_floor(abs((fx + fy) - at.w)) - fl;
          foundPrev = true;
        }
      }
      const int clFrag = __builtin_amdgcn_readlane(fpV, j);
      if (clFrag >= 0) {
        const FragView p = view(clFrag, false);
        int ov = (int)(fragmentLength - (uint32_t)(fy - p.y)) * prm.match;
        if (ov > 0) ov = 0;
        cl = p.cost + ov + indel_penalty(fx, fy, p.x, p.y, prm.sdpIns, prm.sdpDel);
        foundPrev = true;
      }
      const int aboveIdx = __builtin_amdgcn_readlane(faV, j);
      if (aboveIdx >= 0) {
        const FragView p = view(aboveIdx, false);
        ca = (int)((uint32_t)p.cost + (fragmentLength - (uint32_t)(fy - p.y)) * (uint32_t)prm.match +
                   (uint32_t)indel_penalty(fx, fy, p.x, p.y, prm.sdpIns, prm.sdpDel));
        foundPrev = true;
      }
      int mc = cl < ca ? cl : ca;
      if (cp < mc) mc = cp;
      int myCost, myPrev;
      uint32_t myLen;
      if (foundPrev && (t.type == 1 || (t.type == 0 && mc < 0))) {
        myCost = (int)((uint32_t)mc - weight);
        if (mc == cp) myPrev = cpFrag;
        else if (mc == cl) myPrev = clFrag;
        else myPrev = aboveIdx;
        myLen = (uint32_t)view_len(myPrev) + 1u;
      } else {
        myPrev = -1;
        myLen = 1;
        if (t.type == 1)
          myCost = (int)((uint32_t)(fx + fy) * (uint32_t)prm.sdpDel + fragmentLength * (uint32_t)prm.match - weight);
        else myCost = (int)(fragmentLength * (uint32_t)prm.match - weight);
      }
      if (lane == j) {
        cur.cost = myCost;
        cur.len = (int)myLen;
        curPrev = myPrev;
      }
      if (minCost > myCost) {
        minCost = myCost;
        minIdx = i;
      }
      if (myLen > maxLen) {
        maxLen = myLen;
        maxLenIdx = i;
      }
    }
    // memory copies, one coalesced store per window: chainPrev for the back-trace, cost / chainLen for the rare
    // look-ups that reach further back than the previous window
    if (w0 + lane < F) {
      cost[w0 + lane] = cur.cost;
      chainPrev[w0 + lane] = curPrev;
      chainLen[w0 + lane] = cur.len;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  if (lane == 0) lv.chainEnd[idx] = t.type == 0 ? minIdx : maxLenIdx;
}

// ---------------------------------------------------------------------------
// sdp_chain4: the same sweep, FOUR invocations per wavefront.
//
// The sweep scores one fragment at a time, so with one invocation per wavefront 63 lanes wait on one.  Here
// every invocation owns a 16-lane group: a window of 16 fragments (and the previous 16) one per lane, and the
// last 16 entries of its column set, shifted with row DPP operations that never leave the group.  What was a
// v_readlane becomes a ds_bpermute inside the group, what was a scalar becomes a per-group vector value, and the
// four groups step through their fragments in lockstep.  An operation whose position lies below a group's 16
// register entries (a fragment far off the diagonal, about four per 10 kb read) is "deep": it is carried out on
// the array in memory by all 64 lanes for that one group (gcs_* above), then the group reloads its tail.
// ---------------------------------------------------------------------------
// Column-set operations on an array that lies entirely in memory, performed by all 64 lanes for ONE invocation
// (sdp_chain4_kernel's rare "deep" operations: the position lies below the 16 entries a group keeps in registers).
__device__ void gcs_insert(int4 *g, int n, int pos, int4 e) {  // n entries before the insert
  const int lane = lane_id();
  for (int top = n; top > pos; top -= 64) {  // move [pos, n) up by one, highest chunk first
    const int i = top - 1 - lane;
    int4 v = make_int4(0, 0, 0, 0);
    if (i >= pos) v = g[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (i >= pos) g[i + 1] = v;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  if (lane == 0) g[pos] = e;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
__device__ void gcs_erase(int4 *g, int n, int first, int count) {  // removes [first, first + count)
  const int lane = lane_id();
  for (int b = first + count; b < n; b += 64) {
    const int i = b + lane;
    int4 v = make_int4(0, 0, 0, 0);
    if (i < n) v = g[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (i < n) g[i - count] = v;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}
__device__ __forceinline__ int4 *uniform_ptr(int4 *p, int srcLane) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, srcLane);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), srcLane);
  return reinterpret_cast<int4 *>(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) { return (unsigned)__builtin_amdgcn_readlane((int)wave_prefix_sum(v), 63); }
// which record (BLASR_HIP_SDP_KERNEL_*) the sweep kernel of a category of the sorted list fills
__host__ __device__ inline int chain_category_kind(int cat) {
  switch (cat) {
    case 0: return BLASR_HIP_SDP_KERNEL_CHAIN4;
    case 2: return BLASR_HIP_SDP_KERNEL_CHAIN_BITS4;
    case 3: return BLASR_HIP_SDP_KERNEL_CHAIN_BITS_SMALL;
    case 4: return BLASR_HIP_SDP_KERNEL_CHAIN_BITS_MID;
    case 5: return BLASR_HIP_SDP_KERNEL_CHAIN_BITS_LARGE;
    case 6: return BLASR_HIP_SDP_KERNEL_CHAIN_ROWS;
    case 7: return BLASR_HIP_SDP_KERNEL_CHAIN_BITS4_WIDE;
  }
  return -1;  // (1: sdp_chain_wide_kernel, tests only)
}
// where a category starts in the sorted list: behind the categories with a larger number
__device__ __forceinline__ int chain_list_start(const SdpLevelArrays &lv, int cat) {
  int at = 0;
  for (int c = kChainCats - 1; c > cat; c--) at += lv.chainCount[c];
  return at;
}
// The invocations sdp_chain4_kernel takes, as (fragment count, index) pairs for a descending sort: the four of a
// wavefront then have about the same number of fragments (the wavefront runs as long as its longest) and the long
// ones start first.  Entries that are not taken get the key 0 and end up behind the list.
__global__ __launch_bounds__(256) void sdp_chain_list_kernel(SdpLevelArrays lv, uint32_t *keys, int32_t *vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool take = false, scatter = false;
  if (i < lv.n && !lv.fused[i]) {
    if (lv.nFrag[i] == 0) lv.chainEnd[i] = -1;
    else take = true;
    // (a task that is not fused has more than 16 raw fragments, so it was given its column slots)
    scatter = take && lv.colEnt != nullptr && lv.colOff[i + 1] > lv.colOff[i];
  }
  // categories, in the order of the sorted list: 5 / 4 / 3 bitmap sweep, a wavefront each, by the size of the target (the
  // bitmap's LDS decides how many wavefronts a CU holds); 2 bitmap sweep four per wavefront (small target, few fragments:
  // the bulk); 1 the 512-register sweep (tests only: BLASR_HIP_CHAIN_SCATTER=1); 0 sdp_chain4_kernel
  int cat = 0;
  if (scatter) {
    const int tl = lv.tasks[i].tLen;
    const int bySize = tl > kBitsMidTarget ? 5 : (tl > kBitsSmallTarget ? 4 : 3);
    cat = lv.scatterMode == 1 ? 1 : ((bySize > 3 || tl > kBits4Target || lv.nFrag[i] > kBits4MaxFrags || lv.scatterMode == 2) ? bySize : 2);
    // (a target of up to 16384 columns with few fragments: four per wavefront as well, with the larger map -- category 7)
    if (cat == 3 && lv.scatterMode == 0 && lv.nFrag[i] <= min(lv.bits4Wide, kBits4MaxFrags)) cat = 7;
    // the long invocations with several fragments per query row: a row per step (sdp_chain_rows_kernel)
    const int nf = lv.nFrag[i];
    if (lv.scatterMode == 3 || (lv.scatterMode == 0 && lv.rowsMinFrags > 0 && nf >= lv.rowsMinFrags &&
                                (int64_t)2 * nf >= (int64_t)lv.rowsDensityX2 * lv.tasks[i].qLen))
      cat = 6;
  }
  if (i < lv.n) {
    keys[i] = take ? (((uint32_t)cat << 29) | (uint32_t)min(lv.nFrag[i], (1 << 29) - 1)) : 0u;
    vals[i] = i;
  }
  // the category counts (and, for blasr_hip_sdp_last_kernels, what each category's kernel will have swept), summed per
  // WORKGROUP in LDS and added with one round of atomics: a round per wavefront and category took this kernel from 0.3 to 4.6 ms
  // per configs[2] step (ten million invocations a level at the recursion levels, nearly all of them fused)
  __shared__ unsigned long long acc[kChainCats][3];
  if (threadIdx.x < kChainCats * 3) acc[threadIdx.x / 3][threadIdx.x % 3] = 0ull;
  __syncthreads();
  if (__ballot(take) != 0ull && take) {
    atomicAdd(&acc[cat][0], 1ull);
    if (lv.kernelSums != nullptr) {
      atomicAdd(&acc[cat][1], (unsigned long long)(unsigned)lv.nFrag[i]);
      atomicAdd(&acc[cat][2], (unsigned long long)(unsigned)lv.tasks[i].qLen);
    }
  }
  __syncthreads();
  if (threadIdx.x < kChainCats) {
    const int c = (int)threadIdx.x;
    const unsigned long long cntC = acc[c][0];
    if (cntC != 0ull) {
      atomicAdd(lv.chainCount + c, (int)cntC);
      const int kind = chain_category_kind(c);
      if (lv.kernelSums != nullptr && kind >= 0) {
        atomicAdd(lv.kernelSums + 3 * kind + 0, acc[c][1]);
        atomicAdd(lv.kernelSums + 3 * kind + 1, cntC);
        atomicAdd(lv.kernelSums + 3 * kind + 2, acc[c][2]);
      }
    }
  }
}

// The matchers' and the fused small kernels' invocations, for lv.kernelSums: a long target goes through
// sdp_match_index_kernel, any other through sdp_match_kernel (fragments: the raw ones they wrote); a fused invocation
// was ordered and swept by sdp_small4_kernel / sdp_small_kernel (fragments: the kept ones).
__global__ __launch_bounds__(256) void sdp_kernel_sums_kernel(SdpLevelArrays lv) {
  // a fixed grid striding over the level's invocations, every thread with its own nine sums: one round of atomics per
  // WORKGROUP (a round per wavefront of a grid over ten million invocations cost 12 ms a level in contended atomics)
  __shared__ unsigned long long part[4][9];
  unsigned long long acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < lv.n; i += gridDim.x * blockDim.x) {
    const int tLen = lv.tasks[i].tLen, k = lv.tasks[i].k, qLen = lv.tasks[i].qLen;
    const int c = task_uses_index(tLen, k, lv.idxMaxT) ? 0 : 1;
    acc[3 * c + 0] += (unsigned)(lv.nRaw[3 * i] + lv.nRaw[3 * i + 1] + lv.nRaw[3 * i + 2]);
    acc[3 * c + 1] += 1;
    acc[3 * c + 2] += (unsigned)qLen;
    if (lv.fused[i] != 0) {
      acc[6] += (unsigned)lv.nFrag[i];
      acc[7] += 1;
      acc[8] += (unsigned)qLen;
    }
  }
  const int lane = lane_id(), wv = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 9; j++) {
    unsigned long long v = acc[j];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) part[wv][j] = v;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    const int j = (int)threadIdx.x;
    const int kind = j < 3 ? BLASR_HIP_SDP_KERNEL_MATCH_INDEX : (j < 6 ? BLASR_HIP_SDP_KERNEL_MATCH_SHORT : BLASR_HIP_SDP_KERNEL_SMALL);
    const unsigned long long v = part[0][j] + part[1][j] + part[2][j] + part[3][j];
    if (v != 0ull) atomicAdd(lv.kernelSums + 3 * kind + j % 3, v);
  }
}

// The sweep of up to four invocations, one per 16-lane group: `on` says whether this lane's group has one, `idx`
// which (uniform within the group).  Fragments, predCl and above are read from the level's arrays, unless the
// caller holds the whole invocation (at most 16 fragments, lane = fragment) in registers: kPre.
template <bool kPre>
__device__ __forceinline__ void chain4_body(const SdpLevelArrays &lv, const SdpParams &prm, const bool on,
                                            const int idx, const int preF = 0, const int preX = 0,
                                            const int preY = 0, const int preLen = 0, const int prePred = -1,
                                            const int preAbove = -1) {
  const int lane = lane_id(), l = lane & 15, gb = lane & 48;
  const int F = on ? (kPre ? preF : lv.nFrag[idx]) : 0;
  const SdpTask t = lv.tasks[idx];
  const int64_t base = lv.fragOff[idx];
  // per-group arrays are addressed as (uniform array base) + base + index: one 64-bit offset per lane instead of
  // seven pointers
#define C4_FR(i) lv.frags[base + (i)]
#define C4_COST(i) lv.cost[base + (i)]
#define C4_PREV(i) lv.chainPrev[base + (i)]
#define C4_LEN(i) lv.chainLen[base + (i)]
  int4 *colG = lv.colSet + base;
  const int k = t.k, type = t.type;
  auto gbp = [&](int v, int i) { return __builtin_amdgcn_ds_bpermute((gb + (i & 15)) << 2, v); };
  auto gcount = [&](bool flag) {
    const unsigned long long m = __ballot(flag);
    return (int)__popc((unsigned)((m >> gb) & 0xffffull));
  };
  int curX = 0, curY = 0, curCost = 0, curLen = 0, curPrev = -1, prvX = 0, prvY = 0, prvCost = 0, prvLen = 0;
  int4 tail = make_int4(0, 0, 0, 0);
  int n = 0, nBase = 0, fTrail = 0;
  int minCost = INT32_MAX, minIdx = -1, maxLenIdx = -1;
  uint32_t maxLen = 0;
  int maxF = F;
  for (int off = 32; off > 0; off >>= 1) maxF = max(maxF, __shfl_xor(maxF, off));
  for (int w0 = 0; w0 < maxF; w0 += 16) {
    prvX = curX, prvY = curY, prvCost = curCost, prvLen = curLen;
    int flV = 0, fpV = -1, faV = -1;
    // What the two fixed predecessors (the one of the diagonal class, the one above) add to their own cost depends on
    // coordinates alone (:124-137, :150-163): worked out here, lane = fragment, once per window -- not in the sweep's
    // serial part, where every lane of the group would repeat it for every fragment.
    int pX = 0, pY = 0, aX = 0, aY = 0;
    curX = curY = 0;
    if (kPre) {
      const int qx = gbp(preX, prePred), qy = gbp(preY, prePred), rx = gbp(preX, preAbove), ry = gbp(preY, preAbove);
      if (on && l < F) {
        curX = preX;
        curY = preY;
        flV = preLen;
        fpV = prePred;
        faV = preAbove;
        pX = qx, pY = qy, aX = rx, aY = ry;
      }
    } else if (on && w0 + l < F) {
      const Frag f = C4_FR(w0 + l);
      curX = (int)f.x;
      curY = (int)f.y;
      flV = (int)f.len;
      fpV = lv.predCl[base + w0 + l];
      faV = lv.above[base + w0 + l];
      if (fpV >= 0) {
        const Frag q = C4_FR(fpV);
        pX = (int)q.x, pY = (int)q.y;
      }
      if (faV >= 0) {
        const Frag q = C4_FR(faV);
        aX = (int)q.x, aY = (int)q.y;
      }
    }
    int clS, caS;
    {
      int ov = (int)((uint32_t)k - (uint32_t)(curY - pY)) * prm.match;
      if (ov > 0) ov = 0;
      clS = ov + indel_penalty(curX, curY, pX, pY, prm.sdpIns, prm.sdpDel);
      caS = (int)(((uint32_t)k - (uint32_t)(curY - aY)) * (uint32_t)prm.match +
                  (uint32_t)indel_penalty(curX, curY, aX, aY, prm.sdpIns, prm.sdpDel));
    }
    // the cost of fragment p of this group: from the two register windows, else from memory
    auto cost_of = [&](int p, bool need) {
      const int rel = p - w0;
      const int ca = gbp(curCost, rel), cb = gbp(prvCost, rel + 16);
      int vc = rel >= 0 ? ca : cb;
      if (__ballot(need && rel < -16) != 0ull)
        if (need && rel < -16) vc = C4_COST(p);
      return vc;
    };
    for (int j = 0; j < 16; j++) {
      const int i = w0 + j;
      const bool act = on && i < F;
      if (__ballot(act) == 0ull) break;
      const int fx = gbp(curX, j), fy = gbp(curY, j), fl = gbp(flV, j);
      // ---- rows leaving the window: fragments with x <= f.x - k - 2 are retired first (:248-309)
      for (;;) {
        bool need = act && fTrail < i;
        // (the test that ends the loop needs the row alone: column and cost are fetched when some group retires)
        const int pT = need ? fTrail : i, relT = pT - w0;
        int ftx, fty = 0, ftc = 0;
        {
          const int xa = gbp(curX, relT), xb = gbp(prvX, relT + 16);
          ftx = relT >= 0 ? xa : xb;
        }
        const bool farT = need && relT < -16;
        if (__ballot(farT) != 0ull) {
          if (farT) {
            const Frag f = C4_FR(pT);
            ftx = (int)f.x;
            fty = (int)f.y;
            ftc = C4_COST(pT);
          }
        }
        need = need && (int64_t)ftx + k + 2 <= (int64_t)fx;
        if (__ballot(need) == 0ull) break;
        {
          const int ya = gbp(curY, relT), ca = gbp(curCost, relT), yb = gbp(prvY, relT + 16), cb = gbp(prvCost, relT + 16);
          if (!farT) {
            fty = relT >= 0 ? ya : yb;
            ftc = relT >= 0 ? ca : cb;
          }
        }
        const bool validLane = nBase + l < n;
        const int tail0 = gbp(tail.x, 0);
        const int pos = nBase + gcount(validLane && tail.x < fty);
        // "deep": the position lies below the entries held in registers, or the insert would have to go into the
        // half of the tail that is about to be spilled -> the whole retirement is done in memory, see below
        const int atx = gbp(tail.x, pos - nBase), atz = gbp(tail.z, pos - nBase);
        const bool memberT = pos < n && atx == fty;
        const bool deep = need && ((nBase != 0 && !(fty > tail0)) ||
                                   (!memberT && n - nBase == 16 && pos < nBase + 8));
        const bool fast = need && !deep;
        const bool member = fast && memberT;
        const bool store = fast && (!member || atz < ftc);  // sic (:260-267)
        const int4 e = make_int4(fty, fTrail, ftc, ftx + fty);
        if (store && member && l == pos - nBase) tail = e;
        const bool doIns = store && !member;
        const bool full = doIns && n - nBase == 16;
        if (__ballot(full) != 0ull) {  // spill the lower half of the tail
          if (full && l < 8) colG[nBase + l] = tail;
          const int sx = gbp(tail.x, l + 8), sy = gbp(tail.y, l + 8), sz = gbp(tail.z, l + 8), sw = gbp(tail.w, l + 8);
          if (full) {
            tail = make_int4(sx, sy, sz, sw);
            nBase += 8;
          }
        }
        {
          const int p = pos - nBase;
          const int ux = dpp<0x111, 0xf>(0, tail.x), uy = dpp<0x111, 0xf>(0, tail.y);
          const int uz = dpp<0x111, 0xf>(0, tail.z), uw = dpp<0x111, 0xf>(0, tail.w);
          if (doIns && l > p) tail = make_int4(ux, uy, uz, uw);
          if (doIns && l == p) tail = e;
          n += doIns ? 1 : 0;
        }
        // delete the costlier successors; Successor() needs at least two elements (SDPSet.h:72-87)
        int del = 0;
        for (;;) {
          const int sc = gbp(tail.z, pos + 1 + del - nBase);
          const bool more = store && n - del >= 2 && pos + 1 + del < n && sc > ftc;
          if (__ballot(more) == 0ull) break;
          del += more ? 1 : 0;
        }
        if (__ballot(del > 0) != 0ull) {
          const int ex = gbp(tail.x, l + del), ey = gbp(tail.y, l + del), ez = gbp(tail.z, l + del), ew = gbp(tail.w, l + del);
          if (del > 0 && l >= pos + 1 - nBase) tail = make_int4(ex, ey, ez, ew);
          n -= del;
        }
        // ---- deep retirements, one group at a time, with all 64 lanes on the array in memory
        for (unsigned long long dm = __ballot(deep); dm != 0ull;) {
          const int g0 = __builtin_ctzll(dm) & 48;  // first lane of the group
          dm &= ~(0xffffull << g0);
          if (gb == g0 && nBase + l < n) colG[nBase + l] = tail;  // the registers' share joins the array
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          int4 *gp = uniform_ptr(colG, g0);
          int N = __builtin_amdgcn_readlane(n, g0);
          const int key = __builtin_amdgcn_readlane(fty, g0), kc = __builtin_amdgcn_readlane(ftc, g0);
          const int4 ge = make_int4(key, __builtin_amdgcn_readlane(fTrail, g0), kc,
                                    __builtin_amdgcn_readlane(ftx, g0) + key);
          const int gpos = cs_global_lower_bound(gp, N, key);
          bool gmember = false;
          int4 at = make_int4(0, 0, 0, 0);
          if (gpos < N) {
            at = gp[gpos];
            gmember = at.x == key;
          }
          if (!gmember || at.z < kc) {
            if (gmember) {
              if (lane == 0) gp[gpos] = ge;
              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            } else {
              gcs_insert(gp, N, gpos, ge);
              N++;
            }
            int gdel = 0;
            while (N - gdel >= 2 && gpos + 1 + gdel < N && gp[gpos + 1 + gdel].z > kc) gdel++;
            if (gdel) {
              gcs_erase(gp, N, gpos + 1, gdel);
              N -= gdel;
            }
          }
          if (gb == g0) {  // back into registers: the last 16 entries
            n = N;
            nBase = N > 16 ? N - 16 : 0;
            if (nBase + l < n) tail = colG[nBase + l];
          }
        }
        fTrail += need ? 1 : 0;
      }
      // ---- the three predecessors
      int cp = INT32_MAX, cl = INT32_MAX, ca = INT32_MAX, cpFrag = -1;
      bool foundPrev = false;
      {
        const bool q = act && n > 0;
        const bool validLane = nBase + l < n;
        const int tail0 = gbp(tail.x, 0);
        const int pos = nBase + gcount(validLane && tail.x < fy);
        const bool deep = q && nBase != 0 && !(fy > tail0);  // the predecessor lies below the registers' share
        const bool qf = q && !deep;
        const int atx = gbp(tail.x, pos - nBase);
        const bool exact = qf && pos < n && atx == fy;
        bool have = qf && (exact || pos > 0);
        const int sel = (exact ? pos : pos - 1) - nBase;
        int aty = gbp(tail.y, sel), atz = gbp(tail.z, sel), atw = gbp(tail.w, sel);
        for (unsigned long long dm = __ballot(deep); dm != 0ull;) {
          const int g0 = __builtin_ctzll(dm) & 48;
          dm &= ~(0xffffull << g0);
          int4 *gp = uniform_ptr(colG, g0);
          const int NB = __builtin_amdgcn_readlane(nBase, g0), key = __builtin_amdgcn_readlane(fy, g0);
          // entries [0, NB) are in memory; entry NB is the group's first register entry (its column is >= key)
          const int gpos = cs_global_lower_bound(gp, NB, key);
          bool gexact = false;
          int4 at = make_int4(0, 0, 0, 0);
          if (gpos < NB) {
            at = gp[gpos];
            gexact = at.x == key;
          } else if (__builtin_amdgcn_readlane(tail.x, g0) == key) {
            at = make_int4(key, __builtin_amdgcn_readlane(tail.y, g0), __builtin_amdgcn_readlane(tail.z, g0),
                           __builtin_amdgcn_readlane(tail.w, g0));
            gexact = true;
          }
          bool ghave = gexact;
          if (!gexact && gpos > 0) {
            at = gp[gpos - 1];
            ghave = true;
          }
          if (gb == g0) {
            have = ghave;
            aty = at.y;
            atz = at.z;
            atw = at.w;
          }
        }
        if (have) {
          const int d = abs((fx + fy) - atw);
          const int s = isqrt_floor(d);
          cpFrag = aty;
          cp = atz + s - fl;
          foundPrev = true;
        }
      }
      const int clFrag = gbp(fpV, j), aboveIdx = gbp(faV, j);
      {
        const int clSj = gbp(clS, j), caSj = gbp(caS, j);
        const bool needCl = act && clFrag >= 0;
        const int pcl = cost_of(needCl ? clFrag : i, needCl);
        if (needCl) {
          cl = (int)((uint32_t)pcl + (uint32_t)clSj);
          foundPrev = true;
        }
        const bool needCa = act && aboveIdx >= 0;
        const int pca = cost_of(needCa ? aboveIdx : i, needCa);
        if (needCa) {
          ca = (int)((uint32_t)pca + (uint32_t)caSj);
          foundPrev = true;
        }
      }
      int mc = cl < ca ? cl : ca;
      if (cp < mc) mc = cp;
      int myCost, myPrev;
      uint32_t myLen;
      const bool linked = foundPrev && (type == 1 || (type == 0 && mc < 0));
      {
        myPrev = -1;
        if (linked) myPrev = (mc == cp) ? cpFrag : ((mc == cl) ? clFrag : aboveIdx);
        // the chosen predecessor's chain length
        const int p = linked && act ? myPrev : i;
        const int rel = p - w0;
        const int la = gbp(curLen, rel), lb = gbp(prvLen, rel + 16);
        int pl = rel >= 0 ? la : lb;
        if (__ballot(linked && act && rel < -16) != 0ull)
          if (linked && act && rel < -16) pl = C4_LEN(p);
        if (linked) {
          myCost = (int)((uint32_t)mc - (uint32_t)k);
          myLen = (uint32_t)pl + 1u;
        } else {
          myLen = 1;
          if (type == 1) myCost = (int)((uint32_t)(fx + fy) * (uint32_t)prm.sdpDel + (uint32_t)k * (uint32_t)prm.match - (uint32_t)k);
          else myCost = (int)((uint32_t)k * (uint32_t)prm.match - (uint32_t)k);
        }
      }
      if (act && l == j) {
        curCost = myCost;
        curLen = (int)myLen;
        curPrev = myPrev;
      }
      if (act && minCost > myCost) {
        minCost = myCost;
        minIdx = i;
      }
      if (act && myLen > maxLen) {
        maxLen = myLen;
        maxLenIdx = i;
      }
    }
    if (on && w0 + l < F) {
      C4_COST(w0 + l) = curCost;
      C4_PREV(w0 + l) = curPrev;
      C4_LEN(w0 + l) = curLen;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  if (on && l == 0) lv.chainEnd[idx] = type == 0 ? minIdx : maxLenIdx;
#undef C4_FR
#undef C4_COST
#undef C4_PREV
#undef C4_LEN
}

__global__ __launch_bounds__(256, 8) void sdp_chain4_kernel(SdpLevelArrays lv, SdpParams prm) {
  const int lane = lane_id();
  const int nList = lv.chainCount[0];  // the list starts with the scattered ones
  const int nScatter = chain_list_start(lv, 0);
  const int slot = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6))) * 4 + (lane >> 4);
  if (__builtin_amdgcn_readfirstlane(slot) >= nList) return;  // slot of group 0: the whole wavefront is beyond the list
  const bool on = slot < nList;
  chain4_body<false>(lv, prm, on, on ? lv.chainList[nScatter + slot] : 0);
}

// ---------------------------------------------------------------------------
// sdp_chain_bits: the same sweep for invocations whose fragments SCATTER over the rectangle.
//
// chain4's column set keeps its top in registers because an alignment is near-diagonal: fragments retire in (x,y)
// order, so the set is touched at its high end.  The recursion's calls on the gaps of a chance chain (a candidate
// interval on unrelated sequence: a few random k-mer hits chained, then SDPAlign on the rectangles between them
// with a shorter word) see the opposite: hits anywhere, costs that rise along the chain (the square-root term of
// the column predecessor outweighs a fragment's weight), a set of hundreds of entries and 40 % of the operations
// far below its top -- each a shifted array in memory there.  Here the set is a BITMAP over the target's columns
// in LDS (one wavefront per invocation, 2048 words = 65536 columns) with a one-word-per-lane summary in a register,
// and the entries sit in memory at their column: membership, predecessor, successor, insert and erase are a
// constant number of LDS reads and ballots whatever the size of the set and wherever they land.
// ---------------------------------------------------------------------------
constexpr int kBitsWords = kBitsMaxTarget / 32;
struct BitCols {
  uint32_t *bits;    // LDS: bit c of the map = column c is in the set
  int4 *ent;         // memory: the entry of a member column {chain length, fragment, cost, x + y} of its fragment
  uint64_t summary;  // lane i, bit b: word 64 i + b of the map is not empty (64 lanes x 64 words x 32 columns)
  int n;
};
__device__ __forceinline__ uint32_t bc_word(const BitCols &s, int w) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)s.bits[w]);
}
__device__ __forceinline__ uint64_t bc_summary_of(const BitCols &s, int srcLane) {  // uniform lane
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)s.summary, srcLane);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(s.summary >> 32), srcLane);
  return ((uint64_t)hi << 32) | lo;
}
// the largest member <= y, or -1
__device__ __forceinline__ int bc_pred_le(const BitCols &s, int y) {
  const int w = y >> 5, b = y & 31;
  const uint32_t word = bc_word(s, w) & (b == 31 ? 0xffffffffu : ((2u << b) - 1u));
  if (word) return (w << 5) + 31 - __clz((int)word);
  const int lane = lane_id(), li = w >> 6, bi = w & 63;
  uint64_t m = s.summary;
  if (lane == li) m &= (1ull << bi) - 1ull;
  if (lane > li) m = 0;
  const unsigned long long ball = __ballot(m != 0ull);
  if (ball == 0ull) return -1;
  const int hl = 63 - __clzll((long long)ball);
  uint64_t hm = bc_summary_of(s, hl);
  if (hl == li) hm &= (1ull << bi) - 1ull;
  const int w2 = (hl << 6) + 63 - __clzll((long long)hm);
  return (w2 << 5) + 31 - __clz((int)bc_word(s, w2));
}
// the smallest member > y, or -1
__device__ __forceinline__ int bc_succ_gt(const BitCols &s, int y) {
  const int w = y >> 5, b = y & 31;
  const uint32_t word = bc_word(s, w) & (b == 31 ? 0u : ~((2u << b) - 1u));
  if (word) return (w << 5) + __ffs((int)word) - 1;
  const int lane = lane_id(), li = w >> 6, bi = w & 63;
  uint64_t m = s.summary;
  const uint64_t above = bi == 63 ? 0ull : ~((2ull << bi) - 1ull);
  if (lane == li) m &= above;
  if (lane < li) m = 0;
  const unsigned long long ball = __ballot(m != 0ull);
  if (ball == 0ull) return -1;
  const int ll = __builtin_ctzll(ball);
  uint64_t lm = bc_summary_of(s, ll);
  if (ll == li) lm &= above;
  const int w2 = (ll << 6) + __builtin_ctzll(lm);
  return (w2 << 5) + __ffs((int)bc_word(s, w2)) - 1;
}
__device__ __forceinline__ void bc_set(BitCols &s, int y) {
  const int w = y >> 5;
  const uint32_t v = bc_word(s, w) | (1u << (y & 31));
  if (lane_id() == 0) s.bits[w] = v;
  __builtin_amdgcn_wave_barrier();
  if (lane_id() == (w >> 6)) s.summary |= 1ull << (w & 63);
}
__device__ __forceinline__ void bc_clear(BitCols &s, int y) {
  const int w = y >> 5;
  const uint32_t v = bc_word(s, w) & ~(1u << (y & 31));
  if (lane_id() == 0) s.bits[w] = v;
  __builtin_amdgcn_wave_barrier();
  if (v == 0u && lane_id() == (w >> 6)) s.summary &= ~(1ull << (w & 63));
}
// one entry, the same address in every lane (a broadcast load; the scalar cache would not see the wave's own stores)
__device__ __forceinline__ int4 bc_entry(const BitCols &s, int col) {
  const int4 e = s.ent[col];
  return make_int4(__builtin_amdgcn_readfirstlane(e.x), __builtin_amdgcn_readfirstlane(e.y),
                   __builtin_amdgcn_readfirstlane(e.z), __builtin_amdgcn_readfirstlane(e.w));
}

// per-lane variant of bc_pred_le: every lane asks for its own column (the window's speculative queries)
__device__ __forceinline__ uint64_t bperm64(uint64_t v, int srcLane) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(srcLane << 2, (int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(srcLane << 2, (int)(uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ int bc_pred_le_lanes(const BitCols &s, int y, bool ok) {
  const int w = ok ? y >> 5 : 0, b = y & 31;
  const uint32_t word = s.bits[w] & (b == 31 ? 0xffffffffu : ((2u << b) - 1u));
  const int li = w >> 6, bi = w & 63;
  const unsigned long long top = __ballot(s.summary != 0ull);  // lanes whose 64 words are not all empty
  const uint64_t own = bperm64(s.summary, li) & ((1ull << bi) - 1ull);
  const unsigned long long below = top & ((1ull << li) - 1ull);
  const int hl = below ? 63 - __clzll((long long)below) : 0;
  const uint64_t far = bperm64(s.summary, hl);
  int w2 = -1;
  if (below) w2 = (hl << 6) + 63 - __clzll((long long)far);
  if (own) w2 = (li << 6) + 63 - __clzll((long long)own);
  const uint32_t word2 = s.bits[w2 < 0 ? 0 : w2];
  int r = -1;
  if (w2 >= 0) r = (w2 << 5) + 31 - __clz((int)(word2 | 1u));
  if (word) r = (w << 5) + 31 - __clz((int)word);
  return ok ? r : -1;
}

// per-lane variant of bc_succ_gt
__device__ __forceinline__ int bc_succ_gt_lanes(const BitCols &s, int y, bool ok) {
  const int w = ok ? y >> 5 : 0, b = y & 31;
  const uint32_t word = s.bits[w] & (b == 31 ? 0u : ~((2u << b) - 1u));
  const int li = w >> 6, bi = w & 63;
  const unsigned long long top = __ballot(s.summary != 0ull);
  const uint64_t own = bperm64(s.summary, li) & (bi == 63 ? 0ull : ~((2ull << bi) - 1ull));
  const unsigned long long above = li == 63 ? 0ull : top & ~((2ull << li) - 1ull);
  const int ll = above ? __builtin_ctzll(above) : 0;
  const uint64_t far = bperm64(s.summary, ll);
  int w2 = -1;
  if (above) w2 = (ll << 6) + __builtin_ctzll(far | (1ull << 63));
  if (own) w2 = (li << 6) + __builtin_ctzll(own);
  const uint32_t word2 = s.bits[w2 < 0 ? 0 : w2];
  int r = -1;
  if (w2 >= 0 && word2) r = (w2 << 5) + __ffs((int)word2) - 1;
  if (word) r = (w << 5) + __ffs((int)word) - 1;
  return ok ? r : -1;
}

// The sweep is a chain of dependent steps, and an invocation with thousands of scattered fragments is as long as
// its chain: the kernel's time is its longest invocation's.  What is kept off that chain here:
//  * the column predecessors of a whole window of 64 fragments are looked up at once, lane = fragment (bitmap in LDS,
//    then ONE round of 64 entry loads), and a lane's answer stays good until a retirement touches the columns between
//    its predecessor and its own (`dirty`, one vector comparison per change of the set) -- only then is the query
//    repeated when the fragment's turn comes;
//  * the same for the retirements: the first successor of each of the next 64 fragments to retire, and its cost, are
//    fetched at once (lane = fragment) and stay good until the set changes between a fragment's column and its
//    successor; a retirement that finds its answer stale looks for its successor BEFORE it stores its entry (the bit
//    of a greater column does not depend on it), so that the successor's cost and the store are one round trip;
//  * a store is waited for only when the next load of an entry is due.
__device__ __forceinline__ void chain_sweep_bits(const SdpLevelArrays &lv, const SdpParams &prm, int idx, const SdpTask &t,
                                                 int F, uint32_t *ldsBits) {
  const int lane = lane_id();
  const int64_t base = lv.fragOff[idx];
  const Frag *fr = lv.frags + base;
  int *cost = lv.cost + base, *chainPrev = lv.chainPrev + base, *chainLen = lv.chainLen + base;
  const int *predCl = lv.predCl + base, *above = lv.above + base;
  BitCols cols;
  cols.bits = ldsBits;
  cols.ent = lv.colEnt + lv.colOff[idx];
  cols.summary = 0;
  cols.n = 0;
  for (int w = lane; w < ((t.tLen + 31) >> 5); w += 64) ldsBits[w] = 0;
  __builtin_amdgcn_wave_barrier();
  const int k = t.k;
  const uint32_t fragmentLength = (uint32_t)k, weight = (uint32_t)k;
  int fTrail = 0;
  int minCost = INT32_MAX, minIdx = -1, maxLenIdx = -1;
  uint32_t maxLen = 0;
  FragWindow cur, prev;
  int curPrev = -1;
  cur.x = cur.y = cur.cost = cur.len = 0;
  prev = cur;
  bool pendingStore = false;  // an entry was stored and not yet waited for
  auto settle = [&]() {
    if (pendingStore) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    pendingStore = false;
  };
  // the retirements' speculation: lane = fragment rw0 + lane, its column, first successor (-1: none) and that one's cost
  int rw0 = -64, rY = 0, rSucc = -1, rSuccCost = 0;  // (the first retirement fetches the first batch)
  bool rDirty = true;
  auto refresh_retirements = [&](int from) {
    rw0 = from;
    const bool ok = from + lane < F && cols.n > 0;
    rY = from + lane < F ? (int)fr[from + lane].y : 0;
    settle();
    rSucc = bc_succ_gt_lanes(cols, rY, ok);
    rSuccCost = 0;
    if (rSucc >= 0) rSuccCost = cols.ent[rSucc].z;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    rDirty = !ok;  // (an empty set has no answers to keep)
  };
  for (int w0 = 0; w0 < F; w0 += 64) {
    prev = cur;
    int fxV = 0, fyV = 0, flV = 0, fpV = -1, faV = -1;
    int clS = 0, caS = 0;  // what the two fixed predecessors add to their own cost: coordinates only, lane = fragment
    if (w0 + lane < F) {
      const Frag f = fr[w0 + lane];
      fxV = (int)f.x;
      fyV = (int)f.y;
      flV = (int)f.len;
      fpV = predCl[w0 + lane];
      faV = above[w0 + lane];
      if (fpV >= 0) {
        const Frag q = fr[fpV];
        int ov = (int)(fragmentLength - (uint32_t)(fyV - (int)q.y)) * prm.match;
        if (ov > 0) ov = 0;
        clS = ov + indel_penalty(fxV, fyV, (int)q.x, (int)q.y, prm.sdpIns, prm.sdpDel);
      }
      if (faV >= 0) {
        const Frag q = fr[faV];
        caS = (int)((fragmentLength - (uint32_t)(fyV - (int)q.y)) * (uint32_t)prm.match +
                    (uint32_t)indel_penalty(fxV, fyV, (int)q.x, (int)q.y, prm.sdpIns, prm.sdpDel));
      }
    }
    cur.x = fxV;
    cur.y = fyV;
    // ---- the window's column predecessors, speculatively, against the set as it stands now
    settle();
    const bool inWin = w0 + lane < F;
    const int specPred = bc_pred_le_lanes(cols, fyV, inWin && cols.n > 0);
    int4 specEnt = make_int4(0, 0, 0, 0);
    if (specPred >= 0) specEnt = cols.ent[specPred];
    // (waited for here, one round trip per window: a retirement below may store to one of these addresses)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    bool dirty = false;
    auto touched = [&](int col, bool inserted) {  // the set changed at `col`: whose speculative answer is stale?
      dirty = dirty || col == specPred || (inserted && col > specPred && col <= fyV);
      rDirty = rDirty || col == rSucc || (inserted && col > rY && (rSucc < 0 || col < rSucc));
    };
    auto view = [&](int p) {
      FragView v;
      v.len = 0;
      if (p >= w0) {
        v.x = __builtin_amdgcn_readlane(cur.x, p - w0);
        v.y = __builtin_amdgcn_readlane(cur.y, p - w0);
        v.cost = __builtin_amdgcn_readlane(cur.cost, p - w0);
      } else if (p >= w0 - 64) {
        v.x = __builtin_amdgcn_readlane(prev.x, p - w0 + 64);
        v.y = __builtin_amdgcn_readlane(prev.y, p - w0 + 64);
        v.cost = __builtin_amdgcn_readlane(prev.cost, p - w0 + 64);
      } else {
        const Frag f = fr[p];
        v.x = __builtin_amdgcn_readfirstlane((int)f.x);
        v.y = __builtin_amdgcn_readfirstlane((int)f.y);
        v.cost = __builtin_amdgcn_readfirstlane(cost[p]);
      }
      return v;
    };
    auto view_len = [&](int p) {
      if (p >= w0) return __builtin_amdgcn_readlane(cur.len, p - w0);
      if (p >= w0 - 64) return __builtin_amdgcn_readlane(prev.len, p - w0 + 64);
      return __builtin_amdgcn_readfirstlane(chainLen[p]);
    };
    auto view_cost = [&](int p) {
      if (p >= w0) return __builtin_amdgcn_readlane(cur.cost, p - w0);
      if (p >= w0 - 64) return __builtin_amdgcn_readlane(prev.cost, p - w0 + 64);
      return __builtin_amdgcn_readfirstlane(cost[p]);
    };
    const int nHere = min(64, F - w0);
    for (int j = 0; j < nHere; j++) {
      const int i = w0 + j;
      const int fx = __builtin_amdgcn_readlane(fxV, j), fy = __builtin_amdgcn_readlane(fyV, j);
      const int fl = __builtin_amdgcn_readlane(flV, j);
      // rows leaving the window: fragments with x <= f.x - k - 2 are retired before row f.x is scored (:248-309)
      while (fTrail < i) {
        const FragView ft = view(fTrail);
        if ((int64_t)ft.x + k + 2 > (int64_t)fx) break;
        const bool member = ((bc_word(cols, ft.y >> 5) >> (ft.y & 31)) & 1u) != 0u;
        if (fTrail >= rw0 + 64) refresh_retirements(fTrail);
        // Successor() needs at least two elements (SDPSet.h:72-87): the set's size after this fragment went in
        const bool wantSucc = cols.n + (member ? 0 : 1) >= 2;
        int sc = -1, scCost = 0;
        const int rl = fTrail - rw0;
        const bool specOk = rl >= 0 && __builtin_amdgcn_readlane((int)rDirty, rl < 0 ? 0 : rl) == 0;
        if (wantSucc && specOk) {
          sc = __builtin_amdgcn_readlane(rSucc, rl);
          scCost = __builtin_amdgcn_readlane(rSuccCost, rl);
        } else if (wantSucc) {
          sc = bc_succ_gt(cols, ft.y);
          if (sc >= 0) {
            settle();
            scCost = cols.ent[sc].z;  // (issued before the store below: one round trip for both)
          }
        }
        bool store = true;
        if (member) settle();
        if (member) store = bc_entry(cols, ft.y).z < ft.cost;  // sic (:260-267): replaced only when the existing entry is CHEAPER
        if (store) {
          if (lane == 0) cols.ent[ft.y] = make_int4(view_len(fTrail), fTrail, ft.cost, ft.x + ft.y);
          pendingStore = true;
          touched(ft.y, true);
          if (!member) {
            bc_set(cols, ft.y);
            cols.n++;
          }
          // delete the costlier successors
          while (sc >= 0 && __builtin_amdgcn_readfirstlane(scCost) > ft.cost) {
            bc_clear(cols, sc);
            cols.n--;
            touched(sc, false);
            sc = cols.n >= 2 ? bc_succ_gt(cols, ft.y) : -1;
            if (sc >= 0) scCost = cols.ent[sc].z;  // (never the entry just stored: sc > ft.y)
          }
        }
        fTrail++;
      }
      int cp = INT32_MAX, cl = INT32_MAX, ca = INT32_MAX, cpFrag = -1, cpLen = 0;
      bool foundPrev = false;
      if (cols.n > 0) {
        int4 at = make_int4(0, 0, 0, 0);
        int pc;
        if (__builtin_amdgcn_readlane((int)dirty, j) == 0) {
          pc = __builtin_amdgcn_readlane(specPred, j);
          at = lane_bcast(specEnt, j);
        } else {
          pc = bc_pred_le(cols, fy);
          if (pc >= 0) {
            settle();
            at = bc_entry(cols, pc);
          }
        }
        if (pc >= 0) {
          cpFrag = at.y;
          cpLen = at.x;  // the chain length comes with the entry: no second round trip for a far predecessor
          cp = at.z + isqrt_floor(abs((fx + fy) - at.w)) - fl;
          foundPrev = true;
        }
      }
      const int clFrag = __builtin_amdgcn_readlane(fpV, j);
      if (clFrag >= 0) {
        cl = (int)((uint32_t)view_cost(clFrag) + (uint32_t)__builtin_amdgcn_readlane(clS, j));
        foundPrev = true;
      }
      const int aboveIdx = __builtin_amdgcn_readlane(faV, j);
      if (aboveIdx >= 0) {
        ca = (int)((uint32_t)view_cost(aboveIdx) + (uint32_t)__builtin_amdgcn_readlane(caS, j));
        foundPrev = true;
      }
      int mc = cl < ca ? cl : ca;
      if (cp < mc) mc = cp;
      int myCost, myPrev;
      uint32_t myLen;
      if (foundPrev && (t.type == 1 || (t.type == 0 && mc < 0))) {
        myCost = (int)((uint32_t)mc - weight);
        if (mc == cp) myPrev = cpFrag;
        else if (mc == cl) myPrev = clFrag;
        else myPrev = aboveIdx;
        myLen = (uint32_t)(mc == cp ? cpLen : view_len(myPrev)) + 1u;
      } else {
        myPrev = -1;
        myLen = 1;
        if (t.type == 1)
          myCost = (int)((uint32_t)(fx + fy) * (uint32_t)prm.sdpDel + fragmentLength * (uint32_t)prm.match - weight);
        else myCost = (int)(fragmentLength * (uint32_t)prm.match - weight);
      }
      if (lane == j) {
        cur.cost = myCost;
        cur.len = (int)myLen;
        curPrev = myPrev;
      }
      if (minCost > myCost) {
        minCost = myCost;
        minIdx = i;
      }
      if (myLen > maxLen) {
        maxLen = myLen;
        maxLenIdx = i;
      }
    }
    if (w0 + lane < F) {
      cost[w0 + lane] = cur.cost;
      chainPrev[w0 + lane] = curPrev;
      chainLen[w0 + lane] = cur.len;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    pendingStore = false;
  }
  if (lane == 0) lv.chainEnd[idx] = t.type == 0 ? minIdx : maxLenIdx;
}

template <int kWords>
__global__ __launch_bounds__(256) void sdp_chain_bits_kernel(SdpLevelArrays lv, SdpParams prm) {
  __shared__ uint32_t bits[4][kWords];
  const int wave = threadIdx.x >> 6;
  const int slot = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wave));
  constexpr int cat = kWords == kBitsWords ? 5 : (kWords == kBitsMidTarget / 32 ? 4 : 3);  // the list: [.. | large | middle | small | ..]
  if (slot >= lv.chainCount[cat]) return;
  const int idx = lv.chainList[chain_list_start(lv, cat) + slot];
  chain_sweep_bits(lv, prm, idx, lv.tasks[idx], lv.nFrag[idx], bits[wave]);
}

// ---------------------------------------------------------------------------
// sdp_chain_rows: the sweep with a whole ROW per step, for the scattered invocations whose rows hold several fragments
// (the recursion's Global calls with a five- or seven-base word on long rectangles: four to sixteen chance matches per
// query row; the longest of them are what a level waits for).
//
// The reference scores every fragment of sweep row x against ONE state of the column set -- it inserts nothing into colSet
// while it scores a row (SparseDynamicProgramming.h:112-236) -- and retires the fragments of trail row x - k - 1 one after
// the other in y order (:248-309).  Here a row is scored lane = fragment: the column predecessors are independent look-ups
// in a three-level bitmap over the target's columns (LDS), one round of entry loads for the row; the only dependency
// inside a row is `above` when it is the fragment just before it in the row (StoreAbove :55-68), resolved in rounds.  A
// trail row is retired lane = fragment: fragment j reads columns >= y_j only (membership of y_j, the successors it deletes),
// fragment i < j writes [y_i, reach_i] (its column and the successors it deleted), so when reach_i < y_j for every i < j the
// two commute: all pending fragments are evaluated against the same state (reads only), the longest prefix whose members
// commute is applied (disjoint ranges: LDS atomics), the rest is evaluated again.  tests/chain_rows_model.c states this
// formulation on the CPU and compares it with the reference's loop fragment by fragment.
// ---------------------------------------------------------------------------
struct RowBits {
  uint32_t *l0, *l1, *l2;  // LDS: a bit per column / per word of l0 / per word of l1 (kRowsWords / 32 / 1024 words)
};
constexpr int kRowsWords = kBitsMaxTarget / 32;  // 4096 words of 32 columns
constexpr int kRowsCache = 0;                    // entries of the row sweep's cost cache ({column, cost} by column mod this; 0: none.
                                                 //  4096 entries make a lone invocation 8 % faster and the step slower: 32 KB more
                                                 //  LDS per wavefront is LDS the other sweep kernels of the level do not get)
constexpr int kRowsDel = 12;                     // deleted successors a lane records per evaluation (3 KB)
__device__ __forceinline__ uint32_t bits_above(int b) { return b >= 31 ? 0u : (~0u << (b + 1)); }   // bits > b
__device__ __forceinline__ uint32_t bits_upto(int b) { return b >= 31 ? ~0u : ((2u << b) - 1u); }   // bits <= b
__device__ __forceinline__ uint32_t bits_below(int b) { return (1u << b) - 1u; }                      // bits < b (b <= 31)
// the smallest member > y, or -1 (every lane its own column; ok = the lane asks)
__device__ __forceinline__ int rb_succ_gt(const RowBits &s, int y, bool ok) {
  y = ok ? y : 0;
  const int w = y >> 5, w1 = w >> 5, w2 = w1 >> 5;
  const uint32_t m0 = s.l0[w] & bits_above(y & 31);
  // (the next word rides along with the first round of reads: with a member every 20-60 columns the successor is in this
  //  word or the next nine times in ten, and the answer needs no second round trip)
  const uint32_t mNext = s.l0[min(w + 1, kRowsWords - 1)];
  const bool nextOk = w + 1 < kRowsWords && mNext != 0u;
  const uint32_t m1 = s.l1[w1] & bits_above(w & 31);
  const uint4 t2 = *reinterpret_cast<const uint4 *>(s.l2);
  const uint32_t q[4] = {t2.x, t2.y, t2.z, t2.w};
  if (__ballot(ok && m0 == 0u && !nextOk) == 0ull) {
    int r = nextOk ? ((w + 1) << 5) + __ffs((int)mNext) - 1 : -1;
    if (m0) r = (w << 5) + __ffs((int)m0) - 1;
    return ok ? r : -1;
  }
  // the first group (word of l1) with a member word beyond w1
  int g = -1;
#pragma unroll
  for (int i = 3; i >= 0; i--) {
    const uint32_t m = i == w2 ? (q[i] & bits_above(w1 & 31)) : (i > w2 ? q[i] : 0u);
    g = m ? (i << 5) + __ffs((int)m) - 1 : g;
  }
  const uint32_t far1 = s.l1[g < 0 ? 0 : g];
  int wN = -1;
  if (g >= 0) wN = (g << 5) + __ffs((int)(far1 | 0x80000000u)) - 1;  // (far1 != 0 when the structure is consistent)
  if (m1) wN = (w1 << 5) + __ffs((int)m1) - 1;
  const uint32_t far0 = s.l0[wN < 0 ? 0 : wN];
  int r = -1;
  if (wN >= 0 && far0) r = (wN << 5) + __ffs((int)far0) - 1;
  if (m0) r = (w << 5) + __ffs((int)m0) - 1;
  return ok ? r : -1;
}
// the largest member <= y, or -1
__device__ __forceinline__ int rb_pred_le(const RowBits &s, int y, bool ok) {
  y = ok ? y : 0;
  const int w = y >> 5, w1 = w >> 5, w2 = w1 >> 5;
  const uint32_t m0 = s.l0[w] & bits_upto(y & 31);
  const uint32_t mPrev = s.l0[max(w - 1, 0)];
  const bool prevOk = w > 0 && mPrev != 0u;
  const uint32_t m1 = s.l1[w1] & bits_below(w & 31);
  const uint4 t2 = *reinterpret_cast<const uint4 *>(s.l2);
  const uint32_t q[4] = {t2.x, t2.y, t2.z, t2.w};
  if (__ballot(ok && m0 == 0u && !prevOk) == 0ull) {
    int r = prevOk ? ((w - 1) << 5) + 31 - __clz((int)mPrev) : -1;
    if (m0) r = (w << 5) + 31 - __clz((int)m0);
    return ok ? r : -1;
  }
  int g = -1;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t m = i == w2 ? (q[i] & bits_below(w1 & 31)) : (i < w2 ? q[i] : 0u);
    g = m ? (i << 5) + 31 - __clz((int)m) : g;
  }
  const uint32_t far1 = s.l1[g < 0 ? 0 : g];
  int wN = -1;
  if (g >= 0) wN = (g << 5) + 31 - __clz((int)(far1 | 1u));
  if (m1) wN = (w1 << 5) + 31 - __clz((int)m1);
  const uint32_t far0 = s.l0[wN < 0 ? 0 : wN];
  int r = -1;
  if (wN >= 0 && far0) r = (wN << 5) + 31 - __clz((int)far0);
  if (m0) r = (w << 5) + 31 - __clz((int)m0);
  return ok ? r : -1;
}
// (LDS operations of a wavefront are carried out in order; this keeps the compiler from moving reads across the writes of
//  other lanes, which it cannot see as dependencies)
__device__ __forceinline__ void rb_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

#ifdef BLASR_ROWS_CLOCKS  // experiment: where a row step's cycles go (make EXTRA=-DBLASR_ROWS_CLOCKS, tools/chain_bench.py)
__device__ unsigned long long g_rowsClocks[16];
#define ROWS_CLOCK(slot)                         \
  do {                                           \
    const unsigned long long now_ = clock64();   \
    rclk[slot] += now_ - rclk_;                  \
    rclk_ = now_;                                \
  } while (0)
#define ROWS_COUNT(slot, n) (rclk[slot] += (unsigned long long)(n))
void sdp_print_rows_clocks() {
  unsigned long long h[16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rowsClocks), sizeof(h)) != hipSuccess) return;
  fprintf(stderr, "[rows clocks] kcycles: load %llu retire-fetch %llu evaluate %llu walk %llu prefix %llu clear %llu set+store %llu query %llu score %llu "
                  "window-end %llu | rows %llu retired-rows %llu rounds %llu walk-iterations %llu clear-iterations %llu score-rounds %llu\n",
          h[0] >> 10, h[1] >> 10, h[2] >> 10, h[3] >> 10, h[4] >> 10, h[5] >> 10, h[6] >> 10, h[7] >> 10, h[8] >> 10, h[9] >> 10, h[10], h[11],
          h[12], h[13], h[14], h[15]);
  unsigned long long z[16] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rowsClocks), z, sizeof(z));
}
#else
#define ROWS_CLOCK(slot) do { } while (0)
#define ROWS_COUNT(slot, n) do { } while (0)
#endif
__global__ __launch_bounds__(64) void sdp_chain_rows_kernel(SdpLevelArrays lv, SdpParams prm) {
  __shared__ __attribute__((aligned(16))) uint32_t ldsBits[kRowsWords + kRowsWords / 32 + 4];
  // The costs of the entries, a second time, in LDS: {column, cost} by column mod kRowsCache, written (one 8-byte store: tag
  // and cost cannot part) whenever an entry is stored.  A retirement reads costs only -- of its own column's entry, of the
  // successors it walks over -- and with them in LDS a step of that walk is three LDS round trips instead of three and one
  // through L2; a slot that another column has taken since is a miss and goes to memory, which stays the truth.
  __shared__ uint2 costCache[kRowsCache > 0 ? kRowsCache : 1];
  // what a retirement's walk found to delete, per lane (the first kRowsDel of a run; a longer run is walked again)
  __shared__ int delList[kRowsDel][64];
  const int slot = (int)blockIdx.x;
  if (slot >= lv.chainCount[6]) return;
  // These are the level's longest invocations and this is a vector program of ~1000 instructions per row: on a CU it shares
  // with thirty wavefronts of the other sweep kernels it waited its turn for every one of them (the kernel three times as
  // slow as alone).  A raised wave priority halves the kernel's own time and does not shorten the level (profiles/NOTES.md,
  // round 5), so the DEFAULT is the ordinary priority; BLASR_HIP_ROWS_PRIO=1 raises it.
  if (lv.rowsPrio) __builtin_amdgcn_s_setprio(3);
  const int idx = lv.chainList[chain_list_start(lv, 6) + slot];
  const SdpTask t = lv.tasks[idx];
  const int F = lv.nFrag[idx];
  const int lane = lane_id();
  const int64_t base = lv.fragOff[idx];
  const Frag *fr = lv.frags + base;
  int *cost = lv.cost + base, *chainPrev = lv.chainPrev + base, *chainLen = lv.chainLen + base;
  const int *predCl = lv.predCl + base, *above = lv.above + base;
  int4 *ent = lv.colEnt + lv.colOff[idx];  // the entry of a member column: {chain length, fragment, cost, x + y}
  RowBits cols;
  cols.l0 = ldsBits;
  cols.l1 = ldsBits + kRowsWords;
  cols.l2 = ldsBits + kRowsWords + kRowsWords / 32;
  for (int w = lane; w < kRowsWords + kRowsWords / 32 + 4; w += 64) ldsBits[w] = 0;
  for (int w = lane; w < kRowsCache; w += 64) costCache[w] = make_uint2(0xffffffffu, 0u);
  rb_sync();
  // the cost of member column `col` (want: the lane asks)
  auto cost_of_col = [&](int col, bool want) {
    int v = 0;
    if (kRowsCache > 0) {
      const uint2 cv = costCache[want ? col & (kRowsCache - 1) : 0];
      const bool miss = want && cv.x != (uint32_t)col;
      v = (int)cv.y;
      if (__ballot(miss) != 0ull)
        if (miss) v = ent[col].z;
    } else if (want) {
      v = ent[col].z;
    }
    return v;
  };
  const int k = t.k;
  const uint32_t fragmentLength = (uint32_t)k, weight = (uint32_t)k;
  int fTrail = 0;
  // per lane: the best cost / the longest chain among the fragments this lane scored, each at its lowest index (the
  // reference's running minimum and maximum use strict comparisons: the best value at its lowest index)
  int bestCost = INT32_MAX, bestCostIdx = INT32_MAX, bestLenIdx = INT32_MAX;
  uint32_t bestLen = 0;
  int curCost = 0, curLen = 0, curPrev = -1, prevCost = 0, prevLen = 0;
  int rw0 = -64, rxV = 0, ryV = 0;  // the retirements' window: fragments rw0 + lane, their row and column
#ifdef BLASR_ROWS_CLOCKS
  unsigned long long rclk[16] = {}, rclk_ = clock64();
#endif
  for (int w0 = 0; w0 < F; w0 += 64) {
    prevCost = curCost;
    prevLen = curLen;
    int fxV = 0, fyV = 0, flV = 0, fpV = -1, faV = -1, clS = 0, caS = 0;
    const bool inWin = w0 + lane < F;
    if (inWin) {
      const Frag f = fr[w0 + lane];
      fxV = (int)f.x;
      fyV = (int)f.y;
      flV = (int)f.len;
      fpV = predCl[w0 + lane];
      faV = above[w0 + lane];
      if (fpV >= 0) {  // what the two fixed predecessors add to their own cost: coordinates only
        const Frag q = fr[fpV];
        int ov = (int)(fragmentLength - (uint32_t)(fyV - (int)q.y)) * prm.match;
        if (ov > 0) ov = 0;
        clS = ov + indel_penalty(fxV, fyV, (int)q.x, (int)q.y, prm.sdpIns, prm.sdpDel);
      }
      if (faV >= 0) {
        const Frag q = fr[faV];
        caS = (int)((fragmentLength - (uint32_t)(fyV - (int)q.y)) * (uint32_t)prm.match +
                    (uint32_t)indel_penalty(fxV, fyV, (int)q.x, (int)q.y, prm.sdpIns, prm.sdpDel));
      }
    }
    curCost = 0, curLen = 0, curPrev = -1;
    const int nHere = min(64, F - w0);
    ROWS_CLOCK(0);
    // cost / chain length of fragment p (< the row being scored), per lane: this window, the one before, or memory
    auto cost_len_of = [&](int p, bool want, int &c, int &l) {
      const int rel = p - w0;
      const int vc = __builtin_amdgcn_ds_bpermute((rel & 63) << 2, curCost), vl = __builtin_amdgcn_ds_bpermute((rel & 63) << 2, curLen);
      const int pc = __builtin_amdgcn_ds_bpermute((rel & 63) << 2, prevCost), pl = __builtin_amdgcn_ds_bpermute((rel & 63) << 2, prevLen);
      c = rel >= 0 ? vc : pc;
      l = rel >= 0 ? vl : pl;
      const bool far = want && rel < -64;
      if (__ballot(far) != 0ull) {  // (stored at the end of its window, two windows ago or earlier)
        if (far) {
          c = cost[p];
          l = chainLen[p];
        }
      }
    };
    for (int j = 0; j < nHere;) {
      const int fx = __builtin_amdgcn_readlane(fxV, j);
      const unsigned long long rowMask = __ballot(inWin && lane >= j && fxV == fx);  // (sorted by x: contiguous from j)
      const int jEnd = j + (int)__popcll(rowMask);
      // ---- rows leaving the window: fragments with x <= fx - k - 2 are retired before row fx is scored (:248-309)
      while (fTrail < w0 + j) {
        if (fTrail >= rw0 + 64) {
          rw0 = fTrail;
          rxV = 0, ryV = 0;
          if (rw0 + lane < F) {
            const Frag f = fr[rw0 + lane];
            rxV = (int)f.x;
            ryV = (int)f.y;
          }
        }
        const int rl = fTrail - rw0;
        const int tx = __builtin_amdgcn_readlane(rxV, rl);
        if ((int64_t)tx + k + 2 > (int64_t)fx) break;
        const unsigned long long retMask = __ballot(lane >= rl && rw0 + lane < w0 + j && rxV == tx);
        const bool mine = ((retMask >> lane) & 1ull) != 0ull;
        const int fi = rw0 + lane;
        int c = 0, len = 0;
        cost_len_of(fi, mine, c, len);
        ROWS_CLOCK(1);
        ROWS_COUNT(11, 1);
        // -- the row (or the piece of it the window holds), increasing y: evaluate / longest commuting prefix / apply
        unsigned long long pending = retMask;
        while (pending != 0ull) {
          const bool pend = ((pending >> lane) & 1ull) != 0ull;
          const bool member = pend && ((cols.l0[pend ? ryV >> 5 : 0] >> (ryV & 31)) & 1u) != 0u;
          const int exist = cost_of_col(ryV, member);
          const bool store = pend && (!member || exist < c);  // sic (:260-267): replaced only when the existing entry is CHEAPER
          int reach = ryV, nDel = 0;
          int s = rb_succ_gt(cols, ryV, store);
          bool act = store && s >= 0;
          ROWS_CLOCK(2);
          ROWS_COUNT(12, 1);
          while (__ballot(act) != 0ull) {  // delete the costlier successors: every lane walks its own run
            const int sc = cost_of_col(s, act);
            const bool del = act && sc > c;
            if (del && nDel < kRowsDel) delList[nDel][lane] = s;
            nDel += del ? 1 : 0;
            reach = del ? s : reach;
            const int nx = rb_succ_gt(cols, s, del);
            s = del ? nx : s;
            act = del && nx >= 0;
            ROWS_COUNT(13, 1);
          }
          ROWS_CLOCK(3);
          // the longest prefix of pending lanes whose writes stay left of the next lane's column
          const int wr = store ? reach : -1;
          const int incl = wave_prefix_max(pend ? wr : -1);
          const int excl = wave_shr1(incl, -1);
          const unsigned long long dirty = __ballot(pend && excl >= ryV);
          const unsigned long long applyMask = dirty ? (pending & ((1ull << __builtin_ctzll(dirty)) - 1ull)) : pending;
          const bool app = ((applyMask >> lane) & 1ull) != 0ull && store;
          ROWS_CLOCK(4);
          // their writes: the deleted successors' bits first -- the columns the walk listed (no look-up between two of them:
          // the atomics of consecutive list entries are independent), a run beyond the list walked again from its last
          // entry -- then the own column
          {
            rb_sync();
            const int nList = app ? min(nDel, kRowsDel) : 0;
            const int longest = wave_max(nList);
            auto clear_col = [&](int tcol) {
              const int w = tcol >> 5;
              const uint32_t bit = 1u << (tcol & 31);
              const uint32_t old = atomicAnd(&cols.l0[w], ~bit);
              if ((old & ~bit) == 0u) {
                const uint32_t bit1 = 1u << (w & 31);
                const uint32_t old1 = atomicAnd(&cols.l1[w >> 5], ~bit1);
                if ((old1 & ~bit1) == 0u) atomicAnd(&cols.l2[w >> 10], ~(1u << ((w >> 5) & 31)));
              }
            };
            for (int i = 0; i < longest; i++) {
              const int tcol = delList[i][lane];
              if (i < nList) clear_col(tcol);
              ROWS_COUNT(14, 1);
            }
            if (__ballot(app && nDel > kRowsDel) != 0ull) {  // (rare: a run longer than the list)
              rb_sync();
              int tcol = app && nDel > kRowsDel ? delList[kRowsDel - 1][lane] : 0;
              bool go = app && nDel > kRowsDel;
              while (__ballot(go) != 0ull) {
                const int nx = rb_succ_gt(cols, tcol, go && tcol < reach);
                go = go && nx >= 0 && nx <= reach;
                tcol = nx;
                if (go) clear_col(tcol);
                rb_sync();
              }
            }
          }
          rb_sync();
          ROWS_CLOCK(5);
          if (app) {
            const int w = ryV >> 5;
            atomicOr(&cols.l0[w], 1u << (ryV & 31));
            atomicOr(&cols.l1[w >> 5], 1u << (w & 31));
            atomicOr(&cols.l2[w >> 10], 1u << ((w >> 5) & 31));
            ent[ryV] = make_int4(len, fi, c, rxV + ryV);
            if (kRowsCache > 0) costCache[ryV & (kRowsCache - 1)] = make_uint2((uint32_t)ryV, (uint32_t)c);
          }
          rb_sync();  // (the entries are read again by the next evaluation and by the row's queries)
          ROWS_CLOCK(6);
          pending &= ~applyMask;
        }
        fTrail += (int)__popcll(retMask);
      }
      // ---- score the row's lanes: column predecessors against the set as it stands, then the three candidates (:112-236)
      {
        const bool row = ((rowMask >> lane) & 1ull) != 0ull;
        const int pc = rb_pred_le(cols, fyV, row);
        int4 at = make_int4(0, 0, 0, 0);
        if (pc >= 0) at = ent[pc];
        int clCost = 0, clLen = 0, caCost = 0, caLen = 0;
        cost_len_of(fpV, row && fpV >= 0, clCost, clLen);
        // `above` inside this very piece of the row (the fragment just before it in the row) waits for its turn
        bool wait = row && faV >= w0 + j;
        cost_len_of(faV, row && faV >= 0 && !wait, caCost, caLen);
        bool todo = row;
        ROWS_CLOCK(7);
        ROWS_COUNT(10, 1);
        while (__ballot(todo) != 0ull) {
          // (a lane whose `above` is in this piece of the row goes once that lane HAS BEEN scored, i.e. in a later round than
          //  it; the permutes are carried out by every lane: a lane that is switched off reads as zero)
          const int aRel = (faV - w0) & 63;
          const bool aboveTodo = __builtin_amdgcn_ds_bpermute(aRel << 2, (int)todo) != 0;
          const int aCost = __builtin_amdgcn_ds_bpermute(aRel << 2, curCost), aLen = __builtin_amdgcn_ds_bpermute(aRel << 2, curLen);
          const bool now = todo && !(wait && aboveTodo);
          if (wait) {
            caCost = aCost;
            caLen = aLen;
          }
          int cp = INT32_MAX, cl = INT32_MAX, ca = INT32_MAX;
          bool found = false;
          if (pc >= 0) {
            cp = at.z + isqrt_floor(abs((fxV + fyV) - at.w)) - flV;
            found = true;
          }
          if (fpV >= 0) {
            cl = (int)((uint32_t)clCost + (uint32_t)clS);
            found = true;
          }
          if (faV >= 0) {
            ca = (int)((uint32_t)caCost + (uint32_t)caS);
            found = true;
          }
          int mc = cl < ca ? cl : ca;
          if (cp < mc) mc = cp;
          int myCost, myPrev;
          uint32_t myLen;
          if (found && (t.type == 1 || (t.type == 0 && mc < 0))) {
            myCost = (int)((uint32_t)mc - weight);
            if (mc == cp) myPrev = at.y, myLen = (uint32_t)at.x + 1u;
            else if (mc == cl) myPrev = fpV, myLen = (uint32_t)clLen + 1u;
            else myPrev = faV, myLen = (uint32_t)caLen + 1u;
          } else {
            myPrev = -1;
            myLen = 1;
            if (t.type == 1)
              myCost = (int)((uint32_t)(fxV + fyV) * (uint32_t)prm.sdpDel + fragmentLength * (uint32_t)prm.match - weight);
            else myCost = (int)(fragmentLength * (uint32_t)prm.match - weight);
          }
          if (now) {
            curCost = myCost;
            curLen = (int)myLen;
            curPrev = myPrev;
            const int i = w0 + lane;
            if (bestCost > myCost) bestCost = myCost, bestCostIdx = i;
            if (myLen > bestLen) bestLen = myLen, bestLenIdx = i;
          }
          todo = todo && !now;
          ROWS_COUNT(15, 1);
        }
        ROWS_CLOCK(8);
      }
      j = jEnd;
    }
    if (inWin) {
      cost[w0 + lane] = curCost;
      chainPrev[w0 + lane] = curPrev;
      chainLen[w0 + lane] = curLen;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    ROWS_CLOCK(9);
  }
#ifdef BLASR_ROWS_CLOCKS
  if (lane == 0)
    for (int i = 0; i < 16; i++) atomicAdd(&g_rowsClocks[i], rclk[i]);
#endif
  // the minimum cost (Local) / the longest chain (Global), each at its lowest index
  int endIdx;
  if (t.type == 0) {
    int v = bestCost;
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off));
    int i = bestCost == v ? bestCostIdx : INT32_MAX;
    for (int off = 32; off > 0; off >>= 1) i = min(i, __shfl_xor(i, off));
    endIdx = i == INT32_MAX ? -1 : i;
  } else {
    uint32_t v = bestLen;
    for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off));
    int i = (bestLen == v && v > 0) ? bestLenIdx : INT32_MAX;
    for (int off = 32; off > 0; off >>= 1) i = min(i, __shfl_xor(i, off));
    endIdx = i == INT32_MAX ? -1 : i;
  }
  if (lane == 0) lv.chainEnd[idx] = endIdx;
}

// ---------------------------------------------------------------------------
// sdp_chain_bits4: the bitmap sweep, FOUR invocations per wavefront (one per 16-lane group, as sdp_chain4_kernel).
// One invocation per wavefront is a scalar program (two thirds of its instructions run on the CU's one scalar unit)
// and is bound by that unit; four per wavefront do the same work as vector instructions, a fourth of them per
// fragment.  Per group: kWords words of bitmap in LDS, the summary one word per lane (16 lanes x 32 bits = 512 words
// at most), the windows of 16 + 16 fragments of chain4_body.  The four step in lockstep, so this kernel takes the
// many SHORT scattered invocations (sorted by size, four of a kind per wavefront); the long ones -- a wavefront is as
// slow as its longest -- keep a wavefront each (sdp_chain_bits_kernel).
// ---------------------------------------------------------------------------
template <int kWords>
__device__ __forceinline__ void chain_bits4_body(const SdpLevelArrays &lv, const SdpParams &prm, const bool on,
                                                 const int idx, uint32_t *bits /* this group's words */) {
  const int lane = lane_id(), l = lane & 15, gb = lane & 48;
  const int F = on ? lv.nFrag[idx] : 0;
  const SdpTask t = lv.tasks[idx];
  const int64_t base = lv.fragOff[idx];
  int4 *ent = lv.colEnt + (on ? lv.colOff[idx] : 0);
  const int k = t.k, type = t.type;
  auto gbp = [&](int v, int i) { return __builtin_amdgcn_ds_bpermute((gb + (i & 15)) << 2, v); };
  auto gballot = [&](bool flag) { return (uint32_t)((__ballot(flag) >> gb) & 0xffffull); };
  for (int w = l; w < kWords; w += 16) bits[w] = 0;
  __builtin_amdgcn_wave_barrier();
  uint32_t summary = 0;  // lane l, bit b: word 32 l + b of the group's map is not empty
  int n = 0, fTrail = 0;
  int curX = 0, curY = 0, curCost = 0, curLen = 0, curPrev = -1, prvX = 0, prvY = 0, prvCost = 0, prvLen = 0;
  int minCost = INT32_MAX, minIdx = -1, maxLenIdx = -1;
  uint32_t maxLen = 0;
  int maxF = F;
  for (int off = 32; off > 0; off >>= 1) maxF = max(maxF, __shfl_xor(maxF, off));
  // the largest member <= y (-1: none) / the smallest member > y (-1: none); `ok` lanes only, the others read word 0
  auto pred_le = [&](int y, bool ok) {
    const int w = ok ? y >> 5 : 0, b = y & 31;
    const uint32_t word = bits[w] & (b == 31 ? 0xffffffffu : ((2u << b) - 1u));
    const int li = w >> 5, bi = w & 31;
    uint32_t m = summary;
    m = l == li ? (m & ((1u << bi) - 1u)) : (l > li ? 0u : m);
    const uint32_t gm = gballot(m != 0u);
    const int hl = 31 - __clz((int)(gm | 1u));
    const uint32_t hm = (uint32_t)gbp((int)m, hl);
    const int w2 = gm ? (hl << 5) + 31 - __clz((int)(hm | 1u)) : 0;
    const uint32_t word2 = bits[w2];
    int r = -1;
    if (gm) r = (w2 << 5) + 31 - __clz((int)(word2 | 1u));
    if (word) r = (w << 5) + 31 - __clz((int)word);
    return ok ? r : -1;
  };
  auto succ_gt = [&](int y, bool ok) {
    const int w = ok ? y >> 5 : 0, b = y & 31;
    const uint32_t word = bits[w] & (b == 31 ? 0u : ~((2u << b) - 1u));
    const int li = w >> 5, bi = w & 31;
    uint32_t m = summary;
    m = l == li ? (bi == 31 ? 0u : (m & ~((2u << bi) - 1u))) : (l < li ? 0u : m);
    const uint32_t gm = gballot(m != 0u);
    const int ll = gm ? __ffs((int)gm) - 1 : 0;
    const uint32_t lm = (uint32_t)gbp((int)m, ll);
    const int w2 = gm ? (ll << 5) + __ffs((int)lm) - 1 : 0;
    const uint32_t word2 = bits[w2];
    int r = -1;
    if (gm && word2) r = (w2 << 5) + __ffs((int)word2) - 1;
    if (word) r = (w << 5) + __ffs((int)word) - 1;
    return ok ? r : -1;
  };
  for (int w0 = 0; w0 < maxF; w0 += 16) {
    prvX = curX, prvY = curY, prvCost = curCost, prvLen = curLen;
    int flV = 0, fpV = -1, faV = -1;
    int pX = 0, pY = 0, aX = 0, aY = 0;  // the two fixed predecessors' coordinates (chain4_body says why)
    curX = curY = 0;
    if (on && w0 + l < F) {
      const Frag f = lv.frags[base + w0 + l];
      curX = (int)f.x;
      curY = (int)f.y;
      flV = (int)f.len;
      fpV = lv.predCl[base + w0 + l];
      faV = lv.above[base + w0 + l];
      if (fpV >= 0) {
        const Frag q = lv.frags[base + fpV];
        pX = (int)q.x, pY = (int)q.y;
      }
      if (faV >= 0) {
        const Frag q = lv.frags[base + faV];
        aX = (int)q.x, aY = (int)q.y;
      }
    }
    int clS, caS;  // what they add to their own cost, lane = fragment
    {
      int ov = (int)((uint32_t)k - (uint32_t)(curY - pY)) * prm.match;
      if (ov > 0) ov = 0;
      clS = ov + indel_penalty(curX, curY, pX, pY, prm.sdpIns, prm.sdpDel);
      caS = (int)(((uint32_t)k - (uint32_t)(curY - aY)) * (uint32_t)prm.match +
                  (uint32_t)indel_penalty(curX, curY, aX, aY, prm.sdpIns, prm.sdpDel));
    }
    // fragment p of this group: cost and chain length, from the two register windows, else from memory
    auto view2 = [&](int p, bool need, int &vc, int &vl) {
      const int rel = p - w0;
      const int ca = gbp(curCost, rel), la = gbp(curLen, rel);
      const int cb = gbp(prvCost, rel + 16), lb = gbp(prvLen, rel + 16);
      vc = rel >= 0 ? ca : cb;
      vl = rel >= 0 ? la : lb;
      if (__ballot(need && rel < -16) != 0ull) {
        if (need && rel < -16) {
          vc = lv.cost[base + p];
          vl = lv.chainLen[base + p];
        }
      }
    };
    for (int j = 0; j < 16; j++) {
      const int i = w0 + j;
      const bool act = on && i < F;
      if (__ballot(act) == 0ull) break;
      const int fx = gbp(curX, j), fy = gbp(curY, j), fl = gbp(flV, j);
      // ---- rows leaving the window: fragments with x <= f.x - k - 2 are retired first (:248-309)
      for (;;) {
        bool need = act && fTrail < i;
        // (the test that ends the loop needs the row alone)
        const int pT = need ? fTrail : i, relT = pT - w0;
        int ftx, fty = 0, ftc, ftl;
        {
          const int xa = gbp(curX, relT), xb = gbp(prvX, relT + 16);
          ftx = relT >= 0 ? xa : xb;
        }
        const bool farT = need && relT < -16;
        if (__ballot(farT) != 0ull) {
          if (farT) {
            const Frag f = lv.frags[base + pT];
            ftx = (int)f.x;
            fty = (int)f.y;
          }
        }
        need = need && (int64_t)ftx + k + 2 <= (int64_t)fx;
        if (__ballot(need) == 0ull) break;
        {
          const int ya = gbp(curY, relT), yb = gbp(prvY, relT + 16);
          if (!farT) fty = relT >= 0 ? ya : yb;
        }
        view2(pT, need, ftc, ftl);
        const int w = need ? fty >> 5 : 0;
        const uint32_t word = bits[w], bit = 1u << (fty & 31);
        const bool member = need && (word & bit) != 0u;
        // the first successor, before the entry goes in (the bit of a greater column does not depend on it): its
        // cost and the store below are one round trip.  Successor() needs at least two elements (SDPSet.h:72-87)
        int sc = succ_gt(fty, need && n + (member ? 0 : 1) >= 2);
        int atCost = 0, scCost = 0;
        if (__ballot(member || sc >= 0) != 0ull) {
          if (member) atCost = ent[fty].z;
          if (sc >= 0) scCost = ent[sc].z;
        }
        bool store = need && (!member || atCost < ftc);  // sic (:260-267): replaced only when the existing entry is CHEAPER
        if (store && l == 0) ent[fty] = make_int4(ftl, fTrail, ftc, ftx + fty);
        const bool ins = store && !member;
        if (ins && l == 0) bits[w] = word | bit;
        if (ins && l == (w >> 5)) summary |= 1u << (w & 31);
        n += ins ? 1 : 0;
        __builtin_amdgcn_wave_barrier();
        // delete the costlier successors
        for (;;) {
          const bool del = store && sc >= 0 && scCost > ftc;
          if (__ballot(del) == 0ull) break;
          const int ws = del ? sc >> 5 : 0;
          const uint32_t wv = bits[ws] & ~(1u << (sc & 31));
          if (del && l == 0) bits[ws] = wv;
          if (del && wv == 0u && l == (ws >> 5)) summary &= ~(1u << (ws & 31));
          n -= del ? 1 : 0;
          __builtin_amdgcn_wave_barrier();
          store = del;  // a group goes on only while it deletes
          sc = succ_gt(fty, del && n >= 2);
          if (__ballot(sc >= 0) != 0ull)
            if (sc >= 0) scCost = ent[sc].z;  // (never the entry just stored: sc > fty)
        }
        // the entry must have landed before a later step loads it
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        fTrail += need ? 1 : 0;
      }
      // ---- the three predecessors
      int cp = INT32_MAX, cl = INT32_MAX, ca = INT32_MAX, cpFrag = -1, cpLen = 0;
      bool foundPrev = false;
      {
        const bool q = act && n > 0;
        const int pc = pred_le(fy, q);
        const bool have = q && pc >= 0;
        int4 at = make_int4(0, 0, 0, 0);
        if (__ballot(have) != 0ull)
          if (have) at = ent[pc];
        if (have) {
          const int d = abs((fx + fy) - at.w);
          const int sq = isqrt_floor(d);
          cpFrag = at.y;
          cpLen = at.x;
          cp = at.z + sq - fl;
          foundPrev = true;
        }
      }
      const int clFrag = gbp(fpV, j), aboveIdx = gbp(faV, j);
      int clLen = 0, caLen = 0;
      {
        const int clSj = gbp(clS, j), caSj = gbp(caS, j);
        int pc;
        const bool needCl = act && clFrag >= 0;
        view2(needCl ? clFrag : i, needCl, pc, clLen);
        if (needCl) {
          cl = (int)((uint32_t)pc + (uint32_t)clSj);
          foundPrev = true;
        }
        const bool needCa = act && aboveIdx >= 0;
        view2(needCa ? aboveIdx : i, needCa, pc, caLen);
        if (needCa) {
          ca = (int)((uint32_t)pc + (uint32_t)caSj);
          foundPrev = true;
        }
      }
      int mc = cl < ca ? cl : ca;
      if (cp < mc) mc = cp;
      int myCost, myPrev = -1;
      uint32_t myLen = 1;
      const bool linked = foundPrev && (type == 1 || (type == 0 && mc < 0));
      if (linked) {
        myPrev = (mc == cp) ? cpFrag : ((mc == cl) ? clFrag : aboveIdx);
        myLen = (uint32_t)((mc == cp) ? cpLen : ((mc == cl) ? clLen : caLen)) + 1u;
        myCost = (int)((uint32_t)mc - (uint32_t)k);
      } else if (type == 1) {
        myCost = (int)((uint32_t)(fx + fy) * (uint32_t)prm.sdpDel + (uint32_t)k * (uint32_t)prm.match - (uint32_t)k);
      } else {
        myCost = (int)((uint32_t)k * (uint32_t)prm.match - (uint32_t)k);
      }
      if (act && l == j) {
        curCost = myCost;
        curLen = (int)myLen;
        curPrev = myPrev;
      }
      if (act && minCost > myCost) {
        minCost = myCost;
        minIdx = i;
      }
      if (act && myLen > maxLen) {
        maxLen = myLen;
        maxLenIdx = i;
      }
    }
    if (on && w0 + l < F) {
      lv.cost[base + w0 + l] = curCost;
      lv.chainPrev[base + w0 + l] = curPrev;
      lv.chainLen[base + w0 + l] = curLen;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  if (on && l == 0) lv.chainEnd[idx] = type == 0 ? minIdx : maxLenIdx;
}

// category 2 of the chain list (small targets, few fragments), four per wavefront; kWide: category 7, the targets of up to
// kBitsSmallTarget columns with as few fragments (32 KB of maps per workgroup: five workgroups per CU instead of eight)
template <bool kWide>
__global__ __launch_bounds__(256, kWide ? 4 : 8) void sdp_chain_bits4_kernel(SdpLevelArrays lv, SdpParams prm) {
  constexpr int kWords = (kWide ? kBitsSmallTarget : kBits4Target) / 32, kCat = kWide ? 7 : 2;
  __shared__ uint32_t bits[4][4][kWords];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int nList = lv.chainCount[kCat];
  const int slot = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wave)) * 4 + (lane >> 4);
  if (__builtin_amdgcn_readfirstlane(slot) >= nList) return;
  const bool on = slot < nList;
  chain_bits4_body<kWords>(lv, prm, on, on ? lv.chainList[chain_list_start(lv, kCat) + slot] : 0, bits[wave][lane >> 4]);
}

// sdp_chain_wide: one scattered invocation per wavefront, the sweep of sdp_chain_kernel with 256 entries of the column
// set in registers (ColSetT<4>)
__global__ __launch_bounds__(256) void sdp_chain_wide_kernel(SdpLevelArrays lv, SdpParams prm) {
  const int slot = wave_task_index();
  if (slot >= lv.chainCount[1]) return;
  const int idx = lv.chainList[chain_list_start(lv, 1) + slot];
  chain_sweep<false, kWideSets>(lv, prm, idx, lv.tasks[idx], lv.nFrag[idx], 0, 0, 0, -1, -1);
}

// sdp_small4_kernel: what sdp_small_kernel does, for invocations of at most kSmall4Max raw fragments (most of
// the recursion levels' millions) and FOUR of them per wavefront, one per 16-lane group: ranks by all-pairs
// comparison inside the group, duplicates dropped, StoreAbove and the sweep set's predecessor the same way; the
// sweep itself is chain4_body's.
constexpr int kSmall4Max = 16;
// order[] lists the level's invocations by their number of raw fragments, descending (those with more than
// kSmall4Max first): the four of a wavefront are of a size -- its loops run to the largest -- and the big ones go
// first.  sdp_small4_keys_kernel writes the sort keys.
__global__ __launch_bounds__(256) void sdp_small4_keys_kernel(SdpLevelArrays lv, uint32_t *keys, int32_t *vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int nAll = i < lv.n ? lv.nRaw[3 * i] + lv.nRaw[3 * i + 1] + lv.nRaw[3 * i + 2] : 0;
  if (i < lv.n) {
    keys[i] = (uint32_t)min(nAll, kSmall4Max + 1);
    vals[i] = i;
  }
  // the invocations no small kernel fuses (more than 64 raw fragments), as a list: the general kernels take them from
  // there beside the small ones (general_task), instead of a wavefront per invocation of the level that mostly finds nothing
  if (lv.bigGenList != nullptr) {
    const bool big = nAll > 64;
    const unsigned long long m = __ballot(big);
    if (m != 0ull) {
      int at = 0;
      if (lane_id() == 0) at = atomicAdd(lv.bigGenCount, (int)__popcll(m));
      at = __builtin_amdgcn_readfirstlane(at);
      if (big) lv.bigGenList[at + (int)__popcll(m & ((1ull << lane_id()) - 1ull))] = i;
    }
  }
}
__device__ __forceinline__ void small4_slots(const SdpLevelArrays &lv, const SdpParams &prm, const int32_t *order, const int slot0);
__global__ __launch_bounds__(256, 8) void sdp_small4_kernel(SdpLevelArrays lv, SdpParams prm, const int32_t *order) {
  // (a grid of fewer workgroups than the level has slots strides over them: launch_sdp_small)
  for (int b = (int)blockIdx.x; b * 16 < lv.n; b += (int)gridDim.x)
    small4_slots(lv, prm, order, __builtin_amdgcn_readfirstlane((int)(b * 4 + (threadIdx.x >> 6))) * 4);
}
__device__ __forceinline__ void small4_slots(const SdpLevelArrays &lv, const SdpParams &prm, const int32_t *order, const int slot0) {
  const int lane = lane_id(), l = lane & 15, gb = lane & 48;
  if (slot0 >= lv.n) return;
  const int slot = slot0 + (lane >> 4);
  const int idx0 = order[slot0];  // any valid invocation, for the lanes without one
  const int idx = slot < lv.n ? order[slot] : idx0;
  int nAll = 0;
  if (slot < lv.n) nAll = lv.nRaw[3 * idx] + lv.nRaw[3 * idx + 1] + lv.nRaw[3 * idx + 2];
  const bool mine = slot < lv.n && nAll <= kSmall4Max;
  if (__ballot(mine) == 0ull) return;
  const int64_t base = mine ? lv.fragOff[idx] : 0;
  auto gbp = [&](int v, int i) { return __builtin_amdgcn_ds_bpermute((gb + (i & 15)) << 2, v); };
  auto gmask = [&](bool flag) { return (unsigned)((__ballot(flag) >> gb) & 0xffffull); };
  uint32_t x = 0xffffffffu, y = 0xffffffffu, len = 0;
  const bool raw = mine && l < nAll;
  int maxN = mine ? nAll : 0;  // the loops below stop at the largest invocation of the wavefront
  for (int off = 32; off >= 16; off >>= 1) maxN = max(maxN, __shfl_xor(maxN, off));
  maxN = __builtin_amdgcn_readfirstlane(maxN);
  if (raw) {
    const Frag f = lv.raw[base + l];
    x = f.x;
    y = f.y;
    len = f.len;
  }
  // ---- rank by (x,y), equal keys in input order
  int rank = 0;
  bool dup = false;
  for (int j = 0; j < maxN; j++) {
    const uint32_t xj = (uint32_t)gbp((int)x, j), yj = (uint32_t)gbp((int)y, j);
    const bool eq = xj == x && yj == y;
    const bool less = xj < x || (xj == x && yj < y);
    rank += (less || (eq && j < l)) ? 1 : 0;
    dup = dup || (eq && j < l);
  }
  dup = dup && raw;
  const unsigned dupMask = gmask(dup);
  const int F = mine ? nAll - (int)__popc(dupMask) : 0;  // (a group without an invocation must permute onto itself)
  // ---- drop the later duplicates, place the survivors at their rank among the survivors
  int dest = 16;  // lanes without a fragment and dropped duplicates park beyond the survivors
  {
    int droppedBefore = 0;
    if (__ballot(dup) != 0ull) {
      for (int j = 0; j < maxN; j++) {
        const uint32_t xj = (uint32_t)gbp((int)x, j), yj = (uint32_t)gbp((int)y, j);
        const bool dj = ((dupMask >> j) & 1u) != 0u;
        droppedBefore += (dj && (xj < x || (xj == x && (yj < y || (yj == y && j < l))))) ? 1 : 0;
      }
    }
    if (raw && !dup) dest = rank - droppedBefore;
  }
  const unsigned liveMask = gmask(dest < 16);
  if (dest >= 16) dest = F + (int)__popc(~liveMask & ((1u << l) - 1u) & 0xffffu);
  const int fx = __builtin_amdgcn_ds_permute((gb + dest) * 4, (int)x), fy = __builtin_amdgcn_ds_permute((gb + dest) * 4, (int)y);
  const int fl = __builtin_amdgcn_ds_permute((gb + dest) * 4, (int)len);
  const bool on = mine && l < F;
  if (on) {
    Frag f;
    f.x = (uint32_t)fx;
    f.y = (uint32_t)fy;
    f.len = (uint32_t)fl;
    f.pad = 0;
    lv.frags[base + l] = f;
  }
  if (mine && l == 0) {
    lv.nFrag[idx] = F;
    lv.fused[idx] = 1;
  }
  const int k = lv.tasks[idx].k;
  // ---- StoreAbove (SparseDynamicProgramming.h:55-68): the neighbour in (y,x) order
  int r2 = 0;
  for (int j = 0; j < maxN; j++) {
    const uint32_t xj = (uint32_t)gbp(fx, j), yj = (uint32_t)gbp(fy, j);
    r2 += (j < F && (yj < (uint32_t)fy || (yj == (uint32_t)fy && xj < (uint32_t)fx))) ? 1 : 0;
  }
  if (!on) r2 = l;  // keep the permutation a bijection
  const int byYx = __builtin_amdgcn_ds_permute((gb + r2) * 4, l);  // lane r: the fragment with (y,x) rank r
  int above = -1;
  {
    const int a = gbp(byYx, max(r2 - 1, 0));
    const uint32_t ax = (uint32_t)gbp(fx, a), ay = (uint32_t)gbp(fy, a), al = (uint32_t)gbp(fl, a);
    if (on && r2 > 0 && ax <= (uint32_t)fx && ax + al > (uint32_t)fx && ay < (uint32_t)fy) above = a;
  }
  // ---- sweepSet.Predecessor: the largest (diag, x) with diag <= f.diag among x in [f.x-k-1, f.x-1]
  int pred = -1;
  {
    const int fdiag = fy - fx;
    const uint32_t xLow = (uint32_t)fx > (uint32_t)(k + 1) ? (uint32_t)fx - (uint32_t)(k + 1) : 0u;
    int bestDiag = 0;
    uint32_t bestX = 0;
    for (int j = 0; j < maxN; j++) {
      const uint32_t xj = (uint32_t)gbp(fx, j);
      const int d = gbp(fy, j) - (int)xj;
      if (j < l && xj >= xLow && xj < (uint32_t)fx && d <= fdiag &&
          (pred < 0 || d > bestDiag || (d == bestDiag && xj > bestX))) {
        pred = j;
        bestDiag = d;
        bestX = xj;
      }
    }
    if (!on) pred = -1;
  }
  chain4_body<true>(lv, prm, mine, mine ? idx : 0, F, fx, fy, fl, pred, above);
}

__global__ __launch_bounds__(256) void sdp_chain_kernel(SdpLevelArrays lv, SdpParams prm) {
  const int idx = wave_task_index();
  if (idx >= lv.n) return;
  if (lv.fused[idx]) return;  // sdp_small_kernel's
  chain_sweep<false>(lv, prm, idx, lv.tasks[idx], lv.nFrag[idx], 0, 0, 0, -1, -1);
}

// ---------------------------------------------------------------------------
// sdp_small: an invocation with at most 64 raw fragments (nearly all of the recursion levels: 7 and 1.3
// fragments on average at configs[1]) does everything between the match and the blocks in ONE kernel, with the
// fragments one per lane: sort + keep-first (SDPAlign.h:256-269), StoreAbove, the sweep-set predecessor and
// the sweep itself.  All-pairs comparisons through v_readlane replace the sorts.
// An invocation whose lists tie on an (x,y) AND hold more than 16 fragments needs std::sort's schedule
// replayed; it is left to the general kernels (lv.fused[idx] = 0).  With <= 16 elements std::sort is a plain
// insertion sort, i.e. stable, so the first of the duplicates in prefix | middle | suffix order survives.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void small_task(const SdpLevelArrays &lv, const SdpParams &prm, const int idx, const int minAll);
// With sdp_small4_kernel's order (the invocations sorted by min(fragments, 17), largest first) the kernel strides over the
// head of that list with a fixed grid and stops at the first invocation that is sdp_small4_kernel's: at the recursion levels
// nine invocations in ten are, and a wavefront each that returned at once was most of this kernel's time.
__global__ __launch_bounds__(256) void sdp_small_kernel(SdpLevelArrays lv, SdpParams prm, int minAll, const int32_t *order) {
  if (order != nullptr) {
    for (int e = wave_task_index(); e < lv.n; e += (int)gridDim.x * 4) {
      const int idx = order[e];
      if (lv.nRaw[3 * idx] + lv.nRaw[3 * idx + 1] + lv.nRaw[3 * idx + 2] < minAll) return;
      small_task(lv, prm, idx, minAll);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  const int idx = wave_task_index();
  if (idx >= lv.n) return;
  small_task(lv, prm, idx, minAll);
}
__device__ __forceinline__ void small_task(const SdpLevelArrays &lv, const SdpParams &prm, const int idx, const int minAll) {
  const int lane = lane_id();
  const int nAll = lv.nRaw[3 * idx] + lv.nRaw[3 * idx + 1] + lv.nRaw[3 * idx + 2];
  if (nAll < minAll) return;  // sdp_small4_kernel's
  if (nAll > 64) {
    if (lane == 0) lv.fused[idx] = 0;
    return;
  }
  const int64_t base = lv.fragOff[idx];
  uint32_t x = 0xffffffffu, y = 0xffffffffu, len = 0;
  if (lane < nAll) {
    const Frag f = lv.raw[base + lane];
    x = f.x;
    y = f.y;
    len = f.len;
  }
  // ---- rank by (x,y), equal keys in input order
  int rank = 0;
  bool dup = false;
  for (int j = 0; j < nAll; j++) {
    const uint32_t xj = (uint32_t)__builtin_amdgcn_readlane((int)x, j), yj = (uint32_t)__builtin_amdgcn_readlane((int)y, j);
    const bool eq = xj == x && yj == y;
    const bool less = xj < x || (xj == x && yj < y);
    rank += (less || (eq && j < lane)) ? 1 : 0;
    dup = dup || (eq && j < lane);
  }
  dup = dup && lane < nAll;
  const unsigned long long dupMask = __ballot(dup);
  if (dupMask != 0ull && nAll > 16) {
    if (lane == 0) {
      lv.fused[idx] = 0;
      if (lv.leftList != nullptr) lv.leftList[atomicAdd(lv.leftCount, 1)] = idx;  // (general_task, mode 2)
    }
    return;
  }
  if (lane == 0) lv.fused[idx] = 1;
  // ---- drop the later duplicates, place the survivors at their rank among the survivors
  int dest = 64;  // lanes without a fragment and dropped duplicates park beyond the survivors
  const int F = nAll - (int)__popcll(dupMask);
  if (lane < nAll && !dup) {
    // rank counts every smaller-or-earlier element, duplicates included: subtract the dropped ones among them
    int droppedBefore = 0;
    for (unsigned long long m = dupMask; m != 0ull; m &= m - 1ull) {
      const int j = __builtin_ctzll(m);
      const uint32_t xj = (uint32_t)__builtin_amdgcn_readlane((int)x, j), yj = (uint32_t)__builtin_amdgcn_readlane((int)y, j);
      droppedBefore += (xj < x || (xj == x && (yj < y || (yj == y && j < lane)))) ? 1 : 0;
    }
    dest = rank - droppedBefore;
  }
  // ds_permute pushes a lane's value to lane dest; parked lanes must not collide with survivors
  const unsigned long long liveMask = __ballot(dest < 64);
  if (dest >= 64) dest = F + (int)__popcll(~liveMask & ((1ull << lane) - 1ull));
  const int fx = __builtin_amdgcn_ds_permute(dest * 4, (int)x), fy = __builtin_amdgcn_ds_permute(dest * 4, (int)y);
  const int fl = __builtin_amdgcn_ds_permute(dest * 4, (int)len);
  const bool on = lane < F;
  if (on) {
    Frag f;
    f.x = (uint32_t)fx;
    f.y = (uint32_t)fy;
    f.len = (uint32_t)fl;
    f.pad = 0;
    lv.frags[base + lane] = f;
  }
  if (lane == 0) lv.nFrag[idx] = F;
  const SdpTask t = lv.tasks[idx];
  // ---- StoreAbove (SparseDynamicProgramming.h:55-68): the neighbour in (y,x) order
  int r2 = 0;
  for (int j = 0; j < F; j++) {
    const uint32_t xj = (uint32_t)__builtin_amdgcn_readlane(fx, j), yj = (uint32_t)__builtin_amdgcn_readlane(fy, j);
    r2 += (yj < (uint32_t)fy || (yj == (uint32_t)fy && xj < (uint32_t)fx)) ? 1 : 0;
  }
  if (!on) r2 = lane;  // keep the permutation a bijection
  const int byYx = __builtin_amdgcn_ds_permute(r2 * 4, lane);  // lane r: the fragment with (y,x) rank r
  int above = -1;
  {
    const int a = __builtin_amdgcn_ds_bpermute(max(r2 - 1, 0) * 4, byYx);
    const uint32_t ax = (uint32_t)__builtin_amdgcn_ds_bpermute(a * 4, fx), ay = (uint32_t)__builtin_amdgcn_ds_bpermute(a * 4, fy);
    const uint32_t al = (uint32_t)__builtin_amdgcn_ds_bpermute(a * 4, fl);
    if (on && r2 > 0 && ax <= (uint32_t)fx && ax + al > (uint32_t)fx && ay < (uint32_t)fy) above = a;
  }
  // ---- sweepSet.Predecessor: the largest (diag, x) with diag <= f.diag among x in [f.x-k-1, f.x-1]
  int pred = -1;
  {
    const int k = t.k;
    const int fdiag = fy - fx;
    const uint32_t xLow = (uint32_t)fx > (uint32_t)(k + 1) ? (uint32_t)fx - (uint32_t)(k + 1) : 0u;
    int bestDiag = 0;
    uint32_t bestX = 0;
    for (int j = 0; j < F; j++) {
      const uint32_t xj = (uint32_t)__builtin_amdgcn_readlane(fx, j);
      const int d = __builtin_amdgcn_readlane(fy, j) - (int)xj;
      if (j < lane && xj >= xLow && xj < (uint32_t)fx && d <= fdiag &&
          (pred < 0 || d > bestDiag || (d == bestDiag && xj > bestX))) {
        pred = j;
        bestDiag = d;
        bestX = xj;
      }
    }
    if (!on) pred = -1;
  }
  chain_sweep<true>(lv, prm, idx, t, F, fx, fy, fl, pred, above);
}

// ---------------------------------------------------------------------------
// sdp_blocks: chain -> blocks (SDPAlign.h:308-335), trim (:349-357), filter (:372-413), and the
// work lists of :423-536 (front fill for Global, fills between consecutive blocks)
// ---------------------------------------------------------------------------
// One wavefront per task.  Only the chain back-trace is serial (a linked list); run merging, trimming,
// filtering, compaction and the emission of blocks / gap-fill jobs are lane-parallel with wave prefix
// sums, and every 64-block chunk reserves its output slots with ONE atomic per work list.
__global__ __launch_bounds__(256) void sdp_blocks_build_kernel(SdpLevelArrays lv, SdpParams prm, int32_t *counts) {
  const int idx = wave_task_index();
  if (idx >= lv.n) return;
  const int lane = lane_id();
  const int F = lv.nFrag[idx];
  if (lane == 0) counts[4 * idx] = counts[4 * idx + 1] = counts[4 * idx + 2] = counts[4 * idx + 3] = 0;
  if (F == 0) return;
  const SdpTask t = lv.tasks[idx];
  const int64_t base = lv.fragOff[idx];
  const Frag *fr = lv.frags + base;
  const int *chainPrev = lv.chainPrev + base;
  int *chainRev = lv.scratchA + base;                                     // chain, last fragment first
  uint32_t *cb = reinterpret_cast<uint32_t *>(lv.scratchB + 3 * base);   // candidate blocks (q,t,end)
  uint32_t *fb = reinterpret_cast<uint32_t *>(lv.cost + base);           // final blocks reuse cost/chainLen/predCl
  uint32_t *fbT = reinterpret_cast<uint32_t *>(lv.chainLen + base);
  uint32_t *fbL = reinterpret_cast<uint32_t *>(lv.predCl + base);
  if (F <= 64) {
    // ---- the usual invocation of the recursion levels: all of it with lane = fragment, nothing parked in HBM
    // between the steps (the general path below makes five dependent round trips)
    __shared__ int stage[4][3][64];
    int(*sc)[64] = stage[threadIdx.x >> 6];
    const int link = lane < F ? chainPrev[lane] : -1;
    Frag f;
    f.x = f.y = f.len = 0;
    if (lane < F) f = fr[lane];
    int revSrc = 0, nChain = 0;  // lane h: the fragment h hops before the chain's end
    for (int cur = lv.chainEnd[idx]; cur >= 0 && nChain < F;) {
      if (lane == nChain) revSrc = cur;
      nChain++;
      cur = __builtin_amdgcn_readlane(link, cur);
    }
    if (nChain == 0) return;
    const bool on = lane < nChain;
    const int src = __builtin_amdgcn_ds_bpermute((nChain - 1 - lane) << 2, revSrc);  // chain order: lane i = i-th
    const int fx = __builtin_amdgcn_ds_bpermute(src << 2, (int)f.x), fy = __builtin_amdgcn_ds_bpermute(src << 2, (int)f.y);
    const int fl = __builtin_amdgcn_ds_bpermute(src << 2, (int)f.len);
    auto next_lane = [](int v) { return dpp<0x130, 0xf>(0, v); };  // wave_shl:1, lane i <- lane i+1
    // runs of +1/+1 fragments become candidate blocks (:308-335)
    const int px = wave_shr1(fx, 0), py = wave_shr1(fy, 0), nx = next_lane(fx), ny = next_lane(fy);
    const bool start = on && (lane == 0 || !(px + 1 == fx && py + 1 == fy));
    const bool last = on && (lane + 1 >= nChain || !(fx + 1 == nx && fy + 1 == ny));
    const unsigned incl = wave_prefix_sum(start ? 1u : 0u);
    const int ncb = __builtin_amdgcn_readlane((int)incl, 63);
    if (start) {
      sc[0][incl - 1] = fx;
      sc[1][incl - 1] = fy;
    }
    if (last) sc[2][incl - 1] = fx + fl;  // block.length = x_last + len_last - x_first (:332)
    __builtin_amdgcn_wave_barrier();
    uint32_t q = 0, tt = 0, len = 0;
    if (lane < ncb) {
      q = (uint32_t)sc[0][lane];
      tt = (uint32_t)sc[1][lane];
      len = (uint32_t)sc[2][lane] - q;
    }
    __builtin_amdgcn_wave_barrier();
    // trim overlaps (:349-357), drop zero-length and off-diagonal blocks (:372-413), compact
    bool good = false;
    {
      const uint32_t nq = (uint32_t)next_lane((int)q), nt = (uint32_t)next_lane((int)tt);
      const int pq = wave_shr1((int)q, 0), pt = wave_shr1((int)tt, 0);
      if (lane < ncb) {
        if (lane + 1 < ncb) {
          if (q + len > nq) len = nq - q;
          if (tt + len > nt) len = nt - tt;
          if (lane > 0) {
            const int prevDiag = abs(((int)tt - (int)q) - (pt - pq));
            const int prevDist = (int)min(tt - (uint32_t)pt, q - (uint32_t)pq);
            const int nextDiag = abs(((int)nt - (int)nq) - ((int)tt - (int)q));
            const int nextDist = (int)min(nt - tt, nq - q);
            good = !(prevDist * prm.indelRate < prevDiag && nextDist * prm.indelRate < nextDiag);
          } else
            good = true;
        } else
          good = true;
        good = good && len != 0;
      }
    }
    const unsigned incl2 = wave_prefix_sum(good ? 1u : 0u);
    const int nfb = __builtin_amdgcn_readlane((int)incl2, 63);
    if (nfb == 0) return;
    if (good) {
      const int w = (int)incl2 - 1;
      fb[w] = q;
      fbT[w] = tt;
      fbL[w] = len;
      sc[0][w] = (int)q;
      sc[1][w] = (int)tt;
      sc[2][w] = (int)len;
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t bq = 0, bt = 0, bl = 0;
    if (lane < nfb) {
      bq = (uint32_t)sc[0][lane];
      bt = (uint32_t)sc[1][lane];
      bl = (uint32_t)sc[2][lane];
    }
    // count what this task will emit: blocks, SWAlign fills, recursive SDPAlign jobs (:423-536)
    const uint32_t nru = (uint32_t)prm.noRecurseUnder;
    int nSw = 0, nChild = 0;
    const uint32_t q00 = (uint32_t)__builtin_amdgcn_readlane((int)bq, 0), t00 = (uint32_t)__builtin_amdgcn_readlane((int)bt, 0);
    const bool ends = t.type == 1 || (t.flags & kSdpExtendFront);  // :437, :546
    if (ends && q00 > 0 && t00 > 0) {  // front fill
      if (t.recurse == 0 && q00 * t00 < nru) nSw++;
      else if (t.recurse != 0) nChild++;
    }
    {
      const uint32_t nq = (uint32_t)next_lane((int)bq), nt = (uint32_t)next_lane((int)bt);
      int kind = 0;
      if (lane + 1 < nfb && (t.flags & kSdpDetailed)) {
        const uint32_t q0 = bq + bl, t0 = bt + bl;
        const uint32_t fq = nq - q0, ft = nt - t0;
        if (fq > 0 && ft > 0) kind = (fq * ft < nru) ? 1 : (t.recurse != 0 ? 2 : 0);
      }
      nSw += __popcll(__ballot(kind == 1));
      nChild += __popcll(__ballot(kind == 2));
    }
    if (t.flags & kSdpExtendFront) {  // tail fill (:546-598: only under extendFrontByLocalAlignment)
      const uint32_t qe = (uint32_t)__builtin_amdgcn_readlane((int)(bq + bl), nfb - 1);
      const uint32_t te = (uint32_t)__builtin_amdgcn_readlane((int)(bt + bl), nfb - 1);
      const uint32_t fq = (uint32_t)t.qLen - qe, ft = (uint32_t)t.tLen - te;
      if (fq > 0 && ft > 0) {
        if (t.recurse == 0 && fq * ft < nru) nSw++;
        else if (t.recurse != 0) nChild++;
      }
    }
    if (lane == 0) {
      counts[4 * idx] = nfb;
      counts[4 * idx + 1] = nSw;
      counts[4 * idx + 2] = nChild;
    }
    return;
  }
  // ---- back-trace: a linked list, but chainPrev[i] < i and usually close to i, so the wavefront loads a
  // window of 64 links ending at the current fragment and hops inside it with v_readlane; a new HBM
  // round trip is needed only when a hop leaves the window ----
  int nChain = 0;
  for (int cur = lv.chainEnd[idx]; cur != -1 && nChain < F;) {
    const int wBase = cur > 63 ? cur - 63 : 0;
    const int link = (wBase + lane <= cur) ? chainPrev[wBase + lane] : -1;
    while (cur >= wBase && nChain < F) {
      if (lane == 0) chainRev[nChain] = cur;
      nChain++;
      cur = __builtin_amdgcn_readlane(link, cur - wBase);
      if (cur < 0) break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (nChain == 0) return;
  // ---- runs of +1/+1 fragments become candidate blocks (:308-335) ----
  int ncb = 0;
  for (int i0 = 0; i0 < nChain; i0 += 64) {
    const int i = i0 + lane;
    bool start = false, last = false;
    Frag f;
    f.x = f.y = f.len = 0;
    if (i < nChain) {
      f = fr[chainRev[nChain - 1 - i]];
      start = true;
      if (i > 0) {
        const Frag p = fr[chainRev[nChain - i]];
        start = !(p.x + 1 == f.x && p.y + 1 == f.y);
      }
      last = true;
      if (i + 1 < nChain) {
        const Frag nx = fr[chainRev[nChain - 2 - i]];
        last = !(f.x + 1 == nx.x && f.y + 1 == nx.y);
      }
    }
    const unsigned incl = wave_prefix_sum(start ? 1u : 0u);
    const int b = ncb + (int)incl - 1;  // block this fragment belongs to
    if (start) {
      cb[3 * b] = f.x;
      cb[3 * b + 1] = f.y;
    }
    if (last) cb[3 * b + 2] = f.x + f.len;  // block.length = x_last + len_last - x_first (:332)
    ncb += __builtin_amdgcn_readlane((int)incl, 63);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  // ---- trim overlaps (:349-357), drop zero-length and off-diagonal blocks (:372-413), compact ----
  int nfb = 0;
  for (int b0 = 0; b0 < ncb; b0 += 64) {
    const int b = b0 + lane;
    bool good = false;
    uint32_t q = 0, tt = 0, len = 0;
    if (b < ncb) {
      q = cb[3 * b];
      tt = cb[3 * b + 1];
      len = cb[3 * b + 2] - q;
      if (b + 1 < ncb) {
        const uint32_t nq = cb[3 * b + 3], nt = cb[3 * b + 4];
        if (q + len > nq) len = nq - q;
        if (tt + len > nt) len = nt - tt;
        if (b > 0) {
          const int pq = (int)cb[3 * b - 3], pt = (int)cb[3 * b - 2];
          const int prevDiag = abs(((int)tt - (int)q) - (pt - pq));
          const int prevDist = (int)min(tt - (uint32_t)pt, q - (uint32_t)pq);
          const int nextDiag = abs(((int)nt - (int)nq) - ((int)tt - (int)q));
          const int nextDist = (int)min(nt - tt, nq - q);
          good = !(prevDist * prm.indelRate < prevDiag && nextDist * prm.indelRate < nextDiag);
        } else
          good = true;
      } else
        good = true;
      good = good && len != 0;
    }
    const unsigned incl = wave_prefix_sum(good ? 1u : 0u);
    if (good) {
      const int w = nfb + (int)incl - 1;
      fb[w] = q;
      fbT[w] = tt;
      fbL[w] = len;
    }
    nfb += __builtin_amdgcn_readlane((int)incl, 63);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (nfb == 0) return;
  // ---- count what this task will emit: blocks, SWAlign fills, recursive SDPAlign jobs (:423-536) ----
  const uint32_t nru = (uint32_t)prm.noRecurseUnder;
  int nSw = 0, nChild = 0;
  const bool ends = t.type == 1 || (t.flags & kSdpExtendFront);  // :437, :546
  if (ends && fb[0] > 0 && fbT[0] > 0) {  // front fill
    const uint32_t fq = fb[0], ft = fbT[0];
    if (t.recurse == 0 && fq * ft < nru) nSw++;
    else if (t.recurse != 0) nChild++;
  }
  if (t.flags & kSdpExtendFront) {  // tail fill (:546-598: only under extendFrontByLocalAlignment)
    const uint32_t fq = (uint32_t)t.qLen - (fb[nfb - 1] + fbL[nfb - 1]), ft = (uint32_t)t.tLen - (fbT[nfb - 1] + fbL[nfb - 1]);
    if (fq > 0 && ft > 0) {
      if (t.recurse == 0 && fq * ft < nru) nSw++;
      else if (t.recurse != 0) nChild++;
    }
  }
  for (int b0 = 0; b0 + 1 < nfb; b0 += 64) {
    const int b = b0 + lane;
    int kind = 0;
    if (b + 1 < nfb && (t.flags & kSdpDetailed)) {
      const uint32_t q0 = fb[b] + fbL[b], t0 = fbT[b] + fbL[b];
      const uint32_t fq = fb[b + 1] - q0, ft = fbT[b + 1] - t0;
      if (fq > 0 && ft > 0) kind = (fq * ft < nru) ? 1 : (t.recurse != 0 ? 2 : 0);
    }
    nSw += __popcll(__ballot(kind == 1));
    nChild += __popcll(__ballot(kind == 2));
  }
  if (lane == 0) {
    counts[4 * idx] = nfb;
    counts[4 * idx + 1] = nSw;
    counts[4 * idx + 2] = nChild;
  }
}

// Second half: every task knows where its blocks / fills / recursive jobs go (exclusive scans of the
// counts), so nothing contends on a global counter.
__global__ __launch_bounds__(256) void sdp_blocks_emit_kernel(SdpLevelArrays lv, SdpParams prm, SdpEmit e,
                                                              const int32_t *counts, const int64_t *offPool,
                                                              const int64_t *offSw, const int64_t *offChild) {
  const int idx = wave_task_index();
  if (idx >= lv.n) return;
  const int lane = lane_id();
  const int nfb = counts[4 * idx];
  if (nfb == 0) return;
  const SdpTask t = lv.tasks[idx];
  const int64_t base = lv.fragOff[idx];
  const uint32_t *fb = reinterpret_cast<const uint32_t *>(lv.cost + base);
  const uint32_t *fbT = reinterpret_cast<const uint32_t *>(lv.chainLen + base);
  const uint32_t *fbL = reinterpret_cast<const uint32_t *>(lv.predCl + base);
  int64_t atPool = offPool[idx], atSw = offSw[idx], atChild = offChild[idx];
  const uint32_t nru = (uint32_t)prm.noRecurseUnder;
  auto putSw = [&](int64_t at, uint32_t q0, uint32_t t0, uint32_t fq, uint32_t ft) {
    SwTask sw;
    sw.qOff = t.qOff + q0;
    sw.tOff = t.tOff + t0;
    sw.qLen = (int)fq;
    sw.tLen = (int)ft;
    sw.top = t.top;
    sw.qOrg = t.qOrg + (int)q0;
    sw.tOrg = t.tOrg + (int)t0;
    sw.pad = 0;
    e.sw[at] = sw;
  };
  auto putChild = [&](int64_t at, uint32_t q0, uint32_t t0, uint32_t fq, uint32_t ft, int k, int prefix, int maxPerPos,
                      int flags) {
    SdpTask c;
    c.qOff = t.qOff + q0;
    c.tOff = t.tOff + t0;
    c.qLen = (int)fq;
    c.tLen = (int)ft;
    c.k = k;
    c.type = 1;  // Global
    c.prefix = prefix;
    c.recurse = t.recurse - 1;
    c.maxPerPos = maxPerPos;
    c.top = t.top;
    c.qOrg = t.qOrg + (int)q0;
    c.tOrg = t.tOrg + (int)t0;
    c.flags = flags;
    c.pad = 0;
    e.children[at] = c;
  };
  const int kMinus4 = t.k - 4 > 5 ? t.k - 4 : 5;
  const bool ends = t.type == 1 || (t.flags & kSdpExtendFront);
  // :423-474 front fill
  if (ends && fb[0] > 0 && fbT[0] > 0) {
    const uint32_t fq = fb[0], ft = fbT[0];
    if (t.recurse == 0 && fq * ft < nru) {
      if (lane == 0) putSw(atSw, 0, 0, fq, ft);
      atSw++;
    } else if (t.recurse != 0) {
      // smithWatermanAlignType lands in the maxMatchesPerPosition slot of this call (:456) -- EndAnchored = 6 unless the
      // caller said otherwise (AlignSubstring, Blasr.cpp:1089: Global = 1) -- and the child's own is the default again
      if (lane == 0) putChild(atChild, 0, 0, fq, ft, kMinus4, t.prefix, sdp_sw_type(t.flags), t.flags & 0xff);
      atChild++;
    }
  }
  // :546-598 tail fill, only under extendFrontByLocalAlignment: word size k / 2 (sic), the call's own maxMatchesPerPosition
  if (t.flags & kSdpExtendFront) {
    const uint32_t q0 = fb[nfb - 1] + fbL[nfb - 1], t0 = fbT[nfb - 1] + fbL[nfb - 1];
    const uint32_t fq = (uint32_t)t.qLen - q0, ft = (uint32_t)t.tLen - t0;
    if (fq > 0 && ft > 0) {
      if (t.recurse == 0 && fq * ft < nru) {
        if (lane == 0) putSw(atSw, q0, t0, fq, ft);
        atSw++;
      } else if (t.recurse != 0) {
        if (lane == 0) putChild(atChild, q0, t0, fq, ft, t.k / 2 > 5 ? t.k / 2 : 5, t.prefix, t.maxPerPos, t.flags);
        atChild++;
      }
    }
  }
  // :481-536 blocks and the fills between consecutive blocks
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int b0 = 0; b0 < nfb; b0 += 64) {
    const int b = b0 + lane;
    const bool on = b < nfb;
    uint32_t q = 0, tt = 0, len = 0, q0 = 0, t0 = 0, fq = 0, ft = 0;
    int kind = 0;  // 1 = SWAlign fill, 2 = recursive SDPAlign
    if (on) {
      q = fb[b];
      tt = fbT[b];
      len = fbL[b];
      if (b + 1 < nfb) {
        q0 = q + len;
        t0 = tt + len;
        fq = fb[b + 1] - q0;
        ft = fbT[b + 1] - t0;
        if (fq > 0 && ft > 0 && (t.flags & kSdpDetailed)) kind = (fq * ft < nru) ? 1 : (t.recurse != 0 ? 2 : 0);
      }
      e.poolKeys[atPool + b] = ((uint64_t)(uint32_t)t.top << 32) | (uint32_t)(t.qOrg + (int)q);
      e.poolVals[atPool + b] = ((uint64_t)(uint32_t)(t.tOrg + (int)tt) << 32) | len;
    }
    const unsigned long long mSw = __ballot(kind == 1), mCh = __ballot(kind == 2);
    if (kind == 1) putSw(atSw + __popcll(mSw & lt), q0, t0, fq, ft);
    if (kind == 2) putChild(atChild + __popcll(mCh & lt), q0, t0, fq, ft, kMinus4, 0, 0, t.flags & kSdpDetailed);
    atSw += __popcll(mSw);
    atChild += __popcll(mCh);
  }
}

// ---------------------------------------------------------------------------
// sw: SWAlign, Global (SWAlign.h:19-389); only the Diagonal runs are kept (SDPAlign.h:529-534)
// ---------------------------------------------------------------------------
// Most fills are a few bases on either side (between consecutive k-mer anchors of the deepest recursion
// level).  Those with both sides <= 63 are sorted by (target length, query length), so that the 64 fills
// of a wavefront have the same shape, and run one per lane with the score row in registers
// (sw_small_kernel).  The others take the general kernel further down.
constexpr int kSwSmallMax = 63;
constexpr uint32_t kSwBigKey = 1u << 12;
constexpr int kSwArrowWords = kSwSmallMax * 4 * 64;        // per wavefront: 63 rows x 4 words x 64 lanes
constexpr int kSwWaveScratchWords = kSwArrowWords + 64 * 64;  // + 64 parked runs per lane

__global__ __launch_bounds__(256) void sw_sort_keys_kernel(const SwTask *tasks, int64_t n, int smallOk,
                                                           uint32_t *keys, uint32_t *vals, int32_t *nBig) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool big = false;
  if (i < n) {
    const int q = tasks[i].qLen, t = tasks[i].tLen;
    big = !smallOk || q > kSwSmallMax || t > kSwSmallMax;
    keys[i] = big ? kSwBigKey : (uint32_t)((t << 6) | q);
    vals[i] = (uint32_t)i;
  }
  const unsigned long long mask = __ballot(big);
  if (mask != 0ull && lane_id() == 0) atomicAdd(nBig, (int)__popcll(mask));
}

__global__ __launch_bounds__(256) void sw_small_kernel(const SwTask *tasks, const uint32_t *order, int64_t nAll,
                                                       const int32_t *nBigPtr, const uint8_t *seq,
                                                       const int32_t *matrix, SdpParams prm, uint32_t *scratch,
                                                       SdpEmit e) {
  __shared__ int m4[25];               // 4 x score: the low two bits of a cell key hold the arrow
  __shared__ uint8_t qRow[4][64 * 64];  // [wave][row * 64 + lane] = byte offset of the query base's row in m4
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  if (threadIdx.x < 25) m4[threadIdx.x] = matrix[threadIdx.x] * 4;
  __syncthreads();
  const int64_t nSmall = nAll - *nBigPtr;
  const int64_t nChunks = (nSmall + 63) / 64;
  const int64_t waveId = (int64_t)blockIdx.x * 4 + wv, nWaves = (int64_t)gridDim.x * 4;
  uint32_t *arr = scratch + waveId * kSwWaveScratchWords + lane;
  uint32_t *park = arr + kSwArrowWords;
  const int ins4 = prm.ins * 4, del4 = prm.del * 4;
  for (int64_t chunk = waveId; chunk < nChunks; chunk += nWaves) {
    const int64_t pos = chunk * 64 + lane;
    const bool live = pos < nSmall;
    SwTask t;
    t.qOff = t.tOff = 0;
    t.qLen = t.tLen = t.top = t.qOrg = t.tOrg = t.pad = 0;
    if (live) t = tasks[order[pos]];
    const int qMax = wave_max(t.qLen), tMax = wave_max(t.tLen);
    const int wpr = (tMax + 15) >> 4;
    const uint8_t *q = seq + t.qOff, *tt = seq + t.tOff;
    const int qLast = max(t.qLen - 1, 0), tLast = max(t.tLen - 1, 0);
    // target bases, as byte offsets inside a row of m4, four per register
    uint32_t tp[16];
#pragma unroll
    for (int j = 0; j < 16; j++) tp[j] = 0;
#pragma unroll
    for (int g = 0; g < 16; g++) {
      if (g * 4 < tMax) {
#pragma unroll
        for (int cc = 0; cc < 4; cc++)
          tp[g] |= (uint32_t)(base_code(tt[min(g * 4 + cc, tLast)]) * 4) << (8 * cc);
      }
    }
    for (int r = 0; r < qMax; r++) qRow[wv][r * 64 + lane] = (uint8_t)(base_code(q[min(r, qLast)]) * 20);
    // the lanes of a wavefront all compute qMax x tMax cells; what lies outside a lane's own q x t
    // never flows into it (dependencies point left and up only) and is ignored by its traceback
    int row[64];
#pragma unroll
    for (int c = 0; c < 64; c++) row[c] = del4 * (c + 1);
    for (int r = 0; r < qMax; r++) {
      const int qOff = qRow[wv][r * 64 + lane];
      int left = ins4 * (r + 1), diag = ins4 * r;
      uint32_t acc = 0;
      // columns in groups of four (one register of target offsets): the four score look-ups are issued
      // together, then the four dependent cells; a group past tMax is skipped by a scalar branch
#pragma unroll
      for (int g = 0; g < 16; g++) {
        if (g * 4 < tMax) {
          int sc[4];
#pragma unroll
          for (int cc = 0; cc < 4; cc++) {
            const int tOff = (int)((tp[g] >> (8 * cc)) & 0xffu);
            sc[cc] = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(m4) + qOff + tOff);
          }
#pragma unroll
          for (int cc = 0; cc < 4; cc++) {
            const int c = g * 4 + cc;
            const int up = row[c];
            // Diagonal (0) < Up (1) < Left (2) on equal scores, as the if-chain of SWAlign.h:196-207
            const int key = min(min(sc[cc] + diag, up + ins4 + 1), left + del4 + 2);
            diag = up;
            left = key & ~3;
            row[c] = left;
            acc |= (uint32_t)(key & 3) << (2 * (c & 15));
          }
          if ((g & 3) == 3) {
            arr[(r * wpr + (g >> 2)) * 64] = acc;
            acc = 0;
          }
        }
      }
      if (((tMax + 3) >> 2) & 3) arr[(r * wpr + (tMax >> 4)) * 64] = acc;
    }
    // traceback (q,t) -> (0,0); boundary arrows: row 0 Left, column 0 Up (:49-64)
    int nRuns = 0;
    if (live) {
      int r = t.qLen, c = t.tLen, runLen = 0, cached = -1;
      uint32_t w = 0;
      while (r > 0 || c > 0) {
        int a;
        if (r == 0) a = 2;
        else if (c == 0) a = 1;
        else {
          const int at = (r - 1) * wpr + ((c - 1) >> 4);
          if (at != cached) {
            w = arr[at * 64];
            cached = at;
          }
          a = (int)((w >> (2 * ((c - 1) & 15))) & 3u);
        }
        if (a == 0) {
          r--;
          c--;
          runLen++;
        } else {
          if (runLen) {
            park[nRuns * 64] = ((uint32_t)r << 16) | ((uint32_t)c << 8) | (uint32_t)runLen;
            nRuns++;
            runLen = 0;
          }
          if (a == 1) r--; else c--;
        }
      }
      if (runLen) {
        park[nRuns * 64] = (uint32_t)runLen;
        nRuns++;
      }
    }
    // the wavefront reserves pool space for all of its lanes with ONE atomic
    const unsigned incl = wave_prefix_sum((unsigned)nRuns);
    unsigned long long basePool = 0;
    if (lane == 63 && incl > 0) basePool = atomicAdd(e.poolCount, (unsigned long long)incl);
    basePool = __shfl(basePool, 63);
    const unsigned long long at0 = basePool + incl - (unsigned)nRuns;
    for (int j = 0; j < nRuns; j++) {
      const uint32_t pk = park[j * 64];
      if ((int64_t)(at0 + j) < e.poolCap) {
        e.poolKeys[at0 + j] = ((uint64_t)(uint32_t)t.top << 32) | (uint32_t)(t.qOrg + (int)(pk >> 16));
        e.poolVals[at0 + j] = ((uint64_t)(uint32_t)(t.tOrg + (int)((pk >> 8) & 0xffu)) << 32) | (pk & 0xffu);
      } else
        *e.overflow = 1;
    }
  }
}

// The general kernel: one lane per fill, rows and arrows in the lane's slice of a scratch buffer.
__global__ __launch_bounds__(256) void sw_kernel(const SwTask *tasks, const uint32_t *order, int64_t nAll,
                                                 const int32_t *nBigPtr, const uint8_t *seq,
                                                 const int32_t *matrix, SdpParams prm, uint8_t *scratch,
                                                 int64_t scratchPerLane, int64_t parkOff, SdpEmit e) {
  __shared__ int m[25];
  if (threadIdx.x < 25) m[threadIdx.x] = matrix[threadIdx.x];
  __syncthreads();
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  uint8_t *mine = scratch + gid * scratchPerLane;
  // the fills that do not fit sw_small_kernel: the last *nBigPtr entries of the sorted order
  const int64_t n = *nBigPtr;
  order += nAll - n;
  // all lanes of a wavefront iterate together (the pool reservation below is a wave operation)
  for (int64_t i0 = gid - (gid & 63); i0 < n; i0 += stride) {
    const int64_t i = i0 + (gid & 63);
    const bool live = i < n;
    SwTask t;
    t.qOff = t.tOff = 0;
    t.qLen = t.tLen = t.top = t.qOrg = t.tOrg = t.pad = 0;
    if (live) t = tasks[order[i]];
    const uint8_t *q = seq + t.qOff, *tt = seq + t.tOff;
    const int nCols = t.tLen + 1;
    int *prev = reinterpret_cast<int *>(mine);
    int *cur = prev + nCols;
    uint8_t *arrows = reinterpret_cast<uint8_t *>(cur + nCols);  // (qLen+1) x nCols, 0 Diagonal 1 Up 2 Left
    for (int c = 0; c < nCols; c++) prev[c] = prm.del * c;
    for (int r = 0; r < t.qLen; r++) {
      const int qc = base_code(q[r]);
      cur[0] = prm.ins * (r + 1);
      for (int c = 0; c < t.tLen; c++) {
        const int mt = m[qc * 5 + base_code(tt[c])] + prev[c];
        const int qGap = prev[c + 1] + prm.ins;
        const int tGap = cur[c] + prm.del;
        int mn = mt < qGap ? mt : qGap;
        if (tGap < mn) mn = tGap;
        cur[c + 1] = mn;
        arrows[(r + 1) * nCols + c + 1] = (mn == mt) ? 0 : (mn == qGap) ? 1 : 2;  // Diagonal, Up, Left (:196-207)
      }
      int *tmp = prev;
      prev = cur;
      cur = tmp;
    }
    // traceback (q,t) -> (0,0); boundary arrows: row 0 Left, column 0 Up (:49-64).  The Diagonal runs are
    // parked in the lane's scratch ...
    uint64_t *parkK = reinterpret_cast<uint64_t *>(mine + parkOff);  // <= min(q,t) runs, 16 bytes each
    int nRuns = 0;
    if (live) {
      int r = t.qLen, c = t.tLen, runLen = 0;
      while (r > 0 || c > 0) {
        const int a = r == 0 ? 2 : (c == 0 ? 1 : arrows[r * nCols + c]);
        if (a == 0) {
          r--;
          c--;
          runLen++;
        } else {
          if (runLen) {
            parkK[2 * nRuns] = ((uint64_t)(uint32_t)t.top << 32) | (uint32_t)(t.qOrg + r);
            parkK[2 * nRuns + 1] = ((uint64_t)(uint32_t)(t.tOrg + c) << 32) | (uint32_t)runLen;
            nRuns++;
            runLen = 0;
          }
          if (a == 1) r--; else c--;
        }
      }
      if (runLen) {
        parkK[2 * nRuns] = ((uint64_t)(uint32_t)t.top << 32) | (uint32_t)t.qOrg;
        parkK[2 * nRuns + 1] = ((uint64_t)(uint32_t)t.tOrg << 32) | (uint32_t)runLen;
        nRuns++;
      }
    }
    // ... and the wavefront reserves pool space for all of its lanes with ONE atomic
    const unsigned incl = wave_prefix_sum((unsigned)nRuns);
    unsigned long long basePool = 0;
    if (lane_id() == 63 && incl > 0) basePool = atomicAdd(e.poolCount, (unsigned long long)incl);
    basePool = __shfl(basePool, 63);
    const unsigned long long at0 = basePool + incl - (unsigned)nRuns;
    for (int j = 0; j < nRuns; j++) {
      if ((int64_t)(at0 + j) < e.poolCap) {
        e.poolKeys[at0 + j] = parkK[2 * j];
        e.poolVals[at0 + j] = parkK[2 * j + 1];
      } else
        *e.overflow = 1;
    }
  }
}

// ---------------------------------------------------------------------------
// assembly + statistics of the top-level candidates
// ---------------------------------------------------------------------------
// off[c] = index of the first pool entry of candidate c (keys sorted, candidate in the high word), off[n] = nPool
__global__ __launch_bounds__(256) void pool_offsets_kernel(const uint64_t *poolKeys, int64_t nPool, int n, int64_t *off) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > nPool) return;
  const int64_t top = i < nPool ? (int64_t)(poolKeys[i] >> 32) : (int64_t)n;
  const int64_t prev = i > 0 ? (int64_t)(poolKeys[i - 1] >> 32) : -1;
  for (int64_t c = prev + 1; c <= top; c++) off[c] = i;
}

// sizes of the variable-length per-task arrays, as int64, for the device offset scans
__global__ __launch_bounds__(256) void sdp_gather_sizes_kernel(SdpLevelArrays lv, int what, int64_t *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= lv.n) return;
  int64_t v;
  if (what == 0) v = task_needs_keys(lv.tasks[i].tLen, lv.tasks[i].k, lv.idxMaxT) ? lv.tasks[i].tLen : 0;
  else if (what == 3) v = lv.tasks[i].qLen;
  else if (what == 1) v = (int64_t)lv.nRaw[3 * i] + lv.nRaw[3 * i + 1] + lv.nRaw[3 * i + 2];
  else if (what == 4) v = task_may_scatter(lv.tasks[i], lv.nRaw[3 * i] + lv.nRaw[3 * i + 1] + lv.nRaw[3 * i + 2]) ? lv.tasks[i].tLen : 0;
  else v = lv.nFrag[i];
  out[i] = v;
}
__global__ __launch_bounds__(256) void sdp_segment_ends_kernel(SdpLevelArrays lv, int64_t *ends) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < lv.n) ends[i] = lv.fragOff[i] + lv.nFrag[i];
}

// Per top-level candidate (one wavefront): SDPAlign.h:602-624 (Local re-basing, return score) followed by
// ComputeAlignmentStats on an alignment WITHOUT gap lists (AlignmentUtils.h:408-446, :494-529): the gap
// between two blocks is spelled as the query surplus, then the target surplus, then the common part as
// aligned columns.
__global__ __launch_bounds__(256) void sdp_finish_kernel(int n, const SdpTask *tasks, const uint8_t *seq,
                                                         const uint64_t *poolKeys, const uint64_t *poolVals,
                                                         const int64_t *blkOff,
                                                         const int32_t *matrix, int ins, int del, int open, int ext,
                                                         SdpResult *results, int32_t *blocks, int64_t blocksCap,
                                                         int64_t blkBase) {
  __shared__ int m[25];
  __shared__ uint8_t codeOf[256];
  if (threadIdx.x < 25) m[threadIdx.x] = matrix[threadIdx.x];
  codeOf[threadIdx.x] = (uint8_t)base_code((unsigned)threadIdx.x);
  __syncthreads();
  const int idx = wave_task_index();
  if (idx >= n) return;
  const int lane = lane_id();
  const SdpTask t = tasks[idx];
  const uint8_t *qSeq = seq + t.qOff, *tSeq = seq + t.tOff;
  const int64_t b0 = blkOff[idx];
  const int nb = (int)(blkOff[idx + 1] - b0);
  SdpResult res;
  res.status = 0;
  res.score = res.q_pos = res.t_pos = res.stats_score = res.n_match = res.n_mismatch = res.n_ins = res.n_del = 0;
  res.n_blocks = nb;
  res.blk_off = b0 + blkBase;  // blkBase: the blocks of the chunks before this one (the arena pointer is already past them)
  if (nb == 0) {
    if (lane == 0) results[idx] = res;
    return;
  }
  int qPos = 0, tPos = 0;
  if (t.type == 0) {
    qPos = (int)(uint32_t)poolKeys[b0];
    tPos = (int)(poolVals[b0] >> 32);
  }
  const bool fits = b0 + nb <= blocksCap;
  int blockScore = 0, nMatch = 0, nMismatch = 0, nIns = 0, nDel = 0, gapScore = 0, commonScore = 0;
  auto columns = [&](int q, int tt, int len, int &scoreAcc) {
    score_columns(qSeq + q, tSeq + tt, len, m, codeOf, scoreAcc, nMatch, nMismatch);
  };
  auto gapRun = [&](int len, bool deletion) {
    if (len <= 0) return;
    const int linear = (deletion ? del : ins) * len, affine = len * ext + open;
    gapScore += linear < affine ? linear : affine;
    if (deletion) nDel += len; else nIns += len;
  };
  for (int b = lane; b < nb; b += 64) {
    const uint64_t pk = poolKeys[b0 + b], pv = poolVals[b0 + b];
    const int q = (int)(uint32_t)pk, tt = (int)(pv >> 32), len = (int)(uint32_t)pv;
    if (fits) {
      blocks[3 * (b0 + b)] = q - qPos;
      blocks[3 * (b0 + b) + 1] = tt - tPos;
      blocks[3 * (b0 + b) + 2] = len;
    }
    columns(q, tt, len, blockScore);
    if (b + 1 < nb) {
      int qGap = (int)(uint32_t)poolKeys[b0 + b + 1] - (q + len), tGap = (int)(poolVals[b0 + b + 1] >> 32) - (tt + len);
      if (qGap > 0 || tGap > 0) {
        int common = qGap;
        if (qGap > tGap) common = tGap;
        qGap -= common;
        tGap -= common;
        gapRun(qGap, false);
        gapRun(tGap, true);
        columns(q + len + qGap, tt + len + tGap, common, commonScore);
      }
    }
    if (b == 0 && (q - qPos > 0 || tt - tPos > 0)) {
      // leading offset of a Global alignment (:408-446): common part aligned, then target-, then query surplus
      int qp = q - qPos, tp = tt - tPos;
      int common = qp;
      if (common > tp) common = tp;
      columns(qPos, tPos, common, commonScore);
      gapRun(tp - common, true);
      gapRun(qp - common, false);
    }
  }
  int vals[7] = {blockScore, commonScore, gapScore, nMatch, nMismatch, nIns, nDel};
  for (int v = 0; v < 7; v++)
    for (int off = 32; off > 0; off >>= 1) vals[v] += __shfl_xor(vals[v], off);
  if (lane == 0) {
    res.status = fits ? 0 : 8;
    res.score = vals[0];
    res.q_pos = qPos;
    res.t_pos = tPos;
    res.stats_score = vals[0] + vals[1] + vals[2];
    res.n_match = vals[3];
    res.n_mismatch = vals[4];
    res.n_ins = vals[5];
    res.n_del = vals[6];
    results[idx] = res;
  }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
hipEvent_t SdpSpans::take() {
  if (pool == nullptr) return nullptr;
  if (used == pool->size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    pool->push_back(e);
  }
  return (*pool)[used++];
}
void SdpSpans::begin(int kind, hipStream_t s, int concurrent) {
  cur = Span{kind, concurrent, take(), take()};
  if (cur.a != nullptr && cur.b != nullptr) (void)hipEventRecord(cur.a, s);
}
void SdpSpans::end(hipStream_t s) {
  if (cur.kind < 0 || cur.a == nullptr || cur.b == nullptr) return;
  (void)hipEventRecord(cur.b, s);
  spans.push_back(cur);
  cur.kind = -1;
}
void SdpSpans::collect(float *ms, int32_t *launches, int32_t *concurrent, int kinds) {
  for (const Span &sp : spans) {
    float t = 0;
    if (sp.kind >= kinds || hipEventElapsedTime(&t, sp.a, sp.b) != hipSuccess) {
      (void)hipGetLastError();
      continue;
    }
    if (getenv("BLASR_HIP_SDP_TRACE")) fprintf(stderr, "[sdp]   span %s: %.2f ms\n", blasr_hip_sdp_kernel_name(sp.kind), t);
    ms[sp.kind] += t;
    launches[sp.kind]++;
    concurrent[sp.kind] = std::max(concurrent[sp.kind], sp.concurrent);
  }
  spans.clear();
  used = 0;
}
namespace {
struct SpanScope {  // brackets the launches of one kind on one stream
  SdpSpans *sp;
  hipStream_t s;
  SpanScope(const SdpLevelArrays &lv, int kind, hipStream_t st, int concurrent = 1) : sp(lv.spans), s(st) {
    if (sp) sp->begin(kind, s, concurrent);
  }
  ~SpanScope() {
    if (sp) sp->end(s);
  }
};
}  // namespace

static dim3 wave_grid(int n) { return dim3((unsigned)((n + 3) / 4)); }
// (the general kernels' grid: a wavefront per invocation, or -- general_task's mode 2 -- a fixed one that strides over a list)
static dim3 general_grid(const SdpLevelArrays &lv) { return lv.generalMode == 2 ? dim3(2048) : wave_grid(lv.n); }

void launch_sdp_keys(const SdpLevelArrays &lv, const uint8_t *seq, hipStream_t st) {
  if (lv.n > 0) hipLaunchKernelGGL(sdp_keys_kernel, dim3((unsigned)std::min((lv.n + 3) / 4, 8192)), dim3(256), 0, st, lv, seq);
}
void launch_sdp_query_words(const SdpLevelArrays &lv, const uint8_t *seq, hipStream_t st) {
  if (lv.n > 0 && lv.qWords != nullptr)
    hipLaunchKernelGGL(sdp_query_words_kernel, dim3((unsigned)((lv.n + 255) / 256)), dim3(256), 0, st, lv, seq);
}
void launch_sdp_sort_keys(const SdpLevelArrays &lv, uint64_t *tmp, hipStream_t st) {
  if (lv.n > 0) hipLaunchKernelGGL(sdp_sort_keys_kernel, dim3((unsigned)std::min((lv.n + 3) / 4, 8192)), dim3(256), 0, st, lv, tmp);
}
void launch_sdp_sort_yx(const SdpLevelArrays &lv, uint64_t *tmpKeys, uint32_t *tmpIdx, hipStream_t st) {
  if (lv.n > 0) hipLaunchKernelGGL(sdp_sort_yx_kernel, general_grid(lv), dim3(256), 0, st, lv, tmpKeys, tmpIdx);
}
void launch_sdp_match(const SdpLevelArrays &lv, const uint8_t *seq, bool write, hipStream_t st, hipStream_t side) {
  if (lv.n <= 0) return;
  const int conc = side ? 3 : 1;  // (beside the index matcher's two streams, launch_sdp_match_all)
  if (write) {
    SpanScope sc(lv, BLASR_HIP_SDP_KERNEL_MATCH_SHORT, st, conc);
    hipLaunchKernelGGL((sdp_match_kernel<true, false>), wave_grid(lv.n), dim3(256), 0, st, lv, seq);
  } else {
    {
      SpanScope sc(lv, BLASR_HIP_SDP_KERNEL_MATCH_SHORT, st, conc);
      hipLaunchKernelGGL((sdp_match_kernel<false, false>), wave_grid(lv.n), dim3(256), 0, st, lv, seq);
    }
    SpanScope sc(lv, BLASR_HIP_SDP_KERNEL_MATCH_SHORT, side ? side : st, conc);
    hipLaunchKernelGGL((sdp_match_kernel<false, true>), dim3((unsigned)std::min((lv.n + 3) / 4, 4096)), dim3(256), 0,
                       side ? side : st, lv, seq);  // (a fixed grid over its list)
  }
}
// The matchers take disjoint sets of invocations -- tiny and short targets (a wavefront each, bound by the latency of its
// few loads with every wavefront slot taken), targets from 256 bases on (LDS bucket index) and the long ones (LDS k-mer
// index, three size classes) -- so they run side by side: aux / ev as in launch_sdp_chain, or nullptr.
void launch_sdp_match_all(const SdpLevelArrays &lv, const uint8_t *seq, bool write, hipStream_t st, hipStream_t *aux,
                          hipEvent_t *ev) {
  if (lv.n <= 0) return;
  const bool fork = aux != nullptr && ev != nullptr;
  if (!fork) {
    launch_sdp_match(lv, seq, write, st, nullptr);
    launch_sdp_match_index(lv, seq, write, st);
    return;
  }
  (void)hipEventRecord(ev[0], st);
  for (int i = 0; i < 3; i++) (void)hipStreamWaitEvent(aux[i], ev[0], 0);
  launch_sdp_match_index(lv, seq, write, aux[0], aux[2]);  // (the index's larger size classes on a stream of their own)
  launch_sdp_match(lv, seq, write, st, aux[1]);
  for (int i = 0; i < 3; i++) {
    (void)hipEventRecord(ev[1 + i], aux[i]);
    (void)hipStreamWaitEvent(st, ev[1 + i], 0);
  }
}
void launch_sdp_classify(const SdpLevelArrays &lv, hipStream_t st) {
  if (lv.n <= 0) return;
  (void)hipMemsetAsync(lv.bigCount, 0, sizeof(int32_t), st);
  (void)hipMemsetAsync(lv.midCount, 0, sizeof(int32_t), st);
  (void)hipMemsetAsync(lv.keyCount, 0, sizeof(int32_t), st);
  hipLaunchKernelGGL(sdp_classify_kernel, dim3((unsigned)((lv.n + 255) / 256)), dim3(256), 0, st, lv);
}
int sdp_match_index_max_target(int device) {
  // the largest target one workgroup can index: bounded by the LDS a workgroup may allocate
  int plain = 0, optin = 0;
  if (hipDeviceGetAttribute(&plain, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, device) != hipSuccess) optin = 0;
  (void)hipGetLastError();
  auto capFor = [](int bytes) {
    int tCap = 0;
    for (int c = 2048; c <= 61440; c += 1024)
      if (idx_lds_bytes(c) <= bytes) tCap = c;
    return tCap;
  };
  int tCap = capFor(std::min(plain, 65536));
  const int want = std::max(plain, optin);
  if (want > 65536) {
    const int big = capFor(want);
    const int bytes = (int)idx_lds_bytes(big);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&sdp_match_index_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(&sdp_match_index_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess)
      tCap = big;
    (void)hipGetLastError();
  }
  if (getenv("BLASR_HIP_SDP_TRACE"))
    fprintf(stderr, "[sdp] LDS per workgroup %d (opt-in %d) -> indexed targets up to %d bases\n", plain, optin, tCap);
  return tCap;
}
void launch_sdp_match_index(const SdpLevelArrays &lv, const uint8_t *seq, bool write, hipStream_t st, hipStream_t side) {
  if (lv.n <= 0 || lv.idxMaxT <= 0) return;
  // three size classes, so that the usual targets (<= kIdxSmallCap) get four workgroups per CU
  const int grid = std::min(lv.n, 1024);
  const int caps[3] = {std::min(kIdxSmallCap, lv.idxMaxT), std::min(kIdxMidCap, lv.idxMaxT), lv.idxMaxT};
  // (one bracket per stream that gets a launch: the copy kernel and the first size class on `st`, the larger classes -- where
  //  the level has any -- on `side`; `concurrent` = the streams this kernel's launches are spread over, + 1 for the matchers
  //  of the shorter targets that run beside them)
  const bool sideUsed = side != nullptr && caps[1] > caps[0];
  const int conc = side ? (sideUsed ? 3 : 2) : 1;
  for (int part = 0; part < (sideUsed ? 2 : 1); part++) {
    hipStream_t s = part == 0 ? st : side;
    SpanScope sc(lv, BLASR_HIP_SDP_KERNEL_MATCH_INDEX, s, conc);
    if (part == 0 && write) hipLaunchKernelGGL(sdp_match_index_copy_kernel, dim3(std::min(lv.n, 8192)), dim3(256), 0, st, lv, seq);
    for (int c = 0; c < 3; c++) {
      if (c > 0 && caps[c] <= caps[c - 1]) break;
      if (side && (c > 0) != (part == 1)) continue;
      const int minT = c == 0 ? 0 : caps[c - 1];
      const size_t bytes = (size_t)idx_lds_bytes(caps[c]);
      if (write)
        hipLaunchKernelGGL(sdp_match_index_kernel<true>, dim3(grid), dim3(256), bytes, s, lv, seq, caps[c], minT);
      else
        hipLaunchKernelGGL(sdp_match_index_kernel<false>, dim3(grid), dim3(256), bytes, s, lv, seq, caps[c], minT);
    }
  }
}
void launch_sdp_small4_keys(const SdpLevelArrays &lv, uint32_t *keys, int32_t *vals, hipStream_t st) {
  if (lv.n > 0) hipLaunchKernelGGL(sdp_small4_keys_kernel, dim3((unsigned)((lv.n + 255) / 256)), dim3(256), 0, st, lv, keys, vals);
}
void launch_sdp_small(const SdpLevelArrays &lv, const SdpParams &p, const int32_t *order, hipStream_t st) {
  if (lv.n <= 0) return;
  const bool small4 = order != nullptr;
  SpanScope sc(lv, BLASR_HIP_SDP_KERNEL_SMALL, st, lv.generalMode == 0 && lv.leftList != nullptr ? 2 : 1);
  // (sdp_small4_kernel with a fixed grid striding over the level's slots -- ten million invocations a level, 650 k workgroups as a
  //  grid of one each: match + order 133.9 -> 131.3 ms per configs[2] step with 3072 .. 16384 workgroups, 137 with 1024;
  //  BLASR_HIP_SMALL4_GRID=0: a workgroup per sixteen invocations)
  static const int s4Grid = getenv("BLASR_HIP_SMALL4_GRID") ? atoi(getenv("BLASR_HIP_SMALL4_GRID")) : 4096;
  if (small4)
    hipLaunchKernelGGL(sdp_small4_kernel, dim3((unsigned)(s4Grid > 0 ? std::min((lv.n + 15) / 16, s4Grid) : (lv.n + 15) / 16)), dim3(256), 0,
                       st, lv, p, order);
  if (small4)
    hipLaunchKernelGGL(sdp_small_kernel, dim3((unsigned)std::min((lv.n + 3) / 4, 8192)), dim3(256), 0, st, lv, p, kSmall4Max + 1, order);
  else
    hipLaunchKernelGGL(sdp_small_kernel, wave_grid(lv.n), dim3(256), 0, st, lv, p, 0, order);
}
void launch_sdp_order(const SdpLevelArrays &lv, hipStream_t st, bool tiesToo) {
  if (lv.n <= 0) return;
  (void)hipMemsetAsync(lv.tieCount, 0, sizeof(int32_t), st);
  if (lv.cleanList != nullptr) (void)hipMemsetAsync(lv.cleanCount, 0, sizeof(int32_t), st);
  hipLaunchKernelGGL(sdp_order_kernel, general_grid(lv), dim3(256), 0, st, lv);
  if (tiesToo) launch_sdp_order_ties(lv, st);
}
void launch_sdp_order_ties(const SdpLevelArrays &lv, hipStream_t st) {
  if (lv.n <= 0) return;
  hipLaunchKernelGGL(sdp_order_ties_kernel<kTieLdsSmall>, dim3((unsigned)std::min(lv.n, 4096)), dim3(64), 0, st, lv, 0);
  hipLaunchKernelGGL(sdp_order_ties_kernel<kTieLdsCap>, dim3((unsigned)std::min(lv.n, 1024)), dim3(64), 0, st, lv,
                     kTieLdsSmall + 1);
}
void launch_sdp_yx_keys(const SdpLevelArrays &lv, hipStream_t st) {
  if (lv.n > 0) hipLaunchKernelGGL(sdp_yx_keys_kernel, general_grid(lv), dim3(256), 0, st, lv);
}
void launch_sdp_above_predcl(const SdpLevelArrays &lv, hipStream_t st) {
  if (lv.n > 0) hipLaunchKernelGGL(sdp_above_predcl_kernel, general_grid(lv), dim3(256), 0, st, lv);
}
int sdp_chain_bits_max_target() { return kBitsMaxTarget; }
void launch_sdp_kernel_sums(const SdpLevelArrays &lv, hipStream_t st) {
  if (lv.n > 0 && lv.kernelSums != nullptr)
    hipLaunchKernelGGL(sdp_kernel_sums_kernel, dim3((unsigned)std::min((lv.n + 255) / 256, 1024)), dim3(256), 0, st, lv);
}
void launch_sdp_chain_list(const SdpLevelArrays &lv, uint32_t *keys, int32_t *vals, hipStream_t st) {
  if (lv.n <= 0) return;
  (void)hipMemsetAsync(lv.chainCount, 0, kChainCats * sizeof(int32_t), st);  // one count per category
  hipLaunchKernelGGL(sdp_chain_list_kernel, dim3((unsigned)((lv.n + 255) / 256)), dim3(256), 0, st, lv, keys, vals);
}
// counts: the five category counts as the host read them (nullptr: grids sized for the whole level).  At the recursion levels
// one invocation in a hundred is swept by these kernels, and a 64 KB workgroup per four invocations of the level that returns at
// once is not free: the dispatcher was what the four kernels waited for.
void launch_sdp_chain(const SdpLevelArrays &lv, const SdpParams &p, hipStream_t st, hipStream_t *aux, hipEvent_t *ev,
                      const int32_t *counts) {
  if (lv.n <= 0) return;
  auto grid_for = [&](int cat, int perGroup) {  // workgroups of four wavefronts, perGroup invocations per wavefront
    const int n = counts ? counts[cat] : lv.n;
    return dim3((unsigned)std::max((n + 4 * perGroup - 1) / (4 * perGroup), 1));
  };
  if (lv.chainList != nullptr) {  // the list is sorted by category and size (launch_sdp_chain_list + sort)
    if (lv.colEnt != nullptr) {  // the scattered ones (the device knows how many of each kind: the grids are upper bounds)
      // The categories are disjoint sets of invocations and the long ones are serial chains whose kernels end in a
      // tail of a few wavefronts: side by side the throughput-bound kernels (chain4, bits4) fill the device under it.
      const bool fork = aux != nullptr && ev != nullptr;
      hipStream_t s0 = st, s1 = st, s2 = st, s3 = st;
      if (fork) {
        (void)hipEventRecord(ev[0], st);
        for (int i = 0; i < 4; i++) (void)hipStreamWaitEvent(aux[i], ev[0], 0);
        s0 = aux[0], s1 = aux[1], s2 = aux[2], s3 = aux[3];
      }
      int conc = 0;  // how many of the sweep's kernels run side by side
      for (int c : {0, 2, 3, 4, 5, 6, 7}) conc += (!counts || counts[c] > 0) ? 1 : 0;
      if (!fork) conc = 1;
      if (!counts || counts[6] > 0) {  // (the level's longest invocations: first, on the main stream)
        SpanScope sc(lv, BLASR_HIP_SDP_KERNEL_CHAIN_ROWS, st, conc);
        hipLaunchKernelGGL(sdp_chain_rows_kernel, dim3((unsigned)std::max(counts ? counts[6] : lv.n, 1)), dim3(64), 0, st, lv, p);
      }
      // (the largest targets on the MAIN stream, where sdp_chain4_kernel has nothing at the levels that have them: a handful
      //  of invocations, but the longest of the level -- 137 k fragments, 258 ms at configs[4] -- and behind them on one
      //  stream the middle class waited its 250 ms out: 736 ms of sweep instead of ~490)
      if (!counts || counts[5] > 0) {
        SpanScope sc(lv, BLASR_HIP_SDP_KERNEL_CHAIN_BITS_LARGE, st, conc);
        hipLaunchKernelGGL(sdp_chain_bits_kernel<kBitsWords>, grid_for(5, 1), dim3(256), 0, st, lv, p);
      }
      if (!counts || counts[4] > 0) {
        SpanScope sc(lv, BLASR_HIP_SDP_KERNEL_CHAIN_BITS_MID, s0, conc);
        hipLaunchKernelGGL(sdp_chain_bits_kernel<kBitsMidTarget / 32>, grid_for(4, 1), dim3(256), 0, s0, lv, p);
      }
      if (!counts || counts[3] > 0) {
        SpanScope sc(lv, BLASR_HIP_SDP_KERNEL_CHAIN_BITS_SMALL, s1, conc);
        hipLaunchKernelGGL(sdp_chain_bits_kernel<kBitsSmallTarget / 32>, grid_for(3, 1), dim3(256), 0, s1, lv, p);
      }
      // (the two bits4 kernels on a stream each: one after the other -- as kernels bound by the same unit usually go on this device
      //  -- the level waited for both tails, 144 against 118 ms of sweep per configs[2] step)
      if (!counts || counts[7] > 0) {
        SpanScope sc(lv, BLASR_HIP_SDP_KERNEL_CHAIN_BITS4_WIDE, s3, conc);
        hipLaunchKernelGGL(sdp_chain_bits4_kernel<true>, grid_for(7, 4), dim3(256), 0, s3, lv, p);
      }
      if (!counts || counts[2] > 0) {
        SpanScope sc(lv, BLASR_HIP_SDP_KERNEL_CHAIN_BITS4, s2, conc);
        hipLaunchKernelGGL(sdp_chain_bits4_kernel<false>, grid_for(2, 4), dim3(256), 0, s2, lv, p);
      }
      if (lv.scatterMode == 1) hipLaunchKernelGGL(sdp_chain_wide_kernel, grid_for(1, 1), dim3(256), 0, s2, lv, p);
      if (!counts || counts[0] > 0) {
        SpanScope sc(lv, BLASR_HIP_SDP_KERNEL_CHAIN4, st, conc);
        hipLaunchKernelGGL(sdp_chain4_kernel, grid_for(0, 4), dim3(256), 0, st, lv, p);
      }
      if (fork)
        for (int i = 0; i < 4; i++) {
          (void)hipEventRecord(ev[1 + i], aux[i]);
          (void)hipStreamWaitEvent(st, ev[1 + i], 0);
        }
      return;
    }
    SpanScope sc(lv, BLASR_HIP_SDP_KERNEL_CHAIN4, st);
    hipLaunchKernelGGL(sdp_chain4_kernel, grid_for(0, 4), dim3(256), 0, st, lv, p);
  } else {  // one invocation per wavefront (BLASR_HIP_NO_CHAIN4)
    hipLaunchKernelGGL(sdp_chain_kernel, wave_grid(lv.n), dim3(256), 0, st, lv, p);
  }
}
void launch_sdp_blocks_build(const SdpLevelArrays &lv, const SdpParams &p, int32_t *counts, hipStream_t st) {
  if (lv.n > 0) hipLaunchKernelGGL(sdp_blocks_build_kernel, wave_grid(lv.n), dim3(256), 0, st, lv, p, counts);
}
void launch_sdp_blocks_emit(const SdpLevelArrays &lv, const SdpParams &p, const SdpEmit &e, const int32_t *counts,
                            const int64_t *offPool, const int64_t *offSw, const int64_t *offChild, hipStream_t st) {
  if (lv.n > 0)
    hipLaunchKernelGGL(sdp_blocks_emit_kernel, wave_grid(lv.n), dim3(256), 0, st, lv, p, e, counts, offPool, offSw,
                       offChild);
}
void launch_sw_sort_keys(const SwTask *tasks, int64_t n, bool smallOk, uint32_t *keys, uint32_t *vals, int32_t *nBig,
                         hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(sw_sort_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tasks, n,
                     smallOk ? 1 : 0, keys, vals, nBig);
}
int64_t sw_small_scratch_words(int blocks) { return (int64_t)blocks * 4 * kSwWaveScratchWords; }
int sw_small_blocks(int64_t n) { return (int)std::min<int64_t>((n + 255) / 256, 2048); }
void launch_sw_small(const SwTask *tasks, const uint32_t *order, int64_t n, const int32_t *nBig, const uint8_t *seq,
                     const int32_t *matrix, const SdpParams &p, uint32_t *scratch, const SdpEmit &e, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(sw_small_kernel, dim3(sw_small_blocks(n)), dim3(256), 0, st, tasks, order, n, nBig, seq, matrix, p,
                     scratch, e);
}
void launch_sw(const SwTask *tasks, const uint32_t *order, int64_t n, const int32_t *nBig, int blocks,
               const uint8_t *seq, const int32_t *matrix, const SdpParams &p, uint8_t *scratch, int64_t scratchPerLane,
               int64_t parkOff, const SdpEmit &e, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(sw_kernel, dim3(blocks), dim3(256), 0, st, tasks, order, n, nBig, seq, matrix, p, scratch,
                     scratchPerLane, parkOff, e);
}
void launch_pool_offsets(const uint64_t *poolKeys, int64_t nPool, int n, int64_t *off, hipStream_t st) {
  hipLaunchKernelGGL(pool_offsets_kernel, dim3((unsigned)((nPool + 1 + 255) / 256)), dim3(256), 0, st, poolKeys, nPool, n,
                     off);
}
void launch_sdp_gather_sizes(const SdpLevelArrays &lv, int what, int64_t *out, hipStream_t st) {
  if (lv.n > 0)
    hipLaunchKernelGGL(sdp_gather_sizes_kernel, dim3((unsigned)((lv.n + 255) / 256)), dim3(256), 0, st, lv, what, out);
}
void launch_sdp_segment_ends(const SdpLevelArrays &lv, int64_t *ends, hipStream_t st) {
  if (lv.n > 0)
    hipLaunchKernelGGL(sdp_segment_ends_kernel, dim3((unsigned)((lv.n + 255) / 256)), dim3(256), 0, st, lv, ends);
}
void launch_sdp_finish(int n, const SdpTask *tasks, const uint8_t *seq, const uint64_t *poolKeys,
                       const uint64_t *poolVals, const int64_t *blkOff,
                       const int32_t *matrix, int ins, int del, int open, int ext, SdpResult *results, int32_t *blocks,
                       int64_t blocksCap, int64_t blkBase, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(sdp_finish_kernel, wave_grid(n), dim3(256), 0, st, n, tasks, seq, poolKeys, poolVals, blkOff,
                     matrix, ins, del, open, ext, results, blocks, blocksCap, blkBase);
}

}  // namespace blasr_hip
