// intervals_device.hip -- next-1: from the sorted anchor lists of a block of reads to their candidate intervals, with
// the anchors staying in device memory (blasr_hip_find_intervals_device).
//
// What MapRead does between SortMatchPosList and AlignIntervals (alignment/Blasr.cpp:2342-2500) touches every anchor a
// few times -- RemoveOverlappingAnchors, the sliding window of StoreLargestIntervals -- and then works on the handful of
// clusters the window found: GlobalChain over a cluster's anchors, the chain's p-value, the read's interval set.  Here
// the per-anchor steps and the chains are kernels over all lists of the block (a "list" = one strand of one read, the
// anchors sorted by (t, q) as blasr_hip_anchor_reads_device leaves them); what crosses PCIe is one 32-byte record per
// CLUSTER (a few hundred thousand per 100 k reads, where the anchors are 14 GB), and the host finishes with the p-values
// (double-precision logarithms of its libm) and the interval sets (blasr_hip_intervals_from_clusters, intervals_api.hip).
//
//   iv_mark_kernel      RemoveOverlappingAnchors (algorithms/anchoring/FindMaxInterval.h:282-315): lane = anchor.  The
//                       reference walks a list backwards and zeroes the length of an anchor nested in one of the
//                       previous target position; whether anchor m goes depends only on anchors BEFORE it, which the
//                       backward walk has not touched yet -- so every anchor decides for itself.
//   scan + iv_compact   RemoveZeroLengthAnchors (:267-280): one device-wide exclusive scan of the keep flags (a list's
//                       survivors stay contiguous), the element the reference then finds one past the end of its list
//                       (:493, :581, :617: the list's own old entry at that index) kept per list.
//   iv_windows_kernel   StoreLargestIntervals (:394-640, the non-overlapping branch) with AdvanceIndexToPastInterval
//                       (:230-265): a sequential two-pointer sweep whose segmentation depends on its own best window so
//                       far -- lane = list, once to count the clusters of every list and once to write them.
//   iv_chain_kernel     GlobalChain (GlobalChain.h:101-207) for every cluster of at least two anchors, one wavefront per
//                       cluster, everything in LDS: the endpoints sorted by (x, y) -- by rank when all are distinct,
//                       else by the replay of libstdc++'s std::sort (introsort_replay.h), since the order of tied
//                       endpoints is that schedule's -- and then the sweep over the endpoints with the reference's
//                       priority search tree AS IT BEHAVES (PrioritySearchTree.h:53-253): built over the endpoints in
//                       (x, y) order but navigated by y, with an Activate that sends a displaced point down the NEW
//                       point's path.  The chain it finds is not always the heaviest one, and the intervals are its
//                       first and last anchor: so the structure is simulated step by step (lane 0; the tree is implicit
//                       in the index ranges, one int per node).
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "../../include/blasr_hip.h"
#include "ctx.h"
#include "introsort_replay.h"
#include "wave_utils.h"

namespace blasr_hip {
namespace {

struct IvLists {
  const blasr_hip_anchor *anchors;  // all lists back to back
  const int64_t *listOff;           // nLists + 1
  int nLists;
};

__device__ __forceinline__ int list_of(const int64_t *off, int nLists, int64_t i) {  // largest L with off[L] <= i
  int lo = 0, hi = nLists;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

// RemoveOverlappingAnchors for anchor i: keep[i] = 0 when the reference zeroes its length (or it had none)
__global__ __launch_bounds__(256) void iv_mark_kernel(IvLists L, int64_t nAnchors, uint32_t *keep) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nAnchors) return;
  // the list of the wavefront's first anchor by one search on the scalar unit; a lane beyond that list's end steps on
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + (int64_t)(__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) << 6);
  int li = list_of(L.listOff, L.nLists, i0);
  while (li + 1 < L.nLists && i >= L.listOff[li + 1]) li++;
  const blasr_hip_anchor *list = L.anchors + L.listOff[li];
  const int64_t m = i - L.listOff[li];
  const blasr_hip_anchor am = list[m];
  bool merged = false;
  if (m > 0) {
    int64_t n = m - 1;
    while (n > 0 && list[n].t == am.t) n--;
    const uint32_t tn = list[n].t;
    while (!merged && n > 0 && list[n].t == tn) {
      const blasr_hip_anchor an = list[n];
      if (an.q < am.q && an.t < am.t && an.l + an.q >= am.l + am.q && an.l + an.t >= am.l + am.t) merged = true;
      n--;
    }
  }
  keep[i] = (!merged && am.l > 0) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void iv_compact_kernel(IvLists L, int64_t nAnchors, const uint32_t *keep, const uint32_t *slot,
                                                         blasr_hip_anchor *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nAnchors) return;
  if (keep[i]) out[slot[i]] = L.anchors[i];
}

// per list: where its survivors start, how many there are, and the element one past them as the reference's vector
// holds it after the in-place compaction (the list's old entry at that index, its length zeroed if it was removed)
__global__ __launch_bounds__(256) void iv_lists_kernel(IvLists L, const uint32_t *keep, const uint32_t *slot, uint32_t total,
                                                       uint32_t *newOff, uint2 *past) {
  const int li = blockIdx.x * blockDim.x + threadIdx.x;
  if (li > L.nLists) return;
  const int64_t a = li < L.nLists ? L.listOff[li] : 0;
  const int64_t nAll = L.listOff[L.nLists];
  const uint32_t s0 = li == L.nLists ? total : (a < nAll ? slot[a] : total);
  newOff[li] = s0;
  if (li == L.nLists) return;
  const int64_t b = L.listOff[li + 1];
  const uint32_t s1 = b < nAll ? slot[b] : total;
  const int64_t n = b - a, nPos = (int64_t)(s1 - s0);
  uint2 p = make_uint2(0u, 0u);  // nothing removed: the reference reads past its vector's end -- zeros here (blasr_hip.h: stale)
  if (nPos < n) {
    const blasr_hip_anchor e = L.anchors[a + nPos];
    p = make_uint2(e.t, keep[a + nPos] ? e.l : 0u);
  }
  past[li] = p;
}

// (SequenceIndexDatabase::SearchForStartBoundary, SequenceIndexDatabase.h:108-144, is boundary_of in iv_windows_kernel)
struct IvCluster {
  int32_t list;
  uint32_t start, end;  // indices into the list's survivors
};

// StoreLargestIntervals for one list per lane.  kEmit = false: count[list] = its number of clusters, the first
// kFirstFew of them parked in firstFew; true: the clusters are written at clusterOff[list] -- from firstFew when that
// holds them all (nearly always: the sweep is not run again), by a second sweep otherwise.
constexpr int kFirstFew = 4;
template <bool kEmit>
__global__ __launch_bounds__(64) void iv_windows_kernel(const blasr_hip_anchor *pos0, const uint32_t *newOff, const uint2 *past,
                                                        int nLists, const uint32_t *seqStart, int nSeqPos,
                                                        const uint32_t *intervalLength, int minSize, uint32_t *count,
                                                        const uint32_t *clusterOff, IvCluster *clusters, IvCluster *firstFew) {
  const int li = blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= nLists) return;
  if (kEmit && count[li] <= (uint32_t)kFirstFew) {
    for (uint32_t i = 0; i < count[li]; i++) clusters[clusterOff[li] + i] = firstFew[(size_t)li * kFirstFew + i];
    return;
  }
  const blasr_hip_anchor *pos = pos0 + newOff[li];
  const uint32_t nPos = newOff[li + 1] - newOff[li];
  const uint2 pastEl = past[li];
  const uint32_t L = intervalLength[li >> 1];
  uint32_t nOut = 0;
  IvCluster *out = kEmit ? clusters + clusterOff[li] : nullptr;
  auto emit = [&](uint32_t s, uint32_t e) {
    IvCluster c;
    c.list = li;
    c.start = s;
    c.end = e;
    if (kEmit) out[nOut] = c;
    else if (nOut < (uint32_t)kFirstFew) firstFew[(size_t)li * kFirstFew + nOut] = c;
    nOut++;
  };
  auto T = [&](uint32_t i) { return i < nPos ? pos[i].t : pastEl.x; };   // (pos[nPos] is read by the reference)
  auto Len = [&](uint32_t i) { return i < nPos ? pos[i].l : pastEl.y; };
  // The contig a position lies in changes a handful of times along a list: the last answer is kept with the range of
  // positions it holds for (one for the window's start, one for its end), and the table is searched only outside it.
  struct Range {
    uint32_t lo, hi, val;  // start_boundary(p) == val for lo <= p <= hi
  };
  Range rc = {1u, 0u, 0u}, re = {1u, 0u, 0u};
  auto boundary_of = [&](Range &r, uint32_t p) {
    if (p >= r.lo && p <= r.hi) return r.val;
    if (nSeqPos == 1) {
      r.lo = 0u, r.hi = 0xffffffffu, r.val = seqStart[0];
      return r.val;
    }
    int lo = 1, hi = nSeqPos;  // first index in [1, nSeqPos) with seqStart > p
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (seqStart[mid] > p) hi = mid; else lo = mid + 1;
    }
    r.val = seqStart[lo - 1];
    r.lo = lo == 1 ? 0u : r.val;
    r.hi = lo < nSeqPos ? seqStart[lo] - 1u : 0xffffffffu;
    return r.val;
  };
  if (nPos == 1) {
    emit(0, 0);
  } else if (nPos > 1) {
    uint32_t cur = 0, endIndex = 1;
    uint32_t tCur = T(cur), tEnd = T(endIndex);  // T(cur), T(endIndex): kept in step with the two indices
    uint32_t curBoundary = boundary_of(rc, tCur);
    uint32_t endIndexBoundary = boundary_of(re, tEnd);  // (endIndex = 1 < nPos)
    auto advance = [&](uint32_t tStart, uint32_t startBoundary) {  // AdvanceIndexToPastInterval
      if (endIndex >= nPos) return;
      endIndexBoundary = boundary_of(re, tEnd);
      while (endIndex < nPos && tEnd - tStart < L && endIndexBoundary == startBoundary) {
        endIndex++;
        tEnd = T(endIndex);
        if (endIndex < nPos) endIndexBoundary = boundary_of(re, tEnd);
      }
    };
    auto sum = [&](uint32_t s, uint32_t e) {
      int v = 0;
      for (uint32_t i = s; i < e; i++) v += (int)pos[i].l;
      return v;
    };
    advance(tCur, curBoundary);
    uint32_t maxStart = cur, maxEndIndex = endIndex;
    uint32_t tMaxStart = tCur, tMaxEnd = tEnd;  // T(maxStart), T(maxEndIndex)
    int maxSize = sum(cur, endIndex);
    int curSize = maxSize;
    if (curSize > minSize) emit(cur, endIndex);
    while (cur < nPos) {
      if (tCur >= tMaxStart && maxEndIndex > 0 && tCur < tMaxEnd && curBoundary == endIndexBoundary) {
        if (curSize > maxSize) {
          maxSize = curSize;
          maxStart = cur, tMaxStart = tCur;
          maxEndIndex = endIndex, tMaxEnd = tEnd;
        }
      } else {
        if (maxSize > minSize) emit(maxStart, maxEndIndex);
        maxStart = cur, tMaxStart = tCur;
        maxEndIndex = endIndex, tMaxEnd = tEnd;
        maxSize = curSize;
      }
      if (curBoundary != endIndexBoundary) {
        cur = endIndex, tCur = tEnd;
        curBoundary = endIndexBoundary;
        if (endIndex < nPos) {
          endIndex++;
          tEnd = T(endIndex);
          advance(tCur, curBoundary);
          maxSize = 0;
          maxStart = cur, tMaxStart = tCur;
          maxEndIndex = endIndex, tMaxEnd = tEnd;
        }
        maxSize = curSize = sum(cur, endIndex);
      } else {
        while (cur < endIndex && tCur + L < tEnd) {
          curSize -= (int)Len(cur);
          cur++;
          tCur = T(cur);
        }
        if (cur >= nPos) break;
        curSize += (int)Len(endIndex);
        endIndex++;
        tEnd = T(endIndex);
      }
      if (endIndex > nPos) break;
      curBoundary = boundary_of(rc, tCur);
      endIndexBoundary = boundary_of(re, tEnd);
    }
    if (curSize > minSize) emit(maxStart, maxEndIndex);
  }
  if (!kEmit) count[li] = nOut;
}

// ---------------------------------------------------------------------------------------------------------------
// GlobalChain
// ---------------------------------------------------------------------------------------------------------------
struct Point {       // BasicEndpoint: p = (x, y) = (q, t) of an anchor's start or end
  uint32_t x, y;
  uint32_t fs;       // fragment * 2 + side (0 Start, 1 End)
};
struct PointLess {   // Coordinate::operator< (BasicEndpoint.h:28-31)
  __device__ __forceinline__ bool operator()(const Point &a, const Point &b) const {
    return a.x == b.x ? a.y < b.y : a.x < b.x;
  }
};

// The chain of one cluster (n >= 2 anchors at frag[0..n)), by one wavefront; P0 / P1: 2n points each, score / prev: n,
// tree: 4n ints.  Returns through lane 0's out.
__device__ void chain_cluster(const blasr_hip_anchor *frag, int n, Point *P0, Point *P1, int32_t *score, int32_t *prev,
                              int32_t *tree, blasr_hip_iv_cluster &out, const bool byRank, uint32_t *chainOut) {
  const int lane = lane_id();
  const int N = 2 * n;
  for (int f = lane; f < n; f += 64) {  // FragmentSetToEndpoints (GlobalChain.h:8-21), scores = lengths (:114-116)
    const blasr_hip_anchor a = frag[f];
    score[f] = (int32_t)a.l;
    prev[f] = -1;
    Point s, e;
    s.x = a.q, s.y = a.t, s.fs = (uint32_t)(2 * f);
    e.x = a.q + a.l, e.y = a.t + a.l, e.fs = (uint32_t)(2 * f + 1);
    P0[2 * f] = s;
    P0[2 * f + 1] = e;
  }
  for (int v = lane; v < 2 * N - 1; v += 64) tree[v] = -1;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // ---- std::sort of the endpoints (:137).  Distinct (x, y): the sorted order is unique, every point is placed by its
  // rank.  Ties: their order is what libstdc++'s introsort makes of this very sequence -- replayed.
  bool tie = !byRank;  // (a very large cluster: its quadratic rank pass would take longer than the replay)
  for (int i = lane; byRank && i < N; i += 64) {
    const Point p = P0[i];
    int rank = 0;
    for (int j = 0; j < N; j++) {
      const Point o = P0[j];
      const bool less = o.x == p.x ? o.y < p.y : o.x < p.x;
      rank += less ? 1 : 0;
      tie = tie || (j != i && o.x == p.x && o.y == p.y);
    }
    P1[rank] = p;  // (with ties some slots are written twice and some not at all: P1 is not used then)
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  Point *P = P1;
  if (__ballot(tie) != 0ull) {
    if (lane == 0) introsort_replay(P0, (long)N, PointLess());
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    P = P0;
  }
  if (lane != 0) return;
  // ---- the sweep (:142-173) over the priority search tree.  A node is (index v, point range [s, e)): a leaf when
  // e - s == 1, else its children are (v + 1, [s, med)) and (v + 2 (med - s), [med, e)) with med = (s + e) / 2 (the
  // preorder of CreateTree, PrioritySearchTree.h:53-110); its medianKey is the key -- the y -- of point med - 1;
  // tree[v] = the node's maxScoreNode.
  auto pscore = [&](int p) { return score[P[p].fs >> 1]; };
  int best = -1;
  for (int p = 0; p < N; p++) {
    const uint32_t fs = P[p].fs;
    const int f = (int)(fs >> 1);
    if ((fs & 1u) == 0u) {
      // FindIndexOfMaxPoint(root, key(p)) (:112-175): depth first, left before right, `>=` lets the later one win
      const uint32_t maxKey = P[p].y;
      int maxValue = 0, maxIndex = 0, found = 0;
      if (tree[0] != -1) {
        int stV[40], stS[40], stE[40], sp = 0;
        stV[0] = 0, stS[0] = 0, stE[0] = N, sp = 1;
        while (sp > 0) {
          --sp;
          const int v = stV[sp], s = stS[sp], e = stE[sp];
          const int msn = tree[v];
          if (msn == -1) continue;
          if (P[msn].y < maxKey) {
            const int sc = pscore(msn);
            if (sc >= maxValue) {
              maxValue = sc;
              maxIndex = msn;
              found = 1;
            }
            continue;
          }
          if (e - s == 1) continue;
          const int med = (s + e) / 2;
          if (!(maxKey <= P[med - 1].y)) {  // both children, the left one first: the right one waits on the stack
            stV[sp] = v + 2 * (med - s), stS[sp] = med, stE[sp] = e;
            sp++;
          }
          stV[sp] = v + 1, stS[sp] = s, stE[sp] = med;
          sp++;
        }
      }
      if (found) {
        prev[f] = (int32_t)(P[maxIndex].fs >> 1);
        score[f] = pscore(maxIndex) + score[f];
      } else {
        prev[f] = -1;
      }
    } else {
      // Activate (:225-253): the displaced point travels on along the path of the point that displaced it, compared
      // with THAT point's score
      const int pointScore = score[f];
      const uint32_t pointKey = P[p].y;
      int pi = p, v = 0, s = 0, e = N;
      while (pi != -1 && e - s != 1) {
        const int msn = tree[v];
        if (msn == -1 || pscore(msn) <= pointScore) {
          tree[v] = pi;
          pi = msn;
        }
        const int med = (s + e) / 2;
        if (pointKey <= P[med - 1].y) {
          v = v + 1;
          e = med;
        } else {
          v = v + 2 * (med - s);
          s = med;
        }
      }
      if (best < 0 || pscore(best) < score[f]) best = p;
    }
  }
  // ---- the chain, from its last anchor back (:190-204)
  int f = (int)(P[best].fs >> 1);
  const blasr_hip_anchor last = frag[f];
  int sumL = 0, count = 0, firstF = f;
  while (f != -1) {
    firstF = f;
    sumL += (int)frag[f].l;
    count++;
    f = prev[f];
  }
  const blasr_hip_anchor first = frag[firstF];
  if (chainOut) {  // the chain itself, first anchor first (WeightedInterval::matches; only -rbao asks for it)
    int pos = count - 1;
    for (int f2 = (int)(P[best].fs >> 1); f2 != -1; f2 = prev[f2], pos--)
      if (lane == 0) {
        const blasr_hip_anchor a = frag[f2];
        chainOut[3 * pos] = a.t, chainOut[3 * pos + 1] = a.q, chainOut[3 * pos + 2] = a.l;
      }
  }
  out.first_t = first.t, out.first_q = first.q, out.first_l = first.l;
  out.last_t = last.t, out.last_q = last.q, out.last_l = last.l;
  out.sum_l = sumL;
  out.count = count;
}

constexpr int kChainSmall = 128;   // anchors of a cluster: four wavefronts per workgroup, 9 KB of LDS each
constexpr int kChainMid = 1024;    // one wavefront per workgroup, 72 KB

// bytes of a cluster of up to `cap` anchors: P0, P1 (2 cap points each), score, prev (cap), tree (4 cap)
__host__ __device__ constexpr size_t chain_bytes(int cap) { return (size_t)cap * (2 * 2 * sizeof(Point) + 2 * 4 + 4 * 4); }

// cls: 0 clusters of 0 / 1 anchors and the small ones, 1 the mid-size ones, 2 the rest (scratch in global memory)
template <int kCls>
__global__ __launch_bounds__(256) void iv_chain_kernel(const blasr_hip_anchor *pos0, const uint32_t *newOff, const IvCluster *clusters,
                                                       const int32_t *order, const int32_t *classCount, char *globalScratch,
                                                       const int64_t *scratchOff, blasr_hip_iv_cluster *out, uint32_t *chains,
                                                       const int64_t *chainSlot) {
  extern __shared__ uint4 ivChainLds[];
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int wavesPerBlock = blockDim.x >> 6;
  const int slot = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * wavesPerBlock + wave));
  const int first = kCls == 0 ? 0 : (kCls == 1 ? classCount[0] : classCount[0] + classCount[1]);
  if (slot >= classCount[kCls]) return;
  const int ci = order[first + slot];
  const IvCluster c = clusters[ci];
  const blasr_hip_anchor *frag = pos0 + newOff[c.list] + c.start;
  const int n = (int)(c.end - c.start);
  blasr_hip_iv_cluster res;
  res.first_t = res.first_q = res.first_l = res.last_t = res.last_q = res.last_l = 0;
  res.sum_l = res.count = 0;
  uint32_t *chainOut = chains ? chains + 3 * chainSlot[ci] : nullptr;  // room for the cluster's n anchors
  if (n == 1) {  // (:717-719: the chain of a single anchor is that anchor)
    const blasr_hip_anchor a = frag[0];
    if (chainOut && lane == 0) chainOut[0] = a.t, chainOut[1] = a.q, chainOut[2] = a.l;
    res.first_t = res.last_t = a.t;
    res.first_q = res.last_q = a.q;
    res.first_l = res.last_l = a.l;
    res.sum_l = (int32_t)a.l;
    res.count = 1;
  } else if (n >= 2) {
    const int cap = kCls == 0 ? kChainSmall : (kCls == 1 ? kChainMid : n);
    char *mem = kCls == 2 ? globalScratch + scratchOff[slot] : reinterpret_cast<char *>(ivChainLds) + (size_t)wave * chain_bytes(kChainSmall);
    Point *P0 = reinterpret_cast<Point *>(mem);
    Point *P1 = P0 + 2 * cap;
    int32_t *score = reinterpret_cast<int32_t *>(P1 + 2 * cap);
    int32_t *prev = score + cap;
    int32_t *tree = prev + cap;
    chain_cluster(frag, n, P0, P1, score, prev, tree, res, kCls != 2, chainOut);
  }
  if (lane == 0) out[ci] = res;
}

// the chains of the clusters that became intervals, back to back: one wavefront per interval
__global__ __launch_bounds__(256) void iv_gather_chains_kernel(int n, const int64_t *src, const int64_t *dst, const uint32_t *chains,
                                                               uint32_t *out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int64_t a = 3 * src[i], b = 3 * dst[i], words = 3 * (dst[i + 1] - dst[i]);
  for (int64_t w = threadIdx.x & 63; w < words; w += 64) out[b + w] = chains[a + w];
}
__global__ __launch_bounds__(256) void iv_cluster_sizes_kernel(const IvCluster *clusters, int nClusters, int64_t *sizes) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < nClusters) sizes[i] = (int64_t)(clusters[i].end - clusters[i].start);
}

// cluster sizes -> class of every cluster, class counts (wave-aggregated), and for class 2 its scratch bytes
// midCap: the largest cluster that may take the 72 KB LDS kernel (kChainMid where the device grants that much to a
// workgroup, kChainSmall -- no cluster at all -- where it does not: those then run with their scratch in global memory)
__global__ __launch_bounds__(256) void iv_classify_kernel(const IvCluster *clusters, int nClusters, int32_t *cls, int32_t *classCount,
                                                          int midCap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int c = -1;
  if (i < nClusters) {
    const int n = (int)(clusters[i].end - clusters[i].start);
    c = n <= kChainSmall ? 0 : (n <= midCap ? 1 : 2);
    cls[i] = c;
  }
  for (int k = 0; k < 3; k++) {
    const unsigned long long m = __ballot(c == k);
    if (m != 0ull && lane_id() == 0) atomicAdd(classCount + k, (int)__popcll(m));
  }
}

}  // namespace
}  // namespace blasr_hip

using namespace blasr_hip;

extern "C" {

int blasr_hip_find_intervals_device_chains(blasr_hip_ctx *ctx, const uint8_t *genome, int64_t genome_length,
                                           const uint32_t *seq_start_pos, int32_t n_seq_pos, const int32_t *count_table,
                                           int32_t count_tuple_size, int32_t n_reads, const blasr_hip_read *reads,
                                           const blasr_hip_anchor *anchors_device, const int64_t *list_off,
                                           const blasr_hip_interval_params *params, blasr_hip_interval *intervals,
                                           float *p_values, int64_t capacity, int64_t *n_intervals, int64_t *read_interval_off,
                                           uint32_t *chains, int64_t chains_capacity, int64_t *chains_used, int64_t *chain_off) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  const bool wantChains = chains_used != nullptr;
  if (wantChains) {
    *chains_used = 0;
    if (chains_capacity < 0 || (chains_capacity > 0 && !chains) || (capacity > 0 && !chain_off))
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "find_intervals_device: bad chain arguments");
    for (int64_t i = 0; i <= capacity && chain_off; i++) chain_off[i] = 0;
  }
  if (!genome || genome_length <= 0 || genome_length >= (1ll << 32) || !seq_start_pos || n_seq_pos < 1 || !count_table ||
      count_tuple_size < 1 || count_tuple_size > 14 || n_reads < 0 || !params || !n_intervals || !read_interval_off ||
      capacity < 0 || (capacity > 0 && !intervals) || (n_reads > 0 && (!reads || !list_off)))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "find_intervals_device: bad arguments");
  if (params->n_candidates < 0 || params->min_interval < 0)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "find_intervals_device: negative parameter");
  *n_intervals = 0;
  for (int32_t r = 0; r <= n_reads; r++) read_interval_off[r] = 0;
  if (n_reads == 0) return BLASR_HIP_OK;
  const int nLists = 2 * n_reads;
  for (int32_t r = 0; r < n_reads; r++)
    if (reads[r].subread_start > reads[r].subread_end || reads[r].subread_end > reads[r].length ||
        list_off[2 * r + 1] < list_off[2 * r] || list_off[2 * r + 2] < list_off[2 * r + 1])
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "find_intervals_device: read %d: bad subread or list offsets", r);
  const int64_t base = list_off[0], nAnchors = list_off[nLists] - base;
  if (nAnchors >= (1ll << 32)) return blasr_hip_fail(ctx, BLASR_HIP_ERR_UNSUPPORTED, "find_intervals_device: %lld anchors in one call", (long long)nAnchors);
  if (nAnchors > 0 && !anchors_device) return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "find_intervals_device: no anchors");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  // BLASR_HIP_IV_TRACE: wall time between the call's synchronisation points
  const bool ivTrace = getenv("BLASR_HIP_IV_TRACE") != nullptr;
  auto ivNow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double ivT = ivNow();
  auto ivMark = [&](const char *what) {
    if (!ivTrace) return;
    const double t = ivNow();
    fprintf(stderr, "[iv] %-28s %8.2f ms\n", what, t - ivT);
    ivT = t;
  };
  // ---- host-side tables: list offsets from 0, the window length of every read (Blasr.cpp:2402-2406)
  std::vector<int64_t> off((size_t)nLists + 1);
  for (int i = 0; i <= nLists; i++) off[(size_t)i] = list_off[i] - base;
  std::vector<uint32_t> ivLen((size_t)n_reads);
  for (int32_t r = 0; r < n_reads; r++) {
    const uint32_t span = reads[r].subread_end - reads[r].subread_start;
    uint32_t L = (uint32_t)(span * (1 + params->indel_rate));
    if (L - span > 50000u) L = span + 50000u;
    ivLen[(size_t)r] = L;
  }
  // ---- device memory: a few arrays per anchor, from the context's workspace
  size_t scanBytes = 0;
  {
    uint32_t *nullU = nullptr;
    HIP_TRY(ctx, rocprim::exclusive_scan(nullptr, scanBytes, nullU, nullU, 0u, (size_t)std::max<int64_t>(nAnchors, nLists + 1) + 1,
                                         rocprim::plus<uint32_t>(), st));
  }
  size_t at = 0;
  auto carve = [&](size_t bytes) {
    const size_t o = at;
    at += (bytes + 255) & ~(size_t)255;
    return o;
  };
  const size_t oOff = carve(sizeof(int64_t) * ((size_t)nLists + 1)), oLen = carve(sizeof(uint32_t) * (size_t)n_reads),
               oSeq = carve(sizeof(uint32_t) * (size_t)n_seq_pos), oKeep = carve(sizeof(uint32_t) * ((size_t)nAnchors + 1)),
               oSlot = carve(sizeof(uint32_t) * ((size_t)nAnchors + 1)), oPos = carve(sizeof(blasr_hip_anchor) * ((size_t)nAnchors + 1)),
               oNewOff = carve(sizeof(uint32_t) * ((size_t)nLists + 1)), oPast = carve(sizeof(uint2) * (size_t)nLists),
               oCount = carve(sizeof(uint32_t) * ((size_t)nLists + 1)), oCOff = carve(sizeof(uint32_t) * ((size_t)nLists + 1)),
               oFew = carve(sizeof(IvCluster) * (size_t)kFirstFew * (size_t)nLists),
               oScan = carve(scanBytes), oTotals = carve(64);
  const size_t fixedBytes = at;
  {
    const int rc = blasr_hip_grow_workspace(ctx, fixedBytes + (64u << 20));
    if (rc != BLASR_HIP_OK) return rc;
  }
  char *w = ctx->workspace;
  int64_t *dOff = (int64_t *)(w + oOff);
  uint32_t *dLen = (uint32_t *)(w + oLen), *dSeq = (uint32_t *)(w + oSeq), *dKeep = (uint32_t *)(w + oKeep),
           *dSlot = (uint32_t *)(w + oSlot), *dNewOff = (uint32_t *)(w + oNewOff), *dCount = (uint32_t *)(w + oCount),
           *dCOff = (uint32_t *)(w + oCOff);
  blasr_hip_anchor *dPos = (blasr_hip_anchor *)(w + oPos);
  uint2 *dPast = (uint2 *)(w + oPast);
  uint32_t *dTotals = (uint32_t *)(w + oTotals);
  HIP_TRY(ctx, hipMemcpyAsync(dOff, off.data(), sizeof(int64_t) * off.size(), hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dLen, ivLen.data(), sizeof(uint32_t) * ivLen.size(), hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dSeq, seq_start_pos, sizeof(uint32_t) * (size_t)n_seq_pos, hipMemcpyHostToDevice, st));
  IvLists L;
  L.anchors = anchors_device + base;
  L.listOff = dOff;
  L.nLists = nLists;
  // ---- RemoveOverlappingAnchors, the survivors back to back
  uint32_t totalKept = 0;
  if (nAnchors > 0) {
    const unsigned grid = (unsigned)((nAnchors + 255) / 256);
    hipLaunchKernelGGL(iv_mark_kernel, dim3(grid), dim3(256), 0, st, L, nAnchors, dKeep);
    HIP_TRY(ctx, hipMemsetAsync(dKeep + nAnchors, 0, sizeof(uint32_t), st));
    HIP_TRY(ctx, rocprim::exclusive_scan(w + oScan, scanBytes, dKeep, dSlot, 0u, (size_t)nAnchors + 1, rocprim::plus<uint32_t>(), st));
    hipLaunchKernelGGL(iv_compact_kernel, dim3(grid), dim3(256), 0, st, L, nAnchors, dKeep, dSlot, dPos);
    HIP_TRY(ctx, hipMemcpyAsync(&totalKept, dSlot + nAnchors, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
  }
  ivMark("tables, mark, scan, compact");
  hipLaunchKernelGGL(iv_lists_kernel, dim3((unsigned)((nLists + 256) / 256)), dim3(256), 0, st, L, dKeep, dSlot, totalKept, dNewOff, dPast);
  // ---- StoreLargestIntervals: count, scan, write
  const unsigned gridL = (unsigned)((nLists + 63) / 64);
  hipLaunchKernelGGL((iv_windows_kernel<false>), dim3(gridL), dim3(64), 0, st, dPos, dNewOff, dPast, nLists, dSeq, n_seq_pos, dLen,
                     params->min_interval, dCount, nullptr, nullptr, (IvCluster *)(w + oFew));
  HIP_TRY(ctx, hipMemsetAsync(dCount + nLists, 0, sizeof(uint32_t), st));
  HIP_TRY(ctx, rocprim::exclusive_scan(w + oScan, scanBytes, dCount, dCOff, 0u, (size_t)nLists + 1, rocprim::plus<uint32_t>(), st));
  std::vector<uint32_t> cOff((size_t)nLists + 1);
  HIP_TRY(ctx, hipMemcpyAsync(cOff.data(), dCOff, sizeof(uint32_t) * cOff.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  HIP_TRY(ctx, hipGetLastError());
  const int nClusters = (int)cOff[(size_t)nLists];
  ivMark("lists, windows (count), scan");
  std::vector<blasr_hip_iv_cluster> summary((size_t)nClusters);
  struct FreeDev {
    void *p = nullptr;
    ~FreeDev() {
      if (p) (void)hipFree(p);
    }
  } chainsMem, chainSlotMem;        // -rbao only: every cluster's chain, in a slot as long as the cluster
  std::vector<int64_t> chainSlot;   // the slots' offsets (triplets), nClusters + 1
  if (nClusters > 0) {
    // cluster records, classes and results behind the fixed part (the per-anchor arrays stay: the chains read dPos)
    size_t at2 = fixedBytes;
    auto carve2 = [&](size_t bytes) {
      const size_t o = at2;
      at2 += (bytes + 255) & ~(size_t)255;
      return o;
    };
    const size_t oCl = carve2(sizeof(IvCluster) * (size_t)nClusters), oCls = carve2(sizeof(int32_t) * (size_t)nClusters),
                 oIdx = carve2(sizeof(int32_t) * (size_t)nClusters), oIdxOut = carve2(sizeof(int32_t) * (size_t)nClusters),
                 oClsOut = carve2(sizeof(int32_t) * (size_t)nClusters), oRes = carve2(sizeof(blasr_hip_iv_cluster) * (size_t)nClusters);
    size_t sortBytes = 0;
    {
      int32_t *nullI = nullptr;
      HIP_TRY(ctx, rocprim::radix_sort_pairs(nullptr, sortBytes, nullI, nullI, nullI, nullI, (size_t)nClusters, 0u, 2u, st));
    }
    const size_t oSort = carve2(sortBytes);
    if (at2 > ctx->workspaceBytes) {
      // (growing the workspace loses its contents: run again from the top with the room known -- rare, first call only)
      const int rc = blasr_hip_grow_workspace(ctx, at2 + at2 / 4);
      if (rc != BLASR_HIP_OK) return rc;
      return blasr_hip_find_intervals_device_chains(ctx, genome, genome_length, seq_start_pos, n_seq_pos, count_table, count_tuple_size,
                                                    n_reads, reads, anchors_device, list_off, params, intervals, p_values, capacity,
                                                    n_intervals, read_interval_off, chains, chains_capacity, chains_used, chain_off);
    }
    IvCluster *dCl = (IvCluster *)(w + oCl);
    int32_t *dCls = (int32_t *)(w + oCls), *dIdx = (int32_t *)(w + oIdx), *dIdxOut = (int32_t *)(w + oIdxOut),
            *dClsOut = (int32_t *)(w + oClsOut);
    blasr_hip_iv_cluster *dRes = (blasr_hip_iv_cluster *)(w + oRes);
    hipLaunchKernelGGL((iv_windows_kernel<true>), dim3(gridL), dim3(64), 0, st, dPos, dNewOff, dPast, nLists, dSeq, n_seq_pos, dLen,
                       params->min_interval, dCount, dCOff, dCl, (IvCluster *)(w + oFew));
    HIP_TRY(ctx, hipMemsetAsync(dTotals, 0, 64, st));
    // the mid-size kernel needs 72 KB of LDS per workgroup, above the 64 KB a kernel gets unasked: the opt-in is a property
    // of the (function, device) pair, so it is made on every call for the context's device and its answer is looked at
    int midCap = kChainMid;
    {
      int optin = 0;
      if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, ctx->device) != hipSuccess) optin = 0;
      if (optin < (int)chain_bytes(kChainMid) ||
          hipFuncSetAttribute(reinterpret_cast<const void *>(&iv_chain_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)chain_bytes(kChainMid)) != hipSuccess)
        midCap = kChainSmall;
      (void)hipGetLastError();
    }
    hipLaunchKernelGGL(iv_classify_kernel, dim3((unsigned)((nClusters + 255) / 256)), dim3(256), 0, st, dCl, nClusters, dCls,
                       (int32_t *)dTotals, midCap);
    {  // order = the clusters by class (a stable two-bit sort of their indices)
      std::vector<int32_t> iota((size_t)nClusters);
      for (int i = 0; i < nClusters; i++) iota[(size_t)i] = i;
      HIP_TRY(ctx, hipMemcpyAsync(dIdx, iota.data(), sizeof(int32_t) * iota.size(), hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, rocprim::radix_sort_pairs(w + oSort, sortBytes, dCls, dClsOut, dIdx, dIdxOut, (size_t)nClusters, 0u, 2u, st));
      HIP_TRY(ctx, hipStreamSynchronize(st));  // (iota goes out of scope)
    }
    int32_t classCount[3] = {0, 0, 0};
    HIP_TRY(ctx, hipMemcpy(classCount, dTotals, sizeof(classCount), hipMemcpyDeviceToHost));
    ivMark("windows (write), classes");
    if (wantChains) {  // a slot per cluster (an exclusive scan of the clusters' sizes), then the memory
      HIP_TRY(ctx, hipMalloc(&chainSlotMem.p, sizeof(int64_t) * 2 * ((size_t)nClusters + 1)));
      int64_t *dSizes = (int64_t *)chainSlotMem.p, *dSlot = dSizes + nClusters + 1;
      HIP_TRY(ctx, hipMemsetAsync(dSizes + nClusters, 0, sizeof(int64_t), st));
      hipLaunchKernelGGL(iv_cluster_sizes_kernel, dim3((unsigned)((nClusters + 255) / 256)), dim3(256), 0, st, dCl, nClusters, dSizes);
      size_t sb = 0;
      HIP_TRY(ctx, rocprim::exclusive_scan(nullptr, sb, dSizes, dSlot, (int64_t)0, (size_t)nClusters + 1, rocprim::plus<int64_t>(), st));
      FreeDev scanTmp;
      HIP_TRY(ctx, hipMalloc(&scanTmp.p, std::max<size_t>(sb, 256)));
      HIP_TRY(ctx, rocprim::exclusive_scan(scanTmp.p, sb, dSizes, dSlot, (int64_t)0, (size_t)nClusters + 1, rocprim::plus<int64_t>(), st));
      chainSlot.resize((size_t)nClusters + 1);
      HIP_TRY(ctx, hipMemcpyAsync(chainSlot.data(), dSlot, sizeof(int64_t) * chainSlot.size(), hipMemcpyDeviceToHost, st));
      HIP_TRY(ctx, hipStreamSynchronize(st));
      HIP_TRY(ctx, hipMalloc(&chainsMem.p, sizeof(uint32_t) * 3 * (size_t)std::max<int64_t>(chainSlot[(size_t)nClusters], 1)));
    }
    uint32_t *dChains = (uint32_t *)chainsMem.p;
    const int64_t *dChainSlot = wantChains ? (const int64_t *)chainSlotMem.p + nClusters + 1 : nullptr;
    // class 2: scratch in global memory, sized per cluster
    char *dScratch = nullptr;
    int64_t *dScratchOff = nullptr;
    struct Free {
      void *p = nullptr;
      ~Free() {
        if (p) (void)hipFree(p);
      }
    } scratchMem, scratchOffMem;
    if (classCount[2] > 0) {
      std::vector<int32_t> big((size_t)classCount[2]);
      std::vector<IvCluster> allCl((size_t)nClusters);
      HIP_TRY(ctx, hipMemcpy(big.data(), dIdxOut + classCount[0] + classCount[1], sizeof(int32_t) * big.size(), hipMemcpyDeviceToHost));
      HIP_TRY(ctx, hipMemcpy(allCl.data(), dCl, sizeof(IvCluster) * allCl.size(), hipMemcpyDeviceToHost));
      std::vector<int64_t> so(big.size());
      int64_t total = 0;
      for (size_t i = 0; i < big.size(); i++) {
        so[i] = total;
        const IvCluster &c = allCl[(size_t)big[i]];
        total += (int64_t)((chain_bytes((int)(c.end - c.start)) + 255) & ~(size_t)255);
      }
      HIP_TRY(ctx, hipMalloc(&scratchMem.p, (size_t)std::max<int64_t>(total, 256)));
      HIP_TRY(ctx, hipMalloc(&scratchOffMem.p, sizeof(int64_t) * so.size()));
      HIP_TRY(ctx, hipMemcpy(scratchOffMem.p, so.data(), sizeof(int64_t) * so.size(), hipMemcpyHostToDevice));
      dScratch = (char *)scratchMem.p;
      dScratchOff = (int64_t *)scratchOffMem.p;
    }
    // the mid-size clusters' kernel holds two wavefronts a CU (72 KB of LDS each): it runs on a side stream beside the small
    // clusters' one instead of in front of it (17 + 12 ms one after the other at configs[2])
    const bool beside = classCount[0] > 0 && classCount[1] > 0 && ctx->auxStream[0] && ctx->auxEvent[0] && ctx->auxEvent[1];
    hipStream_t stMid = beside ? ctx->auxStream[0] : st;
    if (beside) {
      HIP_TRY(ctx, hipEventRecord(ctx->auxEvent[0], st));
      HIP_TRY(ctx, hipStreamWaitEvent(stMid, ctx->auxEvent[0], 0));
    }
    if (classCount[1] > 0)
      hipLaunchKernelGGL((iv_chain_kernel<1>), dim3((unsigned)classCount[1]), dim3(64), chain_bytes(kChainMid), stMid, dPos, dNewOff, dCl,
                         dIdxOut, (const int32_t *)dTotals, nullptr, nullptr, dRes, dChains, dChainSlot);
    if (classCount[0] > 0)
      hipLaunchKernelGGL((iv_chain_kernel<0>), dim3((unsigned)((classCount[0] + 3) / 4)), dim3(256), 4 * chain_bytes(kChainSmall), st, dPos,
                         dNewOff, dCl, dIdxOut, (const int32_t *)dTotals, nullptr, nullptr, dRes, dChains, dChainSlot);
    if (beside) {
      HIP_TRY(ctx, hipEventRecord(ctx->auxEvent[1], stMid));
      HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->auxEvent[1], 0));
    }
    if (classCount[2] > 0)
      hipLaunchKernelGGL((iv_chain_kernel<2>), dim3((unsigned)classCount[2]), dim3(64), 0, st, dPos, dNewOff, dCl, dIdxOut,
                         (const int32_t *)dTotals, dScratch, dScratchOff, dRes, dChains, dChainSlot);
    HIP_TRY(ctx, hipMemcpyAsync(summary.data(), dRes, sizeof(blasr_hip_iv_cluster) * summary.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    HIP_TRY(ctx, hipGetLastError());
    if (ivTrace) fprintf(stderr, "[iv] %d clusters: %d small, %d mid, %d large\n", nClusters, classCount[0], classCount[1], classCount[2]);
    ivMark("chains");
  }
  // ---- the host's end: p-values, the tests of FindMaxInterval.h:805-825, the interval sets
  std::vector<int64_t> cOff64((size_t)nLists + 1);
  for (int i = 0; i <= nLists; i++) cOff64[(size_t)i] = (int64_t)cOff[(size_t)i];
  std::vector<int64_t> clusterOf(wantChains ? (size_t)std::max<int64_t>(capacity, 1) : 0);
  const int rc = blasr_hip_intervals_from_clusters(genome, genome_length, seq_start_pos, n_seq_pos, count_table, count_tuple_size,
                                                   n_reads, reads, summary.data(), cOff64.data(), params, intervals, p_values, capacity,
                                                   n_intervals, read_interval_off, wantChains ? clusterOf.data() : nullptr);
  ivMark("host: p-values, interval sets");
  if (rc != BLASR_HIP_OK) return rc == BLASR_HIP_ERR_CAPACITY ? blasr_hip_fail(ctx, rc, "find_intervals_device: %lld intervals, room for %lld",
                                                                                (long long)*n_intervals, (long long)capacity) : rc;
  if (wantChains) {  // the chains of the clusters that became intervals, gathered on the device, one download
    const int64_t nIv = *n_intervals;
    std::vector<int64_t> src((size_t)nIv + 1, 0), dst((size_t)nIv + 1, 0);
    for (int64_t k = 0; k < nIv; k++) {
      const int64_t c = clusterOf[(size_t)k];
      src[(size_t)k] = chainSlot[(size_t)c];
      dst[(size_t)k + 1] = dst[(size_t)k] + summary[(size_t)c].count;
    }
    const int64_t total = dst[(size_t)nIv];
    *chains_used = total;
    for (int64_t k = 0; k <= nIv; k++) chain_off[k] = dst[(size_t)k];
    if (total > chains_capacity)
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_CAPACITY, "find_intervals_device: %lld chained anchors, room for %lld", (long long)total,
                            (long long)chains_capacity);
    if (total > 0) {
      FreeDev dIdx, dOut;
      HIP_TRY(ctx, hipMalloc(&dIdx.p, sizeof(int64_t) * 2 * ((size_t)nIv + 1)));
      HIP_TRY(ctx, hipMalloc(&dOut.p, sizeof(uint32_t) * 3 * (size_t)total));
      int64_t *dSrc = (int64_t *)dIdx.p, *dDst = dSrc + nIv + 1;
      HIP_TRY(ctx, hipMemcpyAsync(dSrc, src.data(), sizeof(int64_t) * ((size_t)nIv + 1), hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipMemcpyAsync(dDst, dst.data(), sizeof(int64_t) * ((size_t)nIv + 1), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(iv_gather_chains_kernel, dim3((unsigned)((nIv + 3) / 4)), dim3(256), 0, st, (int)nIv, dSrc, dDst,
                         (const uint32_t *)chainsMem.p, (uint32_t *)dOut.p);
      HIP_TRY(ctx, hipMemcpyAsync(chains, dOut.p, sizeof(uint32_t) * 3 * (size_t)total, hipMemcpyDeviceToHost, st));
      HIP_TRY(ctx, hipStreamSynchronize(st));
      HIP_TRY(ctx, hipGetLastError());
    }
  }
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CTX(ctx)

int blasr_hip_find_intervals_device(blasr_hip_ctx *ctx, const uint8_t *genome, int64_t genome_length, const uint32_t *seq_start_pos,
                                    int32_t n_seq_pos, const int32_t *count_table, int32_t count_tuple_size, int32_t n_reads,
                                    const blasr_hip_read *reads, const blasr_hip_anchor *anchors_device, const int64_t *list_off,
                                    const blasr_hip_interval_params *params, blasr_hip_interval *intervals, float *p_values,
                                    int64_t capacity, int64_t *n_intervals, int64_t *read_interval_off) try {
  return blasr_hip_find_intervals_device_chains(ctx, genome, genome_length, seq_start_pos, n_seq_pos, count_table, count_tuple_size,
                                                n_reads, reads, anchors_device, list_off, params, intervals, p_values, capacity,
                                                n_intervals, read_interval_off, nullptr, 0, nullptr, nullptr);
} BLASR_HIP_CATCH_CTX(ctx)

}  // extern "C"
