// next-1, device half: anchoring reads in the suffix array of the genome.
//
// Replaces MapReadToGenome (common/algorithms/anchoring/MapBySuffixArray.h:217-311) for both strands of every read
// of a batch, and SortMatchPosList (datastructures/anchoring/MatchPos.h:166-168) on the lists:
//   anchor_search_kernel   one lane per read position p of one strand: StoreLCPBounds (SuffixArray.h:928-1070) --
//                          the lookup table's range for the first bases, then one SearchLeftBound / SearchRightBound
//                          pair (:733-816) per further base -- and the heuristics of LocateAnchorBoundsInSuffixArray
//                          (MapBySuffixArray.h:93-204): extend a unique match along the genome, back a dead-end
//                          unique match up by one base, back a repeated one up by `expand`.  The last few ranges
//                          are kept in registers (the only ones those heuristics revisit).  Output: the range
//                          [low, low + count) of the index array and the match length, per position.
//   anchor_advance_kernel  advanceExactMatches: the positions the reference's loop skips, one lane per strand
//   (rocprim scan)         exclusive scan of the counts -> where every position's anchors go
//   anchor_emit_kernel     one wavefront per 64 positions, one lane per anchor: (index[low + i], p, length)
//   (rocprim segmented radix sort on (t, q)) + anchor_pack_kernel   SortMatchPosList
// The genome (ASCII), the index array and the lookup table stay resident: 4 bytes of index per genome base, sized for
// a whole human genome in HBM.  The work is latency / random-sector bound: every probe of a binary search is a load of
// index[m] followed by a dependent load of genome[index[m] + offset].
#include <hip/hip_runtime.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "../../include/blasr_hip.h"
#include "ctx.h"
#include "wave_utils.h"

namespace {
using namespace blasr_hip;

// ThreeBit (NucConversion.h:48-84), rebuilt from its defining property: codes 0..4 map to themselves, '$' to 5,
// ACGT in both cases to 0..3, the IUPAC ambiguity letters the table lists to 4, everything else 255.
struct NucTable {
  uint8_t v[256];
};
constexpr NucTable make_three_bit() {
  NucTable t{};
  for (int i = 0; i < 256; i++) t.v[i] = 255;
  for (int i = 0; i < 5; i++) t.v[i] = (uint8_t)i;
  t.v['$'] = 5;
  const char acgt[4] = {'A', 'C', 'G', 'T'};
  for (int i = 0; i < 4; i++) {
    t.v[(int)acgt[i]] = (uint8_t)i;
    t.v[(int)acgt[i] + 32] = (uint8_t)i;
  }
  const char upper[] = "BDHKMNRSUVWY_";
  for (int i = 0; upper[i]; i++) t.v[(int)upper[i]] = 4;
  const char lower[] = "bdhkmnrsuvwxy";
  for (int i = 0; lower[i]; i++) t.v[(int)lower[i]] = 4;
  return t;
}
__constant__ NucTable kThreeBit = make_three_bit();

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

struct AnchorSeg {     // one strand of one read
  int64_t off;         // first base of the forward read in the pool
  uint32_t len;        // read.length
  uint32_t start;      // subreadStart on this strand (Blasr.cpp:3449-3450 for the reverse one)
  uint32_t matchEnd;   // subreadEnd - minPrefixMatchLength + 1 (MapBySuffixArray.h:46)
  uint32_t strand;
};

struct AnchorArgs {
  const uint8_t *genome;
  uint32_t tlen;
  const uint32_t *index, *startPos, *endPos;
  const uint2 *ranges;  // {startPos, endPos} of a lookup word side by side: one random sector per position instead of two
  const uint16_t *ext;  // per array position: the kExtBases bases after the lookup prefix (or null)
  int exp;              // BLASR_HIP_ANCHOR_EXP: timing experiments with wrong results
  int noWin;            // BLASR_HIP_ANCHOR_NO_WIN: every probe of the companion array from global memory (as in round 2)
  int prefixLen;
  const uint8_t *pool;
  const AnchorSeg *seg;
  const int64_t *posOff;  // nSeg + 1: first position of every segment in this piece
  int nSeg;
  int64_t nPos;
  uint32_t minMatch, minPrefix;
  int expand, maxAnchors, advance, maxLcp, useLookup, stopUnique;
  uint32_t *lo, *cnt, *len;
  uint32_t *lcp;                // per position: lcpLength as the advance step sees it (only when advance != 0)
  unsigned long long *probes;   // optional: probes of the index array
};

// ThreeBit of the strand's base at `at`: the forward read's byte through the table of its strand (three: 512 entries,
// the second half for the complement)
__device__ __forceinline__ int strand_code(const uint8_t *pool, const AnchorSeg &s, uint32_t at, const uint8_t *three) {
  const uint32_t raw = pool[s.off + (int64_t)(s.strand ? s.len - 1 - at : at)];
  return three[raw + (s.strand ? 256u : 0u)];
}

// the segment a position of the piece belongs to: the last s with posOff[s] <= g
__device__ __forceinline__ int segment_of(const int64_t *posOff, int nSeg, int64_t g) {
  int lo = 0, hi = nSeg;  // posOff[lo] <= g < posOff[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (posOff[mid] <= g)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

// The companion of the index array: for the suffix at array position m, the kExtBases bases that follow its first
// `prefixLen` ones, in 16 bits -- bits 2k, 2k+1 = base k, bits 12-14 = how many of them are ACGT before the first
// that is not, bit 15 = that first one is the end of the genome (not an N).  Inside a search range every suffix shares
// the bases matched so far with the read, so the symbol at the next offset is always among those: ONE load per probe,
// and 32 suffixes to a 64-byte sector -- the range a 13-base prefix leaves on a human-size genome (~46 suffixes) is two
// sectors, and the steps after the first stay inside what the first one fetched.  The kernel is bound by HBM sectors
// (DESIGN.md 2.8), so the entry is kept this small; searches deeper than prefix + kExtBases read index[] + genome[].
constexpr int kExtBases = 6;
// A lane whose lookup range is short copies that stretch of the companion array into LDS once (nine 16-byte loads, issued
// together) and every later probe inside it reads LDS: the ~20 dependent sector loads of a position's binary searches
// (each one a sector of its own by the time it is issued: PMC) become three sectors fetched side by side.
constexpr int kWinEntries = 72;  // entries staged per lane: the range (~46 suffixes at hg19 size) + alignment to 16 bytes
constexpr int kWinStride = 37;   // dwords per lane (36 + 1: an odd stride keeps the lanes' slots on different banks)
typedef uint64_t u64u __attribute__((aligned(1)));  // an 8-byte load from any address
struct ExtWin {
  const uint16_t *lds;  // this lane's slot
  uint32_t base, count; // array positions [base, base + count) are in it
};

__global__ __launch_bounds__(256) void anchor_ext_kernel(const uint8_t *genome, uint32_t tlen, const uint32_t *index,
                                                         int64_t n, int prefixLen, uint16_t *ext) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= n) return;
  const uint64_t base = (uint64_t)index[m] + (uint64_t)prefixLen;
  uint32_t bits = 0, vl = 0, end = 0;
  for (int k = 0; k < kExtBases; k++) {
    if (base + k >= tlen) {
      end = 1;
      break;
    }
    const int c = kThreeBit.v[genome[base + k]];
    if (c >= 4) break;
    bits |= (uint32_t)c << (2 * k);
    vl++;
  }
  ext[m] = (uint16_t)(bits | (vl << 12) | (end << 15));
}

// What the suffix at array position m holds `offset` bases in: 0..3 that base, 4 a base that is not ACGT, 5 nothing
// (the suffix ends exactly there), 6 the suffix is shorter still.
__device__ __forceinline__ int suffix_symbol(const AnchorArgs &a, const uint8_t *three, const ExtWin &win, uint32_t m,
                                             uint32_t offset, unsigned &probes) {
  const uint32_t k = offset - (uint32_t)a.prefixLen;  // (wraps to a large number below the prefix)
  if (a.ext && k < (uint32_t)kExtBases) {
    uint32_t w;
    if (m - win.base < win.count) {
      w = win.lds[m - win.base];
    } else {
      probes++;
      w = a.ext[m];
    }
    const uint32_t vl = (w >> 12) & 7u;
    if (k < vl) return (int)((w >> (2 * k)) & 3u);
    if (k == vl) return (w >> 15) ? 5 : 4;
    // (k > vl: not for a suffix of the current range; the general path answers)
  }
  probes++;
  const uint32_t im = a.index[m];
  const uint32_t sufLen = a.tlen - im;
  if (sufLen == offset) return 5;
  if (sufLen < offset) return 6;
  const int t = three[a.genome[(uint64_t)im + offset]];
  return t < 4 ? t : 4;
}

// SearchLeftBound (SuffixArray.h:733-775)
__device__ __forceinline__ uint32_t left_bound(const AnchorArgs &a, const uint8_t *three, const ExtWin &win, uint32_t offset,
                                               int tq, uint32_t l, uint32_t r, unsigned &probes) {
  uint32_t ll = l, lr = r;
  while (ll < lr) {
    const uint32_t m = ll + ((lr - ll) >> 1);
    const int sym = suffix_symbol(a, three, win, m, offset, probes);
    if (sym == 5) {
      ll = m + 1;
      continue;
    }
    const int comp = sym == 6 ? -1 : sym - tq;
    if (comp < 0)
      ll = m + 1;
    else
      lr = m;
  }
  return ll;
}

// SearchRightBound (SuffixArray.h:777-816), with its early stop at a suffix that ends exactly at `offset`
__device__ __forceinline__ uint32_t right_bound(const AnchorArgs &a, const uint8_t *three, const ExtWin &win, uint32_t offset,
                                                int tq, uint32_t l, uint32_t r, unsigned &probes) {
  uint32_t rl = l, rr = r;
  while (rl < rr) {
    const uint32_t m = rl + ((rr - rl) >> 1);
    const int sym = suffix_symbol(a, three, win, m, offset, probes);
    if (sym == 5) {
      rr = m;
      break;
    }
    if (sym == 6) {
      rr = m;
    } else if (sym - tq <= 0) {
      rl = m + 1;
    } else {
      rr = m;
    }
  }
  return rr;
}

// SearchLeftBound followed by SearchRightBound on its result are the lower and the upper bound of `tq` among the
// symbols of [l, r), which the array's order keeps sorted: one descent finds a suffix that holds `tq`, then the two
// bounds are looked for on either side of it -- the same two numbers with about a third fewer probes.  A suffix that
// ends at or before `offset` (symbols 5, 6: only possible without the lookup table, at the first bases) is where the
// two searches of the reference differ from a plain ordering, so the pair is then computed as the reference does.
// Returns true when the pair came from the plain descent (then, if l < r, the suffix at l holds tq at `offset`).
__device__ __forceinline__ bool equal_range(const AnchorArgs &a, const uint8_t *three, const ExtWin &win, uint32_t offset,
                                            int tq, uint32_t &l, uint32_t &r, unsigned &probes) {
  uint32_t lo = l, hi = r;
  bool plain = true;
  while (lo < hi && plain) {
    const uint32_t m = lo + ((hi - lo) >> 1);
    const int s = suffix_symbol(a, three, win, m, offset, probes);
    if (s >= 5) {
      plain = false;
    } else if (s < tq) {
      lo = m + 1;
    } else if (s > tq) {
      hi = m;
    } else {
      uint32_t x = lo, y = m;  // the first suffix that holds tq
      while (x < y && plain) {
        const uint32_t mm = x + ((y - x) >> 1);
        const int s2 = suffix_symbol(a, three, win, mm, offset, probes);
        if (s2 >= 5)
          plain = false;
        else if (s2 < tq)
          x = mm + 1;
        else
          y = mm;
      }
      uint32_t u = m + 1, v = hi;  // the first suffix beyond them
      while (u < v && plain) {
        const uint32_t mm = u + ((v - u) >> 1);
        const int s2 = suffix_symbol(a, three, win, mm, offset, probes);
        if (s2 >= 5)
          plain = false;
        else if (s2 <= tq)
          u = mm + 1;
        else
          v = mm;
      }
      if (plain) {
        l = x;
        r = u;
        return true;
      }
    }
  }
  if (plain) {  // nothing holds tq: both bounds fall on the first larger symbol
    l = r = lo;
    return true;
  }
  l = left_bound(a, three, win, offset, tq, l, r, probes);
  r = right_bound(a, three, win, offset, tq, l, r, probes);
  return false;
}

// kHist: how many of the latest (low, high) pairs stay in registers = max(expand, 1) + 1
template <int kHist>
__global__ __launch_bounds__(256) void anchor_search_kernel(AnchorArgs a) {
  // three[c]: ThreeBit of a base; three[256 + c]: ThreeBit of its reverse complement; rcChar[c]: the reverse complement
  // itself -- a read base of the minus strand is then ONE table read like one of the plus strand (ReverseComplementNuc as
  // eleven compare-and-select pairs cost a minus-strand position ~600 of its ~2 000 vector instructions: thirteen times for
  // the lookup word, once per descent, eight times per step of the extension)
  __shared__ uint8_t three[512];
  __shared__ uint8_t rcChar[256];
  __shared__ uint32_t extWin[256 * kWinStride];
  three[threadIdx.x] = kThreeBit.v[threadIdx.x];
  rcChar[threadIdx.x] = rc_nuc((uint8_t)threadIdx.x);
  three[256 + threadIdx.x] = kThreeBit.v[rc_nuc((uint8_t)threadIdx.x)];
  __syncthreads();
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= a.nPos) return;
  ExtWin win;
  win.lds = reinterpret_cast<const uint16_t *>(extWin + threadIdx.x * kWinStride);
  win.base = 0;
  win.count = 0;
  // the segment of the wavefront's first position by one search on the scalar unit; a lane beyond that segment's end
  // (a wavefront that straddles lists) steps on from there
  const int64_t g0 = (int64_t)blockIdx.x * 256 + (int64_t)(__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) << 6);
  int si = segment_of(a.posOff, a.nSeg, g0);
  while (si + 1 < a.nSeg && g >= a.posOff[si + 1]) si++;
  const AnchorSeg s = a.seg[si];
  const uint32_t p = s.start + (uint32_t)(g - a.posOff[si]);
  const uint32_t queryLength = s.matchEnd - p;  // MapBySuffixArray.h:66

  uint32_t hl[kHist], hr[kHist];  // [0] = the latest pair
#pragma unroll
  for (int i = 0; i < kHist; i++) hl[i] = hr[i] = 0;
  uint32_t n = 0;  // pairs stored (lowMatchBound.size())
  uint32_t l = 0, r = a.tlen, lcp = 0;
  unsigned probes = 0;
  bool searching = true;
  bool prefixUnchecked = true;  // the reported length may reach over bases nobody compared (see the emission)
  if (a.useLookup) {  // SuffixArray.h:954-990; DNATuple::FromStringLR (tuples/DNATuple.h:24-52)
    uint64_t tuple = 0;
    bool ok = true;
    if (a.prefixLen >= 8 && a.prefixLen <= 16) {
      // the word's bases by two (overlapping) 8-byte loads inside the read instead of a byte load each: bytes [0, 8) and
      // [prefixLen - 8, prefixLen) of the word -- on the minus strand the same stretch of the forward read, walked backwards
      const int n = a.prefixLen;
      const int64_t at0 = s.off + (int64_t)(s.strand ? s.len - 1 - (p + (uint32_t)n - 1u) : p);
      const uint64_t w0 = *reinterpret_cast<const u64u *>(a.pool + at0), w1 = *reinterpret_cast<const u64u *>(a.pool + at0 + (n - 8));
      const uint32_t tab = s.strand ? 256u : 0u;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (i >= n) break;
        const int k = s.strand ? n - 1 - i : i;  // byte of the stretch that holds base i of the word
        const uint32_t raw = (uint32_t)((k < 8 ? w0 >> (8 * k) : w1 >> (8 * (k - (n - 8)))) & 0xffu);
        const int c = three[raw + tab];
        ok = ok && c < 4;
        tuple = (tuple << 2) + (uint64_t)(c & 3);
      }
    } else {
      for (int i = 0; i < a.prefixLen; i++) {
        const int c = strand_code(a.pool, s, p + (uint32_t)i, three);
        ok = ok && c < 4;
        tuple = (tuple << 2) + (uint64_t)(c & 3);
      }
    }
    if (ok) {
      if (a.ranges) {
        const uint2 lr = a.ranges[tuple];
        l = lr.x;
        r = lr.y;
      } else {
        l = a.startPos[tuple];
        r = a.endPos[tuple];
      }
      lcp = (uint32_t)a.prefixLen;
      prefixUnchecked = tuple == 0 || a.minPrefix > (uint32_t)a.prefixLen;
    }
    if (ok && l < r) {
      hl[0] = l;
      hr[0] = r;
      n = 1;
      if (a.exp & 4) searching = false;
      if (a.ext && !a.noWin && !(a.exp & 8) && r - l <= (uint32_t)kWinEntries) {  // the range's stretch of the companion array -> this lane's slot
        const uint32_t base = l & ~7u;
        const uint32_t chunks = min(min((a.tlen - base) >> 3, (r - base + 7) >> 3), (uint32_t)(kWinEntries / 8));
        const uint4 *src = reinterpret_cast<const uint4 *>(a.ext + base);
        uint32_t *dst = extWin + threadIdx.x * kWinStride;
        uint4 v[kWinEntries / 8];
#pragma unroll
        for (int c = 0; c < kWinEntries / 8; c++) v[c] = (uint32_t)c < chunks ? src[c] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int c = 0; c < kWinEntries / 8; c++) {
          dst[4 * c] = v[c].x;
          dst[4 * c + 1] = v[c].y;
          dst[4 * c + 2] = v[c].z;
          dst[4 * c + 3] = v[c].w;
        }
        win.base = base;
        win.count = chunks * 8;
        probes += (chunks * 16 + 63) / 64;  // sectors fetched
      }
    } else {
      lcp = 0;
      searching = false;
    }
  }
  while (searching && l < r && lcp < queryLength) {  // SuffixArray.h:996-1067
    if (a.stopUnique && l == r - 1) break;
    if (a.maxLcp && lcp >= (uint32_t)a.maxLcp) break;
    if ((a.exp & 2) && lcp >= (uint32_t)(a.prefixLen + kExtBases)) break;
    // the next genome base is an N (a genome as Blasr holds it ends in one, so the array never runs past it)
    if (suffix_symbol(a, three, win, l, lcp, probes) >= 4) break;
    const int tq = strand_code(a.pool, s, p + lcp, three);
    const bool plain = equal_range(a, three, win, lcp, tq, l, r, probes);
    if (l == r) break;
    // (:1044-1051: the suffix at l must hold the read's base; after the plain descent it does by construction)
    if (tq >= 4 || (!plain && suffix_symbol(a, three, win, l, lcp, probes) != tq)) break;
#pragma unroll
    for (int i = kHist - 1; i > 0; i--) {
      hl[i] = hl[i - 1];
      hr[i] = hr[i - 1];
    }
    hl[0] = l;
    hr[0] = r;
    n++;
    lcp++;
  }

  // LocateAnchorBoundsInSuffixArray's heuristics (MapBySuffixArray.h:93-204).  (Its loop at :104-109 never runs: a
  // pair is only stored with low < high.)
  uint32_t ml = 0, mh = 0, mlen = 0;
  uint32_t uniqueAt = 0xffffffffu, uniqueT = 0;  // the array position whose suffix the extension looked up, and that suffix
  if (n > 0) {
    ml = hl[0];
    mh = hr[0];
    mlen = a.minPrefix + n - 1;
    if (ml + 1 == mh) {  // unique: follow the genome as far as it agrees with the read
      uint32_t ext = mlen;
      uniqueAt = ml;
      uniqueT = a.index[ml];
      int64_t refPos = (int64_t)uniqueT + ext - 1;
      int64_t queryPos = (int64_t)p + ext - 1;
      bool extended = false, ended = false;
      // eight bases at a time while eight are left on both sides: one (unaligned) 8-byte load each instead of sixteen
      // dependent byte loads; the first difference and the first 'N' by bit tricks
      for (;;) {
        const int64_t gAt = refPos + 1, qAt = queryPos + 1;
        int64_t room = min((int64_t)a.tlen - gAt, (int64_t)s.len - qAt);
        if (a.maxLcp != 0) room = min(room, (int64_t)a.maxLcp - (int64_t)ext);
        if (room < 8 || (a.exp & 1)) break;
        const uint64_t gw = *reinterpret_cast<const u64u *>(a.genome + gAt);
        uint64_t qw;
        if (s.strand) {
          const uint64_t raw = *reinterpret_cast<const u64u *>(a.pool + s.off + ((int64_t)s.len - 1 - qAt) - 7);
          qw = 0;
#pragma unroll
          for (int j = 0; j < 8; j++) qw |= (uint64_t)rcChar[(uint8_t)(raw >> (8 * (7 - j)))] << (8 * j);
        } else {
          qw = *reinterpret_cast<const u64u *>(a.pool + s.off + qAt);
        }
        const uint64_t x = gw ^ qw, y = gw ^ 0x4E4E4E4E4E4E4E4Eull;  // 'N'
        const uint64_t z = (y - 0x0101010101010101ull) & ~y & 0x8080808080808080ull;  // lowest set bit: the first 'N'
        const int firstDiff = x ? (__builtin_ctzll(x) >> 3) : 8, firstN = z ? (__builtin_ctzll(z) >> 3) : 8;
        const int matched = min(firstDiff, firstN);
        refPos += matched;
        queryPos += matched;
        ext += (uint32_t)matched;
        extended = extended || matched > 0;
        if (matched < 8) {
          ended = true;
          break;
        }
      }
      while (!ended && refPos + 1 < (int64_t)a.tlen && queryPos + 1 < (int64_t)s.len) {
        const uint8_t gc = a.genome[refPos + 1];
        const uint8_t rawQ = a.pool[s.off + (int64_t)(s.strand ? s.len - 1 - (uint32_t)(queryPos + 1) : (uint32_t)(queryPos + 1))];
        if (gc == 'N' || gc != (s.strand ? rcChar[rawQ] : rawQ)) break;
        if (!(a.maxLcp == 0 || ext < (uint32_t)a.maxLcp)) break;
        refPos++;
        queryPos++;
        ext++;
        extended = true;
      }
      lcp = ext;
      if (extended) {
        mlen = ext;
      } else if (n > 1) {  // a unique match that ends in an error: one base back
        ml = hl[1];
        mh = hr[1];
        mlen = a.minPrefix + n - 2;
      }
    } else {  // repeated: `expand` bases back, at most to the first pair
      const uint32_t back = n > (uint32_t)a.expand ? (uint32_t)a.expand : n - 1;
#pragma unroll
      for (int i = 1; i < kHist; i++)
        if (back == (uint32_t)i) {
          ml = hl[i];
          mh = hr[i];
        }
      mlen = a.minPrefix + (n - back) - 1;
    }
  }
  // MapReadToGenome's emission test (:284-300); the trimming of a match that runs past the read is applied in place
  // there, so later anchors of the position see the trimmed length
  uint32_t count = 0;
  if (mh - ml <= (uint32_t)a.maxAnchors && mlen >= a.minMatch) {
    count = mh - ml;
    if (count > 0 && mlen + p > s.len) {
      mlen = s.len - p;
      if (mlen < a.minMatch) count = 1;
    }
  }
  // a single anchor whose suffix the extension has already looked up: the emission gets the genome position itself
  // (bit 30 of the length word) instead of the place in the index array to read it from -- a random sector per anchor less
  const bool known = count == 1 && ml == uniqueAt && !(a.exp & 16);
  a.lo[g] = known ? uniqueT : ml;
  a.cnt[g] = count;
  // The reported length is minPrefix + (levels) - 1 or the extension's end: with a lookup table whose prefix is at least
  // minPrefix every base under it was compared with the read's (so it is no N) -- except when the match started from the
  // table's all-A word, under which BuildLookupTable also files the suffixes whose first bases make no tuple (it ignores
  // FromStringLR's failure, SuffixArray.h:234-244).  The emission checks the reference's assert against the genome only
  // where that is not so (bit 31): one random read per anchor less.
  a.len[g] = mlen | (prefixUnchecked ? 0x80000000u : 0u) | (known ? 0x40000000u : 0u);
  if (a.lcp) a.lcp[g] = lcp;
  if (a.probes) atomicAdd(a.probes, (unsigned long long)probes);
}

// advanceExactMatches (MapBySuffixArray.h:206-212): the loop over p skips max(1, lcpLength - advance) further
// positions after each one it looks at; the skipped ones report no match.
__global__ __launch_bounds__(64) void anchor_advance_kernel(AnchorArgs a) {
  const int si = blockIdx.x * 64 + threadIdx.x;
  if (si >= a.nSeg) return;
  const int64_t begin = a.posOff[si], end = a.posOff[si + 1];
  int64_t g = begin;
  while (g < end) {
    const int step = (int)(a.lcp[g] - (uint32_t)a.advance);
    const int64_t next = g + (step > 1 ? step : 1) + 1;
    for (int64_t k = g + 1; k < next && k < end; k++) a.cnt[k] = 0;
    g = next;
  }
}

// first anchor of every segment of the piece (exclusive scan value at the segment's first position)
__global__ __launch_bounds__(256) void anchor_seg_offsets_kernel(int nSeg, const int64_t *posOff,
                                                                 const int64_t *anchorOff, int64_t *segAnchor,
                                                                 uint32_t *segAnchor32) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s > nSeg) return;
  const int64_t v = anchorOff[posOff[s]];
  segAnchor[s] = v;
  segAnchor32[s] = (uint32_t)v;
}

// One wavefront per 64 consecutive positions, one lane per anchor.  sorted: t as the key and (length << 32 | q) as the
// value of a STABLE sort -- a list's anchors leave this kernel in the order of their read positions, so anchors with the
// same t come out of the sort by q, which is SortMatchPosList's order (t, then q) with 32 key bits through the radix passes
// instead of 32 + the bits of q; else the records themselves.
// (Val = uint32_t where every read of the batch is shorter than 65536 bases: length and q in sixteen bits each, a third less
//  for the sort to move.)
template <typename Val>
__global__ __launch_bounds__(256) void anchor_emit_kernel(AnchorArgs a, const int64_t *anchorOff, uint32_t *keys,
                                                          Val *vals, blasr_hip_anchor *out, int *error) {
  __shared__ uint32_t sIncl[4][64], sLo[4][64], sLen[4][64], sQ[4][64];
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int64_t g0 = ((int64_t)blockIdx.x * 4 + w) * 64;
  if (g0 >= a.nPos) return;
  const int64_t g = g0 + lane;
  const uint32_t c = g < a.nPos ? a.cnt[g] : 0;
  const uint32_t incl = wave_prefix_sum(c);
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  if (total == 0) return;
  // the list of the wavefront's first position by one search on the scalar unit; a lane beyond that list's end steps on from
  // there (as in the search kernel: a search per lane was eighteen dependent loads at the head of every wavefront)
  uint32_t q = 0;
  {
    int si = segment_of(a.posOff, a.nSeg, g0);
    if (c) {
      while (si + 1 < a.nSeg && g >= a.posOff[si + 1]) si++;
      q = a.seg[si].start + (uint32_t)(g - a.posOff[si]);
    }
  }
  sIncl[w][lane] = incl;
  sLo[w][lane] = g < a.nPos ? a.lo[g] : 0;
  sLen[w][lane] = g < a.nPos ? a.len[g] : 0;
  sQ[w][lane] = q;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int64_t base = anchorOff[g0];
  for (uint32_t j = lane; j < total; j += 64) {
    int lo = 0, hi = 63;  // the first k with sIncl[k] > j
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (sIncl[w][mid] > j)
        hi = mid;
      else
        lo = mid + 1;
    }
    const uint32_t excl = lo ? sIncl[w][lo - 1] : 0;
    const uint32_t lenWord = sLen[w][lo], len = lenWord & 0x3fffffffu;
    const uint32_t t = (lenWord & 0x40000000u) ? sLo[w][lo] : a.index[sLo[w][lo] + (j - excl)];
    // the reference's asserts (MapBySuffixArray.h:301-302); the base under the match's end is one the search compared
    // with the read's (so no N) unless the match started from the all-A word of the lookup table (anchor_search_kernel)
    if ((uint64_t)t + len > a.tlen || len == 0 || ((lenWord >> 31) != 0 && a.genome[(uint64_t)t + len - 1] == 'N')) atomicOr(error, 1);
    if (keys) {
      keys[base + j] = t;
      vals[base + j] = sizeof(Val) == 8 ? (Val)(((uint64_t)len << 32) | sQ[w][lo]) : (Val)((len << 16) | sQ[w][lo]);
    } else {
      blasr_hip_anchor rec;
      rec.t = t;
      rec.q = sQ[w][lo];
      rec.l = len;
      out[base + j] = rec;
    }
  }
}

template <typename Val>
__global__ __launch_bounds__(256) void anchor_pack_kernel(int64_t n, const uint32_t *keys, const Val *vals,
                                                          blasr_hip_anchor *out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t v = vals[i];
  blasr_hip_anchor rec;
  rec.t = keys[i];
  rec.q = sizeof(Val) == 8 ? (uint32_t)v : (uint32_t)(v & 0xffffu);
  rec.l = sizeof(Val) == 8 ? (uint32_t)(v >> 32) : (uint32_t)(v >> 16);
  out[i] = rec;
}

struct Events {
  hipEvent_t e[8] = {};
  ~Events() {
    for (auto &x : e)
      if (x) (void)hipEventDestroy(x);
  }
};

void launch_search(const AnchorArgs &a, hipStream_t st) {
  const unsigned blocks = (unsigned)((a.nPos + 255) / 256);
  const int hist = std::max(a.expand, 1) + 1;
  if (hist <= 2)
    hipLaunchKernelGGL(anchor_search_kernel<2>, dim3(blocks), dim3(256), 0, st, a);
  else if (hist <= 4)
    hipLaunchKernelGGL(anchor_search_kernel<4>, dim3(blocks), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(anchor_search_kernel<8>, dim3(blocks), dim3(256), 0, st, a);
}

constexpr int64_t kPiecePositions = 192ll << 20;  // read positions searched per piece (24 bytes of arrays each): fewer, larger
                                                   // pieces = fewer launch tails and host round trips (48 M: +18 ms per 100 k reads)
constexpr int64_t kSortLimit = 1ll << 30;        // anchors per segmented sort call (rocprim counts in 32 bits)

}  // namespace

namespace {
__global__ __launch_bounds__(256) void anchor_ranges_kernel(const uint32_t *startPos, const uint32_t *endPos, int64_t n, uint2 *ranges) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) ranges[i] = make_uint2(startPos[i], endPos[i]);
}
}  // namespace

void blasr_hip_drop_index(blasr_hip_ctx *ctx) {
  if (ctx->saOwned) {
    if (ctx->saRanges) (void)hipFree(ctx->saRanges);
    if (ctx->saIndex) (void)hipFree(ctx->saIndex);
    if (ctx->saStartPos) (void)hipFree(ctx->saStartPos);
    if (ctx->saEndPos) (void)hipFree(ctx->saEndPos);
    if (ctx->saExt) (void)hipFree(ctx->saExt);
  }
  ctx->saIndex = ctx->saStartPos = ctx->saEndPos = nullptr;
  ctx->saExt = nullptr;
  ctx->saRanges = nullptr;
  ctx->saExtGenome = ctx->saExtIndex = nullptr;
  ctx->saOwned = true;
  ctx->saBorrowedFrom = nullptr;
  ctx->saBorrowedEpoch = 0;
  ctx->indexEpoch++;
  ctx->saLen = ctx->saTableLen = 0;
  ctx->saPrefixLen = 0;
}

// A borrowed index is only good while its owner still holds what was borrowed (the owner outlives the borrower: blasr_hip.h).
bool blasr_hip_borrowed_index_is_current(const blasr_hip_ctx *ctx) {
  return ctx->saOwned || (ctx->saBorrowedFrom && ctx->saBorrowedFrom->indexEpoch == ctx->saBorrowedEpoch);
}

int blasr_hip_ensure_index_ext(blasr_hip_ctx *ctx, hipStream_t st) {
  if (ctx->saExt && ctx->saExtGenome == ctx->genome && ctx->saExtIndex == ctx->saIndex) return BLASR_HIP_OK;
  if (!ctx->saOwned)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "anchor_reads: the borrowed index was not made for this context's genome");
  if (ctx->saExt) (void)hipFree(ctx->saExt);
  ctx->saExt = nullptr;
  if (ctx->saRanges) (void)hipFree(ctx->saRanges);
  ctx->saRanges = nullptr;
  ctx->indexEpoch++;  // (the companion arrays a borrower holds are gone)
  if (ctx->saTableLen > 0) {  // the lookup table's two arrays as one of pairs (the search reads both ends of a word's range)
    HIP_TRY(ctx, hipMalloc(&ctx->saRanges, sizeof(uint2) * (size_t)ctx->saTableLen));
    hipLaunchKernelGGL(anchor_ranges_kernel, dim3((unsigned)((ctx->saTableLen + 255) / 256)), dim3(256), 0, st, ctx->saStartPos,
                       ctx->saEndPos, ctx->saTableLen, ctx->saRanges);
  }
  HIP_TRY(ctx, hipMalloc(&ctx->saExt, sizeof(uint16_t) * (size_t)ctx->saLen));
  hipLaunchKernelGGL(anchor_ext_kernel, dim3((unsigned)((ctx->saLen + 255) / 256)), dim3(256), 0, st, ctx->genome,
                     (uint32_t)ctx->genomeLen, ctx->saIndex, ctx->saLen, ctx->saPrefixLen, ctx->saExt);
  ctx->saExtGenome = ctx->genome;
  ctx->saExtIndex = ctx->saIndex;
  return BLASR_HIP_OK;
}

extern "C" {

int blasr_hip_borrow_index(blasr_hip_ctx *ctx, blasr_hip_ctx *from) try {
  if (!ctx || !from || ctx == from) return ctx ? blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "borrow_index: bad arguments") : BLASR_HIP_ERR_INVALID;
  if (ctx->device != from->device)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "borrow_index: the contexts are on devices %d and %d", ctx->device, from->device);
  if (!from->genome || !from->saIndex || from->saLen != from->genomeLen)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "borrow_index: the other context holds no genome with its suffix array");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!getenv("BLASR_HIP_ANCHOR_NO_EXT")) {  // made by the owner, so that both search the same array
    const int rc = blasr_hip_ensure_index_ext(from, from->stream);
    if (rc != BLASR_HIP_OK) return blasr_hip_fail(ctx, rc, "borrow_index: %s", from->lastError.c_str());
    HIP_TRY(ctx, hipStreamSynchronize(from->stream));
  }
  std::vector<uint32_t> starts;
  if (from->contigStarts && from->nContigs > 0) {
    starts.resize((size_t)from->nContigs + 1);
    HIP_TRY(ctx, hipMemcpy(starts.data(), from->contigStarts, sizeof(uint32_t) * starts.size(), hipMemcpyDeviceToHost));
  }
  const int rc = blasr_hip_adopt_genome_device(ctx, from->genome, from->genomeLen, starts.empty() ? nullptr : starts.data(),
                                               starts.empty() ? 0 : from->nContigs);
  if (rc != BLASR_HIP_OK) return rc;
  blasr_hip_drop_index(ctx);
  ctx->saIndex = from->saIndex;
  ctx->saStartPos = from->saStartPos;
  ctx->saEndPos = from->saEndPos;
  ctx->saExt = from->saExt;
  ctx->saRanges = from->saRanges;
  ctx->saExtGenome = from->saExt ? ctx->genome : nullptr;
  ctx->saExtIndex = from->saExt ? ctx->saIndex : nullptr;
  ctx->saLen = from->saLen;
  ctx->saTableLen = from->saTableLen;
  ctx->saPrefixLen = from->saPrefixLen;
  ctx->saOwned = false;
  ctx->saBorrowedFrom = from;
  ctx->saBorrowedEpoch = from->indexEpoch;
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CTX(ctx)

int blasr_hip_anchor_default_params(blasr_hip_anchor_params *p) try {
  if (!p) return BLASR_HIP_ERR_INVALID;
  p->min_match_length = 14;
  p->min_prefix_match_length = 8;
  p->expand = 0;
  p->max_anchors_per_position = 1000;
  p->advance_exact_matches = 0;
  p->max_lcp_length = 0;
  p->use_lookup_table = 1;
  p->stop_mapping_once_unique = 1;
  p->forward_only = 0;
  p->sorted = 1;
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_upload_suffix_array(blasr_hip_ctx *ctx, const uint32_t *index, int64_t length,
                                  const uint32_t *start_pos_table, const uint32_t *end_pos_table,
                                  int64_t lookup_table_length, int32_t lookup_prefix_length) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (!index || length <= 0 || length >= (1ll << 32) || lookup_table_length < 0 || lookup_prefix_length < 0 ||
      lookup_prefix_length > 16 ||
      (lookup_table_length > 0 &&
       (!start_pos_table || !end_pos_table || lookup_table_length != (1ll << (2 * lookup_prefix_length)))))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "upload_suffix_array: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  blasr_hip_drop_index(ctx);
  HIP_TRY(ctx, hipMalloc(&ctx->saIndex, sizeof(uint32_t) * (size_t)length));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->saIndex, index, sizeof(uint32_t) * (size_t)length, hipMemcpyHostToDevice,
                              ctx->stream));
  if (lookup_table_length > 0) {
    HIP_TRY(ctx, hipMalloc(&ctx->saStartPos, sizeof(uint32_t) * (size_t)lookup_table_length));
    HIP_TRY(ctx, hipMalloc(&ctx->saEndPos, sizeof(uint32_t) * (size_t)lookup_table_length));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->saStartPos, start_pos_table, sizeof(uint32_t) * (size_t)lookup_table_length,
                                hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->saEndPos, end_pos_table, sizeof(uint32_t) * (size_t)lookup_table_length,
                                hipMemcpyHostToDevice, ctx->stream));
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->saLen = length;
  ctx->saTableLen = lookup_table_length;
  ctx->saPrefixLen = lookup_prefix_length;
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CTX(ctx)

int blasr_hip_anchor_last_timing(const blasr_hip_ctx *ctx, float ms[4], int64_t counters[4]) try {
  if (!ctx || !ms || !counters) return BLASR_HIP_ERR_INVALID;
  for (int i = 0; i < 4; i++) {
    ms[i] = ctx->anchorMs[i];
    counters[i] = ctx->anchorCounters[i];
  }
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_anchor_reads_device(blasr_hip_ctx *ctx, int32_t n_reads, const blasr_hip_read *reads,
                                  const uint8_t *read_pool, const blasr_hip_anchor_params *params,
                                  blasr_hip_anchor *anchors, int64_t capacity, int64_t *n_anchors,
                                  int64_t *list_off) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (n_reads < 0 || !params || !n_anchors || !list_off || capacity < 0 || (capacity > 0 && !anchors) ||
      (n_reads > 0 && (!reads || !read_pool)))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "anchor_reads: bad arguments");
  if (!ctx->genome || !ctx->saIndex)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "anchor_reads: no genome / suffix array uploaded");
  if (ctx->saLen != ctx->genomeLen)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "anchor_reads: the suffix array lists %lld suffixes, the genome has %lld bases",
                          (long long)ctx->saLen, (long long)ctx->genomeLen);
  const blasr_hip_anchor_params &P = *params;
  const bool useLookup = P.use_lookup_table != 0 && ctx->saTableLen > 0;
  const uint32_t trim = std::max(P.min_match_length + 1, (uint32_t)ctx->saPrefixLen + 1);  // MapBySuffixArray.h:270
  if (P.expand < 0 || P.expand > 7 || P.max_anchors_per_position < 0 || P.advance_exact_matches < 0 ||
      P.max_lcp_length < 0)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "anchor_reads: expand outside 0..7, or a negative parameter");
  // positions past the searched ones would be read by the emission loop (the reference's assert at :283)
  if (P.min_prefix_match_length > trim + 1)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "anchor_reads: min_prefix_match_length %u beyond the mapped part of a read",
                          P.min_prefix_match_length);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const auto wall0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 4; i++) {
    ctx->anchorMs[i] = 0;
    ctx->anchorCounters[i] = 0;
  }
  const int strands = P.forward_only ? 1 : 2;
  const int64_t nLists = 2 * (int64_t)n_reads;
  for (int64_t i = 0; i <= nLists; i++) list_off[i] = 0;
  *n_anchors = 0;
  if (n_reads == 0) return BLASR_HIP_OK;

  // every list's searched positions: [subreadStart, min(matchEnd, endOfMapping)) -- the emission loop (:276) never
  // looks at a later one, and a later one never changes an earlier one
  std::vector<AnchorSeg> segs((size_t)nLists);
  std::vector<int64_t> segPos((size_t)nLists, 0);
  uint32_t longestRead = 1;
  for (int32_t r = 0; r < n_reads; r++) {
    const blasr_hip_read &rd = reads[r];
    longestRead = std::max(longestRead, rd.length);
    if (rd.subread_start > rd.subread_end || rd.subread_end > rd.length || rd.off < 0)
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "anchor_reads: read %d has a bad subread", r);
    for (int sd = 0; sd < 2; sd++) {
      AnchorSeg s;
      s.off = rd.off;
      s.len = rd.length;
      s.strand = (uint32_t)sd;
      const uint32_t s0 = sd ? rd.length - rd.subread_end : rd.subread_start;
      const uint32_t s1 = sd ? rd.length - rd.subread_start : rd.subread_end;
      s.start = s0;
      s.matchEnd = s1 + 1 - P.min_prefix_match_length;
      int64_t np = 0;
      const uint32_t span = s1 - s0;
      if (sd < strands && span >= P.min_match_length &&
          !(P.min_prefix_match_length > 0 && span < P.min_prefix_match_length)) {
        const int64_t endOfMapping = s1 < trim ? 0 : (int64_t)s1 - trim;
        const int64_t matchEnd = (int64_t)s1 - P.min_prefix_match_length + 1;
        np = std::max<int64_t>(0, std::min(matchEnd, endOfMapping) - (int64_t)s0);
      }
      segs[(size_t)(2 * r + sd)] = s;
      segPos[(size_t)(2 * r + sd)] = np;
    }
  }

  if (!blasr_hip_borrowed_index_is_current(ctx))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID,
                          "anchor_reads: the context the index was borrowed from has replaced it since (borrow again)");
  // the companion array of the index, made once per (genome, index) pair
  const bool useExt = !getenv("BLASR_HIP_ANCHOR_NO_EXT");
  if (useExt) {
    const int rc = blasr_hip_ensure_index_ext(ctx, st);
    if (rc != BLASR_HIP_OK) return rc;
  }
  {  // pinned results: one int64 per list boundary of the largest piece (never more than all lists)
    const size_t want = sizeof(int64_t) * ((size_t)nLists + 2);
    if (ctx->anchorPinnedBytes < want) {
      if (ctx->anchorPinned) (void)hipHostFree(ctx->anchorPinned);
      ctx->anchorPinned = nullptr;
      ctx->anchorPinnedBytes = 0;
      HIP_TRY(ctx, hipHostMalloc((void **)&ctx->anchorPinned, want, hipHostMallocDefault));
      ctx->anchorPinnedBytes = want;
    }
    if (!ctx->anchorFlags) HIP_TRY(ctx, hipMalloc((void **)&ctx->anchorFlags, 16));
    HIP_TRY(ctx, hipMemsetAsync(ctx->anchorFlags, 0, 16, st));
  }
  unsigned long long *dProbes = ctx->anchorFlags;
  int *dError = (int *)(ctx->anchorFlags + 1);
  struct PieceEvents {
    hipEvent_t e[4];
    bool emitted;
  };
  std::vector<PieceEvents> pieceEvents;
  struct EventGuard {
    std::vector<PieceEvents> &v;
    ~EventGuard() {
      for (auto &p : v)
        for (auto &x : p.e)
          if (x) (void)hipEventDestroy(x);
    }
  } eventGuard{pieceEvents};
  const bool countProbes = getenv("BLASR_HIP_ANCHOR_COUNT_PROBES") != nullptr;
  const bool sorted = P.sorted != 0;
  bool overflow = false;
  int64_t emitted = 0;  // anchors of the pieces so far
  int64_t first = 0;    // first list of the piece
  int64_t pieceLimit = std::min<int64_t>(kPiecePositions, std::max<int64_t>(blasr_hip_workspace_reach(ctx) / 64, 1 << 20));  // (24 B of arrays + the sort's buffers)
  if (const char *e = getenv("BLASR_HIP_ANCHOR_PIECE")) pieceLimit = std::max<int64_t>(1, atoll(e));
  std::vector<int64_t> posOff;
  while (first < nLists) {
    int64_t last = first, nPos = 0;
    while (last < nLists && (last == first || nPos + segPos[(size_t)last] <= pieceLimit)) nPos += segPos[(size_t)last++];
    const int nSeg = (int)(last - first);
    posOff.assign((size_t)nSeg + 1, 0);
    for (int i = 0; i < nSeg; i++) posOff[(size_t)i + 1] = posOff[(size_t)i] + segPos[(size_t)(first + i)];
    if (nPos == 0) {
      for (int64_t i = first; i < last; i++) list_off[i] = emitted;
      first = last;
      continue;
    }
    // workspace: segments, position offsets, per-position arrays, scan output, scan scratch, counters
    size_t scanBytes = 0;
    {
      uint32_t *nullU32 = nullptr;
      int64_t *nullI64 = nullptr;
      HIP_TRY(ctx, rocprim::exclusive_scan(nullptr, scanBytes, nullU32, nullI64, (int64_t)0, (size_t)nPos + 1,
                                           rocprim::plus<int64_t>(), st));
    }
    const size_t need = sizeof(AnchorSeg) * (size_t)nSeg + (sizeof(int64_t) * 2 + 4) * ((size_t)nSeg + 1) +
                        (size_t)(nPos + 1) * (4 * 4 + 8) + scanBytes + 4096;
    if (ctx->workspaceBytes < need) {
      const int rc = blasr_hip_grow_workspace(ctx, need);
      if (rc != BLASR_HIP_OK) return rc;
    }
    char *w = ctx->workspace;
    auto take = [&](size_t bytes) {
      char *p = w;
      w += (bytes + 255) & ~(size_t)255;
      return p;
    };
    // (the 256-byte rounding of each of the ten arrays is inside the 4096 spare bytes)
    AnchorSeg *dSeg = (AnchorSeg *)take(sizeof(AnchorSeg) * (size_t)nSeg);
    int64_t *dPosOff = (int64_t *)take(sizeof(int64_t) * ((size_t)nSeg + 1));
    int64_t *dSegAnchor = (int64_t *)take(sizeof(int64_t) * ((size_t)nSeg + 1));
    uint32_t *dSegAnchor32 = (uint32_t *)take(4 * ((size_t)nSeg + 1));
    uint32_t *dLo = (uint32_t *)take(4 * (size_t)(nPos + 1));
    uint32_t *dCnt = (uint32_t *)take(4 * (size_t)(nPos + 1));
    uint32_t *dLen = (uint32_t *)take(4 * (size_t)(nPos + 1));
    uint32_t *dLcp = (uint32_t *)take(4 * (size_t)(nPos + 1));
    int64_t *dAnchorOff = (int64_t *)take(8 * (size_t)(nPos + 1));
    void *scanTmp = take(scanBytes);
    HIP_TRY(ctx, hipMemcpyAsync(dSeg, segs.data() + first, sizeof(AnchorSeg) * (size_t)nSeg, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(dPosOff, posOff.data(), sizeof(int64_t) * ((size_t)nSeg + 1), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(dCnt + nPos, 0, 4, st));  // the scan's extra element

    AnchorArgs a;
    a.genome = ctx->genome;
    a.tlen = (uint32_t)ctx->genomeLen;
    a.index = ctx->saIndex;
    a.startPos = ctx->saStartPos;
    a.endPos = ctx->saEndPos;
    a.ranges = useLookup && !getenv("BLASR_HIP_ANCHOR_TWO_TABLES") ? ctx->saRanges : nullptr;
    a.ext = useExt ? ctx->saExt : nullptr;
    a.noWin = getenv("BLASR_HIP_ANCHOR_NO_WIN") != nullptr;
    a.exp = getenv("BLASR_HIP_ANCHOR_EXP") ? atoi(getenv("BLASR_HIP_ANCHOR_EXP")) : 0;
    a.prefixLen = ctx->saPrefixLen;
    a.pool = read_pool;
    a.seg = dSeg;
    a.posOff = dPosOff;
    a.nSeg = nSeg;
    a.nPos = nPos;
    a.minMatch = P.min_match_length;
    a.minPrefix = P.min_prefix_match_length;
    a.expand = P.expand;
    a.maxAnchors = P.max_anchors_per_position;
    a.advance = P.advance_exact_matches;
    a.maxLcp = P.max_lcp_length;
    a.useLookup = useLookup ? 1 : 0;
    a.stopUnique = P.stop_mapping_once_unique != 0;
    a.lo = dLo;
    a.cnt = dCnt;
    a.len = dLen;
    a.lcp = P.advance_exact_matches ? dLcp : nullptr;
    a.probes = countProbes ? dProbes : nullptr;

    pieceEvents.push_back(PieceEvents{{nullptr, nullptr, nullptr, nullptr}, false});
    PieceEvents &ev = pieceEvents.back();
    for (auto &e : ev.e) HIP_TRY(ctx, hipEventCreate(&e));
    HIP_TRY(ctx, hipEventRecord(ev.e[0], st));
    launch_search(a, st);
    if (P.advance_exact_matches)
      hipLaunchKernelGGL(anchor_advance_kernel, dim3((unsigned)((nSeg + 63) / 64)), dim3(64), 0, st, a);
    HIP_TRY(ctx, hipEventRecord(ev.e[1], st));
    HIP_TRY(ctx, rocprim::exclusive_scan(scanTmp, scanBytes, dCnt, dAnchorOff, (int64_t)0, (size_t)nPos + 1,
                                         rocprim::plus<int64_t>(), st));
    (void)dSegAnchor;
    hipLaunchKernelGGL(anchor_seg_offsets_kernel, dim3((unsigned)((nSeg + 256) / 256)), dim3(256), 0, st, nSeg, dPosOff,
                       dAnchorOff, ctx->anchorPinned, dSegAnchor32);
    HIP_TRY(ctx, hipStreamSynchronize(st));  // the one wait of the piece: how many anchors it has
    const int64_t *segAnchor = ctx->anchorPinned;
    const int64_t pieceAnchors = segAnchor[(size_t)nSeg];
    for (int i = 0; i < nSeg; i++) list_off[first + i] = emitted + segAnchor[(size_t)i];
    ctx->anchorCounters[0] += nPos;
    ctx->anchorCounters[2] += 1;
    if (emitted + pieceAnchors > capacity) overflow = true;
    if (!overflow && pieceAnchors > 0) {
      blasr_hip_anchor *out = anchors + emitted;
      uint32_t *keys = nullptr, *keysAlt = nullptr;
      uint64_t *vals = nullptr, *valsAlt = nullptr;
      if (sorted) {
        if (pieceAnchors > kSortLimit)
          return blasr_hip_fail(ctx, BLASR_HIP_ERR_NOMEM, "anchor_reads: %lld anchors in one piece of %d lists; lower BLASR_HIP_ANCHOR_PIECE",
                                (long long)pieceAnchors, nSeg);
        for (int b = 0; b < 4; b++) {
          const int rc = blasr_hip_ensure(ctx, ctx->anchorBuf[b], (size_t)pieceAnchors * (b < 2 ? 8 : 4));
          if (rc != BLASR_HIP_OK) return rc;
        }
        vals = (uint64_t *)ctx->anchorBuf[0].p;
        valsAlt = (uint64_t *)ctx->anchorBuf[1].p;
        keys = (uint32_t *)ctx->anchorBuf[2].p;
        keysAlt = (uint32_t *)ctx->anchorBuf[3].p;
      }
      const bool narrowVals = sorted && longestRead < 65536u && !getenv("BLASR_HIP_ANCHOR_WIDE_VALUES");
      if (narrowVals)
        hipLaunchKernelGGL(anchor_emit_kernel<uint32_t>, dim3((unsigned)((nPos + 255) / 256)), dim3(256), 0, st, a, dAnchorOff, keys,
                           (uint32_t *)vals, out, dError);
      else
        hipLaunchKernelGGL(anchor_emit_kernel<uint64_t>, dim3((unsigned)((nPos + 255) / 256)), dim3(256), 0, st, a, dAnchorOff, keys,
                           vals, out, dError);
      HIP_TRY(ctx, hipEventRecord(ev.e[2], st));
      if (sorted) {
        int tBits = 1;
        while (tBits < 32 && (a.tlen >> tBits) != 0) tBits++;
        const unsigned sortBits = (unsigned)tBits;  // (t < tlen)
        // the position arrays are dead once the anchors are out: the sort's scratch goes over them
        void *sortTmp = dLo;
        const size_t sortRoom = (size_t)((char *)ctx->workspace + ctx->workspaceBytes - (char *)dLo);
        auto sort_and_pack = [&](auto *v0, auto *v1) -> int {
          typedef typename std::remove_pointer<decltype(v0)>::type Val;
          rocprim::double_buffer<uint32_t> kb(keys, keysAlt);
          rocprim::double_buffer<Val> vb(v0, v1);
          size_t sortBytes = 0;
          HIP_TRY(ctx, rocprim::segmented_radix_sort_pairs(nullptr, sortBytes, kb, vb, (unsigned)pieceAnchors, (unsigned)nSeg,
                                                           dSegAnchor32, dSegAnchor32 + 1, 0, sortBits, st));
          if (sortRoom < sortBytes) return blasr_hip_fail(ctx, BLASR_HIP_ERR_NOMEM, "anchor_reads: sort scratch of %zu bytes", sortBytes);
          HIP_TRY(ctx, rocprim::segmented_radix_sort_pairs(sortTmp, sortBytes, kb, vb, (unsigned)pieceAnchors, (unsigned)nSeg,
                                                           dSegAnchor32, dSegAnchor32 + 1, 0, sortBits, st));
          hipLaunchKernelGGL(anchor_pack_kernel<Val>, dim3((unsigned)((pieceAnchors + 255) / 256)), dim3(256), 0, st, pieceAnchors,
                             kb.current(), vb.current(), out);
          return BLASR_HIP_OK;
        };
        const int rcS = narrowVals ? sort_and_pack((uint32_t *)vals, (uint32_t *)valsAlt) : sort_and_pack(vals, valsAlt);
        if (rcS != BLASR_HIP_OK) return rcS;
      }
      HIP_TRY(ctx, hipEventRecord(ev.e[3], st));
      ev.emitted = true;
    }
    emitted += pieceAnchors;
    first = last;
  }
  // ---- once per call: the assert flag, the probe count, the device times of the pieces
  {
    unsigned long long flags[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(flags, ctx->anchorFlags, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    for (auto &pe : pieceEvents) {
      float ms = 0;
      HIP_TRY(ctx, hipEventElapsedTime(&ms, pe.e[0], pe.e[1]));
      ctx->anchorMs[0] += ms;
      if (pe.emitted) {
        HIP_TRY(ctx, hipEventElapsedTime(&ms, pe.e[1], pe.e[2]));
        ctx->anchorMs[1] += ms;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, pe.e[2], pe.e[3]));
        ctx->anchorMs[2] += ms;
      }
    }
    ctx->anchorCounters[3] = (int64_t)flags[0];
    if ((int)(flags[1] & 0xffffffffu))
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "anchor_reads: an anchor runs past the genome or ends on an N (the reference asserts, MapBySuffixArray.h:301-302)");
  }
  list_off[nLists] = emitted;
  *n_anchors = emitted;
  ctx->anchorCounters[1] = emitted;
  ctx->anchorMs[3] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  if (overflow)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_CAPACITY, "anchor_reads: %lld anchors, capacity %lld", (long long)emitted,
                          (long long)capacity);
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CTX(ctx)

int blasr_hip_anchor_reads_host(blasr_hip_ctx *ctx, int32_t n_reads, const blasr_hip_read *reads,
                                const uint8_t *read_pool, int64_t read_pool_bytes,
                                const blasr_hip_anchor_params *params, blasr_hip_anchor *anchors, int64_t capacity,
                                int64_t *n_anchors, int64_t *list_off) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (read_pool_bytes < 0 || (read_pool_bytes > 0 && !read_pool) || capacity < 0 || (capacity > 0 && !anchors))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "anchor_reads_host: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  uint8_t *dPool = nullptr;
  blasr_hip_anchor *dAnchors = nullptr;
  struct Free {
    void *&p;
    ~Free() {
      if (p) (void)hipFree(p);
    }
  };
  void *p0 = nullptr, *p1 = nullptr;
  Free f0{p0}, f1{p1};
  HIP_TRY(ctx, hipMalloc(&p0, (size_t)std::max<int64_t>(read_pool_bytes, 1)));
  HIP_TRY(ctx, hipMalloc(&p1, sizeof(blasr_hip_anchor) * (size_t)std::max<int64_t>(capacity, 1)));
  dPool = (uint8_t *)p0;
  dAnchors = (blasr_hip_anchor *)p1;
  if (read_pool_bytes > 0) HIP_TRY(ctx, hipMemcpy(dPool, read_pool, (size_t)read_pool_bytes, hipMemcpyHostToDevice));
  for (int32_t r = 0; r < n_reads && reads; r++)
    if (reads[r].off < 0 || reads[r].off + (int64_t)reads[r].length > read_pool_bytes)
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "anchor_reads_host: read %d outside the pool", r);
  const int rc = blasr_hip_anchor_reads_device(ctx, n_reads, reads, dPool, params, dAnchors, capacity, n_anchors, list_off);
  if (rc != BLASR_HIP_OK) return rc;
  if (*n_anchors > 0)
    HIP_TRY(ctx, hipMemcpy(anchors, dAnchors, sizeof(blasr_hip_anchor) * (size_t)*n_anchors, hipMemcpyDeviceToHost));
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CTX(ctx)

}  // extern "C"
