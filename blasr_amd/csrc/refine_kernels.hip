// refine_kernels.hip -- gfx950 kernels for BLASR's guided affine refinement.
//
// What the reference does per candidate (one CPU thread, one call):
//   AlignmentToGuide + StoreMatrixOffsets   GuidedAlign.h:94-259
//   AffineGuidedAlign fill + traceback      AffineGuidedAlign.h:31-488
//   ArrowPathToAlignment                    datastructures/alignment/Alignment.h:190-254
//   RemoveAlignmentPrefixGaps               AlignmentUtils.h:620-644
//   ComputeAlignmentStats (affine form)     AlignmentUtils.h:535-584, :60-124
//
// How it is laid out here (MI355X, wave64):
//   geometry_kernel  one wavefront per candidate: rows of the ragged band are
//                    derived 64 at a time with lane = row (block lookup by
//                    ballot + mbcnt, band start by a wave prefix-max, matrix
//                    offsets by a wave prefix-sum); emits nCells / widest row.
//   fillp_kernel     FOUR candidates per wavefront (one 16-lane DPP row each, three cells per lane) for the
//                    candidates whose rows all fit 112 cells; the row above in a per-candidate LDS ring.
//   fill_kernel      one wavefront per candidate, lane = band column, for the others.  In both a row is
//                    one pass: the vertical/diagonal candidates are elementwise,
//                    the horizontal (deletion) dependency is closed with two
//                    DPP prefix-min scans, so a whole row costs O(1) wave
//                    instructions instead of a serial sweep.  Only the 1-byte packed arrows (3 bits S,
//                    1 bit I, 1 bit D) and an 8-byte row record go to HBM.
//   traceback_kernel one lane per candidate: walks the packed arrows backwards, eight rows per batch of
//                    prefetched 16-byte windows parked in LDS, one arrow per lane and loop trip.
//   stats_kernel     one wavefront per candidate, lane = op: ComputeAlignmentStats.
//   scan / compact   exclusive scan of op counts and a coalesced copy of the
//                    per-candidate op runs into the caller's arena.
// Integer DP: no MFMA anywhere.  Results are bit-identical to the reference.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "refine_kernels.h"
#include <type_traits>

#include "wave_utils.h"

namespace blasr_hip {

// ---------------------------------------------------------------------------
// Guide geometry (AlignmentToGuide, GuidedAlign.h:104-259) for 64 rows at once.
// Row r (1..Q) belongs to query position qStart + r - 1.  Per row:
//   t      centre column (guide[r].t)
//   lo     first column  = t - tPre;  lo_r = max(t_r - cap_r, lo_{r-1})  because
//          tPre = min(cap, t_r - lo_{r-1})   (:160-173, :216, :245)
//   hi     last column   = t + tPost
// cap is "none" on the first row of a block (:161-165), bandSize inside a block
// (:173), MAX_BAND_SIZE on gap rows (:216, :245).
// ---------------------------------------------------------------------------
struct GuideCursor {
  const int32_t *blocks;  // (q,t,len) triplets
  int nBlocks;
  int qStart, tStart, Q, band;
  int nextBlock;  // first block whose first row is >= the next group's first row
  int loCarry;    // lo of the row preceding the next group
  unsigned cellCarry;  // matrix cells in all rows before the next group
};

__device__ __forceinline__ void guide_begin(GuideCursor &g, const int32_t *blocks, int nBlocks, int band) {
  g.blocks = blocks;
  g.nBlocks = nBlocks;
  g.band = band;
  g.qStart = blocks[0];
  g.tStart = blocks[1];
  g.Q = blocks[3 * (nBlocks - 1)] + blocks[3 * (nBlocks - 1) + 2] - g.qStart;
  g.nextBlock = 0;
  g.loCarry = g.tStart - 1;  // row 0: t = tStart-1, tPre = 0 (:126-135)
  int drift = abs(g.tStart - g.qStart);
  int tPost0 = drift > band ? drift : band;
  g.cellCarry = (unsigned)(tPost0 + 1);
}
__device__ __forceinline__ int guide_row0_tpost(const GuideCursor &g) {
  int drift = abs(g.tStart - g.qStart);
  return drift > g.band ? drift : g.band;
}

// Rows r0 .. r0+63 (lane i <-> row r0+i).  flagScratch: 64 bytes of LDS private to the wave.
__device__ __forceinline__ void guide_group(GuideCursor &g, int r0, unsigned char *flagScratch, int &tOut,
                                            int &loOut, int &hiOut, unsigned &rowStartOut) {
  const int lane = lane_id();
  const int r = r0 + lane;
  const bool rowValid = r <= g.Q;
  // which blocks start inside this group?  (a block of length >= 1 owns >= 1 row, so <= 64 of them)
  flagScratch[lane] = 0;
  __builtin_amdgcn_wave_barrier();
  int bCand = g.nextBlock + lane;
  int rb = INT32_MAX;
  if (bCand < g.nBlocks) rb = g.blocks[3 * bCand] - g.qStart + 1;
  bool starts = rb < r0 + 64;
  if (starts) flagScratch[rb - r0] = 1;
  __builtin_amdgcn_wave_barrier();
  unsigned long long startMask = __ballot(flagScratch[lane] != 0);
  __builtin_amdgcn_wave_barrier();
  // number of block starts at rows <= mine -> my block index
  unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
  int bi = g.nextBlock - 1 + __popcll(startMask & le);
  g.nextBlock += __popcll(__ballot(starts));

  int t = 0, cap = 0, tPost = 0;
  if (rowValid) {
    const int qpos = g.qStart + r - 1;
    const int bq = g.blocks[3 * bi], bt = g.blocks[3 * bi + 1], bl = g.blocks[3 * bi + 2];
    const int off = qpos - bq;
    if (off < bl) {
      t = bt + off;
      if (off == 0) {
        // first row of a block: tPre = reach (uncapped), tPost = band + |drift of the gap before| (:161-165)
        int drift;
        if (bi == 0) {
          drift = abs(g.tStart - g.qStart);
        } else {
          const int pq = g.blocks[3 * (bi - 1)], pt = g.blocks[3 * (bi - 1) + 1], pl = g.blocks[3 * (bi - 1) + 2];
          drift = abs((bt - (pt + pl)) - (bq - (pq + pl)));  // ComputeDrift = tGap - qGap (AlignmentUtils.h:586-600)
        }
        cap = -1;  // none
        tPost = g.band + drift;
      } else {
        cap = g.band;
        tPost = min(250, g.band);
      }
    } else {
      // gap rows between block bi and bi+1 (:193-250): min(qGap,tGap) diagonal rows, then straight down
      const int nq = g.blocks[3 * (bi + 1)], nt = g.blocks[3 * (bi + 1) + 1];
      const int qGap = nq - (bq + bl), tGap = nt - (bt + bl);
      const int diagLen = max(min(qGap, tGap), 0);
      const int gi = off - bl;
      t = bt + bl + min(gi, diagLen);
      cap = 250;
      tPost = min(250, g.band + abs(tGap - qGap));
    }
  }
  // lo_r = max over rows j <= r of (t_j - cap_j), seeded with the carried lo
  int reachFloor = (rowValid && cap >= 0) ? t - cap : INT32_MIN;
  int lo = max(wave_prefix_max(reachFloor), g.loCarry);
  int hi = t + tPost;
  unsigned len = rowValid ? (unsigned)(hi - lo + 1) : 0u;
  unsigned incl = wave_prefix_sum(len);
  rowStartOut = g.cellCarry + incl - len;
  // carries for the next group
  int lastLane = min(63, g.Q - r0);
  g.loCarry = __builtin_amdgcn_readlane(lo, lastLane < 0 ? 0 : lastLane);
  g.cellCarry += (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
  tOut = t;
  loOut = lo;
  hiOut = hi;
}

// ---------------------------------------------------------------------------
// geometry_kernel: validation + nCells + widest row, one wavefront per candidate
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void geometry_kernel(RefineBatchView batch, int band, int first, int count,
                                                         TaskGeometry *geo) {
  __shared__ unsigned char flagScratch[4][64];
  const int wave = threadIdx.x >> 6;
  const int lane = lane_id();
  const int idx = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  if (idx >= count) return;
  const int task = first + idx;
  const int nBlocks = batch.n_blocks[task];
  TaskGeometry out;
  out.rows = 0;
  out.cells = 0;
  out.maxWidth = 0;
  out.status = 0;
  out.tSpan = 0;
  if (nBlocks <= 0) {
    out.status = TASK_EMPTY_GUIDE;
    if (lane == 0) geo[task] = out;
    return;
  }
  const int32_t *blocks = batch.blocks + 3 * batch.blk_off[task];
  const int qLen = batch.q_len[task], tLen = batch.t_len[task];
  // validation: blocks ordered, non-overlapping in q and t, positive length, inside both sequences
  bool bad = false;
  for (int b = lane; b < nBlocks; b += 64) {
    int q = blocks[3 * b], t = blocks[3 * b + 1], l = blocks[3 * b + 2];
    if (q < 0 || t < 0 || l <= 0 || (int64_t)q + l > qLen || (int64_t)t + l > tLen) bad = true;
    if (b + 1 < nBlocks) {
      int nq = blocks[3 * b + 3], nt = blocks[3 * b + 4];
      if (nq < (int64_t)q + l || nt < (int64_t)t + l) bad = true;
    }
  }
  if (__ballot(bad) != 0ull) {
    out.status = TASK_BAD_GUIDE;
    if (lane == 0) geo[task] = out;
    return;
  }
  GuideCursor g;
  guide_begin(g, blocks, nBlocks, band);
  int widest = guide_row0_tpost(g) + 1;
  // the kernels address a candidate's cells with 32 bits (GuideCursor::cellCarry, the row records): count in 64
  // here and refuse a candidate whose band would wrap them (a megabase read with a band of thousands)
  int64_t cells64 = (int64_t)g.cellCarry;
  for (int r0 = 1; r0 <= g.Q; r0 += 64) {
    int t, lo, hi;
    unsigned rs;
    const unsigned before = g.cellCarry;
    guide_group(g, r0, flagScratch[wave], t, lo, hi, rs);
    cells64 += (int64_t)(unsigned)(g.cellCarry - before);
    int w = (r0 + lane <= g.Q) ? hi - lo + 1 : 0;
    widest = max(widest, wave_max(w));
    if (w < 0 || w >= (1 << 24)) cells64 = INT64_MAX / 2;  // (uniform after the ballot below)
  }
  if (__ballot(cells64 >= ((int64_t)1 << 31) - 64) != 0ull) {
    out.status = TASK_BAD_GUIDE;
    if (lane == 0) geo[task] = out;
    return;
  }
  out.rows = g.Q;
  out.cells = (int64_t)g.cellCarry;
  out.maxWidth = widest;
  out.tSpan = blocks[3 * (nBlocks - 1) + 1] + blocks[3 * (nBlocks - 1) + 2] - g.tStart;
  if (lane == 0) geo[task] = out;
}

// ---------------------------------------------------------------------------
// geometry_blocks_kernel: geometry_kernel's answers -- rows, cells, widest row, validation -- WITHOUT the walk over the rows:
// lane = guide block.  Inside a block (and in the gap rows behind it) the centre column moves by 0 or 1 per row and
// lo_r = max(lo_{r-1}, t_r - cap_r) (guide_group), so the rows of block b are a handful of arithmetic series once the lo it
// starts with is known, and that lo is a running maximum over the blocks before it:
//   L0(b) = max(tStart - 1, A(0), .., A(b-1)),   A(b) = the largest t_r - cap_r among b's own rows
//     rows 1 .. len-1 of the block: cap = band, t = bt + off           -> bt + len - 1 - band          (len > 1)
//     the qGap gap rows behind it:  cap = 250,  t = bt + len + min(i, diagLen) -> t_last - 250         (qGap > 0)
//   row 0 of the block (cap none):  width = bt + band + drift - L0 + 1
//   rows 1 .. len-1:                width = bt + off + P - max(L0, bt + off - band) + 1,   P = min(250, band)
//   gap row i:                      width = t_i + G - max(Lb, t_i - 250) + 1,  G = min(250, band + |tGap - qGap|),
//                                   Lb = max(L0, bt + len - 1 - band [len > 1])
// Every piece is linear in the row, so its sum, its smallest and its largest width are closed forms.  The sums must equal
// geometry_rows_kernel's exactly (the arrow slabs are sized from them): BLASR_HIP_GEOMETRY_CHECK runs geometry_kernel as
// well and compares every candidate.
// ---------------------------------------------------------------------------
struct SeriesAcc {
  int64_t cells;
  int minW, maxW;
  __device__ __forceinline__ void constant(int64_t n, int w) {  // n rows of width w
    if (n <= 0) return;
    cells += n * (int64_t)w;
    minW = min(minW, w);
    maxW = max(maxW, w);
  }
  __device__ __forceinline__ void rising(int64_t n, int first) {  // n rows of width first, first + 1, ...
    if (n <= 0) return;
    cells += n * (int64_t)first + n * (n - 1) / 2;
    minW = min(minW, first);
    maxW = max(maxW, first + (int)(n - 1));
  }
};
__global__ __launch_bounds__(256) void geometry_blocks_kernel(RefineBatchView batch, int band, int first, int count,
                                                                TaskGeometry *geo) {
  const int wave = threadIdx.x >> 6;
  const int lane = lane_id();
  const int idx = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  if (idx >= count) return;
  const int task = first + idx;
  const int nBlocks = batch.n_blocks[task];
  TaskGeometry out;
  out.rows = 0;
  out.cells = 0;
  out.maxWidth = 0;
  out.status = 0;
  out.tSpan = 0;
  if (nBlocks <= 0) {
    out.status = TASK_EMPTY_GUIDE;
    if (lane == 0) geo[task] = out;
    return;
  }
  const int32_t *blocks = batch.blocks + 3 * batch.blk_off[task];
  const int qLen = batch.q_len[task], tLen = batch.t_len[task];
  const int qStart = blocks[0], tStart = blocks[1];
  const int P = min(250, band);
  const int drift0 = abs(tStart - qStart);
  SeriesAcc acc;
  acc.cells = 0;
  acc.minW = INT32_MAX;
  acc.maxW = INT32_MIN;
  bool bad = false;
  int loCarry = tStart - 1;  // L0 of the next chunk's first block
  for (int b0 = 0; b0 < nBlocks; b0 += 64) {
    const int b = b0 + lane;
    const bool have = b < nBlocks;
    int bq = 0, bt = 0, bl = 1, nq = 0, nt = 0, pq = 0, pt = 0, pl = 0;
    if (have) {
      bq = blocks[3 * b], bt = blocks[3 * b + 1], bl = blocks[3 * b + 2];
      // validation: blocks ordered, non-overlapping in q and t, positive length, inside both sequences
      if (bq < 0 || bt < 0 || bl <= 0 || (int64_t)bq + bl > qLen || (int64_t)bt + bl > tLen) bad = true;
      if (b + 1 < nBlocks) {
        nq = blocks[3 * b + 3], nt = blocks[3 * b + 4];
        if (nq < (int64_t)bq + bl || nt < (int64_t)bt + bl) bad = true;
      }
      if (b > 0) pq = blocks[3 * b - 3], pt = blocks[3 * b - 2], pl = blocks[3 * b - 1];
    }
    if (__ballot(bad) != 0ull) break;  // (what follows assumes an ordered guide)
    const bool hasNext = have && b + 1 < nBlocks;
    const int qGap = hasNext ? nq - (bq + bl) : 0, tGap = hasNext ? nt - (bt + bl) : 0;
    const int diagLen = max(min(qGap, tGap), 0);
    const int T0 = bt + bl;
    const int tLast = T0 + min(qGap - 1, diagLen);  // centre of the last gap row (qGap > 0)
    int A = INT32_MIN;
    if (have && bl > 1) A = bt + bl - 1 - band;
    if (have && qGap > 0) A = max(A, tLast - 250);
    // exclusive running maximum over the blocks
    const int incl = wave_prefix_max(A);
    int L0 = wave_shr1(incl, INT32_MIN);
    L0 = max(L0, loCarry);
    loCarry = max(loCarry, __builtin_amdgcn_readlane(incl, 63));
    if (have) {
      // row 0 of the block
      const int drift = b == 0 ? drift0 : abs((bt - (pt + pl)) - (bq - (pq + pl)));
      acc.constant(1, bt + band + drift - L0 + 1);
      // rows 1 .. bl-1: width bt + off + P - L0 + 1 while bt + off - band < L0, then P + band + 1
      {
        const int64_t n = bl - 1;
        const int64_t o1 = (int64_t)L0 - bt + band;  // first off with bt + off - band >= L0
        const int64_t nA = min(max(o1 - 1, (int64_t)0), n);
        acc.rising(nA, bt + 1 + P - L0 + 1);
        acc.constant(n - nA, P + band + 1);
      }
      if (qGap > 0) {
        const int Lb = bl > 1 ? max(L0, bt + bl - 1 - band) : L0;
        const int G = min(250, band + abs(tGap - qGap));
        // gap rows 0 .. n1-1: t = T0 + i; then t = T0 + diagLen
        const int64_t n1 = min((int64_t)qGap, (int64_t)diagLen + 1);
        const int64_t g1 = (int64_t)Lb - T0 + 250;  // first i with T0 + i - 250 >= Lb
        const int64_t nA = min(max(g1, (int64_t)0), n1);
        acc.rising(nA, T0 + G - Lb + 1);
        acc.constant(n1 - nA, G + 251);
        const int Tc = T0 + diagLen;
        acc.constant((int64_t)qGap - n1, Tc + G - max(Lb, Tc - 250) + 1);
      }
    }
  }
  if (__ballot(bad) != 0ull) {
    out.status = TASK_BAD_GUIDE;
    if (lane == 0) geo[task] = out;
    return;
  }
  // row 0 of the matrix (AffineGuidedAlign.h:126-135): tPost = max(drift, band), tPre = 0
  const int tPost0 = drift0 > band ? drift0 : band;
  int64_t cells = acc.cells;
  int minW = acc.minW, maxW = acc.maxW;
  for (int off = 32; off > 0; off >>= 1) {
    cells += __shfl_xor(cells, off);
    minW = min(minW, __shfl_xor(minW, off));
    maxW = max(maxW, __shfl_xor(maxW, off));
  }
  cells += tPost0 + 1;
  // the kernels address a candidate's cells with 32 bits: refuse a candidate whose band would wrap them, or with a row of
  // negative or absurd width (geometry_kernel's rule)
  if (minW < 0 || maxW >= (1 << 24) || cells >= ((int64_t)1 << 31) - 64) {
    out.status = TASK_BAD_GUIDE;
    if (lane == 0) geo[task] = out;
    return;
  }
  out.rows = blocks[3 * (nBlocks - 1)] + blocks[3 * (nBlocks - 1) + 2] - qStart;
  out.cells = cells;
  out.maxWidth = max(maxW, tPost0 + 1);
  out.tSpan = blocks[3 * (nBlocks - 1) + 1] + blocks[3 * (nBlocks - 1) + 2] - tStart;
  if (lane == 0) geo[task] = out;
}

// ---------------------------------------------------------------------------
// geometry_rows_kernel: the second walk over the guide, once the chunk plan has given every candidate its place in
// the workspace: the row records {matrix offset of the row's first cell, first column} and, per row, its number of
// cells (those with t >= tEnd left out, AffineGuidedAlign.h:268-269) and the query base's code.  The fill kernels
// read them back 64 rows at a time, coalesced, instead of deriving the geometry again for every batch of rows.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void geometry_rows_kernel(RefineBatchView batch, int band, int first, int count,
                                                              const TaskGeometry *geo, int2 *rowTab,
                                                              const int64_t *rowOff, unsigned short *rowMeta) {
  __shared__ unsigned char flagScratch[4][64];
  const int wave = threadIdx.x >> 6;
  const int lane = lane_id();
  const int idx = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  if (idx >= count) return;
  const int task = first + idx;
  if (geo[task].status != 0) return;
  const int nBlocks = batch.n_blocks[task];
  const int32_t *blocks = batch.blocks + 3 * batch.blk_off[task];
  const uint8_t *qSeq = batch.seq + batch.q_off[task];
  int2 *rows = rowTab + rowOff[task];
  unsigned short *meta = rowMeta + rowOff[task];
  GuideCursor g;
  guide_begin(g, blocks, nBlocks, band);
  const int tEnd = blocks[3 * (nBlocks - 1) + 1] + blocks[3 * (nBlocks - 1) + 2];
  // guide_group's arithmetic (the geometry of rows r0 .. r0 + 63, lane = row) with ONE round trip to memory per group and that
  // one issued a group ahead (round 6).  guide_group reads the first rows of the next 64 blocks, finds every row's block from
  // them, and then GATHERS that block, the one before and the one after it from memory again: two dependent round trips per 64
  // rows, which is what this kernel's 12 ms per configs[2] step were (170 groups per candidate, eight wavefronts per SIMD
  // waiting).  A row's block is one of the 64 whose triplets lane j has just read (block nextBlock + j) or the one the group
  // before ended in, so the three blocks come off the lanes (ds_bpermute) or out of a carried triplet; and the blocks the NEXT
  // group will look at are known as soon as this group has counted its block starts.
  auto load_triplet = [&](int b, int &q, int &t, int &l) {
    q = INT32_MAX;  // (no such block: it starts in no group)
    t = l = 0;
    if (b < nBlocks) {
      q = blocks[3 * b];
      t = blocks[3 * b + 1];
      l = blocks[3 * b + 2];
    }
  };
  int nq, nt, nl;  // block g.nextBlock + lane
  load_triplet(lane, nq, nt, nl);
  int cq = 0, ct = 0, cl = 0;  // the block the group before ended in (uniform; none before the first group)
  for (int r0 = 1; r0 <= g.Q; r0 += 64) {
    const int r = r0 + lane;
    const bool rowValid = r <= g.Q;
    const int bq0 = nq, bt0 = nt, bl0 = nl, base = g.nextBlock;
    // ---- which blocks start inside this group?  (a block owns at least one row, so at most 64 of them)
    flagScratch[wave][lane] = 0;
    __builtin_amdgcn_wave_barrier();
    const int rb = bq0 == INT32_MAX ? INT32_MAX : bq0 - g.qStart + 1;
    const bool starts = rb < r0 + 64;
    if (starts) flagScratch[wave][rb - r0] = 1;
    __builtin_amdgcn_wave_barrier();
    const unsigned long long startMask = __ballot(flagScratch[wave][lane] != 0);
    __builtin_amdgcn_wave_barrier();
    const int nStarts = (int)__popcll(__ballot(starts));
    g.nextBlock = base + nStarts;
    // the next group's blocks: their loads are in flight while this group is worked out
    load_triplet(g.nextBlock + lane, nq, nt, nl);
    const unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    const int rel = (int)__popcll(startMask & le) - 1;  // my block is base + rel: -1 = the carried one
    auto off_lane = [&](int v, int j) { return __builtin_amdgcn_ds_bpermute((j & 63) << 2, v); };
    int t = 0, cap = 0, tPost = 0;
    {
      // (every lane takes part in the permutes; what an invalid row gets is not used)
      const int mq = off_lane(bq0, rel), mt = off_lane(bt0, rel), ml = off_lane(bl0, rel);
      const int pe = off_lane(bq0 + bl0, rel - 1), pf = off_lane(bt0 + bl0, rel - 1);  // the block before: q + len, t + len
      const int sq = off_lane(bq0, rel + 1), st = off_lane(bt0, rel + 1);              // the block after
      const int bq = rel >= 0 ? mq : cq, bt = rel >= 0 ? mt : ct, bl = rel >= 0 ? ml : cl;
      const int pqe = rel >= 1 ? pe : cq + cl, pte = rel >= 1 ? pf : ct + cl;
      if (rowValid) {
        const int qpos = g.qStart + r - 1;
        const int off = qpos - bq;
        if (off < bl) {
          t = bt + off;
          if (off == 0) {  // first row of a block: tPre = reach (uncapped), tPost = band + |drift of the gap before| (:161-165)
            int drift;
            if (base + rel == 0) drift = abs(g.tStart - g.qStart);
            else drift = abs((bt - pte) - (bq - pqe));  // ComputeDrift = tGap - qGap (AlignmentUtils.h:586-600)
            cap = -1;  // none
            tPost = g.band + drift;
          } else {
            cap = g.band;
            tPost = min(250, g.band);
          }
        } else {  // gap rows between this block and the next (:193-250): min(qGap, tGap) diagonal rows, then straight down
          const int qGap = sq - (bq + bl), tGap = st - (bt + bl);
          const int diagLen = max(min(qGap, tGap), 0);
          const int gi = off - bl;
          t = bt + bl + min(gi, diagLen);
          cap = 250;
          tPost = min(250, g.band + abs(tGap - qGap));
        }
      }
    }
    if (nStarts > 0) {  // the block this group ends in
      cq = __builtin_amdgcn_readlane(bq0, nStarts - 1);
      ct = __builtin_amdgcn_readlane(bt0, nStarts - 1);
      cl = __builtin_amdgcn_readlane(bl0, nStarts - 1);
    }
    // lo_r = max over rows j <= r of (t_j - cap_j), seeded with the carried lo
    const int reachFloor = (rowValid && cap >= 0) ? t - cap : INT32_MIN;
    const int lo = max(wave_prefix_max(reachFloor), g.loCarry);
    const int hi = t + tPost;
    const unsigned len = rowValid ? (unsigned)(hi - lo + 1) : 0u;
    const unsigned incl = wave_prefix_sum(len);
    const unsigned rs = g.cellCarry + incl - len;
    const int lastLane = min(63, g.Q - r0);
    g.loCarry = __builtin_amdgcn_readlane(lo, lastLane < 0 ? 0 : lastLane);
    g.cellCarry += (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
    if (rowValid) {
      const int width = min(hi, tEnd - 1) - lo + 1;
      const int qc = base_code(qSeq[g.qStart + r - 1]);
      rows[r] = make_int2((int)rs, lo);
      meta[r] = (unsigned short)(min(max(width, 0), 0xfff) | (qc << 12));  // (a width beyond 4095 saturates: wide rows
    }                                                                       //  are not fillp_kernel's)
  }
}

// ---------------------------------------------------------------------------
// fill_kernel
// ---------------------------------------------------------------------------
// Packed arrow byte written per band cell:
//   bits 0-2  S arrow as its PRIORITY in AffineGuidedAlign.h:326-341:
//             0 Diagonal, 1 Left (del), 2 Up (ins), 3 AffineInsClose, 4 AffineDelClose
//   bit 3     affine-insert matrix: 1 = AffineInsOpen, 0 = AffineInsUp   (:357-364)
//   bit 4     affine-delete matrix: 1 = AffineDelOpen, 0 = AffineDelLeft (:366-373)
enum { AR_DIAG = 0, AR_LEFT = 1, AR_UP = 2, AR_INS_CLOSE = 3, AR_DEL_CLOSE = 4, AR_I_OPEN = 8, AR_D_OPEN = 16 };

// Two inclusive 64-lane prefix-min scans, interleaved so that each DPP read of a
// just-written VGPR has its two wait states covered by the other chain + s_nop.
__device__ __forceinline__ void wave_prefix_min2(int &a, int &b) {
  asm volatile(
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_i32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_i32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_i32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_i32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_min_i32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_min_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_min_i32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(a), "+v"(b));
}
// lane i receives lane i+1; lane 63 receives `edge` (DPP wave_shl:1)
__device__ __forceinline__ int wave_shl1(int v, int edge) { return dpp<0x130, 0xf>(edge, v); }

// Scores are carried scaled by kScale with the low bits free for a tag, so that ONE
// integer min yields both the minimum and, on ties, the reference's arrow priority:
// key = score * 32 + priority.  Needs ext < del (then S == S[c-1]+del exactly when S equals
// the linear-gap chain value) and |score| < 2^24; other parameter
// sets run the compare-based variant (kFast = false) on unscaled scores.
constexpr int kShift = 5;
constexpr int kTagMask = 31;

template <bool kFast>
struct RowConsts {
  int inf;                // "no predecessor"
  int ins, del, open, ext;  // unscaled (compare variant) or scaled by 32 (fast variant)
  int laneDel, laneExt;   // lane * del, lane * ext (same scaling)
};
struct ChunkCarry {
  int a, b, s;  // horizontal minima entering this chunk (del chain, affine chain) and S of the last lane before it
};

// One 64-column chunk of one row.  upS/upI = S and I of the row above at this lane's column,
// diagS = S of the row above at column-1 (all "inf" where that cell does not exist),
// m = substitution score (unscaled).  Returns S and the new I for the active lanes (inf elsewhere)
// and the packed arrow byte.
template <bool kFast, bool kOnlyChunk>
__device__ __forceinline__ void row_chunk(const RowConsts<kFast> &k, ChunkCarry &carry, bool active, int upS, int upI,
                                          int diagS, int m, int &sOut, int &iOut, int &arrowOut) {
  if (kFast) {
    const int matchK = (m << kShift) + diagS;        // tag 0 = Diagonal
    const int insSK = upS + (k.ins + AR_UP);          // tag 2 = Up
    const int insEK = upI + (k.ext + AR_INS_CLOSE);   // tag 3 = AffineInsClose
    const int vKey = min(matchK, min(insSK, insEK));
    int v = vKey & ~kTagMask;
    v = active ? v : k.inf;
    int ia = v - k.laneDel, ib = v - k.laneExt;
    wave_prefix_min2(ia, ib);
    int ea = wave_shr1(ia, k.inf), eb = wave_shr1(ib, k.inf);
    if (!kOnlyChunk) {
      ea = min(ea, carry.a);
      eb = min(eb, carry.b);
    }
    const int aK = ea + (k.laneDel + AR_LEFT);                // S[c-1] + del, tag 1 = Left
    const int bK = eb + (k.laneExt + k.open + AR_DEL_CLOSE);  // D[c-1] + ext, tag 4 = AffineDelClose
    const int sKey = min(vKey, min(aK, bK));
    const int s = sKey & ~kTagMask;
    // I = (S + open < I_up + ext) ? open : extend -- ties extend, so "open" carries the larger tag
    const int openKey = s + (k.open + AR_I_OPEN);
    const int iKey = min(openKey, insEK & ~kTagMask);
    // D = (S + open < D_left + ext) ? open : extend, of which only the flag is kept: with the tags in place
    // (open 8 against close 4 on multiples of 32) the strict comparison of the untagged values is openKey < bK.
    // S tags live in bits 0-2, the open flags in bits 3-4, nothing else below kShift: the byte is the OR, masked
    const int dFlag = openKey < bK ? AR_D_OPEN : 0;
    arrowOut = (sKey | iKey | dFlag) & kTagMask;
    sOut = active ? s : k.inf;
    iOut = active ? (iKey & ~kTagMask) : k.inf;
    if (!kOnlyChunk) {
      carry.a = min(carry.a, __builtin_amdgcn_readlane(ia, 63)) + 64 * k.del;
      carry.b = min(carry.b, __builtin_amdgcn_readlane(ib, 63)) + 64 * k.ext;
    }
  } else {
    const int matchS = diagS + m, insS = upS + k.ins, insE = upI + k.ext;
    int v = min(matchS, min(insS, insE));
    v = active ? v : k.inf;
    int ia = v - k.laneDel, ib = v - k.laneExt;
    wave_prefix_min2(ia, ib);
    int ea = wave_shr1(ia, k.inf), eb = wave_shr1(ib, k.inf);
    if (!kOnlyChunk) {
      ea = min(ea, carry.a);
      eb = min(eb, carry.b);
    }
    const int delExt = eb + k.laneExt + k.open;  // = D[c-1] + ext
    const int s = min(v, min(ea + k.laneDel, delExt));
    const int delS = wave_shr1(s, kOnlyChunk ? k.inf : carry.s) + k.del;  // S[c-1] + del
    int arrow = AR_DEL_CLOSE;
    arrow = (s == insE) ? AR_INS_CLOSE : arrow;
    arrow = (s == insS) ? AR_UP : arrow;
    arrow = (s == delS) ? AR_LEFT : arrow;
    arrow = (s == matchS) ? AR_DIAG : arrow;
    const int openS = s + k.open;
    const bool insOpens = openS < insE, delOpens = openS < delExt;
    arrowOut = arrow | (insOpens ? AR_I_OPEN : 0) | (delOpens ? AR_D_OPEN : 0);
    sOut = active ? s : k.inf;
    iOut = active ? (insOpens ? openS : insE) : k.inf;
    if (!kOnlyChunk) {
      carry.s = __builtin_amdgcn_readlane(s, 63);
      carry.a = min(carry.a, __builtin_amdgcn_readlane(ia, 63)) + 64 * k.del;
      carry.b = min(carry.b, __builtin_amdgcn_readlane(ib, 63)) + 64 * k.ext;
    }
  }
}

template <bool kGlobalRing, bool kFast>
__global__ __launch_bounds__(256) void fill_kernel(RefineBatchView batch, FillArgs args) {
  // per-wave LDS: target ring (one dword per target column: the column of the score matrix, packed,
  // or the base code), {S,I} ring by column (only rows wider than one wavefront go through it)
  __shared__ int2 ldsRing[kGlobalRing ? 1 : 4][kGlobalRing ? 1 : kLdsRing];
  __shared__ uint32_t ldsT[kGlobalRing ? 1 : 4][kGlobalRing ? 1 : kLdsRingT];
  __shared__ unsigned char flagScratch[4][64];
  const int wave = threadIdx.x >> 6;
  const int lane = lane_id();
  const int idx = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  if (idx >= args.count) return;
  const int task = kGlobalRing ? args.wideTasks[idx] : (args.taskList ? args.taskList[idx] : args.first + idx);
  const TaskGeometry geo = args.geo[task];
  if (geo.status != 0) return;
  if (!kGlobalRing && geo.maxWidth > kLdsMaxWidth) return;  // handled by the global-ring launch

  int2 *ring;
  uint32_t *ringT;
  int ringMask, ringMaskT;
  if (kGlobalRing) {
    const int64_t off = args.wideRingOff[idx];
    ringMask = ringMaskT = args.wideRingSize[idx] - 1;
    ring = reinterpret_cast<int2 *>(args.wideRings + off);
    ringT = reinterpret_cast<uint32_t *>(ring + (ringMask + 1));
  } else {
    ring = ldsRing[wave];
    ringT = ldsT[wave];
    ringMask = kLdsRing - 1;
    ringMaskT = kLdsRingT - 1;
  }

  const uint8_t *qSeq = batch.seq + batch.q_off[task];
  const uint8_t *tSeq = batch.seq + batch.t_off[task];
  const int tLen = batch.t_len[task];
  const int32_t *blocks = batch.blocks + 3 * batch.blk_off[task];
  const int nBlocks = batch.n_blocks[task];
  uint8_t *arrows = args.arrows + args.arrowOff[task];
  int2 *rowTab = args.rowTab + args.rowOff[task];

  RowConsts<kFast> k;
  const int sc = kFast ? kShift : 0;
  k.inf = kFast ? (1 << 30) : (1 << 29);
  k.ins = args.ins << sc;
  k.del = args.del << sc;
  k.open = args.open << sc;
  k.ext = args.ext << sc;
  k.laneDel = lane * k.del;
  k.laneExt = lane * k.ext;
  // the 5x5 matrix lives in one VGPR: lane i = entry i  ([query][target])
  const int matV = lane < 25 ? args.matrix[lane] : 0;
  // fast variant: per target code, the matrix column for queries A,C,G,T packed into one dword
  auto packedColumn = [&](int tcode) {
    const int b0 = __builtin_amdgcn_ds_bpermute((0 * 5 + tcode) * 4, matV) & 0xff;
    const int b1 = __builtin_amdgcn_ds_bpermute((1 * 5 + tcode) * 4, matV) & 0xff;
    const int b2 = __builtin_amdgcn_ds_bpermute((2 * 5 + tcode) * 4, matV) & 0xff;
    const int b3 = __builtin_amdgcn_ds_bpermute((3 * 5 + tcode) * 4, matV) & 0xff;
    return (uint32_t)(b0 | (b1 << 8) | (b2 << 16) | (b3 << 24));
  };

  GuideCursor g;
  guide_begin(g, blocks, nBlocks, args.band);
  const int qStart = g.qStart, tStart = g.tStart, Q = g.Q;
  const int lastBlk = nBlocks - 1;
  const int tEnd = blocks[3 * lastBlk + 1] + blocks[3 * lastBlk + 2];  // guide[last].t + 1

  // ---- row 0 (AffineGuidedAlign.h:166-208): S = del * distance from tStart-1, I = open ----
  int plo = tStart - 1;
  int phi = plo + guide_row0_tpost(g);
  int tLoaded = plo < 0 ? 0 : plo;  // target columns < tLoaded are staged in ringT
  // S and I of the previous row: in registers (lane j = column plo + j, "inf" beyond the row) while
  // that row is <= 64 wide, otherwise in the ring
  bool prevInRegs = (phi - plo) < 64;
  int prevS = (plo + lane <= phi) ? k.laneDel : k.inf;
  int prevI = (plo + lane <= phi) ? k.open : k.inf;
  if (!prevInRegs) {
    for (int c0 = plo; c0 <= phi; c0 += 64) {
      int c = c0 + lane;
      if (c <= phi) ring[c & ringMask] = make_int2((c - plo) * k.del, k.open);
    }
    if (kGlobalRing) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }

  for (int r0 = 1; r0 <= Q; r0 += 64) {
    int tV, loV, hiV;
    unsigned rsV;
    guide_group(g, r0, flagScratch[wave], tV, loV, hiV, rsV);
    const int nRows = min(64, Q - r0 + 1);
    if (lane < nRows) rowTab[r0 + lane] = make_int2((int)rsV, loV);
    // query codes of these 64 rows
    int qCodeV = 4;
    if (lane < nRows) qCodeV = base_code(qSeq[qStart + r0 - 1 + lane]);

    // The target ring is staged one row ahead of its use, so the LDS read of a row's entries is
    // issued while the previous row is being computed.
    auto stageT = [&](int upTo) {
      while (tLoaded <= upTo) {
        int c = tLoaded + lane;
        int code = (c < tLen) ? base_code(tSeq[c]) : 4;
        ringT[c & ringMaskT] = kFast ? packedColumn(code) : (uint32_t)code;
        tLoaded += 64;
        if (kGlobalRing) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    };
    stageT(min(__builtin_amdgcn_readlane(hiV, 0), tEnd - 1));
    uint32_t tNext = ringT[(__builtin_amdgcn_readlane(loV, 0) + lane) & ringMaskT];

    for (int i = 0; i < nRows; i++) {
      const int lo = __builtin_amdgcn_readlane(loV, i);
      const int hi = __builtin_amdgcn_readlane(hiV, i);
      const unsigned rowStart = (unsigned)__builtin_amdgcn_readlane((int)rsV, i);
      const int qc = __builtin_amdgcn_readlane(qCodeV, i);
      const int hiC = min(hi, tEnd - 1);  // cells with t >= tEnd are skipped (:268-269)
      const uint32_t tCur = tNext;  // first 64 columns of this row, fetched one row ahead
      if (i + 1 < nRows) {
        stageT(min(__builtin_amdgcn_readlane(hiV, i + 1), tEnd - 1));
        tNext = ringT[(__builtin_amdgcn_readlane(loV, i + 1) + lane) & ringMaskT];
      }
      // substitution score of this row's query base against a target-ring entry at column c
      auto subst = [&](uint32_t tEntry, int c) -> int {
        if (kFast) {
          if (qc < 4) return (int)__builtin_amdgcn_sbfe((int)tEntry, qc * 8, 8);
          // query N (rare): the packed column has no N row; read the base again
          const int tcode = (c >= 0 && c < tLen) ? base_code(tSeq[c]) : 4;
          return __builtin_amdgcn_ds_bpermute((20 + tcode) * 4, matV);
        } else {
          return __builtin_amdgcn_ds_bpermute((qc * 5 + (int)tEntry) * 4, matV);
        }
      };
      uint8_t *arrowRow = arrows + rowStart;  // + (column - lo)
      const bool curSingle = (hiC - lo) < 64;
      ChunkCarry carry;
      carry.a = carry.b = carry.s = k.inf;

      if (prevInRegs && curSingle) {
        // ---- register path: the row above is shifted into place with DPP ----
        const int shift = lo - plo;  // >= 0
        int upS, upI, diagS;
        if (shift == 1) {
          upS = wave_shl1(prevS, k.inf);
          upI = wave_shl1(prevI, k.inf);
          diagS = prevS;
        } else if (shift == 0) {
          upS = prevS;
          upI = prevI;
          diagS = wave_shr1(prevS, k.inf);
        } else {
          const bool upOk = lane + shift < 64, diagOk = lane + shift - 1 < 64;
          upS = __builtin_amdgcn_ds_bpermute((lane + shift) * 4, prevS);
          upI = __builtin_amdgcn_ds_bpermute((lane + shift) * 4, prevI);
          diagS = __builtin_amdgcn_ds_bpermute((lane + shift - 1) * 4, prevS);
          upS = upOk ? upS : k.inf;
          upI = upOk ? upI : k.inf;
          diagS = diagOk ? diagS : k.inf;
        }
        int arrow;
        row_chunk<kFast, true>(k, carry, lane <= hiC - lo, upS, upI, diagS, subst(tCur, lo + lane), prevS, prevI,
                               arrow);
        // uniform row base in SGPRs + lane offset: no per-lane 64-bit address arithmetic
        if (lane <= hiC - lo) asm volatile("global_store_byte %0, %1, %2" ::"v"(lane), "v"(arrow), "s"(arrowRow) : "memory");
      } else {
        // ---- ring path: rows wider than one wavefront ----
        if (prevInRegs) {
          const int c = plo + lane;
          if (c <= phi) ring[c & ringMask] = make_int2(prevS, prevI);
          if (kGlobalRing) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        int carryDiag = 0;
        for (int cb = lo; cb <= hiC; cb += 64) {
          const int c = cb + lane;
          const bool active = c <= hiC;
          int2 up = ring[c & ringMask];
          int sd = ring[(c - 1) & ringMask].x;
          if (cb > lo) sd = (lane == 0) ? carryDiag : sd;  // column cb-1 was overwritten by the previous chunk
          carryDiag = __builtin_amdgcn_readlane(up.x, 63);
          // ring entries outside the previous row's columns are stale
          const bool upValid = c <= phi, diagValid = (c - 1 <= phi) && (c > plo);
          up.x = upValid ? up.x : k.inf;
          up.y = upValid ? up.y : k.inf;
          sd = diagValid ? sd : k.inf;
          const uint32_t tEntry = (cb == lo) ? tCur : ringT[c & ringMaskT];
          int s, iNew, arrow;
          row_chunk<kFast, false>(k, carry, active, up.x, up.y, sd, subst(tEntry, c), s, iNew, arrow);
          if (active) {
            ring[c & ringMask] = make_int2(s, iNew);
            arrowRow[c - lo] = (uint8_t)arrow;
          }
          if (kGlobalRing) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          if (cb == lo) {
            prevS = s;
            prevI = iNew;
          }
        }
      }
      prevInRegs = curSingle;
      plo = lo;
      phi = hi;
    }
  }
  // DP score = S[Q][tEnd-1] (AffineGuidedAlign.h:481-483): the last row is still in registers or in the ring
  int finalScore = 0;
  if (prevInRegs) {
    if (plo + lane == tEnd - 1) finalScore = prevS >> sc;
  } else {
    if (lane == 0) finalScore = ring[(tEnd - 1) & ringMask].x >> sc;
  }
  // only one lane holds a non-zero value: a wave sum publishes it
  int fs = finalScore;
  for (int off = 32; off > 0; off >>= 1) fs += __shfl_xor(fs, off);
  if (lane == 0) args.dpScore[task] = fs;
}

// ---------------------------------------------------------------------------
// fillp_kernel<kL>: the same recurrence, 64/kL candidates per wavefront.
//
// A band row is ~33 cells, so one candidate per wavefront leaves half of the 64 lanes idle while the kernel is
// bound by VALU issue.  Here each candidate owns kL lanes (16: one DPP row, 8: half of one) and every lane 48/kL
// consecutive cells (48 columns per pass; wider rows take further passes with carries).  The row above lives in
// a per-candidate LDS ring indexed by target column -- any band shift is just an address -- and is updated in
// place.  The horizontal minima are a lane-local prefix over the lane's cells plus a DPP scan over the kL lanes
// that never crosses into another candidate's lanes.  Tagged keys as in fill_kernel<.., true>; candidates
// outside that variant's domain, or with a row wider than kF4MaxWidth, stay with fill_kernel.
// The fewer lanes per candidate, the more of the per-row bookkeeping (row records, staging, scan, loop) is shared.
// ---------------------------------------------------------------------------
constexpr int kF4Ring = 128;       // {S,I} columns per candidate (a row and the row above overlap in it)
constexpr int kF4RingT = 128;      // packed matrix columns per candidate (a row plus the staging step)
constexpr int kF4MaxWidth = 112;   // widest row: with the staging step it fills the packed-column ring
constexpr int kF4Pass = 48;        // columns per pass

template <int kL>
__device__ __forceinline__ void group_prefix_min2(int &a, int &b);
template <>
__device__ __forceinline__ void group_prefix_min2<16>(int &a, int &b) {
  asm volatile(
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_i32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_i32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_i32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_i32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(a), "+v"(b));
}
// eight lanes: two quad steps (a lane that has no left neighbour reads itself, and min is idempotent), then the
// total of the lower quad reaches the upper one (banks 1 and 3 of each DPP row only)
template <>
__device__ __forceinline__ void group_prefix_min2<8>(int &a, int &b) {
  int ta, tb;
  asm volatile(
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_min_i32_dpp %1, %1, %1 quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_min_i32_dpp %0, %0, %0 quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_min_i32_dpp %1, %1, %1 quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_mov_b32_dpp %2, %0 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %3, %1 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_min_i32_dpp %0, %2, %0 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_min_i32_dpp %1, %3, %1 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "s_nop 1"
      : "+v"(a), "+v"(b), "=&v"(ta), "=&v"(tb));
}

typedef uint32_t __attribute__((aligned(1))) unaligned_u32;
typedef uint16_t __attribute__((aligned(1))) unaligned_u16;

// kExt0: the affine extension penalty is 0 (the reference's default): the D chain has no per-column term and the
// extension of I is the value above itself -- six instructions less per row.
// kHybrid: the candidates with a row wider than kF4MaxWidth.  Their narrow rows (most of them: a wide row is what a
// large gap of the guide leaves behind) run as above, several candidates to a wavefront; a wide row is taken by the
// WHOLE wavefront, 64 columns per step (fill_kernel's ring path), through a per-candidate ring in global memory, while
// the other candidates of the wavefront wait.  The row above moves between the LDS ring and the global ring when the
// width class changes (at most kF4MaxWidth + 2 columns either way).
template <int kL, bool kExt0, bool kHybrid>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void fillp_kernel(RefineBatchView batch, FillArgs args) {
  constexpr int kT = 64 / kL;      // candidates per wavefront
  constexpr int kC = kF4Pass / kL;  // cells per lane and pass
  typedef typename std::conditional<kHybrid, uint32_t, unsigned short>::type RowB;
  constexpr int kQcShift = kHybrid ? 24 : 8;
  constexpr unsigned kWidthMask = kHybrid ? 0xffffffu : 0xffu;
  __shared__ int2 ringLds[4][kT][kF4Ring];
  __shared__ uint32_t ringTLds[4][kT][kF4RingT];
  __shared__ int2 rowA[4][kT][64];          // per row of the current 64-row batch: {matrix offset, first column}
  __shared__ RowB rowB[4][kT][64];          // width | query code << 8 (<< 24 in the hybrid variant)
  __shared__ uint32_t packedCol[8];        // per target code: the matrix column for queries A,C,G,T, one byte each
  if (threadIdx.x < 5) {
    uint32_t v = 0;
    for (int qc = 0; qc < 4; qc++) v |= (uint32_t)(args.matrix[qc * 5 + threadIdx.x] & 0xff) << (8 * qc);
    packedCol[threadIdx.x] = v;
  }
  __shared__ int matN[5];  // the query-N row of the matrix
  if (threadIdx.x < 5) matN[threadIdx.x] = args.matrix[20 + threadIdx.x];
  __syncthreads();
  // hybrid variant: the packed column by target CHARACTER (one LDS read per cell of a wide row instead of the decode)
  __shared__ uint32_t packedByChar[kHybrid ? 256 : 1];
  if (kHybrid) {
    packedByChar[threadIdx.x] = packedCol[base_code((unsigned)threadIdx.x)];
    __syncthreads();
  }
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int grp = lane / kL, l = lane % kL;
  const int slot0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wave)) * kT;
  if (slot0 >= args.count) return;

  // ---- the candidates of this wavefront (uniform values), and each lane's own candidate
  int maxQ = 0;
  int64_t myRowOff = 0;  // (the candidates' uniform values are read off their lanes: from_group below)
  // hybrid variant: what the wavefront needs of a candidate when it takes one of its wide rows is read off that
  // candidate's lanes when the moment comes (from_group below): as uniform arrays for four candidates these values alone
  // were 40 scalar registers held through the whole kernel, and the narrow rows paid for them in spills
  unsigned myCells = 0;
  int myTask = -1, myQ = 0, myTEnd = 0, myTLen = 0, myPlo = 0, myPhi = -1, myTLoaded = 0, myGMask = 0;
  bool myPrevGlobal = false;  // the row above lives in the global ring (the last row was a wide one)
  int mySkipUntil = 0;        // rows up to this one were taken ahead of their turn (a run of wide rows, below)
  int2 *myGRing = nullptr;
  const uint8_t *myTSeq = nullptr;
  uint8_t *myArrows = nullptr;
#pragma unroll
  for (int g = 0; g < kT; g++) {
    if (slot0 + g < args.count && args.taskList[slot0 + g] >= 0) {  // (the hybrid list has empty slots: -1)
      const int task = args.taskList[slot0 + g];
      if (kHybrid && grp == g) {
        myCells = (unsigned)args.geo[task].cells;
        myGMask = args.wideRingSize[slot0 + g] - 1;
        myGRing = reinterpret_cast<int2 *>(args.wideRings + args.wideRingOff[slot0 + g]);
      }
      const int32_t *blocks = batch.blocks + 3 * batch.blk_off[task];
      const int nBlocks = batch.n_blocks[task];
      GuideCursor cur;
      guide_begin(cur, blocks, nBlocks, args.band);
      maxQ = max(maxQ, cur.Q);
      if (grp == g) {
        myRowOff = args.rowOff[task];
        myTask = task;
        myQ = cur.Q;
        myTEnd = blocks[3 * (nBlocks - 1) + 1] + blocks[3 * (nBlocks - 1) + 2];
        myTLen = batch.t_len[task];
        myTSeq = batch.seq + batch.t_off[task];
        myArrows = args.arrows + args.arrowOff[task];
        myPlo = cur.tStart - 1;
        myPhi = myPlo + guide_row0_tpost(cur);
      }
    }
  }
  // a value / a pointer of candidate g, uniform, from its first lane
  auto from_group = [&](int v, int g) { return __builtin_amdgcn_readlane(v, g * kL); };
  auto ptr_from_group = [&](const void *p, int g) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, g * kL);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), g * kL);
    return ((uint64_t)hi << 32) | lo;
  };
  int2 *ring = ringLds[wave][grp];
  uint32_t *ringT = ringTLds[wave][grp];

  const int inf = 1 << 30;
  const int kIns = args.ins << kShift, kDel = args.del << kShift, kOpen = args.open << kShift, kExt = args.ext << kShift;
  int pDel[kC], pExt[kC];
#pragma unroll
  for (int j = 0; j < kC; j++) {
    pDel[j] = (kC * l + j) * kDel;
    pExt[j] = (kC * l + j) * kExt;
  }
  // ---- row 0 (AffineGuidedAlign.h:166-208): S = del * distance from tStart-1, I = open
  if (kHybrid && myTask >= 0 && myPhi - myPlo + 1 > kF4MaxWidth) {  // a wide row 0 goes straight to the global ring
    for (int c = myPlo + l; c <= myPhi; c += kL) myGRing[c & myGMask] = make_int2((c - myPlo) * kDel, kOpen);
    myPrevGlobal = true;
  } else if (myTask >= 0) {
    for (int c = myPlo + l; c <= myPhi; c += kL) ring[c & (kF4Ring - 1)] = make_int2((c - myPlo) * kDel, kOpen);
  }
  myTLoaded = myPlo < 0 ? 0 : myPlo;
  if (kHybrid) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  RowConsts<true> kc;  // the wide rows' constants (row_chunk)
  kc.inf = inf;
  kc.ins = kIns;
  kc.del = kDel;
  kc.open = kOpen;
  kc.ext = kExt;
  kc.laneDel = lane * kDel;
  kc.laneExt = lane * kExt;

  int carryA = inf, carryB = inf, carryDiag = inf, eaKeep = inf, ebKeep = inf;
  for (int r0 = 1; r0 <= maxQ; r0 += 64) {
    // ---- the row records and query codes of the next 64 rows of every candidate (geometry_rows_kernel wrote them)
    int passesV = 1;  // lane i: the passes row r0+i takes, i.e. what its widest candidate needs
    bool spanBad = false;  // some candidate's 64 rows span more target columns than its packed-column ring holds
    int myHmax = 0;        // the last column my candidate touches in these rows
    bool nV = false;       // some query base of these rows is not ACGT
#pragma unroll
    for (int g = 0; g < kT; g++) {
      const int qOfG = from_group(myQ, g);
      if (from_group(myTask, g) >= 0 && r0 <= qOfG) {
        const int nRows = min(64, qOfG - r0 + 1);
        const int64_t rowOffOfG = (int64_t)ptr_from_group((const void *)(uint64_t)myRowOff, g);
        const int2 *rowTabOfG = args.rowTab + rowOffOfG;
        const unsigned short *rowMetaOfG = args.rowMeta + rowOffOfG;
        int lastV = INT32_MIN, loV = 0;
        if (lane < nRows) {
          const int2 rec = rowTabOfG[r0 + lane];
          const unsigned m = rowMetaOfG[r0 + lane];
          int width = (int)(m & 0xfffu);
          const int qc = (int)(m >> 12);
          loV = rec.y;
          if (kHybrid && width == 0xfff) {  // saturated: the row's cells are what lies before the next row's first one
            const unsigned next = r0 + lane < qOfG ? (unsigned)rowTabOfG[r0 + lane + 1].x : (unsigned)from_group((int)myCells, g);
            const int hi = loV + (int)(next - (unsigned)rec.x) - 1;
            width = min(hi, from_group(myTEnd, g) - 1) - loV + 1;
          }
          rowA[wave][g][lane] = rec;
          rowB[wave][g][lane] = (RowB)((unsigned)width | ((unsigned)qc << kQcShift));
          if (!kHybrid || width <= kF4MaxWidth) {
            passesV = max(passesV, (width + kF4Pass - 1) / kF4Pass);
            lastV = loV + width - 1;
          }
          nV = nV || qc >= 4;
        }
        const int hMax = wave_max(lastV);
        spanBad = spanBad || hMax + kL - __builtin_amdgcn_readlane(loV, 0) > kF4RingT;
        if (grp == g) {
          myHmax = hMax;
          // columns left of the batch's first row are never read again (the rows' first columns only grow)
          if (kHybrid) myTLoaded = max(myTLoaded, __builtin_amdgcn_readlane(loV, 0));
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- the packed matrix columns of the target for all 64 rows at once when they fit the ring next to the
    // first row's columns (nearly always); otherwise row by row below
    const bool ahead = !spanBad;
    const bool batchHasN = __ballot(nV) != 0ull;
    if (ahead) {
      const bool on = myTask >= 0 && r0 <= myQ;
      while (__ballot(on && myTLoaded <= myHmax) != 0ull) {
        if (on && myTLoaded <= myHmax) {
          const int c = myTLoaded + l;
          const int code = c < myTLen ? base_code(myTSeq[c]) : 4;
          ringT[c & (kF4RingT - 1)] = packedCol[code];
          myTLoaded += kL;
        }
      }
    }
    const int nRowsMax = min(64, maxQ - r0 + 1);
    for (int i = 0; i < nRowsMax; i++) {
      const int r = r0 + i;
      const int2 rec = rowA[wave][grp][i];
      const unsigned wb = rowB[wave][grp][i];
      const int lo = rec.y, width = (int)(wb & kWidthMask), qc = (int)(wb >> kQcShift);
      const int hiC = lo + width - 1;
      const bool isWide = kHybrid && width > kF4MaxWidth;
      const bool rowAll = myTask >= 0 && r <= myQ && (!kHybrid || r > mySkipUntil);
      if (kHybrid) {
        const unsigned long long wideMask = __ballot(rowAll && isWide);
        if (wideMask != 0ull && args.exp != 2) {
#pragma unroll 1  // (one copy of the wide paths, not one per candidate slot: a quarter of the kernel's code)
          for (int g = 0; g < kT; g++) {
            if (((wideMask >> (g * kL)) & 1ull) == 0ull) continue;
            // ---- one wide row of candidate g, the whole wavefront on it (fill_kernel's ring path)
            const int src = g * kL;
            const int wLo = __builtin_amdgcn_readlane(lo, src), wHi = __builtin_amdgcn_readlane(hiC, src);
            const int wPlo = __builtin_amdgcn_readlane(myPlo, src), wPhi = __builtin_amdgcn_readlane(myPhi, src);
            const int wQc = __builtin_amdgcn_readlane(qc, src);
            const unsigned wRow = (unsigned)__builtin_amdgcn_readlane(rec.x, src);
            const bool wGlobal = __builtin_amdgcn_readlane((int)myPrevGlobal, src) != 0;
            // (explicitly global: as generic pointers these become FLAT accesses, whose lgkmcnt the LDS reads of the
            //  step then wait on -- which would serialise the prefetch of the next step behind this one)
            typedef unsigned long long __attribute__((address_space(1))) GPair;  // {S, I} of a ring entry
            typedef uint8_t __attribute__((address_space(1))) GByte;
            GPair *gring = (GPair *)ptr_from_group(myGRing, g);
            auto pair_of = [](int2 v) { return (unsigned long long)(uint32_t)v.x | ((unsigned long long)(uint32_t)v.y << 32); };
            auto int2_of = [](unsigned long long p) { return make_int2((int)(uint32_t)p, (int)(uint32_t)(p >> 32)); };
            const int gmask = from_group(myGMask, g);
            const int tLenOfG = from_group(myTLen, g);
            const uint64_t arrowsOfG = ptr_from_group(myArrows, g), tSeqOfG = ptr_from_group(myTSeq, g);
            // ---- A RUN of wide rows, the row above in REGISTERS.  The wide rows of a guide come in runs -- the gap rows
            // between two blocks (GuidedAlign.h:193-250), a few hundred columns each, whose first column moves by 0 or 1
            // from row to row.  Through the global ring every one of them waits for the stores of the row before it (a
            // round trip through L2 per row, three candidates of the wavefront idle meanwhile).  Here the whole run is
            // taken at once, ahead of the other candidates' turn: three consecutive cells per lane, 192 columns per step
            // (the narrow pass's arithmetic on 64 lanes), up to three steps per row, the row above held as cell (step,
            // lane, j) = column base + 3 (64 step + lane) + j and handed on by a lane shift.  A row wider than 576
            // columns, or one that starts more than one column further on, ends the run and takes the ring path below.
            {
#ifdef BLASR_WIDE5  // experiment (make EXTRA=-DBLASR_WIDE5): five cells per lane, two steps of 320 columns
              constexpr int kW = 5, kWStep = 64 * kW, kRunSteps = 2, kRunMax = kRunSteps * kWStep;
#else
              constexpr int kW = 3, kWStep = 64 * kW, kRunSteps = 3, kRunMax = kRunSteps * kWStep;
#endif
              const int nG = min(64, from_group(myQ, g) - r0 + 1);  // rows of candidate g in this batch
              int run = 0;
              if (wHi - wLo + 1 <= kRunMax && wPhi - wLo + 1 <= kRunMax && args.exp != 3) {
                int pS[kRunSteps][kW], pI[kRunSteps][kW];
                int base = wLo, prevLo = wPlo, prevHi = wPhi;
#pragma unroll
                for (int k = 0; k < kRunSteps; k++)
#pragma unroll
                  for (int j = 0; j < kW; j++) {
                    const int c = wLo + kW * (64 * k + lane) + j;
                    const bool valid = c >= wPlo && c <= wPhi;
                    int2 v = make_int2(inf, inf);
                    if (wGlobal) v = int2_of(gring[c & gmask]);
                    else if (k == 0) v = ringLds[wave][g][c & (kF4Ring - 1)];  // (a narrow row ends within the first step)
                    pS[k][j] = valid ? v.x : inf;
                    pI[k][j] = valid ? v.y : inf;
                  }
                int leftS = inf;  // S of the row above at column base - 1
                if (wLo - 1 >= wPlo && wLo - 1 <= wPhi)
                  leftS = wGlobal ? (int)(uint32_t)gring[(wLo - 1) & gmask] : ringLds[wave][g][(wLo - 1) & (kF4Ring - 1)].x;
                leftS = __builtin_amdgcn_readfirstlane(leftS);
                GByte *aBase = (GByte *)arrowsOfG;
                const GByte *wT = (const GByte *)tSeqOfG;
                const int wTLen = tLenOfG, tLast = max(wTLen - 1, 0);
                // The target bases of the 576 columns from `base` on, three to a register like the cells (byte j of
                // tP[step] = the base of cell (step, lane, j)), loaded once per run and handed on by the same lane shift
                // as the row above: a load at the head of every step would expose its whole latency, once per step.
                uint64_t tP[kRunSteps];
#pragma unroll
                for (int k = 0; k < kRunSteps; k++) {
                  uint32_t b[kW];
#pragma unroll
                  for (int j = 0; j < kW; j++) {
                    const int c = wLo + kW * (64 * k + lane) + j;
                    b[j] = (uint32_t)wT[min(max(c, 0), tLast)];
                    b[j] = (c >= 0 && c < wTLen) ? b[j] : (uint32_t)'N';  // (beyond the target: N, decided here once)
                  }
                  tP[k] = 0;
#pragma unroll
                  for (int j = 0; j < kW; j++) tP[k] |= (uint64_t)b[j] << (8 * j);
                }
                for (int ii = i; ii < nG; ii++) {
                  const int2 rrec = rowA[wave][g][ii];
                  const unsigned rwb = rowB[wave][g][ii];
                  const int rLo = __builtin_amdgcn_readfirstlane(rrec.y);
                  const int rWidth = __builtin_amdgcn_readfirstlane((int)(rwb & kWidthMask));
                  const int rQc = __builtin_amdgcn_readfirstlane((int)(rwb >> kQcShift));
                  const int rHi = rLo + rWidth - 1, d = rLo - base;
                  if (rWidth <= kF4MaxWidth || rWidth > kRunMax || d < 0 || d > 1) break;
                  GByte *rowArrows = aBase + (unsigned)__builtin_amdgcn_readfirstlane(rrec.x);
                  if (d == 1) {  // the window of bases moves one column on: column base + 576 enters at its end
                    uint32_t entering = (uint32_t)wT[min(max(base + kRunMax, 0), tLast)];
                    entering = (base + kRunMax >= 0 && base + kRunMax < wTLen) ? entering : (uint32_t)'N';
                    uint32_t first[kRunSteps + 1];
#pragma unroll
                    for (int k = 0; k < kRunSteps; k++) first[k] = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tP[k], 0) & 0xffu;
                    first[kRunSteps] = entering;
#pragma unroll
                    for (int k = 0; k < kRunSteps; k++)
                      tP[k] = (tP[k] >> 8) | ((uint64_t)((uint32_t)wave_shl1((int)(uint32_t)tP[k], (int)first[k + 1]) & 0xffu) << (8 * (kW - 1)));
                  }
                  // the values that cross a step's edge are read off just before they are needed: the last cell of the
                  // step before (d = 0; saved before that step overwrote it) or the first cell of the step after (d = 1;
                  // not yet overwritten) -- one or two uniform values alive instead of eleven
                  int ePrevOld = inf;
                  int wCarryA = inf, wCarryB = inf;
#pragma unroll
                  for (int k = 0; k < kRunSteps; k++) {
                    if (rLo + kWStep * k > rHi) break;
                    const int c0 = rLo + kW * (64 * k + lane);
                    int upS[kW], upI[kW], diagS[kW];
                    if (d == 0) {
                      const int eOld = __builtin_amdgcn_readlane(pS[k][kW - 1], 63);
                      int dg = wave_shr1(pS[k][kW - 1], inf);
                      dg = lane == 0 ? (k == 0 ? leftS : ePrevOld) : dg;
                      ePrevOld = eOld;
                      diagS[0] = dg;
#pragma unroll
                      for (int j = 0; j < kW; j++) {
                        upS[j] = pS[k][j];
                        upI[j] = pI[k][j];
                        if (j > 0) diagS[j] = pS[k][j - 1];
                      }
                    } else {
#pragma unroll
                      for (int j = 0; j < kW; j++) {
                        diagS[j] = pS[k][j];
                        if (j + 1 < kW) {
                          upS[j] = pS[k][j + 1];
                          upI[j] = pI[k][j + 1];
                        }
                      }
                      upS[kW - 1] = wave_shl1(pS[k][0], k + 1 < kRunSteps ? __builtin_amdgcn_readlane(pS[k + 1 < kRunSteps ? k + 1 : k][0], 0) : inf);
                      upI[kW - 1] = wave_shl1(pI[k][0], k + 1 < kRunSteps ? __builtin_amdgcn_readlane(pI[k + 1 < kRunSteps ? k + 1 : k][0], 0) : inf);
                    }
                    // cells of the row above that do not exist: only a step that reaches beyond that row's ends has any
                    if (!(rLo + kWStep * k - 1 >= prevLo && rLo + kWStep * (k + 1) - 1 <= prevHi)) {
#pragma unroll
                      for (int j = 0; j < kW; j++) {
                        const int c = c0 + j;
                        const bool upValid = c <= prevHi, diagValid = c - 1 <= prevHi && c - 1 >= prevLo;  // (c >= rLo >= prevLo)
                        upS[j] = upValid ? upS[j] : inf;
                        upI[j] = upValid ? upI[j] : inf;
                        diagS[j] = diagValid ? diagS[j] : inf;
                      }
                    }
                    int m[kW], vKey[kW], insEK[kW], preA[kW], preB[kW];
#pragma unroll
                    for (int j = 0; j < kW; j++) {
                      const int tCur = (int)((tP[k] >> (8 * j)) & 0xffu);
                      if (rQc < 4) m[j] = (int)__builtin_amdgcn_sbfe((int)packedByChar[tCur], (rQc & 3) * 8, 8);  // (uniform)
                      else m[j] = matN[base_code((unsigned)tCur)];
                      const int matchK = (m[j] << kShift) + diagS[j];
                      const int insSK = upS[j] + (kIns + AR_UP);
                      insEK[j] = upI[j] + (kExt + AR_INS_CLOSE);
                      vKey[j] = min(matchK, min(insSK, insEK[j]));
                      const int v = vKey[j] & ~kTagMask;
                      const int ia = v - (kW * lane + j) * kDel;
                      const int ib = kExt0 ? v : v - (kW * lane + j) * kExt;
                      preA[j] = j == 0 ? ia : min(preA[j - 1], ia);
                      preB[j] = j == 0 ? ib : min(preB[j - 1], ib);
                    }
                    int pa = preA[kW - 1], pb = preB[kW - 1];
                    wave_prefix_min2(pa, pb);
                    int ea = wave_shr1(pa, inf), eb = wave_shr1(pb, inf);
                    if (k > 0) {
                      ea = min(ea, wCarryA);
                      eb = min(eb, wCarryB);
                    }
                    uint32_t arrowLo = 0, arrowHi = 0;  // (cells 0 .. kW - 2 in arrowLo, the last one in arrowHi)
#pragma unroll
                    for (int j = 0; j < kW; j++) {
                      const int eaj = j == 0 ? ea : min(ea, preA[j - 1]);
                      const int ebj = j == 0 ? eb : min(eb, preB[j - 1]);
                      const int aK = eaj + ((kW * lane + j) * kDel + AR_LEFT);                              // S[c-1] + del
                      const int bK = ebj + ((kExt0 ? 0 : (kW * lane + j) * kExt) + kOpen + AR_DEL_CLOSE);  // D[c-1] + ext
                      const int sKey = min(vKey[j], min(aK, bK));
                      const int sv = sKey & ~kTagMask;
                      const int openKey = sv + (kOpen + AR_I_OPEN);
                      const int iKey = min(openKey, kExt0 ? upI[j] : insEK[j] & ~kTagMask);
                      const int dFlag = openKey < bK ? AR_D_OPEN : 0;
                      const uint32_t arrow = (uint32_t)(sKey | iKey | dFlag) & kTagMask;
                      if (j < kW - 1) arrowLo |= arrow << (8 * j); else arrowHi = arrow;
                      pS[k][j] = sv;  // (cells beyond the row's end hold what no later row takes for a cell: prevHi)
                      pI[k][j] = iKey & ~kTagMask;
                    }
                    if (c0 <= rHi) {  // three arrows per lane; those beyond the row's end land in the next row's space
                      typedef unaligned_u16 __attribute__((address_space(1))) GU16;
                      GByte *ap = rowArrows + (c0 - rLo);
                      if (kW == 3) {
                        *(GU16 *)ap = (uint16_t)arrowLo;
                      } else {
                        typedef unaligned_u32 __attribute__((address_space(1))) GU32;
                        *(GU32 *)ap = arrowLo;
                      }
                      ap[kW - 1] = (uint8_t)arrowHi;
                    }
                    wCarryA = min(wCarryA, __builtin_amdgcn_readlane(pa, 63)) + kWStep * kDel;
                    wCarryB = min(wCarryB, __builtin_amdgcn_readlane(pb, 63)) + kWStep * kExt;
                  }
                  base = prevLo = rLo;
                  prevHi = rHi;
                  leftS = inf;  // (column base - 1 lies left of the row just made)
                  run++;
                }
                if (run > 0) {
                  // ---- the run's last row: to where the next row will look for it
                  bool toNarrow = false;
                  int keepLo = prevLo, keepHi = prevHi;
                  if (i + run < nG) {
                    const int nLo = __builtin_amdgcn_readfirstlane(rowA[wave][g][i + run].y);
                    const int nWidth = __builtin_amdgcn_readfirstlane((int)((unsigned)rowB[wave][g][i + run] & kWidthMask));
                    if (nWidth <= kF4MaxWidth) {
                      toNarrow = true;
                      keepLo = max(keepLo, nLo - 1);
                      keepHi = min(keepHi, nLo + nWidth - 1);
                    }
                  }
#pragma unroll
                  for (int k = 0; k < kRunSteps; k++)
#pragma unroll
                    for (int j = 0; j < kW; j++) {
                      const int c = base + kW * (64 * k + lane) + j;
                      if (c >= keepLo && c <= keepHi) {
                        if (toNarrow) ringLds[wave][g][c & (kF4Ring - 1)] = make_int2(pS[k][j], pI[k][j]);
                        else gring[c & gmask] = pair_of(make_int2(pS[k][j], pI[k][j]));
                      }
                    }
                  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                  __builtin_amdgcn_wave_barrier();
                  if (grp == g) {
                    myPlo = prevLo;
                    myPhi = prevHi;
                    myPrevGlobal = !toNarrow;
                    mySkipUntil = r0 + i + run - 1;
                  }
                  continue;  // (the next candidate with a wide row at this index, if any)
                }
              }
            }
            if (!wGlobal) {  // the row above is a narrow one: from the LDS ring into the global ring
              for (int c = wPlo + lane; c <= wPhi; c += 64) gring[c & gmask] = pair_of(ringLds[wave][g][c & (kF4Ring - 1)]);
              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            // Which of this row's {S, I} the next row can read: when that row is a narrow one of the same batch, only
            // its own columns (and one to their left); the candidate's last row only feeds the final score.  The
            // others are not stored -- most of a wide row, which nothing below it ever looks at.
            int keepLo = INT32_MIN, keepHi = INT32_MAX;
            if (r == from_group(myQ, g)) {
              keepLo = keepHi = from_group(myTEnd, g) - 1;
            } else if (i + 1 < 64) {
              const int nLo = rowA[wave][g][i + 1].y;
              const int nWidth = (int)((unsigned)rowB[wave][g][i + 1] & kWidthMask);
              if (nWidth <= kF4MaxWidth) {
                keepLo = nLo - 1;
                keepHi = nLo + nWidth - 1;
              }
            }
            GByte *wArrows = (GByte *)(arrowsOfG + wRow);
            const GByte *wT = (const GByte *)tSeqOfG;
            const int wTLen = tLenOfG;
            ChunkCarry carry;
            carry.a = carry.b = carry.s = inf;
            int cDiag = 0;
            // the target bases and the row above run one step ahead of their use (nothing a step stores is read
            // by a later step of the same row, bar column cb-1, which travels in cDiag).  Two steps per trip with two
            // sets of registers taking turns: with ONE set the copy "current = next" at the end of a step made the
            // compiler wait there for the loads the step had just issued (s_waitcnt vmcnt(0) in every step).
            // The loads of the step after are issued UNCONDITIONALLY (clamped target index, any ring slot is memory of
            // the ring) and what they returned is judged when it is used: with loads behind branches the compiler cannot
            // count what is in flight and waits for everything (vmcnt(0)) right after issuing the prefetch.
            const int tLast = max(wTLen - 1, 0);
            int tA = (int)wT[min(max(wLo + lane, 0), tLast)], tB = 'N';
            int2 upA = int2_of(gring[(wLo + lane) & gmask]), upB = make_int2(inf, inf);
            // S of column wLo - 1, the one diagonal neighbour that is nobody's "up": lane 0 carries it into the row
            if (wLo <= wPhi + 1 && lane == 0) cDiag = (int)(uint32_t)gring[(wLo - 1) & gmask];
            auto wideStep = [&](const int cb, const int tRaw, int2 up, int &tNext, int2 &upNext) {
              const int c = cb + lane;
              const bool active = c <= wHi;
              int sd = inf;
              tNext = (int)wT[min(max(c + 64, 0), tLast)];
              upNext = int2_of(gring[(c + 64) & gmask]);
              const int tCur = (c >= 0 && c < wTLen) ? tRaw : 'N';
              if (cb <= wPhi + 1) {
                // S of the row above one column to the left = the left neighbour's "up" (column cb - 1: what the
                // previous step -- or the load above -- left in cDiag; the ring entry itself is overwritten by then)
                sd = wave_shr1(up.x, inf);
                sd = (lane == 0) ? __builtin_amdgcn_readfirstlane(cDiag) : sd;
                cDiag = __builtin_amdgcn_readlane(up.x, 63);
                const bool upValid = c <= wPhi, diagValid = (c - 1 <= wPhi) && (c > wPlo);
                up.x = upValid ? up.x : inf;
                up.y = upValid ? up.y : inf;
                sd = diagValid ? sd : inf;
              } else {  // beyond the row above there is nothing: whatever the ring slot held is not it
                up = make_int2(inf, inf);
              }
              const int mm = wQc < 4 ? (int)__builtin_amdgcn_sbfe((int)packedByChar[tCur & 255], (wQc & 3) * 8, 8)
                                     : matN[base_code((unsigned)tCur)];
              int sv, iv, arrow;
              row_chunk<true, false>(kc, carry, active, up.x, up.y, sd, mm, sv, iv, arrow);
              if (active) {
                if (c >= keepLo && c <= keepHi) gring[c & gmask] = pair_of(make_int2(sv, iv));
                wArrows[c - wLo] = (uint8_t)arrow;
              }
            };
            for (int cb = wLo; cb <= wHi && args.exp != 1; cb += 128) {
              wideStep(cb, tA, upA, tB, upB);
              if (cb + 64 <= wHi) wideStep(cb + 64, tB, upB, tA, upA);
            }
            // (this row's stores are read by the next row only: one wait for all of them)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (grp == g) {
              myPlo = wLo;
              myPhi = wHi;
              myPrevGlobal = true;
            }
          }
        }
        // ---- back to narrow rows: the columns this row reads of the wide row above, into the LDS ring
        if (__ballot(rowAll && !isWide && myPrevGlobal) != 0ull && args.exp != 2) {
          if (rowAll && !isWide && myPrevGlobal) {
            for (int c = lo - 1 + l; c <= min(hiC, myPhi); c += kL)
              if (c >= myPlo) ring[c & (kF4Ring - 1)] = myGRing[c & myGMask];
            myPrevGlobal = false;
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
      const bool rowOn = rowAll && !isWide;
      if (kHybrid && rowOn) myTLoaded = max(myTLoaded, lo);
      // ---- stage the packed matrix columns of the target up to this row's last column
      while (!ahead && __ballot(rowOn && myTLoaded <= hiC) != 0ull) {
        if (rowOn && myTLoaded <= hiC) {
          const int c = myTLoaded + l;
          const int code = c < myTLen ? base_code(myTSeq[c]) : 4;
          ringT[c & (kF4RingT - 1)] = packedCol[code];
          myTLoaded += kL;
        }
      }
      __builtin_amdgcn_wave_barrier();
      const int nPass = __builtin_amdgcn_readlane(passesV, i);
      // (carryA/B/Diag, declared before the row loops, are only read in passes after a row's first)
      uint8_t *arrowRow = myArrows + (unsigned)rec.x;
      for (int pass = 0; pass < nPass; pass++) {
        const int cb = lo + kF4Pass * pass, c0 = cb + kC * l;
        // ---- the row above, by column.  Ring entries right of the previous row's last cell are stale.
        int upS[kC], upI[kC], diagS[kC];
        const int2 dRaw = ring[(c0 - 1) & (kF4Ring - 1)];
        int d0 = dRaw.x;
        if (pass > 0) d0 = (l == 0) ? carryDiag : d0;  // column cb-1 was overwritten by the previous pass
        d0 = (c0 - 1 <= myPhi && c0 - 1 >= myPlo) ? d0 : inf;
        int lastRaw = 0;
#pragma unroll
        for (int j = 0; j < kC; j++) {
          const int c = c0 + j;
          const int2 up = ring[c & (kF4Ring - 1)];
          if (j == kC - 1) lastRaw = up.x;
          const bool upValid = c <= myPhi;  // c >= lo >= plo always
          upS[j] = upValid ? up.x : inf;
          upI[j] = upValid ? up.y : inf;
        }
        diagS[0] = d0;
#pragma unroll
        for (int j = 1; j < kC; j++) diagS[j] = upS[j - 1];
        // ---- substitution scores
        int m[kC];
#pragma unroll
        for (int j = 0; j < kC; j++) {
          const uint32_t te = ringT[(c0 + j) & (kF4RingT - 1)];
          m[j] = (int)__builtin_amdgcn_sbfe((int)te, (qc & 3) * 8, 8);
        }
        if (batchHasN && __ballot(rowOn && qc >= 4) != 0ull) {  // query N (rare): the packed column has no N row
#pragma unroll
          for (int j = 0; j < kC; j++)
            if (rowOn && qc >= 4) {
              const int c = c0 + j;
              const int tcode = (c >= 0 && c < myTLen) ? base_code(myTSeq[c]) : 4;
              m[j] = matN[tcode];
            }
        }
        // ---- vertical candidates and the two horizontal chains.  Cells beyond the row's end are not masked:
        // they lie right of every real cell, the horizontal minima only flow left to right, and what they leave
        // in the ring is never read as a cell (the next row's validity test above).
        int vKey[kC], insEK[kC], ia[kC], ib[kC], preA[kC], preB[kC];
#pragma unroll
        for (int j = 0; j < kC; j++) {
          const int matchK = (m[j] << kShift) + diagS[j];
          const int insSK = upS[j] + (kIns + AR_UP);
          insEK[j] = upI[j] + (kExt + AR_INS_CLOSE);
          vKey[j] = min(matchK, min(insSK, insEK[j]));
          const int v = vKey[j] & ~kTagMask;
          ia[j] = v - pDel[j];
          ib[j] = kExt0 ? v : v - pExt[j];
          preA[j] = j == 0 ? ia[0] : min(preA[j - 1], ia[j]);  // inclusive over this lane's cells
          preB[j] = j == 0 ? ib[0] : min(preB[j - 1], ib[j]);
        }
        int pa = preA[kC - 1], pb = preB[kC - 1];
        group_prefix_min2<kL>(pa, pb);  // inclusive over the lanes of the candidate
        // lanes before this one; lane 0 of a DPP row has no source and keeps the old value of the destination,
        // which therefore stays the "inf" it was given before the loops
        eaKeep = dpp<0x111, 0xf>(eaKeep, pa);
        ebKeep = dpp<0x111, 0xf>(ebKeep, pb);
        int ea = eaKeep, eb = ebKeep;
        if (kL < 16) {
          ea = l == 0 ? inf : ea;
          eb = l == 0 ? inf : eb;
        }
        if (pass > 0) {  // (no cap at "inf" otherwise: what exceeds it loses every comparison against a real cell)
          ea = min(ea, carryA);
          eb = min(eb, carryB);
        }
        uint32_t arrowLo = 0, arrowHi = 0;
#pragma unroll
        for (int j = 0; j < kC; j++) {
          const int eaj = j == 0 ? ea : min(ea, preA[j - 1]);
          const int ebj = j == 0 ? eb : min(eb, preB[j - 1]);
          const int aK = eaj + (pDel[j] + AR_LEFT);                 // S[c-1] + del
          const int bK = ebj + ((kExt0 ? 0 : pExt[j]) + kOpen + AR_DEL_CLOSE);    // D[c-1] + ext
          const int sKey = min(vKey[j], min(aK, bK));
          const int sv = sKey & ~kTagMask;
          const int openKey = sv + (kOpen + AR_I_OPEN);
          const int iKey = min(openKey, kExt0 ? upI[j] : insEK[j] & ~kTagMask);  // (ring entries carry no tag)
          const int dFlag = openKey < bK ? AR_D_OPEN : 0;  // = the tag of min(S + open, D + ext), see row_chunk
          // the arrow byte = the tags of the three keys: S tags live in bits 0-2, the open flags in bits 3-4, and
          // nothing else is set below kShift -- so it is the OR of the keys, masked once (after packing, kC == 3)
          const uint32_t arrow = (uint32_t)(sKey | iKey | dFlag);
          if (kC == 3) {
            if (j == 1) arrowLo = __builtin_amdgcn_perm(arrow, arrowLo, 0x0c0c0400u) & 0x1f1fu;  // {cell 0, cell 1}
            else if (j == 0) arrowLo = arrow;
            else arrowHi = arrow & kTagMask;                                                     // cell 2
          } else if (j < 4) {
            arrowLo |= (arrow & kTagMask) << (8 * j);
          } else {
            arrowHi |= (arrow & kTagMask) << (8 * (j - 4));
          }
          // (a third pass must not write beyond the row's end: those slots hold the row's first columns)
          if (rowOn && (pass < 2 || c0 + j <= hiC)) ring[(c0 + j) & (kF4Ring - 1)] = make_int2(sv, iKey & ~kTagMask);
        }
        // ---- arrows: kC consecutive bytes per lane.  Bytes of cells beyond the row's end spill into the
        // next row's space (or the slack after the last row) and are overwritten when that row is stored.
        if (rowOn && c0 <= hiC) {
          uint8_t *ap = arrowRow + (c0 - lo);
          if (kC == 3) {
            *(unaligned_u16 *)ap = (uint16_t)arrowLo;
            ap[2] = (uint8_t)arrowHi;
          } else {
            *(unaligned_u32 *)ap = arrowLo;
            *(unaligned_u16 *)(ap + 4) = (uint16_t)arrowHi;
          }
        }
        if (nPass > 1) {  // carries into the next 48 columns
          const int last = (lane | (kL - 1)) * 4;
          const int ta = __builtin_amdgcn_ds_bpermute(last, pa), tb = __builtin_amdgcn_ds_bpermute(last, pb);
          carryA = (pass > 0 ? min(carryA, ta) : ta) + kF4Pass * kDel;
          carryB = (pass > 0 ? min(carryB, tb) : tb) + kF4Pass * kExt;
          carryDiag = __builtin_amdgcn_ds_bpermute(last, lastRaw);  // the raw S of column cb+47, for lane 0
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (rowOn) {
        myPlo = lo;
        myPhi = hiC;
      }
    }
  }
  // ---- S[Q][tEnd-1] (:481-483): a candidate's ring is not written after its last row
  if (myTask >= 0 && l == 0 && myTEnd - 1 >= myPlo && myTEnd - 1 <= myPhi)
    args.dpScore[myTask] =
        (kHybrid && myPrevGlobal ? myGRing[(myTEnd - 1) & myGMask].x : ring[(myTEnd - 1) & (kF4Ring - 1)].x) >> kShift;
}
// ---------------------------------------------------------------------------
// fill8_kernel: the candidates of fillp_kernel<16, true, false> (every row <= kF4MaxWidth cells, extension penalty 0) once
// more, EIGHT to a wavefront and each AT ITS OWN ROW.
//
// 72 % of the narrow rows are the band's own 33 cells and 94 % have at most 40: eight lanes x five cells cover such a row in one
// pass with 7 idle slots where 16 x 3 has 15, and the per-pass bookkeeping (row record, scan, addresses, stores) is paid once for
// eight rows.  What made the 8-lane instantiation of fillp_kernel lose was the lockstep: a row beyond the pass width sent the
// whole wavefront through another pass, and with eight candidates 41 % of the row steps had one.  Here a candidate that needs
// another pass simply stays on its row while the others go on -- row, pass and carries are per-lane state, the loop runs until the
// last candidate is through (the list is sorted by rows, so they end together).
// LDS is what limits the resident wavefronts, so a candidate has exactly 1280 bytes of it (four workgroups per CU):
//   * the {S, I} ring of 128 columns (1024 bytes), updated in place as in fillp_kernel;
//   * the staging ring holds the target's CODES x 6, one byte per column (128 bytes); a row's substitution scores are then 6-bit
//     fields of ONE register, the matrix row of its query base (any query code, N included; matrix entries within [-32, 31],
//     checked by the caller);
//   * a window of 16 row records, 8 bytes each (128 bytes): {arrow offset, first column - tStart + 1 : 20 | width : 8 | query
//     code : 3} (target spans below 2^20, checked by the caller).
// The loop itself issues no loads from memory (a load inside it would wait for the arrow stores before it, every step): when one
// candidate has run out of records or of staged columns, ALL of them top up as far as their window / ring holds.
// The recurrence, the tagged keys and the arrow byte are fillp_kernel's.
// ---------------------------------------------------------------------------
constexpr int kF8Pass = 40;
constexpr int kF8Rows = 16;
constexpr int kF8Clean = 8;   // = the lanes of a candidate: one entry each
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void fill8_kernel(RefineBatchView batch, FillArgs args) {
  constexpr int kL = 8, kT = 8, kC = kF8Pass / kL, kPass = kF8Pass;
  __shared__ int2 ringLds[4][kT][kF4Ring];
  __shared__ uint8_t ringTLds[4][kT][kF4RingT];
  __shared__ uint2 rowWin[4][kT][kF8Rows];
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int grp = lane / kL, l = lane % kL;
  const int slot0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wave)) * kT;
  if (slot0 >= args.count) return;
  int2 *ring = ringLds[wave][grp];
  uint8_t *ringT = ringTLds[wave][grp];
  uint2 *win = rowWin[wave][grp];
  // lane qc (0..4, N = 4; any other code reads N's) holds the matrix row of query code qc, 6 bits per target code: a row's
  // register is fetched from that lane (ds_bpermute: the crossbar, no LDS behind it)
  uint32_t row6Table = 0;
#pragma unroll
  for (int tc = 0; tc < 5; tc++) row6Table |= (uint32_t)(args.matrix[min(lane, 4) * 5 + tc] & 0x3f) << (6 * tc);

  int myTask = -1, myQ = 0, myTEnd = 0, myTLen = 0, myPlo = 0, myPhi = -1, myBase = 0;
  const int2 *tab = nullptr;
  const unsigned short *meta = nullptr;
  const uint8_t *myTSeq = nullptr;
  uint8_t *myArrows = nullptr;
  if (slot0 + grp < args.count) myTask = args.taskList[slot0 + grp];
  if (myTask >= 0) {
    const int32_t *blocks = batch.blocks + 3 * batch.blk_off[myTask];
    const int nBlocks = batch.n_blocks[myTask];
    GuideCursor cur;
    guide_begin(cur, blocks, nBlocks, args.band);
    myQ = cur.Q;
    myTEnd = blocks[3 * (nBlocks - 1) + 1] + blocks[3 * (nBlocks - 1) + 2];
    myTLen = batch.t_len[myTask];
    myTSeq = batch.seq + batch.t_off[myTask];
    myArrows = args.arrows + args.arrowOff[myTask];
    tab = args.rowTab + args.rowOff[myTask];
    meta = args.rowMeta + args.rowOff[myTask];
    myBase = cur.tStart - 1;
    myPlo = myBase;
    myPhi = myPlo + guide_row0_tpost(cur);
  }
  const int inf = 1 << 30;
  const int kIns = args.ins << kShift, kDel = args.del << kShift, kOpen = args.open << kShift;
  int pDel[kC];
#pragma unroll
  for (int j = 0; j < kC; j++) pDel[j] = (kC * l + j) * kDel;
  // ---- row 0 (AffineGuidedAlign.h:166-208): S = del * distance from tStart-1, I = open
  // The kF8Clean entries behind a row's last cell are set to "no cell" when the row is done: the next row reads that far beyond
  // the row above without a test per cell (a row that reaches further sends the wavefront through the tests, below).
  if (myTask >= 0) {
    for (int c = myPlo + l; c <= myPhi; c += kL) ring[c & (kF4Ring - 1)] = make_int2((c - myPlo) * kDel, kOpen);
    ring[(myPhi + 1 + l) & (kF4Ring - 1)] = make_int2(inf, inf);
  }
  int myTLoaded = (myPlo < 0 ? 0 : myPlo) & ~3;  // columns below it are staged; whole dwords of codes
  const int tLast = max(myTLen - 1, 0);
  __builtin_amdgcn_wave_barrier();

  int myRow = 1, myPass = 0, stagedEnd = 1;  // records of rows below stagedEnd (and not below myRow) are in the window
  int carryA = inf, carryB = inf, carryDiag = inf, eaKeep = inf, ebKeep = inf;
  bool alive = myTask >= 0 && myRow <= myQ;
  uint2 recCur = make_uint2(0u, 0u);
  while (__ballot(alive) != 0ull) {
    // ---- row records: when one candidate has used up its window, all of them fill theirs
    if (__ballot(alive && myRow >= stagedEnd) != 0ull) {
      if (alive) {
        const int newEnd = min(myRow + kF8Rows, myQ + 1);
#pragma unroll
        for (int k = 0; k < kF8Rows / kL; k++) {
          const int r = stagedEnd + l + kL * k;
          if (r < newEnd) {
            const int2 rec = tab[r];
            const unsigned m = meta[r];
            win[r & (kF8Rows - 1)] =  // (the query code as the byte address of its lane in row6Table)
                make_uint2((unsigned)rec.x, (unsigned)(rec.y - myBase) | ((m & 0xffu) << 20) | (min(m >> 12, 4u) << 28));
          }
        }
        stagedEnd = newEnd;
      }
      __builtin_amdgcn_wave_barrier();
      recCur = win[myRow & (kF8Rows - 1)];
    }
    const uint2 recNext = win[(myRow + 1) & (kF8Rows - 1)];  // (the row after this one, if the window has it)
    const int lo = myBase + (int)(recCur.y & 0xfffffu), width = (int)((recCur.y >> 20) & 0xffu);
    const int hiC = lo + width - 1;
    // ---- the target's codes: when one candidate has run out, all of them stage as far as their ring holds
    if (__ballot(alive && hiC >= myTLoaded) != 0ull) {
      if (alive) {
        myTLoaded = max(myTLoaded, lo & ~3);
        const int limit = (lo & ~3) + kF4RingT;  // the first column whose slot still holds one this row may read
        const int c = myTLoaded + 16 * l;
        uint32_t b[16];
#pragma unroll
        for (int k = 0; k < 16; k++) b[k] = (uint32_t)myTSeq[min(c + k, tLast)];
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) {
          uint32_t w = 0;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int cc = c + 4 * k4 + k;
            const int code = cc < myTLen ? base_code(b[4 * k4 + k]) : 4;
            w |= (uint32_t)(code * 6) << (8 * k);
          }
          if (c + 4 * k4 < limit) *reinterpret_cast<uint32_t *>(&ringT[(c + 4 * k4) & (kF4RingT - 1)]) = w;
        }
        myTLoaded = min(myTLoaded + 16 * kL, limit);
      }
      __builtin_amdgcn_wave_barrier();
    }
    // ---- one pass of my candidate's row: columns cb .. cb + 39
    uint8_t *arrowRow = myArrows + recCur.x;
    const int cb = lo + kPass * myPass, c0 = cb + kC * l;
    int upS[kC], upI[kC], diagS[kC];
    int d0 = ring[(c0 - 1) & (kF4Ring - 1)].x;
    d0 = (myPass > 0 && l == 0) ? carryDiag : d0;  // column cb-1 was overwritten by the pass before
    d0 = c0 - 1 >= myPlo ? d0 : inf;               // (lane 0 of a row that starts where the row above did)
    int lastRaw = 0;
#pragma unroll
    for (int j = 0; j < kC; j++) {
      const int2 up = ring[(c0 + j) & (kF4Ring - 1)];
      if (j == kC - 1) lastRaw = up.x;
      upS[j] = up.x;
      upI[j] = up.y;
    }
    if (__ballot(alive && hiC > myPhi + kF8Clean) != 0ull) {  // (rare) entries beyond the cleaned ones are stale
      d0 = c0 - 1 <= myPhi ? d0 : inf;
#pragma unroll
      for (int j = 0; j < kC; j++) {
        const bool upValid = c0 + j <= myPhi;  // c >= lo >= plo always
        upS[j] = upValid ? upS[j] : inf;
        upI[j] = upValid ? upI[j] : inf;
      }
    }
    diagS[0] = d0;
#pragma unroll
    for (int j = 1; j < kC; j++) diagS[j] = upS[j - 1];
    const uint32_t row6 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((recCur.y >> 26) & 0x1cu), (int)row6Table);
    int vKey[kC], preA[kC], preB[kC];
#pragma unroll
    for (int j = 0; j < kC; j++) {
      const int m = (int)__builtin_amdgcn_sbfe((int)row6, (unsigned)ringT[(c0 + j) & (kF4RingT - 1)], 6);
      const int matchK = (m << kShift) + diagS[j];
      const int insSK = upS[j] + (kIns + AR_UP);
      const int insEK = upI[j] + AR_INS_CLOSE;
      vKey[j] = min(matchK, min(insSK, insEK));
      const int v = vKey[j] & ~kTagMask;
      const int ia = v - pDel[j];
      preA[j] = j == 0 ? ia : min(preA[j - 1], ia);  // inclusive over this lane's cells
      preB[j] = j == 0 ? v : min(preB[j - 1], v);
    }
    int pa = preA[kC - 1], pb = preB[kC - 1];
    group_prefix_min2<kL>(pa, pb);  // inclusive over the lanes of the candidate
    eaKeep = dpp<0x111, 0xf>(eaKeep, pa);
    ebKeep = dpp<0x111, 0xf>(ebKeep, pb);
    int ea = l == 0 ? inf : eaKeep, eb = l == 0 ? inf : ebKeep;
    ea = myPass > 0 ? min(ea, carryA) : ea;
    eb = myPass > 0 ? min(eb, carryB) : eb;
    uint32_t arrowLo = 0, arrowHi = 0;
#pragma unroll
    for (int j = 0; j < kC; j++) {
      const int eaj = j == 0 ? ea : min(ea, preA[j - 1]);
      const int ebj = j == 0 ? eb : min(eb, preB[j - 1]);
      const int aK = eaj + (pDel[j] + AR_LEFT);           // S[c-1] + del
      const int bK = ebj + (kOpen + AR_DEL_CLOSE);         // D[c-1] + ext
      const int sKey = min(vKey[j], min(aK, bK));
      const int sv = sKey & ~kTagMask;
      const int openKey = sv + (kOpen + AR_I_OPEN);
      const int iKey = min(openKey, upI[j]);               // (ring entries carry no tag)
      const int dFlag = openKey < bK ? AR_D_OPEN : 0;
      // the arrow byte = the tags of the three keys (S tags in bits 0-2, the open flags in bits 3-4, nothing else below
      // kShift): the OR of the keys, its low byte picked by v_perm, masked once per dword
      const uint32_t arrow = (uint32_t)(sKey | iKey | dFlag);
      if (j == 0) arrowLo = arrow;
      else if (j == 1) arrowLo = __builtin_amdgcn_perm(arrow, arrowLo, 0x0c0c0400u);   // {cell 0, cell 1, 0, 0}
      else if (j == 2) arrowHi = arrow;
      else if (j == 3) arrowLo = __builtin_amdgcn_perm(__builtin_amdgcn_perm(arrow, arrowHi, 0x04000c0cu), arrowLo, 0x07060100u) & 0x1f1f1f1fu;
      else arrowHi = arrow & kTagMask;
      if (alive) ring[(c0 + j) & (kF4Ring - 1)] = make_int2(sv, iKey & ~kTagMask);
    }
    // ---- arrows: five consecutive bytes per lane; bytes beyond the row's end spill into the next row's space (or the slack
    // after the last row) and are overwritten when that row is stored
    if (alive && c0 <= hiC) {
      uint8_t *ap = arrowRow + (c0 - lo);
      *(unaligned_u32 *)ap = arrowLo;
      ap[4] = (uint8_t)arrowHi;
    }
    const bool more = alive && cb + kPass <= hiC;
    if (alive && !more) ring[(hiC + 1 + l) & (kF4Ring - 1)] = make_int2(inf, inf);  // (after the cells' own writes)
    if (__ballot(more) != 0ull) {  // carries into the next 40 columns
      const int last = (lane | (kL - 1)) * 4;
      const int ta = __builtin_amdgcn_ds_bpermute(last, pa), tb = __builtin_amdgcn_ds_bpermute(last, pb);
      carryA = (myPass > 0 ? min(carryA, ta) : ta) + kPass * kDel;
      carryB = myPass > 0 ? min(carryB, tb) : tb;
      carryDiag = __builtin_amdgcn_ds_bpermute(last, lastRaw);  // the raw S of column cb+39, for lane 0
    }
    __builtin_amdgcn_wave_barrier();
    if (alive) {
      if (more) {
        myPass++;
      } else {
        myPass = 0;
        myRow++;
        myPlo = lo;
        myPhi = hiC;
        recCur = recNext;
      }
    }
    alive = alive && myRow <= myQ;
  }
  // ---- S[Q][tEnd-1] (:481-483): a candidate's ring is not written after its last row
  if (myTask >= 0 && l == 0 && myTEnd - 1 >= myPlo && myTEnd - 1 <= myPhi)
    args.dpScore[myTask] = ring[(myTEnd - 1) & (kF4Ring - 1)].x >> kShift;
}

// ---------------------------------------------------------------------------
// fillh8_kernel: the hybrid list (candidates with a row wider than kF4MaxWidth) in fill8_kernel's form.
//
// The hybrid instantiation of fillp_kernel steps its four candidates through their narrow rows in lockstep: 131 vector
// instructions per pass of four rows, and 40 % of those row steps carry a second pass because ONE of the four rows is wider
// than 48 columns (the hybrid list's narrow rows: 57 % at most 33 cells, 19 % 34-40, 4 % 41-48, 8 % 49-112) -- 0.97
// wavefront instructions per narrow cell where fill8_kernel spends 0.81.  Here the narrow rows ARE fill8_kernel's: eight
// candidates per wavefront, eight lanes x five cells each, every candidate at its own row and pass, 1280 bytes of LDS per
// candidate (ring, staged codes, a window of sixteen row records) and no load from memory inside the loop.  A candidate that
// meets a WIDE row hands the wavefront to the wide paths -- the other seven wait, as the other three did; the time the
// wavefront spends there is the same either way -- which are fillp_kernel's: a RUN of wide rows with the row above in
// registers (three cells per lane, 192 columns per step, up to 576), or one row through the candidate's ring in global memory.
// What those paths need of the following rows (a run is found by looking at them) comes from memory when the wavefront
// ENTERS the wide mode, lane = row, sixty-four records in one round trip, and is read off the lanes (v_readlane) -- the
// 8-lane layout has no LDS for fillp_kernel's 64-row batches.  A piece (a run, or one row) per entry; the candidate comes back
// to its narrow rows with the row above in its LDS ring, the eight entries behind it set to "no cell" as fill8_kernel's rows
// leave them.  The wide paths take their substitution scores like the narrow rows: 6-bit fields of ONE register, the matrix
// row of the query base (uniform in a wide row), indexed by the target's code x 6 -- no table in LDS, so the kernel has exactly
// fill8_kernel's 40 KB per workgroup.  Domain: fill8_kernel's (extension penalty 0, matrix entries within [-32, 31], target
// spans below 2^20); other parameters keep fillp_kernel<16, ., true>.
// ---------------------------------------------------------------------------
constexpr int kH8Wide = 255;  // the width field of a window record whose row is a wide one (the true width comes from memory)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void fillh8_kernel(RefineBatchView batch, FillArgs args) {
  constexpr int kL = 8, kT = 8, kC = kF8Pass / kL, kPass = kF8Pass;
  __shared__ int2 ringLds[4][kT][kF4Ring];
  __shared__ uint8_t ringTLds[4][kT][kF4RingT];
  __shared__ uint2 rowWin[4][kT][kF8Rows];
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int grp = lane / kL, l = lane % kL;
  int2 *ring = ringLds[wave][grp];
  uint8_t *ringT = ringTLds[wave][grp];
  uint2 *win = rowWin[wave][grp];
  // lane qc (0..4, N = 4) holds the matrix row of query code qc, 6 bits per target code (fill8_kernel)
  uint32_t row6Table = 0;
#pragma unroll
  for (int tc = 0; tc < 5; tc++) row6Table |= (uint32_t)(args.matrix[min(lane, 4) * 5 + tc] & 0x3f) << (6 * tc);

  int myTask = -1, myQ = 0, myTEnd = 0, myTLen = 0, myPlo = 0, myPhi = -1, myBase = 0, myGMask = 0;
  unsigned myCells = 0;
  bool myPrevGlobal = false;  // the row above lives in the candidate's ring in global memory
  const int2 *tab = nullptr;
  const unsigned short *meta = nullptr;
  const uint8_t *myTSeq = nullptr;
  uint8_t *myArrows = nullptr;
  int2 *myGRing = nullptr;
  const int inf = 1 << 30;
  const int kIns = args.ins << kShift, kDel = args.del << kShift, kOpen = args.open << kShift;
  int pDel[kC];
#pragma unroll
  for (int j = 0; j < kC; j++) pDel[j] = (kC * l + j) * kDel;
  int myTLoaded = 0, tLast = 0;
  // a value / a pointer of candidate g, uniform, from its first lane
  auto from_group = [&](int v, int g) { return __builtin_amdgcn_readlane(v, g * kL); };
  auto ptr_from_group = [&](const void *p, int g) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo32 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, g * kL);
    const uint32_t hi32 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), g * kL);
    return ((uint64_t)hi32 << 32) | lo32;
  };

  int myRow = 1, myPass = 0, stagedEnd = 1;  // records of rows below stagedEnd (and not below myRow) are in the window
  int carryA = inf, carryB = inf, carryDiag = inf, eaKeep = inf, ebKeep = inf;
  bool alive = false;
  // The list is a QUEUE (args.queue counts what was handed out): a group of eight lanes that is through with its candidate
  // takes the next one, whatever the other seven groups of its wavefront are doing.  A wavefront's narrow rows then never wait
  // for its longest candidate (with eight fixed candidates per wavefront a fifth of the narrow lane-passes of configs[2] were
  // idle ones) and the launch has no last round of half-empty compute units: the wavefronts end together, when the list does.
  // The list is sorted by cells, largest first.  args.groups (1 .. 8): the groups of a wavefront that take candidates -- a list
  // too short to give every resident wavefront eight uses fewer, so that wide rows of different candidates do not queue up
  // behind each other in one wavefront while other compute units are empty.
  bool drained = grp >= args.groups;
  bool firstFetch = true;
  uint2 recCur = make_uint2(0u, 0u);
  for (;;) {
    if (__ballot(!alive && !drained) != 0ull) {
      if (!alive && !drained) {
        // ---- S[Q][tEnd-1] of the candidate that is done (:481-483): its ring is not written after its last row
        if (myTask >= 0 && l == 0 && myTEnd - 1 >= myPlo && myTEnd - 1 <= myPhi)
          args.dpScore[myTask] =
              (myPrevGlobal ? myGRing[(myTEnd - 1) & myGMask].x : ring[(myTEnd - 1) & (kF4Ring - 1)].x) >> kShift;
        myTask = -1;
        // The FIRST candidate of a group has a fixed place in the list -- wavefront w, group g: slot w * groups + g -- so that
        // the host can deal the heaviest candidates to different wavefronts (a wavefront takes its candidates' wide rows one
        // after the other: with the eight heaviest in one of them a list of a few ten thousand candidates took as long as
        // that wavefront); the others come from the queue, which starts behind the first round.
        int slot = 0;
        if (firstFetch) {
          slot = (int)(blockIdx.x * 4 + wave) * args.groups + grp;
          if (slot >= args.firstRound) slot = args.count;  // (no such slot: from the queue next time round)
          firstFetch = false;
          if (slot >= args.count) {
            if (l == 0) slot = args.firstRound + atomicAdd(args.queue, 1);
            slot = __builtin_amdgcn_ds_bpermute((lane & ~(kL - 1)) * 4, slot);
          }
        } else {
          if (l == 0) slot = args.firstRound + atomicAdd(args.queue, 1);
          slot = __builtin_amdgcn_ds_bpermute((lane & ~(kL - 1)) * 4, slot);
        }
        if (slot >= args.count) {
          drained = true;
        } else {
          myTask = args.taskList[slot];
        }
        if (myTask >= 0) {
          const int32_t *blocks = batch.blocks + 3 * batch.blk_off[myTask];
          const int nBlocks = batch.n_blocks[myTask];
          GuideCursor cur;
          guide_begin(cur, blocks, nBlocks, args.band);
          myQ = cur.Q;
          myTEnd = blocks[3 * (nBlocks - 1) + 1] + blocks[3 * (nBlocks - 1) + 2];
          myTLen = batch.t_len[myTask];
          myTSeq = batch.seq + batch.t_off[myTask];
          myArrows = args.arrows + args.arrowOff[myTask];
          tab = args.rowTab + args.rowOff[myTask];
          meta = args.rowMeta + args.rowOff[myTask];
          myBase = cur.tStart - 1;
          myPlo = myBase;
          myPhi = myPlo + guide_row0_tpost(cur);
          myCells = (unsigned)args.geo[myTask].cells;
          myGMask = args.wideRingSize[slot] - 1;
          myGRing = reinterpret_cast<int2 *>(args.wideRings + args.wideRingOff[slot]);
          // ---- row 0 (AffineGuidedAlign.h:166-208): S = del * distance from tStart-1, I = open; a wide row 0 goes to the global ring
          myPrevGlobal = myPhi - myPlo + 1 > kF4MaxWidth;
          if (myPrevGlobal) {
            for (int c = myPlo + l; c <= myPhi; c += kL) myGRing[c & myGMask] = make_int2((c - myPlo) * kDel, kOpen);
          } else {
            for (int c = myPlo + l; c <= myPhi; c += kL) ring[c & (kF4Ring - 1)] = make_int2((c - myPlo) * kDel, kOpen);
            ring[(myPhi + 1 + l) & (kF4Ring - 1)] = make_int2(inf, inf);
          }
          myTLoaded = (myPlo < 0 ? 0 : myPlo) & ~3;  // columns below it are staged; whole dwords of codes
          tLast = max(myTLen - 1, 0);
          myRow = 1;
          myPass = 0;
          stagedEnd = 1;
          alive = myRow <= myQ;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      continue;  // (a group whose candidate has no rows takes another)
    }
    if (__ballot(alive) == 0ull) break;  // (every group is through and the list is empty)
    // ---- row records: when one candidate has used up its window, all of them fill theirs
    if (__ballot(alive && myRow >= stagedEnd) != 0ull) {
      if (alive) {
        const int newEnd = min(myRow + kF8Rows, myQ + 1);
#pragma unroll
        for (int k = 0; k < kF8Rows / kL; k++) {
          const int r = stagedEnd + l + kL * k;
          if (r < newEnd) {
            const int2 rec = tab[r];
            const unsigned m = meta[r];
            const unsigned w = m & 0xfffu;
            win[r & (kF8Rows - 1)] = make_uint2((unsigned)rec.x, (unsigned)(rec.y - myBase) |
                                                                     ((w > (unsigned)kF4MaxWidth ? (unsigned)kH8Wide : w) << 20) |
                                                                     (min(m >> 12, 4u) << 28));
          }
        }
        stagedEnd = newEnd;
      }
      __builtin_amdgcn_wave_barrier();
      recCur = win[myRow & (kF8Rows - 1)];
    }
    const int lo = myBase + (int)(recCur.y & 0xfffffu), width = (int)((recCur.y >> 20) & 0xffu);
    const int hiC = lo + width - 1;
    // ---- wide rows: the wavefront takes them one candidate after the other, a piece (a run, or one row) each
    const unsigned long long wideMask = __ballot(alive && width > kF4MaxWidth);
    if (wideMask != 0ull) {
#pragma unroll 1
      for (int g = 0; g < kT; g++) {
        if (((wideMask >> (g * kL)) & 1ull) == 0ull) continue;
        const int src = g * kL;
        const int wRow = __builtin_amdgcn_readlane(myRow, src), wQ = __builtin_amdgcn_readlane(myQ, src);
        const int nG = min(64, wQ - wRow + 1);  // rows of candidate g from this one on, as far as the lanes hold them
        const int tEndOfG = from_group(myTEnd, g), tLenOfG = from_group(myTLen, g), gmask = from_group(myGMask, g);
        const int wPlo = __builtin_amdgcn_readlane(myPlo, src), wPhi = __builtin_amdgcn_readlane(myPhi, src);
        const bool wGlobal = __builtin_amdgcn_readlane((int)myPrevGlobal, src) != 0;
        const uint64_t arrowsOfG = ptr_from_group(myArrows, g), tSeqOfG = ptr_from_group(myTSeq, g);
        // ---- the records of rows wRow .. wRow + 63, lane = row: {arrow offset, first column, width, query code}
        int recX = 0, recLo = 0, recW = 0, recQc = 0;
        {
          const int2 *tabG = (const int2 *)ptr_from_group(tab, g);
          const unsigned short *metaG = (const unsigned short *)ptr_from_group(meta, g);
          if (lane < nG) {
            const int2 rec = tabG[wRow + lane];
            const unsigned m = metaG[wRow + lane];
            int w = (int)(m & 0xfffu);
            if (w == 0xfff) {  // saturated: the row's cells are what lies before the next row's first one
              const unsigned next = wRow + lane < wQ ? (unsigned)tabG[wRow + lane + 1].x : (unsigned)from_group((int)myCells, g);
              const int hi = rec.y + (int)(next - (unsigned)rec.x) - 1;
              w = min(hi, tEndOfG - 1) - rec.y + 1;
            }
            recX = rec.x;
            recLo = rec.y;
            recW = w;
            recQc = (int)min(m >> 12, 4u);
          }
        }
        const int wLo = __builtin_amdgcn_readlane(recLo, 0), wHi = wLo + __builtin_amdgcn_readlane(recW, 0) - 1;
        // (explicitly global: as generic pointers these become FLAT accesses, whose lgkmcnt the LDS reads then wait on)
        typedef unsigned long long __attribute__((address_space(1))) GPair;  // {S, I} of a ring entry
        typedef uint8_t __attribute__((address_space(1))) GByte;
        GPair *gring = (GPair *)ptr_from_group(myGRing, g);
        auto pair_of = [](int2 v) { return (unsigned long long)(uint32_t)v.x | ((unsigned long long)(uint32_t)v.y << 32); };
        auto int2_of = [](unsigned long long p) { return make_int2((int)(uint32_t)p, (int)(uint32_t)(p >> 32)); };
        // target code x 6 of column c (N = 4 beyond the target's ends too)
        auto code6_at = [&](const GByte *wT, int c) {
          const uint32_t b = (uint32_t)wT[min(max(c, 0), max(tLenOfG - 1, 0))];
          return (uint32_t)((c >= 0 && c < tLenOfG) ? base_code(b) : 4) * 6u;
        };
        int consumed = 0;     // rows this piece takes
        bool toNarrow = false;
        int newPlo = wPlo, newPhi = wPhi;
        // ---- A RUN of wide rows, the row above in REGISTERS (fillp_kernel's run path: three consecutive cells per lane, 192
        // columns per step, up to three steps per row; cell (step, lane, j) = column base + 3 (64 step + lane) + j)
        {
          constexpr int kW = 3, kWStep = 64 * kW, kRunSteps = 3, kRunMax = kRunSteps * kWStep;
          int run = 0;
          if (wHi - wLo + 1 <= kRunMax && wPhi - wLo + 1 <= kRunMax) {
            int pS[kRunSteps][kW], pI[kRunSteps][kW];
            int base = wLo, prevLo = wPlo, prevHi = wPhi;
#pragma unroll
            for (int k = 0; k < kRunSteps; k++)
#pragma unroll
              for (int j = 0; j < kW; j++) {
                const int c = wLo + kW * (64 * k + lane) + j;
                const bool valid = c >= wPlo && c <= wPhi;
                int2 v = make_int2(inf, inf);
                if (wGlobal) v = int2_of(gring[c & gmask]);
                else if (k == 0) v = ringLds[wave][g][c & (kF4Ring - 1)];  // (a narrow row ends within the first step)
                pS[k][j] = valid ? v.x : inf;
                pI[k][j] = valid ? v.y : inf;
              }
            int leftS = inf;  // S of the row above at column base - 1
            if (wLo - 1 >= wPlo && wLo - 1 <= wPhi)
              leftS = wGlobal ? (int)(uint32_t)gring[(wLo - 1) & gmask] : ringLds[wave][g][(wLo - 1) & (kF4Ring - 1)].x;
            leftS = __builtin_amdgcn_readfirstlane(leftS);
            GByte *aBase = (GByte *)arrowsOfG;
            const GByte *wT = (const GByte *)tSeqOfG;
            // the target's codes x 6 of the 576 columns from `base` on, three to a register like the cells, loaded once per
            // run and handed on by the same lane shift as the row above
            uint32_t tP[kRunSteps];
#pragma unroll
            for (int k = 0; k < kRunSteps; k++) {
              tP[k] = 0;
#pragma unroll
              for (int j = 0; j < kW; j++) tP[k] |= code6_at(wT, wLo + kW * (64 * k + lane) + j) << (8 * j);
            }
            for (int ii = 0; ii < nG; ii++) {
              const int rLo = __builtin_amdgcn_readlane(recLo, ii), rWidth = __builtin_amdgcn_readlane(recW, ii);
              const int rHi = rLo + rWidth - 1, d = rLo - base;
              if (rWidth <= kF4MaxWidth || rWidth > kRunMax || d < 0 || d > 1) break;
              const uint32_t row6 = (uint32_t)__builtin_amdgcn_readlane((int)row6Table, __builtin_amdgcn_readlane(recQc, ii));
              GByte *rowArrows = aBase + (unsigned)__builtin_amdgcn_readlane(recX, ii);
              if (d == 1) {  // the window of codes moves one column on: column base + 576 enters at its end
                uint32_t first[kRunSteps + 1];
#pragma unroll
                for (int k = 0; k < kRunSteps; k++) first[k] = (uint32_t)__builtin_amdgcn_readlane((int)tP[k], 0) & 0xffu;
                first[kRunSteps] = (uint32_t)__builtin_amdgcn_readfirstlane((int)code6_at(wT, base + kRunMax));
#pragma unroll
                for (int k = 0; k < kRunSteps; k++)
                  tP[k] = (tP[k] >> 8) | (((uint32_t)wave_shl1((int)tP[k], (int)first[k + 1]) & 0xffu) << (8 * (kW - 1)));
              }
              int ePrevOld = inf;
              int wCarryA = inf, wCarryB = inf;
#pragma unroll
              for (int k = 0; k < kRunSteps; k++) {
                if (rLo + kWStep * k > rHi) break;
                const int c0 = rLo + kW * (64 * k + lane);
                int upS[kW], upI[kW], diagS[kW];
                if (d == 0) {
                  const int eOld = __builtin_amdgcn_readlane(pS[k][kW - 1], 63);
                  int dg = wave_shr1(pS[k][kW - 1], inf);
                  dg = lane == 0 ? (k == 0 ? leftS : ePrevOld) : dg;
                  ePrevOld = eOld;
                  diagS[0] = dg;
#pragma unroll
                  for (int j = 0; j < kW; j++) {
                    upS[j] = pS[k][j];
                    upI[j] = pI[k][j];
                    if (j > 0) diagS[j] = pS[k][j - 1];
                  }
                } else {
#pragma unroll
                  for (int j = 0; j < kW; j++) {
                    diagS[j] = pS[k][j];
                    if (j + 1 < kW) {
                      upS[j] = pS[k][j + 1];
                      upI[j] = pI[k][j + 1];
                    }
                  }
                  upS[kW - 1] = wave_shl1(pS[k][0], k + 1 < kRunSteps ? __builtin_amdgcn_readlane(pS[k + 1 < kRunSteps ? k + 1 : k][0], 0) : inf);
                  upI[kW - 1] = wave_shl1(pI[k][0], k + 1 < kRunSteps ? __builtin_amdgcn_readlane(pI[k + 1 < kRunSteps ? k + 1 : k][0], 0) : inf);
                }
                // cells of the row above that do not exist: only a step that reaches beyond that row's ends has any
                if (!(rLo + kWStep * k - 1 >= prevLo && rLo + kWStep * (k + 1) - 1 <= prevHi)) {
#pragma unroll
                  for (int j = 0; j < kW; j++) {
                    const int c = c0 + j;
                    const bool upValid = c <= prevHi, diagValid = c - 1 <= prevHi && c - 1 >= prevLo;  // (c >= rLo >= prevLo)
                    upS[j] = upValid ? upS[j] : inf;
                    upI[j] = upValid ? upI[j] : inf;
                    diagS[j] = diagValid ? diagS[j] : inf;
                  }
                }
                int vKey[kW], preA[kW], preB[kW];
#pragma unroll
                for (int j = 0; j < kW; j++) {
                  const int m = (int)__builtin_amdgcn_sbfe((int)row6, (tP[k] >> (8 * j)) & 0xffu, 6);
                  const int matchK = (m << kShift) + diagS[j];
                  const int insSK = upS[j] + (kIns + AR_UP);
                  const int insEK = upI[j] + AR_INS_CLOSE;
                  vKey[j] = min(matchK, min(insSK, insEK));
                  const int v = vKey[j] & ~kTagMask;
                  const int ia = v - (kW * lane + j) * kDel;
                  preA[j] = j == 0 ? ia : min(preA[j - 1], ia);
                  preB[j] = j == 0 ? v : min(preB[j - 1], v);
                }
                int pa = preA[kW - 1], pb = preB[kW - 1];
                wave_prefix_min2(pa, pb);
                int ea = wave_shr1(pa, inf), eb = wave_shr1(pb, inf);
                if (k > 0) {
                  ea = min(ea, wCarryA);
                  eb = min(eb, wCarryB);
                }
                uint32_t arrowLo = 0, arrowHi = 0;  // (cells 0 .. kW - 2 in arrowLo, the last one in arrowHi)
#pragma unroll
                for (int j = 0; j < kW; j++) {
                  const int eaj = j == 0 ? ea : min(ea, preA[j - 1]);
                  const int ebj = j == 0 ? eb : min(eb, preB[j - 1]);
                  const int aK = eaj + ((kW * lane + j) * kDel + AR_LEFT);  // S[c-1] + del
                  const int bK = ebj + (kOpen + AR_DEL_CLOSE);              // D[c-1] + ext
                  const int sKey = min(vKey[j], min(aK, bK));
                  const int sv = sKey & ~kTagMask;
                  const int openKey = sv + (kOpen + AR_I_OPEN);
                  const int iKey = min(openKey, upI[j]);
                  const int dFlag = openKey < bK ? AR_D_OPEN : 0;
                  const uint32_t arrow = (uint32_t)(sKey | iKey | dFlag) & kTagMask;
                  if (j < kW - 1) arrowLo |= arrow << (8 * j); else arrowHi = arrow;
                  pS[k][j] = sv;  // (cells beyond the row's end hold what no later row takes for a cell: prevHi)
                  pI[k][j] = iKey & ~kTagMask;
                }
                if (c0 <= rHi) {  // three arrows per lane; those beyond the row's end land in the next row's space
                  typedef unaligned_u16 __attribute__((address_space(1))) GU16;
                  GByte *ap = rowArrows + (c0 - rLo);
                  *(GU16 *)ap = (uint16_t)arrowLo;
                  ap[kW - 1] = (uint8_t)arrowHi;
                }
                wCarryA = min(wCarryA, __builtin_amdgcn_readlane(pa, 63)) + kWStep * kDel;
                wCarryB = min(wCarryB, __builtin_amdgcn_readlane(pb, 63));
              }
              base = prevLo = rLo;
              prevHi = rHi;
              leftS = inf;  // (column base - 1 lies left of the row just made)
              run++;
            }
            if (run > 0) {
              // ---- the run's last row: to where the next row will look for it
              int keepLo = prevLo, keepHi = prevHi;
              if (run < nG) {
                const int nLo = __builtin_amdgcn_readlane(recLo, run), nWidth = __builtin_amdgcn_readlane(recW, run);
                if (nWidth <= kF4MaxWidth) {
                  toNarrow = true;
                  keepLo = max(keepLo, nLo - 1);
                  keepHi = min(keepHi, nLo + nWidth - 1);
                }
              }
#pragma unroll
              for (int k = 0; k < kRunSteps; k++)
#pragma unroll
                for (int j = 0; j < kW; j++) {
                  const int c = base + kW * (64 * k + lane) + j;
                  if (c >= keepLo && c <= keepHi) {
                    if (toNarrow) ringLds[wave][g][c & (kF4Ring - 1)] = make_int2(pS[k][j], pI[k][j]);
                    else gring[c & gmask] = pair_of(make_int2(pS[k][j], pI[k][j]));
                  }
                }
              // (the entries behind it that the narrow rows read without a test: "no cell")
              if (toNarrow && lane < kF8Clean) ringLds[wave][g][(keepHi + 1 + lane) & (kF4Ring - 1)] = make_int2(inf, inf);
              consumed = run;
              newPlo = prevLo;
              newPhi = toNarrow ? keepHi : prevHi;
            }
          }
        }
        if (consumed == 0) {
          // ---- ONE wide row through the candidate's ring in global memory (fill_kernel's step, 64 columns at a time): a row
          // wider than a run holds, or one that starts more than a column further on
          const int wQc = __builtin_amdgcn_readlane(recQc, 0);
          const uint32_t row6 = (uint32_t)__builtin_amdgcn_readlane((int)row6Table, wQc);
          if (!wGlobal) {  // the row above is a narrow one: from the LDS ring into the global ring
            for (int c = wPlo + lane; c <= wPhi; c += 64) gring[c & gmask] = pair_of(ringLds[wave][g][c & (kF4Ring - 1)]);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          }
          // which of this row's {S, I} the next row can read: a narrow one only its own columns (and one to their left); the
          // candidate's last row only feeds the final score
          int keepLo = INT32_MIN, keepHi = INT32_MAX;
          if (wRow == wQ) {
            keepLo = keepHi = tEndOfG - 1;
          } else if (nG > 1) {
            const int nLo = __builtin_amdgcn_readlane(recLo, 1), nWidth = __builtin_amdgcn_readlane(recW, 1);
            if (nWidth <= kF4MaxWidth) {
              keepLo = nLo - 1;
              keepHi = nLo + nWidth - 1;
            }
          }
          GByte *wArrows = (GByte *)(arrowsOfG + (unsigned)__builtin_amdgcn_readlane(recX, 0));
          const GByte *wT = (const GByte *)tSeqOfG;
          RowConsts<true> kc;
          kc.inf = inf;
          kc.ins = kIns;
          kc.del = kDel;
          kc.open = kOpen;
          kc.ext = 0;
          kc.laneDel = lane * kDel;
          kc.laneExt = 0;
          ChunkCarry carry;
          carry.a = carry.b = carry.s = inf;
          int cDiag = 0;
          // the target bases and the row above run one step ahead of their use, two sets of registers taking turns, the
          // loads of the step after issued unconditionally (fillp_kernel's wide step)
          const int tLastG = max(tLenOfG - 1, 0);
          int tA = (int)wT[min(max(wLo + lane, 0), tLastG)], tB = 'N';
          int2 upA = int2_of(gring[(wLo + lane) & gmask]), upB = make_int2(inf, inf);
          if (wLo <= wPhi + 1 && lane == 0) cDiag = (int)(uint32_t)gring[(wLo - 1) & gmask];
          auto wideStep = [&](const int cb, const int tRaw, int2 up, int &tNext, int2 &upNext) {
            const int c = cb + lane;
            const bool active = c <= wHi;
            int sd = inf;
            tNext = (int)wT[min(max(c + 64, 0), tLastG)];
            upNext = int2_of(gring[(c + 64) & gmask]);
            const int code = (c >= 0 && c < tLenOfG) ? base_code((unsigned)tRaw) : 4;
            if (cb <= wPhi + 1) {
              sd = wave_shr1(up.x, inf);
              sd = (lane == 0) ? __builtin_amdgcn_readfirstlane(cDiag) : sd;
              cDiag = __builtin_amdgcn_readlane(up.x, 63);
              const bool upValid = c <= wPhi, diagValid = (c - 1 <= wPhi) && (c > wPlo);
              up.x = upValid ? up.x : inf;
              up.y = upValid ? up.y : inf;
              sd = diagValid ? sd : inf;
            } else {  // beyond the row above there is nothing: whatever the ring slot held is not it
              up = make_int2(inf, inf);
            }
            const int mm = (int)__builtin_amdgcn_sbfe((int)row6, (unsigned)(code * 6), 6);
            int sv, iv, arrow;
            row_chunk<true, false>(kc, carry, active, up.x, up.y, sd, mm, sv, iv, arrow);
            if (active) {
              if (c >= keepLo && c <= keepHi) gring[c & gmask] = pair_of(make_int2(sv, iv));
              wArrows[c - wLo] = (uint8_t)arrow;
            }
          };
          for (int cb = wLo; cb <= wHi; cb += 128) {
            wideStep(cb, tA, upA, tB, upB);
            if (cb + 64 <= wHi) wideStep(cb + 64, tB, upB, tA, upA);
          }
          consumed = 1;
          newPlo = wLo;
          newPhi = wHi;
        }
        // (this piece's stores are read by the rows after it only: one wait for all of them)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (grp == g) {
          myPlo = newPlo;
          myPhi = newPhi;
          myPrevGlobal = !toNarrow;
          myRow += consumed;
          myPass = 0;
          alive = myRow <= myQ;
          if (myRow >= stagedEnd) stagedEnd = myRow;       // (the window is filled from there at the head of the loop)
          else recCur = win[myRow & (kF8Rows - 1)];
        }
      }
      continue;
    }
    // ---- back to narrow rows from a wide one that went to the global ring: the columns this row reads of it, into the LDS ring
    if (__ballot(alive && myPrevGlobal) != 0ull) {
      if (alive && myPrevGlobal) {
        const int last = min(hiC, myPhi);
        for (int c = lo - 1 + l; c <= last; c += kL)
          if (c >= myPlo) ring[c & (kF4Ring - 1)] = myGRing[c & myGMask];
        ring[(last + 1 + l) & (kF4Ring - 1)] = make_int2(inf, inf);
        myPhi = last;
        myPrevGlobal = false;
      }
      __builtin_amdgcn_wave_barrier();
    }
    const uint2 recNext = win[(myRow + 1) & (kF8Rows - 1)];  // (the row after this one, if the window has it)
    // ---- the target's codes: when one candidate has run out, all of them stage as far as their ring holds
    if (__ballot(alive && hiC >= myTLoaded) != 0ull) {
      if (alive) {
        myTLoaded = max(myTLoaded, lo & ~3);
        const int limit = (lo & ~3) + kF4RingT;  // the first column whose slot still holds one this row may read
        const int c = myTLoaded + 16 * l;
        uint32_t b[16];
#pragma unroll
        for (int k = 0; k < 16; k++) b[k] = (uint32_t)myTSeq[min(c + k, tLast)];
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) {
          uint32_t w = 0;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int cc = c + 4 * k4 + k;
            const int code = cc < myTLen ? base_code(b[4 * k4 + k]) : 4;
            w |= (uint32_t)(code * 6) << (8 * k);
          }
          if (c + 4 * k4 < limit) *reinterpret_cast<uint32_t *>(&ringT[(c + 4 * k4) & (kF4RingT - 1)]) = w;
        }
        myTLoaded = min(myTLoaded + 16 * kL, limit);
      }
      __builtin_amdgcn_wave_barrier();
    }
    // ---- one pass of my candidate's row: columns cb .. cb + 39 (fill8_kernel's)
    uint8_t *arrowRow = myArrows + recCur.x;
    const int cb = lo + kPass * myPass, c0 = cb + kC * l;
    int upS[kC], upI[kC], diagS[kC];
    int d0 = ring[(c0 - 1) & (kF4Ring - 1)].x;
    d0 = (myPass > 0 && l == 0) ? carryDiag : d0;  // column cb-1 was overwritten by the pass before
    d0 = c0 - 1 >= myPlo ? d0 : inf;               // (lane 0 of a row that starts where the row above did)
    int lastRaw = 0;
#pragma unroll
    for (int j = 0; j < kC; j++) {
      const int2 up = ring[(c0 + j) & (kF4Ring - 1)];
      if (j == kC - 1) lastRaw = up.x;
      upS[j] = up.x;
      upI[j] = up.y;
    }
    if (__ballot(alive && hiC > myPhi + kF8Clean) != 0ull) {  // (rare) entries beyond the cleaned ones are stale
      d0 = c0 - 1 <= myPhi ? d0 : inf;
#pragma unroll
      for (int j = 0; j < kC; j++) {
        const bool upValid = c0 + j <= myPhi;  // c >= lo >= plo always
        upS[j] = upValid ? upS[j] : inf;
        upI[j] = upValid ? upI[j] : inf;
      }
    }
    diagS[0] = d0;
#pragma unroll
    for (int j = 1; j < kC; j++) diagS[j] = upS[j - 1];
    const uint32_t row6 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((recCur.y >> 26) & 0x1cu), (int)row6Table);
    int vKey[kC], preA[kC], preB[kC];
#pragma unroll
    for (int j = 0; j < kC; j++) {
      const int m = (int)__builtin_amdgcn_sbfe((int)row6, (unsigned)ringT[(c0 + j) & (kF4RingT - 1)], 6);
      const int matchK = (m << kShift) + diagS[j];
      const int insSK = upS[j] + (kIns + AR_UP);
      const int insEK = upI[j] + AR_INS_CLOSE;
      vKey[j] = min(matchK, min(insSK, insEK));
      const int v = vKey[j] & ~kTagMask;
      const int ia = v - pDel[j];
      preA[j] = j == 0 ? ia : min(preA[j - 1], ia);  // inclusive over this lane's cells
      preB[j] = j == 0 ? v : min(preB[j - 1], v);
    }
    int pa = preA[kC - 1], pb = preB[kC - 1];
    group_prefix_min2<kL>(pa, pb);  // inclusive over the lanes of the candidate
    eaKeep = dpp<0x111, 0xf>(eaKeep, pa);
    ebKeep = dpp<0x111, 0xf>(ebKeep, pb);
    int ea = l == 0 ? inf : eaKeep, eb = l == 0 ? inf : ebKeep;
    ea = myPass > 0 ? min(ea, carryA) : ea;
    eb = myPass > 0 ? min(eb, carryB) : eb;
    uint32_t arrowLo = 0, arrowHi = 0;
#pragma unroll
    for (int j = 0; j < kC; j++) {
      const int eaj = j == 0 ? ea : min(ea, preA[j - 1]);
      const int ebj = j == 0 ? eb : min(eb, preB[j - 1]);
      const int aK = eaj + (pDel[j] + AR_LEFT);           // S[c-1] + del
      const int bK = ebj + (kOpen + AR_DEL_CLOSE);         // D[c-1] + ext
      const int sKey = min(vKey[j], min(aK, bK));
      const int sv = sKey & ~kTagMask;
      const int openKey = sv + (kOpen + AR_I_OPEN);
      const int iKey = min(openKey, upI[j]);               // (ring entries carry no tag)
      const int dFlag = openKey < bK ? AR_D_OPEN : 0;
      const uint32_t arrow = (uint32_t)(sKey | iKey | dFlag);
      if (j == 0) arrowLo = arrow;
      else if (j == 1) arrowLo = __builtin_amdgcn_perm(arrow, arrowLo, 0x0c0c0400u);   // {cell 0, cell 1, 0, 0}
      else if (j == 2) arrowHi = arrow;
      else if (j == 3) arrowLo = __builtin_amdgcn_perm(__builtin_amdgcn_perm(arrow, arrowHi, 0x04000c0cu), arrowLo, 0x07060100u) & 0x1f1f1f1fu;
      else arrowHi = arrow & kTagMask;
      if (alive) ring[(c0 + j) & (kF4Ring - 1)] = make_int2(sv, iKey & ~kTagMask);
    }
    if (alive && c0 <= hiC) {
      uint8_t *ap = arrowRow + (c0 - lo);
      *(unaligned_u32 *)ap = arrowLo;
      ap[4] = (uint8_t)arrowHi;
    }
    const bool more = alive && cb + kPass <= hiC;
    if (alive && !more) ring[(hiC + 1 + l) & (kF4Ring - 1)] = make_int2(inf, inf);  // (after the cells' own writes)
    if (__ballot(more) != 0ull) {  // carries into the next 40 columns
      const int last = (lane | (kL - 1)) * 4;
      const int ta = __builtin_amdgcn_ds_bpermute(last, pa), tb = __builtin_amdgcn_ds_bpermute(last, pb);
      carryA = (myPass > 0 ? min(carryA, ta) : ta) + kPass * kDel;
      carryB = myPass > 0 ? min(carryB, tb) : tb;
      carryDiag = __builtin_amdgcn_ds_bpermute(last, lastRaw);  // the raw S of column cb+39, for lane 0
    }
    __builtin_amdgcn_wave_barrier();
    if (alive) {
      if (more) {
        myPass++;
      } else {
        myPass = 0;
        myRow++;
        myPlo = lo;
        myPhi = hiC;
        recCur = recNext;
      }
    }
    alive = alive && myRow <= myQ;
  }
}

template __global__ void fillp_kernel<16, true, false>(RefineBatchView, FillArgs);
template __global__ void fillp_kernel<16, false, false>(RefineBatchView, FillArgs);
template __global__ void fillp_kernel<16, true, true>(RefineBatchView, FillArgs);
template __global__ void fillp_kernel<16, false, true>(RefineBatchView, FillArgs);

template __global__ void fill_kernel<false, true>(RefineBatchView, FillArgs);
template __global__ void fill_kernel<false, false>(RefineBatchView, FillArgs);
template __global__ void fill_kernel<true, true>(RefineBatchView, FillArgs);
template __global__ void fill_kernel<true, false>(RefineBatchView, FillArgs);

// ---------------------------------------------------------------------------
// traceback_kernel: one lane per candidate (AffineGuidedAlign.h:377-468), fused with
// ArrowPathToAlignment (RLE, Alignment.h:190-254) and RemoveAlignmentPrefixGaps.
//
// The walk is a chain of dependent byte loads, so it is organised around memory latency: rows are taken
// kTbRows at a time.  At the start of a batch the lane issues, back to back, the 16-byte arrow windows of the
// rows ahead (the path moves at most one column per row unless it runs through a deletion, so the
// window of the j-th row ends 8 columns right of the column it would reach on a pure diagonal) and the row
// records of the following batch; the single wait for these loads is the only exposed memory latency per
// batch.  The windows go to LDS and the lane walks them there.  A path that leaves its window (a long deletion)
// fetches again.
// ---------------------------------------------------------------------------
constexpr int kTbWin = 16;  // bytes per prefetched arrow window

__device__ __forceinline__ uint4 load_window(const uint8_t *p) {
  // unaligned 16-byte load (global memory allows it); p + 16 stays inside the candidate's arrow slab.
  // (Round 5 measured the aligned alternatives -- the window starts at a byte address -- at configs[2]: two 16-byte-aligned
  //  loads and a funnel shift through selects 119 ms (115 registers, scratch), a dword-aligned 16-byte load + one dword and
  //  four v_alignbit 43.1 ms, this load 39.1 ms: the misalignment is not what keeps the texture addresser busy.)
  uint4 v;
  __builtin_memcpy(&v, p, 16);
  return v;
}
__device__ __forceinline__ unsigned window_byte(const uint4 &w, int i) {  // 0 <= i < 16
  const uint32_t lo = (i & 4) ? w.y : w.x, hi = (i & 4) ? w.w : w.z;
  const uint32_t d = (i & 8) ? hi : lo;
  return (d >> ((i & 3) * 8)) & 0xffu;
}

// Candidates per wavefront.  The walk is a chain of dependent steps, so the device wants many resident wavefronts
// rather than full ones: with 64 candidates per wavefront configs[1] yields only 1.5 wavefronts per SIMD.
constexpr int kTbLanes = 32;

// One step of the walk, without a branch: what the arrow at (row, column) makes of the state.
//   index   state 0 (match matrix): the S arrow, 0..7;  state 1 (affine insertion): 8 + its "open" bit;
//           state 2 (affine deletion): 10 + its "open" bit
//   entry   bits 0-1 the op the step emits (3 = none), bit 2 column step, bit 3 the step leaves the row,
//           bits 4-5 next state, bit 6 impossible arrow
#define TB_E(op, dc, leave, st, bad) ((uint32_t)((op) | ((dc) << 2) | ((leave) << 3) | ((st) << 4) | ((bad) << 6)))
__device__ __forceinline__ unsigned tb_entry(int idx) {
  constexpr uint32_t kS03 = TB_E(OP_M, 1, 1, 0, 0) | TB_E(OP_D, 1, 0, 0, 0) << 8 |     // Diagonal, Left
                            TB_E(OP_I, 0, 1, 0, 0) << 16 | TB_E(OP_I, 0, 1, 1, 0) << 24;  // Up, AffineInsClose
  constexpr uint32_t kS47 = TB_E(OP_D, 1, 0, 2, 0) | TB_E(3, 0, 1, 0, 1) << 8 |        // AffineDelClose, -
                            TB_E(3, 0, 1, 0, 1) << 16 | TB_E(3, 0, 1, 0, 1) << 24;
  constexpr uint32_t kID = TB_E(OP_I, 0, 1, 1, 0) | TB_E(3, 0, 0, 0, 0) << 8 |         // I: up / open
                           TB_E(OP_D, 1, 0, 2, 0) << 16 | TB_E(3, 0, 0, 0, 0) << 24;   // D: left / open
  const uint32_t d = idx < 4 ? kS03 : (idx < 8 ? kS47 : kID);
  return (d >> ((idx & 3) * 8)) & 0xffu;
}
#undef TB_E
static_assert(AR_DIAG == 0 && AR_LEFT == 1 && AR_UP == 2 && AR_INS_CLOSE == 3 && AR_DEL_CLOSE == 4, "tb_entry order");

// Where a candidate's op codes go: two bits per op in walk order, sixteen to a dword, dword k at codesEnd - 4 (k+1)
// -- the tail of the candidate's own arrow slab.  The walk has consumed those arrows before the codes get there:
// every op crosses at least one cell (one byte of arrows) and costs a quarter of a byte.
__device__ __forceinline__ uint32_t *tb_codes_end(uint8_t *arrows, int64_t cells) {
  return reinterpret_cast<uint32_t *>(arrows + ((cells + 16) & ~(int64_t)3));
}

// Rows per batch of the walk: their 16-byte arrow windows are fetched together (one exposed round trip per batch) and
// parked in LDS, where each lane then walks at its own pace.  (Sixteen rows per batch: 56 ms where eight take 41 at
// configs[2] -- the record and window registers of a batch double.)
constexpr int kTbRows = 8;

// The walk is EVENT driven: a trip of the loop is one step of every lane's walk (one arrow), wherever that lane is in
// its batch of rows -- a lane in a deletion run or at an affine "open" does not hold up the others row by row; they only
// meet again at the end of the batch.  A step is two LDS reads and a handful of integer instructions: the arrow byte
// from the lane's window of its current row, and the decoded step {op, column step, leaves the row, next state,
// impossible} from a 96-entry table indexed by state and arrow byte.  Windows are NOT clamped to the row's first
// column (the bytes left of it belong to the row before, or to the slack before the slab, and a correct walk never
// looks at them): the byte index of the walk's cell in the window of row j follows from the column alone.
template <int kB>
__global__ __launch_bounds__(64) void traceback_kernel(RefineBatchView batch, TraceArgs args) {
  __shared__ uint4 winLds[kB][kTbLanes];
  __shared__ uint32_t stepTab[96];
  // stepTab[state * 32 + arrow byte]: bits 0-1 the op to emit (0 when none), bit 2 column step, bit 3 leaves the row,
  // bit 6 impossible arrow, bit 7 emits an op, bits 8.. the byte offset of the next state's 32 entries
  for (int t = threadIdx.x; t < 96; t += 64) {
    const int st = t >> 5, a = t & 31;
    const int at = st == 0 ? (a & 7) : (st == 1 ? 8 + ((a >> 3) & 1) : 10 + ((a >> 4) & 1));
    const unsigned e = tb_entry(at), op = e & 3u;
    stepTab[t] = (op != 3u ? op | 0x80u : 0u) | (e & 0x4cu) | ((((e >> 4) & 3u) * 128u) << 8);
  }
  __syncthreads();
  const int idx = blockIdx.x * kTbLanes + threadIdx.x;
  const bool inRange = idx < args.count && threadIdx.x < kTbLanes;
  const int task = args.first + (inRange ? idx : 0);
  const int slot = threadIdx.x & (kTbLanes - 1);
  const TaskGeometry geo = args.geo[task];
  bool alive = inRange && geo.status == 0;
  const int32_t *blocks = batch.blocks + 3 * batch.blk_off[task];
  const int nBlocks = alive ? batch.n_blocks[task] : 1;
  const int tStart = alive ? blocks[1] : 0;
  const int Q = alive ? geo.rows : 0;
  const int tEnd = alive ? blocks[3 * (nBlocks - 1) + 1] + blocks[3 * (nBlocks - 1) + 2] : 0;
  uint8_t *arrows = const_cast<uint8_t *>(args.arrows) + args.arrowOff[task];
  const int2 *rowTab = args.rowTab + args.rowOff[task];
  const int2 *anyRec = reinterpret_cast<const int2 *>(args.geo);  // a valid address for records that are not used
  uint32_t *codes = tb_codes_end(arrows, alive ? geo.cells : 0);  // grows downwards

  uint64_t acc = 0;  // ops not yet stored, oldest in the low bits
  int fill = 0;
  uint32_t *const codesEnd = codes;
  int c = tEnd - 1;
  unsigned stateOff = 0;  // 128 x (0 = match matrix, 1 = affine ins, 2 = affine del): where its entries start in stepTab
  unsigned badAcc = 0;
  int guard = 2 * (Q + (tEnd - tStart)) + 64;  // steps a walk can take at most (every step leaves a row, moves left or
  int status = 0;                              //  follows a close / open pair)
  auto flush = [&]() {  // lanes holding sixteen ops or more store a dword
    if (fill >= 16) {
      *--codes = (uint32_t)acc;
      acc >>= 32;
      fill -= 16;
    }
  };
  auto window_at = [&](const int2 rec, int cb) {  // the 16 arrows of a row from column cb on (unclamped, see above)
    return load_window(arrows + (int64_t)(rec.x + (cb - rec.y)));
  };

  // RT[j] = record of the j-th row of the current batch (rows rTop, rTop-1, ...), RTn the same for the next one
  // (two rows' records per 16-byte load: rows a - 1 and a are neighbours in the table, and record 0 exists, unused)
  static_assert(kB % 2 == 0, "row records are fetched in pairs");
  auto load_pair = [&](int a, int2 &recA, int2 &recBelow) {  // the records of rows a and a - 1
    int4 v;
    __builtin_memcpy(&v, (alive && a >= 1) ? reinterpret_cast<const void *>(rowTab + (a - 1)) : reinterpret_cast<const void *>(anyRec), 16);
    recA = make_int2(v.z, v.w);
    recBelow = make_int2(v.x, v.y);
  };
  int2 RT[kB], RTn[kB];
#pragma unroll
  for (int j = 0; j < kB; j += 2) load_pair(Q - j, RT[j], RT[j + 1]);

  int rTop = Q;
  while (__any(alive)) {
    const int nHere = alive ? min(kB, rTop) : 0;  // rows of this batch
    int c0 = c;                                   // the window of row j starts at column c0 - j - 7
    {
      // (addresses are chosen, loads are unconditional: a load behind a branch costs a wait for everything in flight)
      uint4 W[kB];
#pragma unroll
      for (int j = 0; j < kB; j++)
        W[j] = load_window(j < nHere ? arrows + (int64_t)(RT[j].x + (c0 - j - 7 - RT[j].y)) : arrows);
#pragma unroll
      for (int j = 0; j < kB; j += 2) load_pair(rTop - kB - j, RTn[j], RTn[j + 1]);
#pragma unroll
      for (int j = 0; j < kB; j++)
        if (alive) winLds[j][slot] = W[j];
    }
    int j = 0;
    for (int trip = 0;; trip++) {
      const bool on = j < nHere && guard >= 0;  // (nHere = 0 for a lane that is not alive)
      if (__builtin_amdgcn_ballot_w64(on) == 0ull) break;
      if (on) {
        int i = c - c0 + j + 7;
        if ((unsigned)i >= (unsigned)kTbWin) {
          // the path left the prefetched window (a long deletion): this row and the rows of the batch still ahead are
          // fetched again, placed so that the walk's cell is the window's last
          c0 = c + j - (kTbWin - 1 - 7);
          i = kTbWin - 1;
          for (int jj = j; jj < nHere; jj++) winLds[jj][slot] = window_at(rowTab[rTop - jj], c0 - jj - 7);
        }
        const unsigned a = reinterpret_cast<const uint8_t *>(&winLds[j][slot])[i];
        const unsigned e = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(stepTab) + (stateOff + (a & 31u) * 4u));
        acc |= (uint64_t)(e & 3u) << (2 * fill);
        const int emits = (int)((e >> 7) & 1u);
        fill += emits;
        c -= (int)((e >> 2) & 1u);
        j += (int)((e >> 3) & 1u);
        stateOff = e >> 8;
        badAcc |= e;
        guard--;
      }
      // many ops inside one batch (a long deletion run): a batch starts with fewer than sixteen ops pending and a trip
      // adds one at most, so the first fourteen trips need no look
      if (trip >= 14 && __builtin_amdgcn_ballot_w64(on && fill >= 30) != 0ull) flush();
    }
    if (alive) {
      if ((badAcc & 0x40u) != 0u || guard < 0) {  // an arrow the recurrence cannot have written: NoArrow (:394-408)
        status = TASK_BAD_PATH;
        alive = false;
      } else if (rTop <= kB) {
        // row 0 (the deletion row, :178-208): Left arrows down to column tStart-1; the walk ends when
        // q < qStart and t < tStart
        while (c >= tStart) {
          acc |= (uint64_t)OP_D << (2 * fill);
          fill++;
          c--;
          if (fill == 32) flush();
        }
        alive = false;
      }
      flush();
    }
    rTop -= kB;
#pragma unroll
    for (int j = 0; j < kB; j++) RT[j] = RTn[j];
  }
  if (inRange) {
    const int nOps = 16 * (int)(codesEnd - codes) + fill;  // sixteen to every dword stored, and those still pending
    if (geo.status == 0) {
      while (fill > 0) {  // the last, partly filled dwords
        *--codes = (uint32_t)acc;
        acc >>= 32;
        fill -= 16;
      }
    }
    args.opsStart[task] = nOps;                                       // runs_kernel turns both into
    args.results[task].status = geo.status != 0 ? geo.status : status;  // the candidate's record
  }
}

// ---------------------------------------------------------------------------
// runs_kernel: the op codes of the walk (alignment end first) become the run-length encoded op list, one
// wavefront per candidate (ArrowPathToAlignment, Alignment.h:190-254): lane = sixteen ops, a run is written
// when its end is seen, run k of the walk at arena[cap-1-k], so that the list reads forwards.  Gap runs behind
// the last block are dropped (:251-253 clears the trailing gap list) and leading ones fold into qPos / tPos
// (RemoveAlignmentPrefixGaps, AlignmentUtils.h:620-644).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void runs_kernel(RefineBatchView batch, TraceArgs args) {
  const int lane = lane_id();
  const int idx = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (idx >= args.count) return;
  const int task = args.first + idx;
  const TaskGeometry geo = args.geo[task];
  const int status = args.results[task].status;
  RefineResult res;
  res.status = status;
  res.q_pos = res.t_pos = res.dp_score = res.n_cells = res.score = 0;
  res.n_match = res.n_mismatch = res.n_ins = res.n_del = res.n_ops = 0;
  res.reserved = 0;
  res.ops_off = 0;
  if (geo.status != 0) {
    if (lane == 0) {
      args.results[task] = res;
      args.opsStart[task] = 0;
    }
    return;
  }
  const int32_t *blocks = batch.blocks + 3 * batch.blk_off[task];
  const int nBlocks = batch.n_blocks[task];
  const int qStart = blocks[0], tStart = blocks[1];
  const int tEnd = blocks[3 * (nBlocks - 1) + 1] + blocks[3 * (nBlocks - 1) + 2];
  const int cap = geo.rows + (tEnd - tStart) + 2;
  uint32_t *arena = args.arena + args.arenaOff[task];
  const uint32_t *codes = tb_codes_end(const_cast<uint8_t *>(args.arrows) + args.arrowOff[task], geo.cells);
  const int n = args.opsStart[task];  // ops of the walk

  int nRuns = 0, firstM = INT32_MAX;  // boundaries seen so far (= runs begun), first run that is a block
  int lastStart = 0;                  // where the run in progress began
  uint32_t carryWord = 0;             // the dword before this iteration's first
  for (int d0 = 0; 16 * d0 < n; d0 += 64) {
    const int d = d0 + lane;
    const bool have = 16 * d < n;
    const uint32_t word = have ? codes[-1 - d] : 0u;
    const uint32_t before = (uint32_t)wave_shr1((int)word, (int)carryWord);
    carryWord = (uint32_t)__builtin_amdgcn_readlane((int)word, 63);
    // op i starts a run when it differs from op i-1 (op 0 always does)
    const uint32_t shifted = (word << 2) | (before >> 30);
    uint32_t diff = word ^ shifted;
    diff = (diff | (diff >> 1)) & 0x55555555u;
    if (d == 0) diff |= 1u;
    const int valid = have ? min(16, n - 16 * d) : 0;
    if (valid < 16) diff &= (1u << (2 * valid)) - 1u;
    const int nb = __popc(diff);
    const unsigned inclNb = wave_prefix_sum((unsigned)nb);
    int lastB = nb ? 16 * d + ((31 - __clz(diff)) >> 1) : -1;
    const int inclLast = wave_prefix_max(lastB);
    int prevStart = max(wave_shr1(inclLast, -1), lastStart);  // start of the run in progress at this lane's first op
    int k = nRuns + (int)(inclNb - (unsigned)nb);              // index of this lane's first boundary
    for (uint32_t m = diff; m != 0u; m &= m - 1u) {
      const int i = __builtin_ctz(m) >> 1, p = 16 * d + i;
      if (k >= 1) {  // the run that ends here
        const uint32_t ty = i > 0 ? (word >> (2 * (i - 1))) & 3u : before >> 30;
        arena[cap - k] = ((uint32_t)(p - prevStart) << 2) | ty;
        if (ty == OP_M) firstM = min(firstM, k - 1);
      }
      prevStart = p;
      k++;
    }
    nRuns += __builtin_amdgcn_readlane((int)inclNb, 63);
    lastStart = max(lastStart, __builtin_amdgcn_readlane(inclLast, 63));
  }
  if (n > 0 && lane == 0) {  // the last run of the walk
    const uint32_t ty = (codes[-1 - ((n - 1) >> 4)] >> (2 * ((n - 1) & 15))) & 3u;
    arena[cap - nRuns] = ((uint32_t)(n - lastStart) << 2) | ty;
    if (ty == OP_M) firstM = min(firstM, nRuns - 1);
  }
  firstM = -wave_max(-firstM);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (lane == 0) {
    int w = cap, end = cap, leadI = 0, leadD = 0;
    if (firstM != INT32_MAX) {  // without a block nothing is kept
      w = cap - nRuns;
      end = cap - firstM;
      while (w < end && (arena[w] & 3u) != OP_M) {
        if ((arena[w] & 3u) == OP_I) leadI += (int)(arena[w] >> 2); else leadD += (int)(arena[w] >> 2);
        w++;
      }
    }
    res.q_pos = qStart + leadI;
    res.t_pos = tStart + leadD;
    res.dp_score = args.dpScore[task];
    res.n_cells = (int32_t)geo.cells;
    res.n_ops = end - w;
    args.results[task] = res;
    args.opsStart[task] = w;
  }
}

// ---------------------------------------------------------------------------
// stats_kernel: ComputeAlignmentStats, affine form (AlignmentUtils.h:535-584 with
// ComputeAlignmentScore :60-124), one wavefront per candidate, lane = op.
// The walk over the alignment columns is replaced by prefix sums over the op list.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stats_kernel(RefineBatchView batch, TraceArgs args) {
  __shared__ int matrix[25];
  __shared__ uint8_t codeOf[256];
  if (threadIdx.x < 25) matrix[threadIdx.x] = args.matrix[threadIdx.x];
  codeOf[threadIdx.x] = (uint8_t)base_code((unsigned)threadIdx.x);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int idx = blockIdx.x * 4 + wave;
  if (idx >= args.count) return;
  const int task = args.first + idx;
  RefineResult res = args.results[task];
  if (res.status & (TASK_EMPTY_GUIDE | TASK_BAD_GUIDE)) return;
  const uint8_t *qSeq = batch.seq + batch.q_off[task];
  const uint8_t *tSeq = batch.seq + batch.t_off[task];
  const uint32_t *ops = args.arena + args.arenaOff[task] + args.opsStart[task];
  const int nOps = res.n_ops;
  int qBase = res.q_pos, tBase = res.t_pos;
  int matchSum = 0, nMatch = 0, nMismatch = 0, nIns = 0, nDel = 0, gapScore = 0;
  for (int c0 = 0; c0 < nOps; c0 += 64) {
    const int i = c0 + lane;
    const uint32_t op = i < nOps ? ops[i] : 0u;
    const int ty = (int)(op & 3u), len = (int)(op >> 2);
    const unsigned qAdv = ty != OP_D ? (unsigned)len : 0u, tAdv = ty != OP_I ? (unsigned)len : 0u;
    const unsigned qIncl = wave_prefix_sum(qAdv), tIncl = wave_prefix_sum(tAdv);
    if (ty == OP_M) {
      const uint8_t *qp = qSeq + qBase + (int)(qIncl - qAdv), *tp = tSeq + tBase + (int)(tIncl - tAdv);
      score_columns(qp, tp, len, matrix, codeOf, matchSum, nMatch, nMismatch);
    } else if (i < nOps) {
      if (ty == OP_I) nIns += len; else nDel += len;
      if ((ops[i + 1] & 3u) == OP_M) {
        // last op of a maximal gap run (the op list ends with a block, so i+1 exists): one penalty per run,
        // linear cost of the run's LAST column type against open + len * extend
        int runLen = len;
        for (int b = i - 1; b >= 0 && (ops[b] & 3u) != OP_M; b--) runLen += (int)(ops[b] >> 2);
        const int linear = (ty == OP_D ? args.del : args.ins) * runLen;
        const int affine = runLen * args.ext + args.open;
        gapScore += (linear < affine) ? linear : affine;
      }
    }
    qBase += (int)__builtin_amdgcn_readlane((int)qIncl, 63);
    tBase += (int)__builtin_amdgcn_readlane((int)tIncl, 63);
  }
  int score = matchSum + gapScore;
  for (int off = 32; off > 0; off >>= 1) {
    score += __shfl_xor(score, off);
    nMatch += __shfl_xor(nMatch, off);
    nMismatch += __shfl_xor(nMismatch, off);
    nIns += __shfl_xor(nIns, off);
    nDel += __shfl_xor(nDel, off);
  }
  if (lane == 0) {
    res.score = score;
    res.n_match = nMatch;
    res.n_mismatch = nMismatch;
    res.n_ins = nIns;
    res.n_del = nDel;
    args.results[task] = res;
  }
}

// ---------------------------------------------------------------------------
// exclusive scan of n_ops over a range of candidates (single workgroup) and
// the coalesced copy of each candidate's op run into the caller's arena
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void scan_ops_kernel(RefineResult *results, int first, int count,
                                                          int64_t capacity, int64_t *total) {
  const int64_t base = *total;  // ops of the candidates before `first`
  __shared__ int64_t partial[1024];
  const int tid = threadIdx.x;
  const int per = (count + 1023) / 1024;
  const int lo = min(count, tid * per), hi = min(count, lo + per);
  int64_t sum = 0;
  for (int i = lo; i < hi; i++) sum += results[first + i].n_ops;
  partial[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int64_t v = tid >= off ? partial[tid - off] : 0;
    __syncthreads();
    partial[tid] += v;
    __syncthreads();
  }
  int64_t run = base + partial[tid] - sum;
  for (int i = lo; i < hi; i++) {
    RefineResult &r = results[first + i];
    r.ops_off = run;
    if (run + r.n_ops > capacity) r.status |= TASK_OPS_OVERFLOW;
    run += r.n_ops;
  }
  if (tid == 1023) *total = base + partial[1023];
}

__global__ __launch_bounds__(256) void compact_ops_kernel(const RefineResult *results, int first, int count,
                                                            const uint32_t *arena, const int64_t *arenaOff,
                                                            const int32_t *opsStart, uint32_t *ops) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int idx = blockIdx.x * 4 + wave;
  if (idx >= count) return;
  const RefineResult r = results[first + idx];
  if (r.status & TASK_OPS_OVERFLOW) return;
  const uint32_t *src = arena + arenaOff[first + idx] + opsStart[first + idx];
  uint32_t *dst = ops + r.ops_off;
  for (int i = lane; i < r.n_ops; i += 64) dst[i] = src[i];
}

// ---------------------------------------------------------------------------
// row_hist_kernel (BLASR_HIP_REFINE_TRACE only): where the cells of the candidates of a task list are, by row width --
// hist[b] = {rows, cells, rows whose row above is a wide one too, 192-column steps} for the width classes of kHistEdges
// ---------------------------------------------------------------------------
__constant__ int kHistEdges[10] = {48, 96, 112, 192, 384, 576, 1024, 2048, 4096, 1 << 30};
__global__ __launch_bounds__(256) void row_hist_kernel(RefineBatchView batch, FillArgs args, unsigned long long *hist) {
  const int lane = lane_id();
  const int idx = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (idx >= args.count) return;
  const int task = args.taskList[idx];
  if (task < 0) return;
  const TaskGeometry geo = args.geo[task];
  const int2 *rowTab = args.rowTab + args.rowOff[task];
  const int32_t *blk = batch.blocks + 3 * batch.blk_off[task];
  const int nb = batch.n_blocks[task];
  const int tEnd = blk[3 * (nb - 1) + 1] + blk[3 * (nb - 1) + 2];
  auto widthOf = [&](int r) {
    const unsigned next = r < geo.rows ? (unsigned)rowTab[r + 1].x : (unsigned)geo.cells;
    const int lo = rowTab[r].y, hi = lo + (int)(next - (unsigned)rowTab[r].x) - 1;
    return min(hi, tEnd - 1) - lo + 1;
  };
  for (int r = 1 + lane; r <= geo.rows; r += 64) {
    const int w = widthOf(r), pw = r > 1 ? widthOf(r - 1) : 0;
    int b = 0;
    while (w > kHistEdges[b]) b++;
    atomicAdd(&hist[4 * b + 0], 1ull);
    atomicAdd(&hist[4 * b + 1], (unsigned long long)max(w, 0));
    if (pw > kF4MaxWidth) atomicAdd(&hist[4 * b + 2], 1ull);
    atomicAdd(&hist[4 * b + 3], (unsigned long long)((max(w, 0) + 191) / 192));
  }
}

// row_profile_kernel (BLASR_HIP_DUMP_HYBRID only): per slot of a task list {passes of 40 columns of its narrow rows, its narrow
// rows, its wide rows, their steps of 192 columns} -- what a wavefront of fillh8_kernel spends on the candidate (tools/hybrid_order.py)
__global__ __launch_bounds__(256) void row_profile_kernel(RefineBatchView batch, FillArgs args, int4 *out) {
  const int lane = lane_id();
  const int idx = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (idx >= args.count) return;
  const int task = args.taskList[idx];
  int passes = 0, narrow = 0, wide = 0, steps = 0;
  if (task >= 0) {
    const TaskGeometry geo = args.geo[task];
    const int2 *rowTab = args.rowTab + args.rowOff[task];
    const int32_t *blk = batch.blocks + 3 * batch.blk_off[task];
    const int nb = batch.n_blocks[task];
    const int tEnd = blk[3 * (nb - 1) + 1] + blk[3 * (nb - 1) + 2];
    for (int r = 1 + lane; r <= geo.rows; r += 64) {
      const unsigned next = r < geo.rows ? (unsigned)rowTab[r + 1].x : (unsigned)geo.cells;
      const int lo = rowTab[r].y, hi = lo + (int)(next - (unsigned)rowTab[r].x) - 1;
      const int w = max(min(hi, tEnd - 1) - lo + 1, 0);
      if (w > kF4MaxWidth) {
        wide++;
        steps += (w + 191) / 192;
      } else {
        narrow++;
        passes += max((w + kF8Pass - 1) / kF8Pass, 1);
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    passes += __shfl_xor(passes, off);
    narrow += __shfl_xor(narrow, off);
    wide += __shfl_xor(wide, off);
    steps += __shfl_xor(steps, off);
  }
  if (lane == 0) out[idx] = make_int4(passes, narrow, wide, steps);
}

// ---------------------------------------------------------------------------
// host-callable launchers
// ---------------------------------------------------------------------------
void launch_row_profile(const RefineBatchView &batch, const FillArgs &args, int4 *out, hipStream_t stream) {
  if (args.count <= 0) return;
  hipLaunchKernelGGL(row_profile_kernel, dim3((args.count + 3) / 4), dim3(256), 0, stream, batch, args, out);
}
void launch_row_hist(const RefineBatchView &batch, const FillArgs &args, unsigned long long *hist, hipStream_t stream) {
  if (args.count <= 0) return;
  hipLaunchKernelGGL(row_hist_kernel, dim3((args.count + 3) / 4), dim3(256), 0, stream, batch, args, hist);
}
void launch_geometry(const RefineBatchView &batch, int band, int first, int count, TaskGeometry *geo,
                     hipStream_t stream, bool walkRows) {
  if (count <= 0) return;
  if (walkRows)
    hipLaunchKernelGGL(geometry_kernel, dim3((count + 3) / 4), dim3(256), 0, stream, batch, band, first, count, geo);
  else
    hipLaunchKernelGGL(geometry_blocks_kernel, dim3((count + 3) / 4), dim3(256), 0, stream, batch, band, first, count, geo);
}
void launch_geometry_rows(const RefineBatchView &batch, int band, int first, int count, const TaskGeometry *geo,
                          int2 *rowTab, const int64_t *rowOff, unsigned short *rowMeta, hipStream_t stream) {
  if (count <= 0) return;
  hipLaunchKernelGGL(geometry_rows_kernel, dim3((count + 3) / 4), dim3(256), 0, stream, batch, band, first, count, geo,
                     rowTab, rowOff, rowMeta);
}
void launch_fill(const RefineBatchView &batch, const FillArgs &args, bool globalRing, bool fast,
                 hipStream_t stream) {
  if (args.count <= 0) return;
  const dim3 grid((args.count + 3) / 4), block(256);
  if (globalRing) {
    if (fast)
      hipLaunchKernelGGL((fill_kernel<true, true>), grid, block, 0, stream, batch, args);
    else
      hipLaunchKernelGGL((fill_kernel<true, false>), grid, block, 0, stream, batch, args);
  } else {
    if (fast)
      hipLaunchKernelGGL((fill_kernel<false, true>), grid, block, 0, stream, batch, args);
    else
      hipLaunchKernelGGL((fill_kernel<false, false>), grid, block, 0, stream, batch, args);
  }
}
void launch_fill4(const RefineBatchView &batch, const FillArgs &args, int lanes, hipStream_t stream) {
  if (args.count <= 0) return;
  if (lanes == 16 && args.ext == 0)
    hipLaunchKernelGGL((fillp_kernel<16, true, false>), dim3((args.count + 15) / 16), dim3(256), 0, stream, batch, args);
  else if (lanes == 16)
    hipLaunchKernelGGL((fillp_kernel<16, false, false>), dim3((args.count + 15) / 16), dim3(256), 0, stream, batch, args);
  else  // (the caller has checked fill8_kernel's domain: extension 0, matrix entries within 6 bits)
    hipLaunchKernelGGL(fill8_kernel, dim3((args.count + 31) / 32), dim3(256), 0, stream, batch, args);
}
// the candidates with wide rows: args.taskList = the list, args.wideRingOff / wideRingSize indexed like it
void launch_fill_hybrid(const RefineBatchView &batch, const FillArgs &args, hipStream_t stream, bool eight) {
  if (args.count <= 0) return;
  if (eight)  // (the list as a queue, args.queue zeroed by the caller, who has checked fill8_kernel's domain)
    hipLaunchKernelGGL(fillh8_kernel, dim3((unsigned)args.gridGroups), dim3(256), 0, stream, batch, args);
  else if (args.ext == 0)
    hipLaunchKernelGGL((fillp_kernel<16, true, true>), dim3((args.count + 15) / 16), dim3(256), 0, stream, batch, args);
  else
    hipLaunchKernelGGL((fillp_kernel<16, false, true>), dim3((args.count + 15) / 16), dim3(256), 0, stream, batch, args);
}
int fill4_max_width() { return kF4MaxWidth; }
void launch_traceback(const RefineBatchView &batch, const TraceArgs &args, hipStream_t stream, hipEvent_t afterWalk) {
  if (args.count <= 0) {
    if (afterWalk) (void)hipEventRecord(afterWalk, stream);
    return;
  }
  hipLaunchKernelGGL(traceback_kernel<kTbRows>, dim3((args.count + kTbLanes - 1) / kTbLanes), dim3(64), 0, stream, batch, args);
  if (afterWalk) (void)hipEventRecord(afterWalk, stream);
  hipLaunchKernelGGL(runs_kernel, dim3((args.count + 3) / 4), dim3(256), 0, stream, batch, args);
  hipLaunchKernelGGL(stats_kernel, dim3((args.count + 3) / 4), dim3(256), 0, stream, batch, args);
}
void launch_scan_ops(RefineResult *results, int first, int count, int64_t capacity, int64_t *total,
                     hipStream_t stream) {
  hipLaunchKernelGGL(scan_ops_kernel, dim3(1), dim3(1024), 0, stream, results, first, count, capacity, total);
}
void launch_compact_ops(const RefineResult *results, int first, int count, const uint32_t *arena,
                        const int64_t *arenaOff, const int32_t *opsStart, uint32_t *ops, hipStream_t stream) {
  if (count <= 0) return;
  hipLaunchKernelGGL(compact_ops_kernel, dim3((count + 3) / 4), dim3(256), 0, stream, results, first, count, arena,
                     arenaOff, opsStart, ops);
}

}  // namespace blasr_hip
