// kband_api.hip -- a-11: AffineKBandAlign, Global (algorithms/alignment/AffineKBandAlign.h:12-401), the
// refiner PairwiseLocalAlign runs under -noUseGuidedAlign (alignment/Blasr.cpp:675-746).
//
// Fixed band of nCols = 2k+1 columns, column = k + t - q; three matrices: S, affine insertion I, affine
// homopolymer insertion HP (extension only where q[i] == q[i-1]), linear deletion.
//   kband_plan_kernel       SetKBoundedLengths (KBandAlign.h:36-56), matrix sizes
//   kband_fill_kernel       one wavefront per task, a row at a time in 64-column chunks.  I and HP only look
//                           up; the horizontal (deletion) dependency is linear, so S of a row is one
//                           prefix-min scan of V[c] - c*del with a carry between chunks.  The rows live in
//                           LDS and are updated in place; one packed arrow byte per cell goes to HBM.
//   kband_traceback_kernel  one lane per task: the three-matrix walk of :299-391, run-length encoded into the
//                           same op arena the guided refiner uses (M / I / D), then the shared compaction.
// Bands wider than the LDS rows (-noUseGuidedAlign gives k ~ 1.3 k at 10 kb reads and ~ 2.7 k at 30 kb,
// Blasr.cpp:991-1001, :699) take the second pair of kernels, which have no limit on k:
//   kband_wide_kernel       the matrix in (q, t) coordinates, cut into strips of 256 target columns; a wavefront
//                           sweeps a strip along anti-diagonals, lane = four adjacent columns, lane l one row
//                           behind lane l-1.  Every neighbour of a cell is then a value of the step before (own
//                           registers, or the lane to the left through one DPP shift): no scan, no LDS rows.
//                           The strip's last column goes to the next strip through a small column buffer; one
//                           arrow NIBBLE per cell goes to HBM, 16 bytes per lane per eight steps.
//   kband_wide_traceback_kernel  one wavefront per task: windows of the nibble matrix staged in LDS (the path
//                           crosses about five d-steps per lane), lane 0 walks them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/blasr_hip.h"
#include "ctx.h"
#include "refine_kernels.h"
#include "wave_utils.h"

namespace blasr_hip {

constexpr int kKbMaxCols = 1088;       // widest band held in LDS (k <= 543)
constexpr int kKbInf = 1 << 28;        // "no such cell": any value above every real score and safe to add to
// arrow byte: bits 0-1 S arrow (pathMat), bit 2 insPathMat is AffineInsOpen, bit 3 hpInsPathMat is AffineHPInsOpen
enum { KB_DIAG = 0, KB_LEFT = 1, KB_INS_CLOSE = 2, KB_HP_CLOSE = 3, KB_I_OPEN = 4, KB_H_OPEN = 8 };

struct KbandGeo {
  int32_t qLen, tLen;  // after SetKBoundedLengths
  int32_t k, nCols;
  int32_t status;
  int32_t wide;        // 1: kband_wide_kernel (strips of the (q, t) matrix), 0: kband_fill_kernel (band rows in LDS)
  int64_t cells;       // (qLen + 1) * nCols
  int64_t arenaWords;  // worst-case op count
  int64_t arrowBytes;  // arrow storage: one byte per band cell, or (wide) nStrips * stepsPad * 128 nibble bytes
  int64_t realCells;   // band cells inside the matrix: sum over q = 1..qLen of min(tLen, q + k) - max(1, q - k) + 1
  int32_t nStrips, stepsPad;  // wide only
};

// ---- the wide kernel's geometry: strips of kKwW target columns, lane = kKwC adjacent columns
constexpr int kKwC = 4, kKwW = 64 * kKwC;

__global__ __launch_bounds__(256) void kband_plan_kernel(int n, const int32_t *qLenIn, const int32_t *tLenIn,
                                                         const int32_t *kIn, KbandGeo *geo, int wideFromCols, int linear,
                                                         int magnitude) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  KbandGeo g;
  const int64_t qLength = qLenIn[i], tLength = tLenIn[i], k = kIn[i];
  g.k = (int32_t)k;
  g.nCols = (int32_t)(2 * k + 1);
  g.status = 0;
  g.wide = (!linear && 2 * k + 1 >= wideFromCols) ? 1 : 0;
  g.nStrips = g.stepsPad = 0;
  g.arrowBytes = 0;
  // every real score stays below 2^25 (the strip kernel keeps them times four), so that kKbInf plus or minus one of
  // them still compares as "no such cell"
  if (qLength <= 0 || tLength <= 0 || k < 1 || (!g.wide && 2 * k + 1 > kKbMaxCols) || k > (1 << 24) ||
      (qLength + tLength + 2) * (int64_t)max(magnitude, 1) >= (1 << 25)) {
    // nothing to align at all: the candidate had no blocks (the callers treat it as the reference does, Blasr.cpp:959-962)
    g.status = (qLength == 0 && tLength == 0 && k >= 1) ? TASK_EMPTY_GUIDE : TASK_BAD_GUIDE;
    g.qLen = g.tLen = 0;
    g.cells = 0;
    g.arenaWords = 0;
    g.realCells = 0;
  } else {
    int64_t qLen = qLength, tLen = tLength;  // KBandAlign.h:43-55
    if (tLength < qLength) qLen = min(qLength, tLength + k);
    else if (qLength < tLength) tLen = min(tLength, qLength + k);
    g.qLen = (int32_t)qLen;
    g.tLen = (int32_t)tLen;
    g.cells = (qLen + 1) * (2 * k + 1);
    g.arenaWords = qLen + tLen + 2 * k + 4;
    g.arrowBytes = g.cells;
    {  // sum of min(tLen, q + k) over q: q + k up to q = tLen - k, tLen after; sum of max(1, q - k): 1 up to q = k + 1
      const int64_t a = max((int64_t)0, min(qLen, tLen - k)), b = min(qLen, k + 1);
      const int64_t hi = a * k + a * (a + 1) / 2 + (qLen - a) * tLen;
      const int64_t lo = b + (qLen - b) * (qLen + b + 1) / 2 - (qLen - b) * k;
      g.realCells = hi - lo + qLen;
    }
    if (g.wide) {
      // a strip's rows: q in [max(1, tA - k), min(qLen, tA + W - 1 + k)], 63 more steps for the lanes' skew
      g.nStrips = (int32_t)((tLen + kKwW - 1) / kKwW);
      g.stepsPad = (int32_t)((min(qLen, (int64_t)kKwW + 2 * k) + 63 + 7) & ~(int64_t)7);
      g.arrowBytes = (int64_t)g.nStrips * g.stepsPad * 128;
    }
  }
  geo[i] = g;
}

struct KbandArgs {
  int first, count;
  int hpInsOpen, hpInsExtend, insOpen, insExtend, del;
  const int32_t *matrix;
  const KbandGeo *geo;
  const uint8_t *seq;
  const int64_t *qOff, *tOff;
  uint8_t *arrows;
  const int64_t *arrowOff;
  uint32_t *arena;
  const int64_t *arenaOff;
  int32_t *opsStart;
  RefineResult *results;
  int32_t *bound;            // wide only: per task nw + 1 columns of (qLen + 2) scores, a strip's last column
  const int64_t *boundOff;
  int64_t cellsPerWave;      // wide only: a task gets its cells / this wavefronts (rounded as waveRound says), at most blockDim / 64
  int64_t waveRound;         // added before the division: cellsPerWave - 1 rounds up, cellsPerWave / 2 to the nearest
  int maxWaves;              // no task gets more (the column buffers between strips are sized for this many + 1)
};

__global__ __launch_bounds__(256) void kband_fill_kernel(KbandArgs a) {
  __shared__ int rowS[4][kKbMaxCols + 64], rowI[4][kKbMaxCols + 64], rowH[4][kKbMaxCols + 64];
  __shared__ int m[25];
  if (threadIdx.x < 25) m[threadIdx.x] = a.matrix[threadIdx.x];
  __syncthreads();
  const int wv = threadIdx.x >> 6, lane = lane_id();
  const int idx = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wv));
  if (idx >= a.count) return;
  const int task = a.first + idx;
  const KbandGeo g = a.geo[task];
  if (g.status != 0 || g.wide) return;
  int *S = rowS[wv], *I = rowI[wv], *H = rowH[wv];
  const uint8_t *qSeq = a.seq + a.qOff[task], *tSeq = a.seq + a.tOff[task];
  uint8_t *arrows = a.arrows + a.arrowOff[task];
  const int k = g.k, nCols = g.nCols, qLen = g.qLen, tLen = g.tLen;
  const int del = a.del;
  // ---- row 0 (:96-150): S = t * del right of the diagonal, I = HP = 0 on it, "infinite" elsewhere
  for (int c = lane; c < nCols + 64; c += 64) {
    S[c] = c > k && c < nCols ? (c - k) * del : 0;
    I[c] = c == k ? 0 : kKbInf;
    H[c] = c == k ? 0 : kKbInf;
    if (c < nCols) arrows[c] = (uint8_t)(c > k ? KB_LEFT : (c == k ? (KB_I_OPEN | KB_H_OPEN) : 0));
  }
  __builtin_amdgcn_wave_barrier();
  for (int q0 = 1; q0 <= qLen; q0 += 64) {
    // raw query bytes of these 64 rows (the homopolymer test compares bytes, :183) and of the row before
    const int qi = q0 - 1 + lane;
    const int qRawV = qi < qLen ? (int)qSeq[qi] : 0;
    const int qPrevRaw0 = q0 >= 2 ? (int)qSeq[q0 - 2] : -1;
    const int nRows = min(64, qLen - q0 + 1);
    for (int r = 0; r < nRows; r++) {
      const int q = q0 + r;
      const int qRaw = __builtin_amdgcn_readlane(qRawV, r);
      const int qPrevRaw = r > 0 ? __builtin_amdgcn_readlane(qRawV, r - 1) : qPrevRaw0;
      const bool sameBase = q > 1 && qRaw == qPrevRaw;
      const int qc = base_code((unsigned)qRaw);
      uint8_t *arrowRow = arrows + (int64_t)q * nCols;
      const bool hasB = q <= k;  // the column of t = 0 is inside the band
      const int cB = k - q;
      if (hasB && lane == 0) {  // boundary cell of this row (:98-101, :122-125, :135-138)
        const int v = q * a.insExtend + a.insOpen;
        S[cB] = v;
        I[cB] = v;
        H[cB] = q * a.hpInsExtend + a.hpInsOpen;
        arrowRow[cB] = (uint8_t)KB_INS_CLOSE;
      }
      __builtin_amdgcn_wave_barrier();
      const int cLo = max(0, k - q + 1), cHi = min(2 * k, k + tLen - q);
      const int cStart = hasB ? cB : cLo;
      int carry = kKbInf, carryS = kKbInf;  // min of V - c*del left of this chunk; S of the cell left of it
      for (int cb = cStart; cb <= cHi; cb += 64) {
        const int c = cb + lane;
        const bool valid = c >= cLo && c <= cHi;
        const bool isB = hasB && c == cB;
        const int sp = S[c];
        const bool upperOk = c < 2 * k;
        const int upS = S[c + 1], upI = I[c + 1], upH = H[c + 1];
        // :169-215 the two insertion matrices look only at the cell above
        const int hpOpen = upperOk ? upS + a.hpInsOpen : kKbInf;
        const int hpExt = (sameBase && upperOk) ? upH + a.hpInsExtend : kKbInf;
        const bool hOpens = hpOpen < hpExt;
        const int minHp = hOpens ? hpOpen : hpExt;
        const int insOpen = upperOk ? upS + a.insOpen : kKbInf;
        const int insExt = upperOk ? upI + a.insExtend : kKbInf;
        const bool iOpens = insOpen < insExt;
        const int minIns = iOpens ? insOpen : insExt;
        int match = kKbInf;
        if (valid) match = sp + m[qc * 5 + base_code((unsigned)tSeq[c - k + q - 1])];
        int v = min(match, min(minIns, minHp));
        v = valid ? v : (isB ? sp : kKbInf);
        // :219-229 linear deletion: S[c] = min(V[c], S[c-1] + del) = c*del + prefix-min(V[j] - j*del)
        int x = v - lane * del;
        x = wave_prefix_min(x);
        const int e = min(wave_shr1(x, kKbInf), carry);      // cells strictly left of this lane
        const int s = min(v, e + lane * del);
        const int left = wave_shr1(s, carryS);                // S of the cell to the left
        const int delScore = (c == cStart) ? kKbInf : left + del;
        int arrow = KB_HP_CLOSE;                              // :246-257
        arrow = (s == minIns) ? KB_INS_CLOSE : arrow;
        arrow = (s == delScore) ? KB_LEFT : arrow;
        arrow = (s == match) ? KB_DIAG : arrow;
        if (valid) {
          S[c] = s;
          I[c] = minIns;
          H[c] = minHp;
          arrowRow[c] = (uint8_t)(arrow | (iOpens ? KB_I_OPEN : 0) | (hOpens ? KB_H_OPEN : 0));
        }
        carry = min(carry, __builtin_amdgcn_readlane(x, 63)) + 64 * del;
        carryS = __builtin_amdgcn_readlane(s, 63);
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  // optScore = scoreMat[qLen][k - (qLen - tLen)] (:301-304, :344)
  if (lane == 0) {
    RefineResult res;
    memset(&res, 0, sizeof(res));
    res.dp_score = S[k - (qLen - tLen)];
    res.n_cells = (int32_t)g.cells;
    a.results[task] = res;
  }
}

__global__ __launch_bounds__(64) void kband_traceback_kernel(KbandArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.count) return;
  const int task = a.first + idx;
  const KbandGeo g = a.geo[task];
  if (g.status != 0) {
    RefineResult res;
    memset(&res, 0, sizeof(res));
    res.status = g.status;
    a.results[task] = res;
    a.opsStart[task] = 0;
    return;
  }
  if (g.wide) return;  // kband_wide_traceback_kernel
  const uint8_t *arrows = a.arrows + a.arrowOff[task];
  uint32_t *arena = a.arena + a.arenaOff[task];
  const int k = g.k, nCols = g.nCols;
  int q = g.qLen, t = k - (g.qLen - g.tLen);
  int pos = (int)g.arenaWords;  // filled from the back, so the ops come out in forward order
  int curType = -1, curLen = 0, mat = 0, status = 0;
  // the walk starts at the alignment's END: gap columns met before the first Diagonal arrow form the gap list behind
  // the last block, which ArrowPathToAlignment empties (Alignment.h:246-250) -- they are not emitted
  auto emit = [&](int type) {
    if (curType < 0 && type != BLASR_HIP_OP_M) return;
    if (type == curType) curLen++;
    else {
      if (curLen) arena[--pos] = ((uint32_t)curLen << 2) | (uint32_t)curType;
      curType = type;
      curLen = 1;
    }
  };
  int guard = g.qLen + g.tLen + 2 * k + 8;
  while (q > 0 || (q == 0 && t > k)) {
    if (--guard < 0 || t < 0 || t >= nCols) {  // cannot happen for a filled matrix
      status = TASK_BAD_PATH;
      break;
    }
    const int ab = arrows[(int64_t)q * nCols + t];
    if (mat == 0) {
      const int ar = ab & 3;
      if (ar == KB_DIAG) {
        emit(BLASR_HIP_OP_M);
        q--;
      } else if (ar == KB_LEFT) {
        emit(BLASR_HIP_OP_D);
        t--;
      } else
        mat = ar == KB_INS_CLOSE ? 1 : 2;
    } else {
      const bool open = mat == 1 ? (ab & KB_I_OPEN) != 0 : (ab & KB_H_OPEN) != 0;
      if (open) mat = 0;
      emit(BLASR_HIP_OP_I);
      q--;
      t++;
    }
  }
  if (curLen) arena[--pos] = ((uint32_t)curLen << 2) | (uint32_t)curType;
  RefineResult res = a.results[task];
  res.status = status;
  res.n_ops = status ? 0 : (int32_t)(g.arenaWords - pos);
  a.results[task] = res;
  a.opsStart[task] = pos;
}


// ---------------------------------------------------------------------------------------------
// The wide band: AffineKBandAlign with no limit on k.
//
// Coordinates are (q, t), 1-based, as in the reference's loops (:160-166); cell (q, t) is in the band when
// |t - q| <= k and t <= tLen.  Strip s holds the columns tA .. tA + 255, tA = 1 + 256 s, and the rows
// qLo = max(1, tA - k) .. qHi = min(qLen, tA + 255 + k) that meet the band in them.  Lane l owns the columns
// tA + 4 l .. + 3 and works on row qLo + d - l at step d: what a cell reads (:171-245) is
//   up   (q-1, t)    the lane's own value of the step before (registers pS / pI / pH),
//   diag (q-1, t-1)  the same for the lane's columns 1..3; for column 0 the left lane's column 3 two steps ago,
//   left (q,   t-1)  this step's value of the column before; for column 0 the left lane's column 3 one step ago,
// so one DPP shift per step moves everything that crosses lanes.  Lane 0's left neighbour is the strip before
// (column t = 0 for strip 0: q * insExtend + insOpen, :98-101, :143-146), read 64 rows at a time from the column
// buffer that strip wrote; lane 63's column 3 is collected the same way and written 64 rows at a time.
// Cells outside the band are computed like any other and hold garbage; every read INTO the band from outside
// it is masked instead: `up` above the band's upper edge (t = q + k, the reference's `t < q + k` tests), `left`
// below its lower edge (t = q - k, :218-220); a diagonal neighbour of a band cell is always a band cell or row 0 /
// column 0.  Groups of eight steps in which no lane touches an edge run without the masks; groups in which some
// lane has not started or has finished its rows run under EXEC so that idle lanes keep their state (row 0, or the
// strip's last row -- the final score is read from there).
// Arrow nibble: bits 0-1 pathMat, bit 2 insPathMat is AffineInsOpen, bit 3 hpInsPathMat is AffineHPInsOpen;
// layout [strip][step / 8][lane] 16 bytes = eight steps x four columns.
// ---------------------------------------------------------------------------------------------
// Arithmetic: every score is kept times four, so that its two low bits can carry a tag.  The four candidates of
// S get the tags 0 (match) 1 (deletion) 2 (insertion matrix) 3 (homopolymer matrix): ONE minimum then yields the
// score and, in the low bits, the arrow with the reference's precedence on ties (:246-257: Diagonal, Left,
// AffineInsClose, AffineHPInsClose); the values of the insertion matrices carry their tag all the time (2 and 3, the
// same on both sides of every comparison between them), S is stripped of it.  Open-or-extend (:190-215, strict <) is the
// sign of open - extend, appended to the arrow word by one v_alignbit.
struct KwLane {
  int pS[kKwC], pI[kKwC], pH[kKwC];  // the row above, this lane's columns (x 4; pI tagged 2, pH tagged 3)
  int leftIn;                        // S(q, t0 - 1) of the current row
  int info;                          // the row's query base: byte offset of its matrix row (code * 20), and bit 28 set
                                     // when the base differs from the one before (no homopolymer extension, :180-188)
  int tb[kKwC];                      // the columns' target bases: byte offset inside a matrix row (code * 4)
  int eTop0;                         // column c may look up   <=> d - eTop0 > c
};                                   //          may look left <=> d - eTop0 - 2k < c
struct KwPenalties {                 // x 4, tags folded in
  int hpOpenT;   // hpInsOpen * 4 + 3
  int hpExt;     // hpInsExtend * 4
  int insOpenT;  // insOpen * 4 + 2
  int insExt;    // insExtend * 4
  int delT;      // del * 4 + 1
};
enum { KW_I_OPEN = 8, KW_H_OPEN = 4 };  // the nibble: insPathMat is Open, hpInsPathMat is Open, two bits of pathMat

__device__ __forceinline__ int kw_append_sign(int acc, int v) {  // (acc << 1) | (v < 0)
  return (int)__builtin_amdgcn_alignbit((unsigned)acc, (unsigned)v, 31);
}

template <bool MASK>
__device__ __forceinline__ unsigned kw_cells(KwLane &L, const int (&m)[kKwC], int info, int d, int twoK, int leftPrev,
                                             const KwPenalties &P) {
  const int hpExtAdd = (info & ~0xff) + P.hpExt;  // + kKbInf outside a homopolymer
  const int x = d - L.eTop0, y = x - twoK;
  int diag = leftPrev, left = L.leftIn;
  int pack = 0;
#pragma unroll
  for (int c = 0; c < kKwC; c++) {
    const int upS = L.pS[c];
    const int match = diag + m[c];
    const int hpOpen = upS + P.hpOpenT, hpExt = L.pH[c] + hpExtAdd;
    int minHp = min(hpOpen, hpExt);
    const int insOpen = upS + P.insOpenT, insExt = L.pI[c] + P.insExt;
    int minIns = min(insOpen, insExt);
    int delScore = left + P.delT;
    if (MASK) {
      const bool upperOk = x > c;
      minHp = upperOk ? minHp : kKbInf;
      minIns = upperOk ? minIns : kKbInf;
      delScore = (y < c) ? delScore : kKbInf;
    }
    const int tagged = min(min(match, delScore), min(minIns, minHp));
    pack = kw_append_sign(pack, insOpen - insExt);
    pack = kw_append_sign(pack, hpOpen - hpExt);
    pack = (pack << 2) | (tagged & 3);
    const int sc = tagged & ~3;
    diag = upS;
    left = sc;
    L.pS[c] = sc;
    L.pI[c] = minIns;
    L.pH[c] = minHp;
  }
  return (unsigned)pack;
}

// the four substitution scores of a step: the row's query base against the lane's target bases (25 ints in LDS)
__device__ __forceinline__ void kw_lookup(const KwLane &L, const char *mBytes, int info, int (&m)[kKwC]) {
  const int qoff = info & 0xff;
#pragma unroll
  for (int c = 0; c < kKwC; c++) m[c] = *reinterpret_cast<const int *>(mBytes + qoff + L.tb[c]);
}

// eight steps d0 .. d0 + 7 of a strip; returns the lane's 16 bytes of arrows (dword j: steps 2j, 2j+1; a step's
// four columns, column 0 in the highest nibble of its half).  The substitution scores of step d + 1 are looked up while
// step d computes (the row's base reaches the lane one step ahead of the row), so that only the first step of a group
// waits for LDS.
template <bool RAMP, bool MASK>
__device__ __forceinline__ uint4 kw_group(KwLane &L, const char *mBytes, int d0, int lane, int nRows, int twoK,
                                          const KwPenalties &P, int qBlock, int bBlock, int &outB, int32_t *bOut) {
  unsigned acc[4] = {0, 0, 0, 0};
  int info = wave_shr1(L.info, __builtin_amdgcn_readlane(qBlock, d0 & 63));
  int m[kKwC];
  kw_lookup(L, mBytes, info, m);
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int d = d0 + r;
    int infoNext = info, mNext[kKwC] = {0, 0, 0, 0};
    if (r < 7) {
      infoNext = wave_shr1(info, __builtin_amdgcn_readlane(qBlock, (d + 1) & 63));
      kw_lookup(L, mBytes, infoNext, mNext);
    }
    const int edgeS = __builtin_amdgcn_readlane(bBlock, d & 63);
    const int leftPrev = L.leftIn;
    L.leftIn = wave_shr1(L.pS[kKwC - 1], edgeS);
    unsigned pack = 0;
    if (RAMP) {
      if ((unsigned)(d - lane) < (unsigned)nRows) pack = kw_cells<MASK>(L, m, info, d, twoK, leftPrev, P);
    } else {
      pack = kw_cells<MASK>(L, m, info, d, twoK, leftPrev, P);
    }
    acc[r >> 1] = (acc[r >> 1] << 16) | pack;
    asm volatile("" : "+v"(acc[r >> 1]));  // the arrow word is built here, not sunk to the store after the eighth step
    // the strip's last column, row rr of the strip: 64 rows per store
    const int rr = d - 63;
    if (!RAMP || (rr >= 0 && rr < nRows)) {
      const int v63 = __builtin_amdgcn_readlane(L.pS[kKwC - 1], 63);
      outB = lane == (rr & 63) ? v63 : outB;
      if ((rr & 63) == 63 || (RAMP && rr == nRows - 1)) {
        if (lane <= (rr & 63)) bOut[(rr & ~63) + lane] = outB;
      }
    }
    info = infoNext;
#pragma unroll
    for (int c = 0; c < kKwC; c++) m[c] = mNext[c];
    __builtin_amdgcn_sched_barrier(0);
  }
  L.info = info;
  return make_uint4(acc[0], acc[1], acc[2], acc[3]);
}

// One workgroup per task, NW = blockDim.x / 64 wavefronts: wavefront w sweeps the strips s = w, w + NW, ... and strip
// s + 1 follows strip s a few 64-row blocks behind (it needs S(q, last column of s) for the rows it enters).  The
// columns cross in HBM (NW + 1 buffers per task: strip s writes buffer s mod (NW + 1), which its reader, strip s + 1,
// has long left when strip s + NW + 1 -- the same wavefront as s + 1 -- writes it again); how far a strip has got is
// published in LDS: {strip, rows of it stored}.  A wavefront only ever waits for a strip with a smaller number, so
// the waits cannot form a cycle; all wavefronts of a workgroup are resident together.
struct KwProgress {
  unsigned long long word[16];  // per wavefront: (strip << 32) | rows of it stored -- ONE word, so that a reader can never
                                // pair a new strip number with the previous strip's row count
};
__device__ __forceinline__ void kw_publish(KwProgress *pg, int w, int strip, int rows, bool newStrip) {
  // the stores of the column (and everything before them) have reached L2 before the count moves
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long v = ((unsigned long long)(unsigned)strip << 32) | (unsigned)(newStrip ? 0 : rows);
  __hip_atomic_store(&pg->word[w], v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void kw_wait(const KwProgress *pg, int w, int strip, int rows) {
  for (;;) {
    const unsigned long long v = __hip_atomic_load(&pg->word[w], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int st = (int)(v >> 32), rw = (int)(unsigned)v;
    if (st > strip || (st == strip && rw >= rows)) break;
    __builtin_amdgcn_s_sleep(8);
  }
}

__global__ __launch_bounds__(1024) void kband_wide_kernel(KbandArgs a) {
  __shared__ int m[32];
  __shared__ KwProgress progress;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
  const int idx = (int)blockIdx.x;
  if (idx >= a.count) return;
  const int task = a.first + idx;
  const KbandGeo g = a.geo[task];
  if (g.status != 0 || !g.wide) return;
  // wavefronts of this task: by its size, so that the launch's wavefronts all carry about the same number of cells (one
  // outlier -- a candidate whose SDP alignment drifted far -- would otherwise run long after the others have finished)
  const int nw = (int)min((int64_t)min(min((int)(blockDim.x >> 6), a.maxWaves), g.nStrips),
                          max((int64_t)1, (g.realCells + a.waveRound) / a.cellsPerWave));
  if (threadIdx.x < 25) m[threadIdx.x] = a.matrix[threadIdx.x] * 4;
  if (threadIdx.x < 16) {
    // "the strip before my first": nothing of strip w is there yet
    progress.word[threadIdx.x] = (unsigned long long)(unsigned)((int)threadIdx.x - nw) << 32;
  }
  __syncthreads();
  if (w >= nw) return;
  const uint8_t *qSeq = a.seq + a.qOff[task], *tSeq = a.seq + a.tOff[task];
  uint4 *arrows = reinterpret_cast<uint4 *>(a.arrows + a.arrowOff[task]);
  int32_t *bound = a.bound + a.boundOff[task];
  const int k = g.k, qLen = g.qLen, tLen = g.tLen, twoK = 2 * g.k, bStride = g.qLen + 2;
  const char *mBytes = reinterpret_cast<const char *>(m);
  const KwPenalties P{a.hpInsOpen * 4 + 3, a.hpInsExtend * 4, a.insOpen * 4 + 2, a.insExtend * 4, a.del * 4 + 1};
  const int del4 = a.del * 4;
  const int wPrev = (w + nw - 1) % nw;  // the wavefront that sweeps the strip to the left of mine
  KwLane L;
  for (int s = w; s < g.nStrips; s += nw) {
    const int tA = 1 + s * kKwW;
    const int qLo = max(1, tA - k), qHi = min(qLen, tA + kKwW - 1 + k);
    const int nRows = qHi - qLo + 1, nSteps = nRows + 63;
    // the strip to the left: its first row, and how many rows it has in all
    const int qLoPrev = max(1, tA - kKwW - k), nRowsPrev = min(qLen, tA - 1 + k) - qLoPrev + 1;
    const int32_t *bIn = bound + (int64_t)((s + nw) % (nw + 1)) * bStride;  // = (s - 1) mod (nw + 1)
    int32_t *bOut = bound + (int64_t)(s % (nw + 1)) * bStride;
    kw_publish(&progress, w, s, 0, true);
    const int t0 = tA + kKwC * lane;
#pragma unroll
    for (int c = 0; c < kKwC; c++) {
      const int t = t0 + c;
      L.tb[c] = (t <= tLen ? base_code((unsigned)tSeq[t - 1]) : 4) * 4;
      // the row above the strip's first: row 0 (:147-150, :127-132) when the strip starts at row 1, otherwise a
      // row whose cells of these columns lie above the band (never looked at: the masks)
      L.pS[c] = qLo == 1 ? t * del4 : kKbInf;
      L.pI[c] = kKbInf;
      L.pH[c] = kKbInf;
    }
    L.eTop0 = t0 - k - qLo + lane;
    int firstDiag;  // S(qLo - 1, tA - 1)
    if (qLo == 1) firstDiag = (tA - 1) * del4;
    else {
      kw_wait(&progress, wPrev, s - 1, min(nRowsPrev, qLo - qLoPrev));
      firstDiag = __hip_atomic_load(bIn + qLo - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    L.leftIn = wave_shr1(L.pS[kKwC - 1], firstDiag);
    L.info = 0;
    int qBlock = 0, bBlock = 0, outB = 0;
    uint4 *arrowStrip = arrows + (int64_t)s * g.stepsPad * 8;
    const int fastFrom = tA - k - qLo + 63 + kKwW, fastTo = tA + k - qLo;  // no lane at an edge: fastFrom <= d < fastTo
    for (int d0 = 0; d0 < nSteps; d0 += 8) {
      if ((d0 & 63) == 0) {  // the next 64 rows entering at lane 0: query bases, the left neighbour's column
        const int q = qLo + d0 + lane;
        if (s > 0) kw_wait(&progress, wPrev, s - 1, min(nRowsPrev, qLo + d0 + 64 - qLoPrev));
        int info = 0, left = kKbInf;
        if (q <= qLen) {
          const int raw = qSeq[q - 1];
          const int prev = q >= 2 ? (int)qSeq[q - 2] : -1;
          info = base_code((unsigned)raw) * 20 | (raw == prev ? 0 : kKbInf);
          left = s == 0 ? (q * a.insExtend + a.insOpen) * 4
                        : __hip_atomic_load(bIn + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        qBlock = info;
        bBlock = left;
      }
      const bool ramp = d0 < 63 || d0 + 7 >= nRows;
      const bool fast = !ramp && d0 >= fastFrom && d0 + 7 < fastTo;
      uint4 out;
      if (ramp) out = kw_group<true, true>(L, mBytes, d0, lane, nRows, twoK, P, qBlock, bBlock, outB, bOut + qLo);
      else if (fast) out = kw_group<false, false>(L, mBytes, d0, lane, nRows, twoK, P, qBlock, bBlock, outB, bOut + qLo);
      else out = kw_group<false, true>(L, mBytes, d0, lane, nRows, twoK, P, qBlock, bBlock, outB, bOut + qLo);
      arrowStrip[(int64_t)(d0 >> 3) * 64 + lane] = out;
      // rows of the last column stored so far: through step d0 + 7, i.e. row d0 + 7 - 63 of the strip, in whole blocks
      const int stored = min(nRows, (d0 + 8 - 63) & ~63);
      if (d0 + 8 >= nSteps) kw_publish(&progress, w, s, nRows, false);
      else if (stored > 0 && ((d0 + 8 - 63) & 63) < 8) kw_publish(&progress, w, s, stored, false);
    }
    if (s == g.nStrips - 1) {  // optScore = S(qLen, tLen) (:301-304, :344): the lanes stopped on the strip's last row
      const int lStar = (tLen - tA) >> 2, cStar = (tLen - tA) & 3;
      int v = L.pS[0];
      v = cStar == 1 ? L.pS[1] : v;
      v = cStar == 2 ? L.pS[2] : v;
      v = cStar == 3 ? L.pS[3] : v;
      if (lane == lStar) {
        RefineResult res;
        memset(&res, 0, sizeof(res));
        res.dp_score = v >> 2;
        res.n_cells = (int32_t)min(g.cells, (int64_t)0x7fffffff);
        a.results[task] = res;
      }
    }
  }
}

// One wavefront per task.  The walk of :299-391 in (q, t): from (qLen, tLen) (Global, :301-304), Diagonal -> (q-1, t-1),
// Left -> (q, t-1), an insertion matrix -> (q-1, t) until its Open arrow.  Column 0 and row 0 hold no stored arrows:
// (q, 0) is AffineInsClose then q insertions (:98-101, :143-146), (0, t) is t Left arrows (:147-150).
// In a strip's [step / 8][lane] layout a move lowers the step by 0..2 and the lane by 0..1, about five steps per lane:
// a window of 16 step-groups x 32 lanes (8 KB) anchored at the current cell holds the next ~25 lanes of the path.
__global__ __launch_bounds__(256) void kband_wide_traceback_kernel(KbandArgs a) {
  __shared__ uint4 win[4][16 * 32];
  const int wv = threadIdx.x >> 6, lane = lane_id();
  const int idx = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wv));
  if (idx >= a.count) return;
  const int task = a.first + idx;
  const KbandGeo g = a.geo[task];
  if (g.status != 0 || !g.wide) return;
  const uint4 *arr = reinterpret_cast<const uint4 *>(a.arrows + a.arrowOff[task]);
  uint32_t *arena = a.arena + a.arenaOff[task];
  const int k = g.k;
  int q = g.qLen, t = g.tLen, mat = 0, status = 0;
  int pos = (int)g.arenaWords, curType = -1, curLen = 0;
  auto emit = [&](int type, int n) {  // (gap columns before the first Diagonal arrow: the emptied last gap list)
    if (n <= 0 || (curType < 0 && type != BLASR_HIP_OP_M)) return;
    if (type == curType) curLen += n;
    else {
      if (curLen) {
        --pos;
        if (lane == 0) arena[pos] = ((uint32_t)curLen << 2) | (uint32_t)curType;
      }
      curType = type;
      curLen = n;
    }
  };
  int guard = 2 * (g.qLen + g.tLen) + 8;
  const uint32_t *w32 = reinterpret_cast<const uint32_t *>(win[wv]);
  while (q > 0 && t > 0 && status == 0) {
    const int s = (t - 1) >> 8, lTop = ((t - 1) & 255) >> 2;
    const int qLo = max(1, 1 + s * kKwW - k);
    const int rowTop = (q - qLo + lTop) >> 3;
    const uint4 *strip = arr + (int64_t)s * g.stepsPad * 8;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int id = lane + 64 * i, row = rowTop - (id >> 5), ln = lTop - (id & 31);
      if (row >= 0 && ln >= 0) win[wv][id] = strip[(int64_t)row * 64 + ln];
    }
    __builtin_amdgcn_wave_barrier();
    while (q > 0 && t > 0) {
      const int col = (t - 1) & 255;
      if (((t - 1) >> 8) != s) break;
      const int l = col >> 2, cc = col & 3, d = q - qLo + l;
      const int wr = rowTop - (d >> 3), wl = lTop - l;
      if (wr > 15 || wl > 31) break;
      if (--guard < 0 || d < 0) {  // cannot happen for a filled matrix
        status = TASK_BAD_PATH;
        break;
      }
      const uint32_t word = (uint32_t)__builtin_amdgcn_readfirstlane((int)w32[(wr * 32 + wl) * 4 + ((d & 7) >> 1)]);
      const int nib = (int)(word >> ((((d & 1) ^ 1) << 4) + (3 - cc) * 4)) & 15;
      if (mat == 0) {
        const int ar = nib & 3;
        if (ar == KB_DIAG) {
          emit(BLASR_HIP_OP_M, 1);
          q--;
          t--;
        } else if (ar == KB_LEFT) {
          emit(BLASR_HIP_OP_D, 1);
          t--;
        } else
          mat = ar == KB_INS_CLOSE ? 1 : 2;
      } else {
        if (nib & (mat == 1 ? KW_I_OPEN : KW_H_OPEN)) mat = 0;
        emit(BLASR_HIP_OP_I, 1);
        q--;
      }
    }
  }
  if (status == 0) {
    if (t == 0) emit(BLASR_HIP_OP_I, q);
    else emit(BLASR_HIP_OP_D, t);
  }
  if (curLen) {
    --pos;
    if (lane == 0) arena[pos] = ((uint32_t)curLen << 2) | (uint32_t)curType;
  }
  if (lane == 0) {
    RefineResult res = a.results[task];
    res.status = status;
    res.n_ops = status ? 0 : (int32_t)(g.arenaWords - pos);
    a.results[task] = res;
    a.opsStart[task] = pos;
  }
}


// ---------------------------------------------------------------------------------------------
// ComputeAlignmentStats (AlignmentUtils.h:535-584) on a k-band result, as PairwiseLocalAlign calls it (Blasr.cpp:740):
// the alignment keeps its first gap list (no RemoveAlignmentPrefixGaps), its columns are gaps[0], then blocks and gap
// lists in turn (CreateAlignmentStrings :448-533); the op list ends with the last block.  One wavefront per task, lane =
// op, prefix sums over the op list instead of the walk.  Score (ComputeAlignmentScore :60-124): the matrix over the
// aligned columns, and per gap column ins / del (non-affine, the form of :740), or per maximal gap run the cheaper of
// the linear cost of its last column's kind and open + length * extend (affine).
// ---------------------------------------------------------------------------------------------
struct KbandStatsArgs {
  int n;
  const uint8_t *seq;
  const int64_t *qOff, *tOff;
  RefineResult *results;
  const uint32_t *ops;
  const int32_t *matrix;
  int ins, del, open, ext, useAffine;
};

__global__ __launch_bounds__(256) void kband_stats_kernel(KbandStatsArgs a) {
  __shared__ int matrix[25];
  if (threadIdx.x < 25) matrix[threadIdx.x] = a.matrix[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (task >= a.n) return;
  RefineResult res = a.results[task];
  if (res.status != 0) return;
  const uint8_t *qSeq = a.seq + a.qOff[task], *tSeq = a.seq + a.tOff[task];
  const uint32_t *ops = a.ops + res.ops_off;
  const int nOps = res.n_ops;
  int qBase = res.q_pos, tBase = res.t_pos;
  int matchSum = 0, nMatch = 0, nMismatch = 0, nIns = 0, nDel = 0, gapScore = 0;
  for (int c0 = 0; c0 < nOps; c0 += 64) {
    const int i = c0 + lane;
    const uint32_t op = i < nOps ? ops[i] : 0u;
    const int ty = (int)(op & 3u), len = (int)(op >> 2);
    const unsigned qAdv = ty != BLASR_HIP_OP_D ? (unsigned)len : 0u, tAdv = ty != BLASR_HIP_OP_I ? (unsigned)len : 0u;
    const unsigned qIncl = wave_prefix_sum(qAdv), tIncl = wave_prefix_sum(tAdv);
    if (ty == BLASR_HIP_OP_M) {
      const uint8_t *qp = qSeq + qBase + (int)(qIncl - qAdv), *tp = tSeq + tBase + (int)(tIncl - tAdv);
      for (int j = 0; j < len; j++) {
        const int qc = base_code(qp[j]), tc = base_code(tp[j]);
        matchSum += matrix[qc * 5 + tc];
        if (qc == tc) nMatch++; else nMismatch++;
      }
    } else if (i < nOps) {
      if (ty == BLASR_HIP_OP_I) nIns += len; else nDel += len;
      if (!a.useAffine) {
        gapScore += len * (ty == BLASR_HIP_OP_I ? a.ins : a.del);
      } else if ((ops[i + 1] & 3u) == BLASR_HIP_OP_M) {  // last op of a maximal gap run (a block follows every run)
        int runLen = len;
        for (int b = i - 1; b >= 0 && (ops[b] & 3u) != BLASR_HIP_OP_M; b--) runLen += (int)(ops[b] >> 2);
        const int linear = (ty == BLASR_HIP_OP_D ? a.del : a.ins) * runLen;
        const int affine = runLen * a.ext + a.open;
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
    a.results[task] = res;
  }
}

// ---------------------------------------------------------------------------------------------
// KBandAlign with a DistanceMatrixScoreFunction (KBandAlign.h:74-403): one matrix, linear gaps, alignment
// types Global / QueryFit / TargetFit / Fit.  The reference fills its matrices with 0 / NoArrow and its loops
// skip some cells that later ones read (row 0 beyond t < tLen, the TargetFit column); to get exactly those
// values the kernel keeps a full image of the current and the previous matrix row in LDS, zero-filled before
// the boundary values and the computed cells go in, and writes an arrow byte for EVERY cell of the matrix.
// ---------------------------------------------------------------------------------------------
enum { KL_DIAG = 0, KL_UP = 1, KL_LEFT = 2, KL_NONE = 3 };
enum { KT_GLOBAL = 1, KT_QUERYFIT = 2, KT_TARGETFIT = 3, KT_FIT = 7 };

struct KbandLinearArgs {
  int first, count;
  int ins, del, fnIns, fnDel, alignType;
  const int32_t *matrix;
  const KbandGeo *geo;
  const uint8_t *seq;
  const int64_t *qOff, *tOff;
  uint8_t *arrows;
  const int64_t *arrowOff;
  int32_t *scores;          // per task: last row image (nCols), S[q][t = tLen] for q = 0..qLen, then S[q][column
                            // k - (qLen - tLen)] for q = 0..qLen (where the reference starts a TargetFit walk)
  const int64_t *scoreOff;
  uint32_t *arena;
  const int64_t *arenaOff;
  int32_t *opsStart;
  RefineResult *results;
};

__global__ __launch_bounds__(256) void kband_linear_fill_kernel(KbandLinearArgs a) {
  __shared__ int rowImg[4][2][kKbMaxCols + 64];
  __shared__ int m[25];
  if (threadIdx.x < 25) m[threadIdx.x] = a.matrix[threadIdx.x];
  __syncthreads();
  const int wv = threadIdx.x >> 6, lane = lane_id();
  const int idx = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wv));
  if (idx >= a.count) return;
  const int task = a.first + idx;
  const KbandGeo g = a.geo[task];
  if (g.status != 0) return;
  const uint8_t *qSeq = a.seq + a.qOff[task], *tSeq = a.seq + a.tOff[task];
  uint8_t *arrows = a.arrows + a.arrowOff[task];
  int32_t *scores = a.scores + a.scoreOff[task];
  const int k = g.k, nCols = g.nCols, qLen = g.qLen, tLen = g.tLen, fnDel = a.fnDel;
  const bool fitQ = a.alignType == KT_QUERYFIT || a.alignType == KT_FIT;
  const bool fitT = a.alignType == KT_TARGETFIT || a.alignType == KT_FIT;
  int *prev = rowImg[wv][0], *cur = rowImg[wv][1];
  // ---- row 0 (:113-141)
  for (int c = lane; c < nCols + 64; c += 64) {
    int v = 0, ar = KL_NONE;
    const int t = c - k;
    if (c < nCols && t >= 1 && t <= k && t < tLen && (a.alignType == KT_GLOBAL || fitQ)) {
      v = a.alignType == KT_GLOBAL ? t * a.del : 0;
      ar = KL_LEFT;
    }
    if (c == k) ar = KL_DIAG;
    prev[c] = v;
    if (c < nCols) arrows[c] = (uint8_t)ar;
  }
  const int cEnd = k - (qLen - tLen);  // the Global end column; TargetFit / Fit keep it for whatever row they pick
  if (lane == 0) {
    if (tLen <= k) scores[nCols] = prev[k + tLen];  // S[0][t = tLen]
    scores[nCols + qLen + 1] = prev[cEnd];
  }
  __builtin_amdgcn_wave_barrier();
  for (int q = 1; q <= qLen; q++) {
    const int qc = base_code((unsigned)qSeq[q - 1]);
    uint8_t *arrowRow = arrows + (int64_t)q * nCols;
    // the zero-filled row with its boundary cells
    for (int c = lane; c < nCols + 64; c += 64) {
      cur[c] = 0;
      if (c < nCols) arrowRow[c] = (uint8_t)KL_NONE;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      if (q <= k) {
        cur[k - q] = q * a.ins;
        arrowRow[k - q] = (uint8_t)KL_UP;
      }
      if (fitT && q <= k && q < qLen) {
        cur[0] = 0;
        arrowRow[0] = (uint8_t)KL_UP;
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int cLo = max(0, k - q + 1), cHi = min(2 * k, k + tLen - q);
    const int cSeed = cLo - 1;  // the cell left of the first computed one (t = 0 or outside the band)
    // carry: min of S[j] + (cb - j)*fnDel over the cells left of the chunk, so that adding lane*fnDel gives the
    // deletion chain's value at the lane; seeded with the cell left of the first computed one
    int carry = cSeed >= 0 ? cur[cSeed] + fnDel : kKbInf;
    int carryS = cSeed >= 0 ? cur[cSeed] : kKbInf;
    for (int cb = cLo; cb <= cHi; cb += 64) {
      const int c = cb + lane;
      const bool valid = c <= cHi;
      int match = kKbInf, insS = kKbInf;
      if (valid) {
        match = prev[c] + m[base_code((unsigned)tSeq[c - k + q - 1]) * 5 + qc];  // scoreMatrix[ref][query]
        if (c < 2 * k) insS = prev[c + 1] + a.fnIns;
      }
      const int v = min(match, insS);
      int x = v - lane * fnDel;
      x = wave_prefix_min(x);
      const int e = min(wave_shr1(x, kKbInf), carry);
      const int s = min(v, e + lane * fnDel);
      const int left = wave_shr1(s, carryS);
      const bool hasLeft = c > 0;  // t == q - k has no deletion (:151-153)
      const int delScore = hasLeft ? left + fnDel : kKbInf;
      const int arrow = (s == match) ? KL_DIAG : ((s == delScore) ? KL_LEFT : KL_UP);
      if (valid) {
        cur[c] = s;
        arrowRow[c] = (uint8_t)arrow;
      }
      carry = min(carry, __builtin_amdgcn_readlane(x, 63)) + 64 * fnDel;
      carryS = __builtin_amdgcn_readlane(s, 63);
      __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
      const int cT = k + tLen - q;  // the cell of t = tLen in this row, if inside the band
      if (cT >= 0 && cT < nCols) scores[nCols + q] = cur[cT];
      scores[nCols + qLen + 1 + q] = cur[cEnd];
    }
    int *tmp = prev;
    prev = cur;
    cur = tmp;
    __builtin_amdgcn_wave_barrier();
  }
  for (int c = lane; c < nCols; c += 64) scores[c] = prev[c];  // the last row
}

__global__ __launch_bounds__(64) void kband_linear_traceback_kernel(KbandLinearArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.count) return;
  const int task = a.first + idx;
  const KbandGeo g = a.geo[task];
  RefineResult res;
  memset(&res, 0, sizeof(res));
  if (g.status != 0) {
    res.status = g.status;
    a.results[task] = res;
    a.opsStart[task] = 0;
    return;
  }
  const uint8_t *arrows = a.arrows + a.arrowOff[task];
  const int32_t *scores = a.scores + a.scoreOff[task];
  uint32_t *arena = a.arena + a.arenaOff[task];
  const int k = g.k, nCols = g.nCols, qLen = g.qLen, tLen = g.tLen, at = a.alignType;
  // ---- end cell (:228-296)
  int q = qLen, t = k - (qLen - tLen);
  const int globalMin = scores[t];
  int minLastCol = globalMin, minLastRow = globalMin, minLastColIdx = 0, minLastRowIdx = 0;
  if (at == KT_QUERYFIT || at == KT_FIT) {
    bool set = false;
    for (int t2 = q - k; t2 < qLen + k + 1; t2++) {
      if (t2 < 1 || t2 > tLen) continue;
      const int v = scores[k + t2 - q];
      if (!set || v < minLastRow) {
        set = true;
        minLastRow = v;
        minLastRowIdx = t2;
      }
    }
    if (set) t = k - (q - minLastRowIdx);
  }
  if (at == KT_TARGETFIT || at == KT_FIT) {
    const int t2 = k - (qLen - tLen);
    bool set = false;
    for (int q2 = qLen; (uint32_t)q2 >= (uint32_t)tLen - (uint32_t)k && q2 > 0; q2--) {  // unsigned, as :279
      const int v = scores[nCols + q2];
      if (!set || v < minLastCol) {
        minLastCol = v;
        set = true;
        minLastColIdx = q2;
      }
    }
    if (at == KT_FIT) {
      if (minLastCol < minLastRow) {
        t = t2;
        q = minLastColIdx;
      }
    } else {
      t = t2;
      q = minLastColIdx;
    }
  }
  // optScore = scoreMat[q][t]: on the last row, or (TargetFit / Fit) in column k - (qLen - tLen) of row q
  res.dp_score = q == qLen ? scores[t] : scores[nCols + qLen + 1 + q];
  // ---- traceback (:317-381)
  int pos = (int)g.arenaWords, curType = -1, curLen = 0;
  auto emit = [&](int type) {
    if (type == curType) curLen++;
    else {
      if (curLen) arena[--pos] = ((uint32_t)curLen << 2) | (uint32_t)curType;
      curType = type;
      curLen = 1;
    }
  };
  int guard = qLen + tLen + 2 * k + 8;
  for (;;) {
    bool go;
    if (at == KT_GLOBAL || at == KT_QUERYFIT) go = q > 0 && (t < k ? (k - t != q) : true);
    else if (at == KT_FIT) go = q > 0 && (t < k ? (k - t != q) : true) && (q <= k ? k - q != t : true);
    else go = q > 0 && (q < k ? k - q != t : true);
    if (!go || --guard < 0 || t < 0 || t >= nCols) break;
    const int ar = arrows[(int64_t)q * nCols + t];
    if (ar == KL_NONE) break;
    if (ar == KL_DIAG) {
      emit(BLASR_HIP_OP_M);
      q--;
    } else if (ar == KL_UP) {
      emit(BLASR_HIP_OP_I);
      q--;
      t++;
    } else {
      emit(BLASR_HIP_OP_D);
      t--;
    }
  }
  if (curLen) arena[--pos] = ((uint32_t)curLen << 2) | (uint32_t)curType;
  res.q_pos = q;
  res.t_pos = t < k ? (k - t) - q : (t - k) - q;
  res.n_cells = (int32_t)g.cells;
  res.n_ops = (int32_t)(g.arenaWords - pos);
  a.results[task] = res;
  a.opsStart[task] = pos;
}

}  // namespace blasr_hip

using namespace blasr_hip;

extern "C" {

}  // extern "C"

namespace {
// p: AffineKBandAlign parameters, or lp: KBandAlign parameters (exactly one is set)
int kband_run(blasr_hip_ctx *ctx, const blasr_hip_kband_batch *b, const blasr_hip_kband_params *p,
              const blasr_hip_kband_linear_params *lp, blasr_hip_refine_result *results, uint32_t *ops,
              int64_t ops_capacity, int64_t *ops_used) {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (!b || (!p && !lp) || !ops_used || b->n < 0 ||
      (b->n > 0 && (!b->seq || !b->q_off || !b->q_len || !b->t_off || !b->t_len || !b->k || !results || !ops)))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "kband: bad arguments");
  *ops_used = 0;
  const int n = b->n;
  if (n == 0) return BLASR_HIP_OK;
  const int32_t *hostMatrix = p ? p->matrix : lp->matrix;
  int64_t magnitude = 0;
  for (int i = 0; i < 25; i++) magnitude = std::max<int64_t>(magnitude, std::abs((int64_t)hostMatrix[i]));
  if (p) {
    for (int v : {p->hp_ins_open, p->hp_ins_extend, p->ins_open, p->ins_extend, p->del})
      magnitude = std::max<int64_t>(magnitude, std::abs((int64_t)v));
  } else {
    for (int v : {lp->ins, lp->del, lp->fn_ins, lp->fn_del}) magnitude = std::max<int64_t>(magnitude, std::abs((int64_t)v));
    if (lp->align_type != 1 && lp->align_type != 2 && lp->align_type != 3 && lp->align_type != 7)
      return blasr_hip_fail(ctx, BLASR_HIP_ERR_UNSUPPORTED, "kband: align_type must be Global, QueryFit, TargetFit or Fit");
  }
  if ((p ? p->del : lp->fn_del) < 0 || magnitude >= (1 << 10))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_UNSUPPORTED, "kband: deletion penalty must be >= 0, |penalties| < 1024");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  struct Dev {
    void *p = nullptr;
    ~Dev() {
      if (p) (void)hipFree(p);
    }
  };
  Dev dGeo, dMatrix, dArrowOff, dArenaOff, dOpsStart, dTotal, dScoreOff;
  HIP_TRY(ctx, hipMalloc(&dGeo.p, sizeof(KbandGeo) * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dMatrix.p, sizeof(int32_t) * 32));
  HIP_TRY(ctx, hipMalloc(&dArrowOff.p, sizeof(int64_t) * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dArenaOff.p, sizeof(int64_t) * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dOpsStart.p, sizeof(int32_t) * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dTotal.p, sizeof(int64_t)));
  HIP_TRY(ctx, hipMalloc(&dScoreOff.p, sizeof(int64_t) * (size_t)n));
  HIP_TRY(ctx, hipMemcpyAsync(dMatrix.p, hostMatrix, sizeof(int32_t) * 25, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemsetAsync(dTotal.p, 0, sizeof(int64_t), st));
  // bands from this many columns on take the strip kernel (any k); below it the band's rows live in LDS (which is also
  // the only form KBandAlign's four alignment types exist in: their bands stay within kKbMaxCols).
  // BLASR_HIP_KBAND_WIDE_COLS moves the switch (tests run the small vectors through the strip kernel with it)
  int wideFromCols = 100;  // measured (tools/exp_kband_sweep.sh, 10 kb reads): 294 against 237 Gcells/s at k = 60, 1174 against 297 at k = 543
  if (const char *e = getenv("BLASR_HIP_KBAND_WIDE_COLS")) wideFromCols = std::max(3, atoi(e));
  hipLaunchKernelGGL(kband_plan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, b->q_len, b->t_len, b->k,
                     static_cast<KbandGeo *>(dGeo.p), wideFromCols, lp ? 1 : 0, (int)magnitude);
  std::vector<KbandGeo> geo((size_t)n);
  HIP_TRY(ctx, hipMemcpyAsync(geo.data(), dGeo.p, sizeof(KbandGeo) * (size_t)n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  // chunks of tasks whose arrow matrices + op arenas fit the workspace
  const int64_t limit = std::max<int64_t>(ctx->workspaceLimit, 1 << 20);
  // wavefronts per task of the strip kernel (one 30 kb read at k = 2.7 k holds 86 MB of arrows: what fits the workspace at
  // once is far fewer tasks than the chip holds wavefronts).  The buffers are sized for the most a task can get; how many
  // it does get is decided per chunk from the chunk's cells (cellsPerWave below).
  int nwMax = 1;
  {
    int64_t nWide = 0;
    int maxStrips = 1;
    for (int i = 0; i < n; i++)
      if (geo[i].status == 0 && geo[i].wide) {
        nWide++;
        maxStrips = std::max(maxStrips, geo[i].nStrips);
      }
    if (nWide > 0) nwMax = std::min(16, maxStrips);
    if (const char *e = getenv("BLASR_HIP_KBAND_WAVES")) nwMax = std::min(16, std::max(1, atoi(e)));
  }
  const int nw = nwMax;
  std::vector<int64_t> arrowOff((size_t)n), arenaOff((size_t)n), scoreOff((size_t)n);
  struct Chunk {
    int first, count;
    int64_t arrowBytes, arenaWords, scoreWords;
    bool wide;
  };
  std::vector<Chunk> chunks;
  int64_t maxBytes = 0;
  // The strip kernel's chunks run as TWO lanes, each on a half of the workspace and on a stream of its own (even chunks /
  // odd chunks): a chunk ends in a tail -- its largest task has all its wavefronts in one workgroup, i.e. on one CU, and
  // runs long after the others are done (578 ms for a chunk whose work is 245 ms at configs[4]) -- and the other lane's
  // chunk fills the device under it.  One chunk (everything fits) stays one chunk.
  bool twoLanes = false;
  if (!lp) {
    int64_t need = 0;
    for (int i = 0; i < n; i++)
      if (geo[i].status == 0)
        need += ((geo[i].arrowBytes + 255) & ~(int64_t)255) + 4 * ((geo[i].arenaWords + 63) & ~(int64_t)63) +
                (geo[i].wide ? 4 * (((nw + 1) * ((int64_t)geo[i].qLen + 2) + 63) & ~(int64_t)63) : 0);
    twoLanes = need > limit && !getenv("BLASR_HIP_KBAND_ONE_LANE");
    // a lane gets half the limit: a task that needs more than that (it would be accepted as a chunk of its own and make
    // the two lanes together exceed the limit the host set) keeps everything on one lane
    for (int i = 0; i < n && twoLanes; i++)
      if (geo[i].status == 0 &&
          ((geo[i].arrowBytes + 255) & ~(int64_t)255) + 4 * ((geo[i].arenaWords + 63) & ~(int64_t)63) +
                  (geo[i].wide ? 4 * (((nw + 1) * ((int64_t)geo[i].qLen + 2) + 63) & ~(int64_t)63) : 0) + 1024 > limit / 2)
        twoLanes = false;
  }
  const int64_t chunkLimit = twoLanes ? limit / 2 : limit;
  for (int i = 0; i < n;) {
    Chunk c{i, 0, 0, 0, 0, false};
    while (i < n) {
      const int64_t ab = (geo[i].arrowBytes + 255) & ~(int64_t)255, aw = (geo[i].arenaWords + 63) & ~(int64_t)63;
      // KBandAlign also keeps the last row and the column of t = tLen for the choice of the end cell; the strip
      // kernel nw + 1 columns of scores between strips (the same offsets array serves both)
      const int64_t sw = lp ? (((int64_t)geo[i].nCols + 2 * ((int64_t)geo[i].qLen + 1) + 63) & ~(int64_t)63)
                            : (geo[i].wide ? (((nw + 1) * ((int64_t)geo[i].qLen + 2) + 63) & ~(int64_t)63) : 0);
      if (geo[i].wide) c.wide = true;
      if (c.count > 0 && c.arrowBytes + ab + 4 * (c.arenaWords + aw + c.scoreWords + sw) > chunkLimit) break;
      if (ab + 4 * (aw + sw) > limit)
        return blasr_hip_fail(ctx, BLASR_HIP_ERR_NOMEM, "kband: task %d needs %lld bytes of workspace", i,
                              (long long)(ab + 4 * (aw + sw)));
      arrowOff[i] = c.arrowBytes;
      arenaOff[i] = c.arenaWords;
      scoreOff[i] = c.scoreWords;
      c.arrowBytes += ab;
      c.arenaWords += aw;
      c.scoreWords += sw;
      c.count++;
      i++;
    }
    maxBytes = std::max(maxBytes, c.arrowBytes + 4 * (c.arenaWords + c.scoreWords) + 1024);
    chunks.push_back(c);
  }
  if (twoLanes) {  // the chunks whose largest task is largest first: the long tails then lie under later chunks, not at the end
    auto largest_of = [&](const Chunk &c) {
      int64_t m = 0;
      for (int i = c.first; i < c.first + c.count; i++)
        if (geo[i].status == 0) m = std::max(m, geo[i].realCells);
      return m;
    };
    std::vector<std::pair<int64_t, Chunk>> keyed;
    for (const Chunk &c : chunks) keyed.emplace_back(-largest_of(c), c);
    std::stable_sort(keyed.begin(), keyed.end(), [](const std::pair<int64_t, Chunk> &x, const std::pair<int64_t, Chunk> &y) { return x.first < y.first; });
    for (size_t i = 0; i < chunks.size(); i++) chunks[i] = keyed[i].second;
  }
  const int64_t laneBytes = (maxBytes + 255) & ~(int64_t)255;  // a lane's share of the workspace
  if (twoLanes) maxBytes = 2 * laneBytes;
  if (maxBytes > limit + 4096)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INTERNAL, "kband: the chunk plan needs %lld bytes, the workspace limit is %lld",
                          (long long)maxBytes, (long long)limit);
  if (ctx->workspaceBytes < (size_t)maxBytes) {
    if (ctx->workspace) (void)hipFree(ctx->workspace);
    ctx->workspace = nullptr;
    ctx->workspaceBytes = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->workspace, (size_t)maxBytes));
    ctx->workspaceBytes = (size_t)maxBytes;
  }
  HIP_TRY(ctx, hipMemcpyAsync(dArrowOff.p, arrowOff.data(), sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dArenaOff.p, arenaOff.data(), sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dScoreOff.p, scoreOff.data(), sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, st));
  RefineResult *dResults = reinterpret_cast<RefineResult *>(results);
  for (int kd = 0; kd < BLASR_HIP_KERNEL_KINDS; kd++) {
    ctx->kernelRec[kd] = blasr_hip_kernel_record();
    ctx->kernelRec[kd].kind = kd;
  }
  for (int i = 0; i < n && !lp; i++) {
    if (geo[i].status != 0) continue;
    blasr_hip_kernel_record &r = ctx->kernelRec[geo[i].wide ? BLASR_HIP_KERNEL_KBAND_STRIPS : BLASR_HIP_KERNEL_KBAND_ROWS];
    r.cells += geo[i].realCells;
    r.candidates += 1;
    r.algorithmic_bytes += (geo[i].wide ? geo[i].realCells / 2 : geo[i].realCells) + geo[i].qLen + geo[i].tLen;
    if (geo[i].wide) {  // the walk: one nibble per step read (as whole 16-byte groups: a window's worth is the traffic)
      blasr_hip_kernel_record &tb = ctx->kernelRec[BLASR_HIP_KERNEL_KBAND_TRACEBACK];
      tb.cells += geo[i].realCells;
      tb.candidates += 1;
      tb.algorithmic_bytes += (int64_t)geo[i].qLen + geo[i].tLen;
    }
  }
  // every way out of the loop below -- an error of a launch, of collect(), of a copy -- leaves BOTH streams idle: the
  // other lane's chunk may still be queued on the side stream, on a workspace the next call reuses (or frees)
  struct JoinLanes {
    blasr_hip_ctx *c;
    bool on;
    ~JoinLanes() {
      if (!on) return;
      (void)hipStreamSynchronize(c->sideStream);
      (void)hipStreamSynchronize(c->stream);
    }
  } joinLanes{ctx, twoLanes};
  hipEvent_t *ev = ctx->events;  // the context's own (one call at a time per context): six per lane
  int inFlight[2] = {-1, -1};    // the chunk each lane is working on
  // waits for a lane's chunk and adds its kernels' times to the records
  auto collect = [&](int ci, int laneIx) -> int {
    const Chunk &cc = chunks[(size_t)ci];
    hipEvent_t *lev = ev + 6 * laneIx;
    HIP_TRY(ctx, hipStreamSynchronize(laneIx ? ctx->sideStream : ctx->stream));
    float t = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&t, lev[0], lev[1]));
    if (ctx->kernelRec[BLASR_HIP_KERNEL_KBAND_ROWS].candidates > 0) {
      ctx->kernelRec[BLASR_HIP_KERNEL_KBAND_ROWS].ms += t;
      ctx->kernelRec[BLASR_HIP_KERNEL_KBAND_ROWS].launches += 1;
    }
    if (cc.wide) {
      HIP_TRY(ctx, hipEventElapsedTime(&t, lev[2], lev[3]));
      if (getenv("BLASR_HIP_KBAND_TRACE")) fprintf(stderr, "[kband]   chunk %d: strip kernel %.1f ms\n", ci, t);
      ctx->kernelRec[BLASR_HIP_KERNEL_KBAND_STRIPS].ms += t;
      ctx->kernelRec[BLASR_HIP_KERNEL_KBAND_STRIPS].launches += 1;
      ctx->kernelRec[BLASR_HIP_KERNEL_KBAND_STRIPS].ext0 = nw;  // most wavefronts a task could get
      HIP_TRY(ctx, hipEventElapsedTime(&t, lev[4], lev[5]));
      ctx->kernelRec[BLASR_HIP_KERNEL_KBAND_TRACEBACK].ms += t;
      ctx->kernelRec[BLASR_HIP_KERNEL_KBAND_TRACEBACK].launches += 1;
    }
    return BLASR_HIP_OK;
  };
  for (const Chunk &c : chunks) {
    if (lp) {
      KbandLinearArgs a;
      a.first = c.first;
      a.count = c.count;
      a.ins = lp->ins;
      a.del = lp->del;
      a.fnIns = lp->fn_ins;
      a.fnDel = lp->fn_del;
      a.alignType = lp->align_type;
      a.matrix = static_cast<int32_t *>(dMatrix.p);
      a.geo = static_cast<KbandGeo *>(dGeo.p);
      a.seq = b->seq;
      a.qOff = b->q_off;
      a.tOff = b->t_off;
      a.arrows = reinterpret_cast<uint8_t *>(ctx->workspace);
      a.arrowOff = static_cast<int64_t *>(dArrowOff.p);
      a.arena = reinterpret_cast<uint32_t *>(ctx->workspace + c.arrowBytes);
      a.arenaOff = static_cast<int64_t *>(dArenaOff.p);
      a.scores = reinterpret_cast<int32_t *>(ctx->workspace + c.arrowBytes + 4 * c.arenaWords);
      a.scoreOff = static_cast<int64_t *>(dScoreOff.p);
      a.opsStart = static_cast<int32_t *>(dOpsStart.p);
      a.results = dResults;
      hipLaunchKernelGGL(kband_linear_fill_kernel, dim3((unsigned)((c.count + 3) / 4)), dim3(256), 0, st, a);
      hipLaunchKernelGGL(kband_linear_traceback_kernel, dim3((unsigned)((c.count + 63) / 64)), dim3(64), 0, st, a);
      launch_scan_ops(dResults, c.first, c.count, ops_capacity, static_cast<int64_t *>(dTotal.p), st);
      launch_compact_ops(dResults, c.first, c.count, a.arena, a.arenaOff, a.opsStart, ops, st);
      HIP_TRY(ctx, hipStreamSynchronize(st));  // the next chunk reuses the workspace
      continue;
    }
    // ---- AffineKBandAlign: chunk ci on lane ci & 1 (its half of the workspace, its stream, its six events)
    const int ci = (int)(&c - chunks.data());
    const int laneIx = twoLanes ? (ci & 1) : 0;
    hipStream_t ls = laneIx ? ctx->sideStream : st;
    hipEvent_t *lev = ev + 6 * laneIx;
    char *wsBase = ctx->workspace + (twoLanes ? laneIx * laneBytes : 0);
    if (inFlight[laneIx] >= 0) {  // the lane's chunk before this one: its times, before its events are recorded again
      const int rcPrev = collect(inFlight[laneIx], laneIx);
      if (rcPrev) return rcPrev;
    }
    KbandArgs a;
    a.first = c.first;
    a.count = c.count;
    a.hpInsOpen = p->hp_ins_open;
    a.hpInsExtend = p->hp_ins_extend;
    a.insOpen = p->ins_open;
    a.insExtend = p->ins_extend;
    a.del = p->del;
    a.matrix = static_cast<int32_t *>(dMatrix.p);
    a.geo = static_cast<KbandGeo *>(dGeo.p);
    a.seq = b->seq;
    a.qOff = b->q_off;
    a.tOff = b->t_off;
    a.arrows = reinterpret_cast<uint8_t *>(wsBase);
    a.arrowOff = static_cast<int64_t *>(dArrowOff.p);
    a.arena = reinterpret_cast<uint32_t *>(wsBase + c.arrowBytes);
    a.arenaOff = static_cast<int64_t *>(dArenaOff.p);
    a.opsStart = static_cast<int32_t *>(dOpsStart.p);
    a.results = dResults;
    a.bound = reinterpret_cast<int32_t *>(wsBase + c.arrowBytes + 4 * c.arenaWords);
    a.boundOff = static_cast<int64_t *>(dScoreOff.p);
    a.cellsPerWave = 1;
    a.waveRound = 0;
    a.maxWaves = nw;
    if (ci == 0 && twoLanes) {  // the side stream starts behind what the main one has queued so far (plan, uploads)
      HIP_TRY(ctx, hipEventRecord(ctx->auxEvent[0], st));
      HIP_TRY(ctx, hipStreamWaitEvent(ctx->sideStream, ctx->auxEvent[0], 0));
    }
    HIP_TRY(ctx, hipEventRecord(lev[0], ls));
    hipLaunchKernelGGL(kband_fill_kernel, dim3((unsigned)((c.count + 3) / 4)), dim3(256), 0, ls, a);
    HIP_TRY(ctx, hipEventRecord(lev[1], ls));
    if (c.wide) {
      // about 4096 wavefronts in all (89 registers: 5 per SIMD, 5120 on the chip), each with the same share of cells
      int64_t chunkCells = 0, largest = 0;
      for (int i = c.first; i < c.first + c.count; i++)
        if (geo[i].status == 0 && geo[i].wide) {
          chunkCells += geo[i].realCells;
          largest = std::max(largest, geo[i].realCells);
        }
      int targetWaves = 4096;  // four per SIMD: 1.26 Tcells/s at 3072-4096, 0.98 at 5120 and beyond (1024 reads of 30 kb, k = 2.7 k);
                               // with two lanes 4096 per lane again (refine stage of configs[4]: 929 ms against 1076 at 3072)
      if (const char *e = getenv("BLASR_HIP_KBAND_TARGET_WAVES")) targetWaves = std::max(1, atoi(e));
      a.cellsPerWave = std::max<int64_t>(chunkCells / targetWaves, 1 << 20);
      if (getenv("BLASR_HIP_KBAND_WAVES")) a.cellsPerWave = 1;  // experiments: every task gets nwMax wavefronts
      a.waveRound = a.cellsPerWave / 2;
      const int nwLaunch = (int)std::min<int64_t>(nw, std::max<int64_t>(1, (largest + a.waveRound) / a.cellsPerWave));
      if (getenv("BLASR_HIP_KBAND_TRACE"))
        fprintf(stderr, "[kband] chunk %d (lane %d) of %d tasks, %.1f Gcells (largest %.2f), %lld cells per wavefront, workgroups of "
                        "%d wavefronts\n", ci, laneIx, c.count, chunkCells / 1e9, largest / 1e9, (long long)a.cellsPerWave, nwLaunch);
      HIP_TRY(ctx, hipMemsetAsync(a.bound, 0, (size_t)(4 * c.scoreWords), ls));
      HIP_TRY(ctx, hipEventRecord(lev[2], ls));
      hipLaunchKernelGGL(kband_wide_kernel, dim3((unsigned)c.count), dim3((unsigned)(64 * nwLaunch)), 0, ls, a);
      HIP_TRY(ctx, hipEventRecord(lev[3], ls));
    }
    hipLaunchKernelGGL(kband_traceback_kernel, dim3((unsigned)((c.count + 63) / 64)), dim3(64), 0, ls, a);
    if (c.wide) {
      HIP_TRY(ctx, hipEventRecord(lev[4], ls));
      hipLaunchKernelGGL(kband_wide_traceback_kernel, dim3((unsigned)((c.count + 3) / 4)), dim3(256), 0, ls, a);
      HIP_TRY(ctx, hipEventRecord(lev[5], ls));
    }
    // the op lists go to the caller's arena in chunk order: this chunk's scan starts from the total the one before left
    if (ci > 0 && twoLanes) HIP_TRY(ctx, hipStreamWaitEvent(ls, ctx->auxEvent[1 + ((ci - 1) & 1)], 0));
    launch_scan_ops(dResults, c.first, c.count, ops_capacity, static_cast<int64_t *>(dTotal.p), ls);
    launch_compact_ops(dResults, c.first, c.count, a.arena, a.arenaOff, a.opsStart, ops, ls);
    if (twoLanes) HIP_TRY(ctx, hipEventRecord(ctx->auxEvent[1 + (ci & 1)], ls));
    inFlight[laneIx] = ci;
  }
  for (int laneIx = 0; laneIx < 2; laneIx++)
    if (inFlight[laneIx] >= 0) {
      const int rcLast = collect(inFlight[laneIx], laneIx);
      if (rcLast) return rcLast;
    }
  if (twoLanes) {  // the main stream goes on behind the side stream's last chunk
    HIP_TRY(ctx, hipEventRecord(ctx->auxEvent[0], ctx->sideStream));
    HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->auxEvent[0], 0));
  }
  int64_t total = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&total, dTotal.p, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  HIP_TRY(ctx, hipGetLastError());
  *ops_used = total;
  if (total > ops_capacity)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_CAPACITY, "kband: %lld ops produced, arena holds %lld", (long long)total,
                          (long long)ops_capacity);
  return BLASR_HIP_OK;
}

int kband_run_host(blasr_hip_ctx *ctx, const blasr_hip_kband_batch *b, const blasr_hip_kband_params *p,
                   const blasr_hip_kband_linear_params *lp, blasr_hip_refine_result *results, uint32_t *ops,
                   int64_t ops_capacity, int64_t *ops_used) {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (!b || (!p && !lp) || !ops_used || b->n < 0 || ops_capacity < 0)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "kband: bad arguments");
  *ops_used = 0;
  const int n = b->n;
  if (n == 0) return BLASR_HIP_OK;
  int64_t seqBytes = 0;
  for (int i = 0; i < n; i++) {
    if (b->q_len[i] < 0 || b->t_len[i] < 0) return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "kband: negative length");
    seqBytes = std::max<int64_t>(seqBytes, std::max(b->q_off[i] + b->q_len[i], b->t_off[i] + b->t_len[i]));
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  struct Dev {
    void *p = nullptr;
    ~Dev() {
      if (p) (void)hipFree(p);
    }
  };
  Dev dSeq, dQOff, dTOff, dQLen, dTLen, dK, dRes, dOps;
  HIP_TRY(ctx, hipMalloc(&dSeq.p, (size_t)std::max<int64_t>(seqBytes, 1)));
  HIP_TRY(ctx, hipMalloc(&dQOff.p, sizeof(int64_t) * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dTOff.p, sizeof(int64_t) * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dQLen.p, sizeof(int32_t) * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dTLen.p, sizeof(int32_t) * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dK.p, sizeof(int32_t) * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dRes.p, sizeof(blasr_hip_refine_result) * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&dOps.p, sizeof(uint32_t) * (size_t)std::max<int64_t>(ops_capacity, 1)));
  HIP_TRY(ctx, hipMemcpyAsync(dSeq.p, b->seq, (size_t)seqBytes, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dQOff.p, b->q_off, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dTOff.p, b->t_off, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dQLen.p, b->q_len, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dTLen.p, b->t_len, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dK.p, b->k, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
  blasr_hip_kband_batch db;
  db.n = n;
  db.seq = static_cast<uint8_t *>(dSeq.p);
  db.q_off = static_cast<int64_t *>(dQOff.p);
  db.q_len = static_cast<int32_t *>(dQLen.p);
  db.t_off = static_cast<int64_t *>(dTOff.p);
  db.t_len = static_cast<int32_t *>(dTLen.p);
  db.k = static_cast<int32_t *>(dK.p);
  const int rc = kband_run(ctx, &db, p, lp, static_cast<blasr_hip_refine_result *>(dRes.p),
                           static_cast<uint32_t *>(dOps.p), ops_capacity, ops_used);
  if (rc != BLASR_HIP_OK && rc != BLASR_HIP_ERR_CAPACITY) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(results, dRes.p, sizeof(blasr_hip_refine_result) * (size_t)n, hipMemcpyDeviceToHost, st));
  const int64_t copyOps = std::min(*ops_used, ops_capacity);
  if (copyOps > 0) HIP_TRY(ctx, hipMemcpyAsync(ops, dOps.p, sizeof(uint32_t) * (size_t)copyOps, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return rc;
}
}  // namespace

extern "C" {

int blasr_hip_kband_device(blasr_hip_ctx *ctx, const blasr_hip_kband_batch *b, const blasr_hip_kband_params *p,
                           blasr_hip_refine_result *results, uint32_t *ops, int64_t ops_capacity, int64_t *ops_used) try {
  if (ctx && !p) return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "kband: bad arguments");
  return kband_run(ctx, b, p, nullptr, results, ops, ops_capacity, ops_used);
} BLASR_HIP_CATCH_CTX(ctx)
int blasr_hip_kband_host(blasr_hip_ctx *ctx, const blasr_hip_kband_batch *b, const blasr_hip_kband_params *p,
                         blasr_hip_refine_result *results, uint32_t *ops, int64_t ops_capacity, int64_t *ops_used) try {
  if (ctx && !p) return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "kband: bad arguments");
  return kband_run_host(ctx, b, p, nullptr, results, ops, ops_capacity, ops_used);
} BLASR_HIP_CATCH_CTX(ctx)
int blasr_hip_kband_stats_device(blasr_hip_ctx *ctx, const blasr_hip_kband_batch *b, const blasr_hip_score_params *score,
                                 int32_t use_affine, blasr_hip_refine_result *results, const uint32_t *ops) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (!b || !score || b->n < 0 || (b->n > 0 && (!b->seq || !b->q_off || !b->t_off || !results || !ops)))
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "kband_stats: bad arguments");
  if (b->n == 0) return BLASR_HIP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  struct Dev {
    void *p = nullptr;
    ~Dev() {
      if (p) (void)hipFree(p);
    }
  } dMatrix;
  HIP_TRY(ctx, hipMalloc(&dMatrix.p, sizeof(int32_t) * 32));
  HIP_TRY(ctx, hipMemcpyAsync(dMatrix.p, score->matrix, sizeof(int32_t) * 25, hipMemcpyHostToDevice, ctx->stream));
  KbandStatsArgs a;
  a.n = b->n;
  a.seq = b->seq;
  a.qOff = b->q_off;
  a.tOff = b->t_off;
  a.results = reinterpret_cast<RefineResult *>(results);
  a.ops = ops;
  a.matrix = static_cast<int32_t *>(dMatrix.p);
  a.ins = score->ins;
  a.del = score->del;
  a.open = score->affine_open;
  a.ext = score->affine_extend;
  a.useAffine = use_affine != 0;
  hipLaunchKernelGGL(kband_stats_kernel, dim3((unsigned)((b->n + 3) / 4)), dim3(256), 0, ctx->stream, a);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipGetLastError());
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CTX(ctx)
int blasr_hip_kband_linear_device(blasr_hip_ctx *ctx, const blasr_hip_kband_batch *b,
                                  const blasr_hip_kband_linear_params *p, blasr_hip_refine_result *results,
                                  uint32_t *ops, int64_t ops_capacity, int64_t *ops_used) try {
  if (ctx && !p) return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "kband: bad arguments");
  return kband_run(ctx, b, nullptr, p, results, ops, ops_capacity, ops_used);
} BLASR_HIP_CATCH_CTX(ctx)
int blasr_hip_kband_linear_host(blasr_hip_ctx *ctx, const blasr_hip_kband_batch *b,
                                const blasr_hip_kband_linear_params *p, blasr_hip_refine_result *results, uint32_t *ops,
                                int64_t ops_capacity, int64_t *ops_used) try {
  if (ctx && !p) return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "kband: bad arguments");
  return kband_run_host(ctx, b, nullptr, p, results, ops, ops_capacity, ops_used);
} BLASR_HIP_CATCH_CTX(ctx)

}  // extern "C"
