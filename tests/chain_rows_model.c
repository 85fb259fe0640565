/* chain_rows_model.c -- CPU model of sdp_chain_rows_kernel's formulation of SDPLongestCommonSubsequence
 * (sdp/SparseDynamicProgramming.h:70-322), TEST INFRASTRUCTURE (tests/test_chain_rows_model.py builds and runs it).
 *
 * The reference scores one fragment at a time and retires one fragment at a time.  The kernel works on whole ROWS:
 *   - scoring: every fragment of sweep row x sees the same column set (the reference inserts nothing into colSet while it
 *     scores a row, :112-236), so their column predecessors are independent look-ups; the only dependency inside a row is
 *     `above` when it lies in the same row (StoreAbove :55-68: the fragment just before it in the row);
 *   - retiring (:248-309): the fragments of trail row x - k - 1 go into the column set in y order.  Fragment j reads
 *     columns >= y_j only (membership of y_j, the successors it deletes) and fragment i < j writes [y_i, reach_i] (its own
 *     column and the successors it deleted): when reach_i < y_j for every i < j the two commute.  All fragments of the row
 *     are evaluated against the SAME set state (lanes of a wavefront), the longest prefix whose members commute is applied,
 *     the rest is evaluated again.
 * This file states that formulation with "lanes" as array slots, every read of a batch done before any of its writes,
 * next to a plain restatement of the reference loop, and compares every fragment's cost / chainPrev / chainLength and the
 * reported chain on random fragment sets (scattered chance matches, near-diagonal alignments, low-complexity sequence);
 * the plain restatement is itself compared with the oracle's (oracle/blasr_oracle.c: bo_sdp_chain).
 * It also counts what the kernel's cost depends on: fragments per row, evaluation rounds per retired row, walk lengths. */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int bo_sdp_chain(int queryLength, const int *xyl, int nFrag, int fragmentLength, int ins, int del, int match, int alignType,
                 int *chain, int cap);

typedef struct {
  int x, y, len;
  int predCl, above;
  int cost, prev, clen;
} frag;

static int indel_penalty(int x1, int y1, int x2, int y2, int ins, int del) {
  const int drift = (x1 - y1) - (x2 - y2);
  if (drift > 0) return drift * ins;
  if (drift < 0) return -drift * del;
  return 0;
}
static int isqrt_floor(int d) { return (int)sqrt((double)d); }

/* ---- the static predecessors (geometry only), as sdp_above_predcl_kernel computes them ---- */
static int cmp_yx(const void *a, const void *b) {
  const frag *fa = *(frag *const *)a, *fb = *(frag *const *)b;
  if (fa->y != fb->y) return fa->y < fb->y ? -1 : 1;
  if (fa->x != fb->x) return fa->x < fb->x ? -1 : 1;
  return 0;
}
static void static_preds(frag *F, int n, int k) {
  frag **byY = (frag **)malloc(sizeof(frag *) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++) {
    byY[i] = &F[i];
    F[i].above = -1;
  }
  qsort(byY, (size_t)n, sizeof(frag *), cmp_yx);
  for (int j = 1; j < n; j++) {
    const frag *a = byY[j - 1];
    frag *b = byY[j];
    if (a->x <= b->x && a->x + a->len > b->x && a->y < b->y) b->above = (int)(a - F);
  }
  free(byY);
  for (int i = 0; i < n; i++) { /* sweepSet.Predecessor: largest (diag, x) with diag <= f.diag among rows [x-k-1, x-1] */
    const int fd = F[i].y - F[i].x, xLow = F[i].x > k + 1 ? F[i].x - (k + 1) : 0;
    int best = -1, bd = 0, bx = 0;
    for (int j = i - 1; j >= 0 && F[j].x >= xLow; j--) {
      if (F[j].x >= F[i].x) continue;
      const int d = F[j].y - F[j].x;
      if (d <= fd && (best < 0 || d > bd || (d == bd && F[j].x > bx))) best = j, bd = d, bx = F[j].x;
    }
    F[i].predCl = best;
  }
}

/* ---- column set: a bitmap over the target's columns + the entry of a member column ---- */
typedef struct {
  int nCols;
  unsigned char *bit;
  int *eFrag, *eCost, *eLen, *eXY;
} colset;
static void cs_init(colset *s, int nCols) {
  s->nCols = nCols;
  s->bit = (unsigned char *)calloc((size_t)nCols + 1, 1);
  s->eFrag = (int *)calloc((size_t)nCols + 1, sizeof(int));
  s->eCost = (int *)calloc((size_t)nCols + 1, sizeof(int));
  s->eLen = (int *)calloc((size_t)nCols + 1, sizeof(int));
  s->eXY = (int *)calloc((size_t)nCols + 1, sizeof(int));
}
static void cs_free(colset *s) {
  free(s->bit), free(s->eFrag), free(s->eCost), free(s->eLen), free(s->eXY);
}
static int cs_pred_le(const colset *s, int y) {
  for (int c = y; c >= 0; c--)
    if (s->bit[c]) return c;
  return -1;
}
static int cs_succ_gt(const colset *s, int y) {
  for (int c = y + 1; c < s->nCols; c++)
    if (s->bit[c]) return c;
  return -1;
}

/* the cost of fragment i from its three candidates (:112-236); cp candidates given */
static void score_one(frag *F, int i, int k, int ins, int del, int match, int type, int pc, const colset *s, int aboveCost,
                      int aboveLen) {
  frag *f = &F[i];
  int cp = INT_MAX, cl = INT_MAX, ca = INT_MAX, cpFrag = -1, cpLen = 0, found = 0;
  if (pc >= 0) {
    cpFrag = s->eFrag[pc];
    cpLen = s->eLen[pc];
    cp = s->eCost[pc] + isqrt_floor(abs((f->x + f->y) - s->eXY[pc])) - f->len;
    found = 1;
  }
  if (f->predCl >= 0) {
    const frag *p = &F[f->predCl];
    int ov = (int)((uint32_t)k - (uint32_t)(f->y - p->y)) * match;
    if (ov > 0) ov = 0;
    cl = p->cost + ov + indel_penalty(f->x, f->y, p->x, p->y, ins, del);
    found = 1;
  }
  if (f->above >= 0) {
    const frag *p = &F[f->above];
    ca = (int)((uint32_t)aboveCost + ((uint32_t)k - (uint32_t)(f->y - p->y)) * (uint32_t)match +
               (uint32_t)indel_penalty(f->x, f->y, p->x, p->y, ins, del));
    found = 1;
  }
  int mc = cl < ca ? cl : ca;
  if (cp < mc) mc = cp;
  if (found && (type == 1 || (type == 0 && mc < 0))) {
    f->cost = (int)((uint32_t)mc - (uint32_t)k);
    if (mc == cp) f->prev = cpFrag, f->clen = cpLen + 1;
    else if (mc == cl) f->prev = f->predCl, f->clen = F[f->predCl].clen + 1;
    else f->prev = f->above, f->clen = aboveLen + 1;
  } else {
    f->prev = -1;
    f->clen = 1;
    if (type == 1) f->cost = (int)((uint32_t)(f->x + f->y) * (uint32_t)del + (uint32_t)k * (uint32_t)match - (uint32_t)k);
    else f->cost = (int)((uint32_t)k * (uint32_t)match - (uint32_t)k);
  }
}

/* ---- the reference loop, one fragment at a time ---- */
static int sweep_serial(frag *F, int n, int k, int nCols, int ins, int del, int match, int type) {
  colset s;
  cs_init(&s, nCols);
  int fTrail = 0, minCost = INT_MAX, minIdx = -1, maxLenIdx = -1;
  unsigned maxLen = 0;
  for (int i = 0; i < n; i++) {
    while (fTrail < i && (int64_t)F[fTrail].x + k + 2 <= (int64_t)F[i].x) {
      const frag *t = &F[fTrail];
      const int member = s.bit[t->y];
      int store = 1;
      if (member) store = s.eCost[t->y] < t->cost; /* sic, :260-267 */
      if (store) {
        s.bit[t->y] = 1;
        s.eFrag[t->y] = fTrail, s.eCost[t->y] = t->cost, s.eLen[t->y] = t->clen, s.eXY[t->y] = t->x + t->y;
        for (int sc = cs_succ_gt(&s, t->y); sc >= 0 && s.eCost[sc] > t->cost; sc = cs_succ_gt(&s, t->y)) s.bit[sc] = 0;
      }
      fTrail++;
    }
    const int pc = cs_pred_le(&s, F[i].y);
    score_one(F, i, k, ins, del, match, type, pc, &s, F[i].above >= 0 ? F[F[i].above].cost : 0,
              F[i].above >= 0 ? F[F[i].above].clen : 0);
    if (minCost > F[i].cost) minCost = F[i].cost, minIdx = i;
    if ((unsigned)F[i].clen > maxLen) maxLen = (unsigned)F[i].clen, maxLenIdx = i;
  }
  cs_free(&s);
  return type == 0 ? minIdx : maxLenIdx;
}

/* ---- the row formulation ---- */
#define LANES 64
static long g_rows, g_rowFrags, g_retRows, g_retRounds, g_walkIters, g_walkRounds, g_retFrags, g_maxRowLanes, g_inRowAbove,
    g_aboveRounds;

/* one batch of retiring fragments [first, first + m) of ONE trail row, m <= LANES, increasing y */
static void retire_batch(frag *F, int first, int m, colset *s) {
  int pending0 = 0; /* first lane still pending */
  g_retRows++;
  g_retFrags += m;
  while (pending0 < m) {
    int store[LANES], reach[LANES];
    g_retRounds++;
    long longest = 0;
    /* phase 1: every pending lane against the set as it stands (reads only) */
    for (int l = pending0; l < m; l++) {
      const frag *t = &F[first + l];
      const int member = s->bit[t->y];
      store[l] = member ? (s->eCost[t->y] < t->cost) : 1;
      reach[l] = -1;
      if (store[l]) {
        reach[l] = t->y;
        long it = 0;
        for (int sc = cs_succ_gt(s, t->y); sc >= 0 && s->eCost[sc] > t->cost; sc = cs_succ_gt(s, sc)) reach[l] = sc, it++;
        g_walkIters += it;
        if (it + 1 > longest) longest = it + 1;
      }
    }
    g_walkRounds += longest;
    /* phase 2: the longest prefix of pending lanes whose members commute */
    int apply = m, pm = -1;
    for (int l = pending0; l < m; l++) {
      if (pm >= F[first + l].y) {
        apply = l;
        break;
      }
      if (reach[l] > pm) pm = reach[l];
    }
    /* phase 3: their writes */
    for (int l = pending0; l < apply; l++) {
      if (!store[l]) continue;
      const frag *t = &F[first + l];
      s->bit[t->y] = 1;
      s->eFrag[t->y] = first + l, s->eCost[t->y] = t->cost, s->eLen[t->y] = t->clen, s->eXY[t->y] = t->x + t->y;
      for (int c = t->y + 1; c <= reach[l]; c++) s->bit[c] = 0;
    }
    pending0 = apply;
  }
}

static int sweep_rows(frag *F, int n, int k, int nCols, int ins, int del, int match, int type) {
  colset s;
  cs_init(&s, nCols);
  int fTrail = 0;
  int laneMin[LANES], laneMinIdx[LANES], laneMaxIdx[LANES];
  unsigned laneMax[LANES];
  for (int l = 0; l < LANES; l++) laneMin[l] = INT_MAX, laneMinIdx[l] = -1, laneMax[l] = 0, laneMaxIdx[l] = -1;
  for (int w0 = 0; w0 < n; w0 += LANES) {
    const int nHere = n - w0 < LANES ? n - w0 : LANES;
    int j = 0;
    while (j < nHere) {
      const int fx = F[w0 + j].x;
      int jEnd = j;
      while (jEnd < nHere && F[w0 + jEnd].x == fx) jEnd++;
      /* retire rows x <= fx - k - 2, a row (or a piece of LANES of it) at a time */
      while (fTrail < w0 + j && (int64_t)F[fTrail].x + k + 2 <= (int64_t)fx) {
        int m = 0;
        while (m < LANES && fTrail + m < w0 + j && F[fTrail + m].x == F[fTrail].x) m++;
        retire_batch(F, fTrail, m, &s);
        fTrail += m;
      }
      /* score the row's lanes: column predecessors first (all against the same set) */
      int pc[LANES];
      for (int l = j; l < jEnd; l++) pc[l] = cs_pred_le(&s, F[w0 + l].y);
      /* lanes whose `above` lies in this very piece of the row wait for it */
      int done[LANES], left = 0;
      for (int l = j; l < jEnd; l++) {
        done[l] = !(F[w0 + l].above >= w0 + j);
        left += !done[l];
        if (!done[l]) g_inRowAbove++;
      }
      for (int l = j; l < jEnd; l++)
        if (done[l])
          score_one(F, w0 + l, k, ins, del, match, type, pc[l], &s, F[w0 + l].above >= 0 ? F[F[w0 + l].above].cost : 0,
                    F[w0 + l].above >= 0 ? F[F[w0 + l].above].clen : 0);
      while (left > 0) {
        int now[LANES];
        g_aboveRounds++;
        for (int l = j; l < jEnd; l++) now[l] = !done[l] && done[F[w0 + l].above - w0];
        for (int l = j; l < jEnd; l++)
          if (now[l]) {
            score_one(F, w0 + l, k, ins, del, match, type, pc[l], &s, F[F[w0 + l].above].cost, F[F[w0 + l].above].clen);
            done[l] = 1;
            left--;
          }
      }
      for (int l = j; l < jEnd; l++) { /* per-lane running best; strict comparisons keep the earliest index */
        const frag *f = &F[w0 + l];
        if (laneMin[l] > f->cost) laneMin[l] = f->cost, laneMinIdx[l] = w0 + l;
        if ((unsigned)f->clen > laneMax[l]) laneMax[l] = (unsigned)f->clen, laneMaxIdx[l] = w0 + l;
      }
      g_rows++;
      g_rowFrags += jEnd - j;
      if (jEnd - j > g_maxRowLanes) g_maxRowLanes = jEnd - j;
      j = jEnd;
    }
  }
  /* the reference's running minimum / maximum with strict comparisons = the best value at its lowest index */
  int minCost = INT_MAX, minIdx = -1, maxLenIdx = -1;
  unsigned maxLen = 0;
  for (int l = 0; l < LANES; l++) {
    if (laneMinIdx[l] >= 0 && (laneMin[l] < minCost || (laneMin[l] == minCost && laneMinIdx[l] < minIdx)))
      minCost = laneMin[l], minIdx = laneMinIdx[l];
    if (laneMaxIdx[l] >= 0 && (laneMax[l] > maxLen || (laneMax[l] == maxLen && laneMaxIdx[l] < maxLenIdx)))
      maxLen = laneMax[l], maxLenIdx = laneMaxIdx[l];
  }
  cs_free(&s);
  return type == 0 ? minIdx : maxLenIdx;
}


/* ---- the batch formulation (sdp_chain_rows_kernel) ----
 * The sweep is a sequence of EVENTS: for every row fx with fragments, first the retirements R_j of the not yet retired
 * fragments with x_j <= fx - k - 2 (index order), then the queries Q_i of the row's fragments.  R_j needs only j's cost, and
 * j lies at least k + 2 rows back: the events of up to k + 2 consecutive rows involve nothing that is computed inside those
 * rows.  A batch = a prefix of the remaining sequence: up to LANES query lanes (fragments fScore ..) and up to LANES retire
 * lanes (fragments fTrail ..).  All pending events are evaluated against the set as it stands; an event is DIRTY when an
 * earlier pending retirement writes ([y_i, reach_i]: its own column and the successors it deletes) into what the event read
 * (a query: [predecessor, y]; a retirement: [y, first surviving successor]); everything before the first dirty event is final
 * (its writes are applied, disjoint by construction), the rest is evaluated again.  Then the rows of the batch are scored in
 * order (the column predecessors are at hand), which is register work only. */
static long g_batches, g_batchEvents, g_batchRounds, g_batchRows, g_writerLoops, g_batchWalkRounds;

typedef struct {
  int frag, t;                 /* fragment, ordinal of the batch row that triggers it */
  int store, reach, stop;      /* evaluation: writes [y, reach], read [y, stop] */
  int done;
} rlane;
typedef struct {
  int frag, t;
  int p;                       /* evaluation: predecessor column or -1; read [p, y] (p < 0: [0, y]) */
  int cpFrag, cpLen, cpCost, cpXY;
  int done;
} qlane;

static int sweep_batches(frag *F, int n, int k, int nCols, int ins, int del, int match, int type) {
  colset s;
  cs_init(&s, nCols);
  int fTrail = 0, fScore = 0;
  int minCost = INT_MAX, minIdx = -1, maxLenIdx = -1;
  unsigned maxLen = 0;
  while (fScore < n) {
    /* ---- the batch: rows from fScore on, as long as their events fit the lanes and involve nothing of the batch itself */
    qlane Q[LANES];
    rlane R[LANES];
    int nq = 0, nr = 0, rows = 0;
    const int X = F[fScore].x;
    int at = fScore, trail = fTrail;
    while (at < n) {
      const int fx = F[at].x;
      if (fx > X + k + 1 && rows > 0) break;
      int rowEnd = at;
      while (rowEnd < n && F[rowEnd].x == fx) rowEnd++;
      int newR = 0;
      while (trail + newR < fScore && (int64_t)F[trail + newR].x + k + 2 <= (int64_t)fx) newR++;
      /* (trail + newR < fScore: a retirement of the batch must be of a fragment scored BEFORE the batch; with fx <= X + k + 1
       *  every fragment the row retires is) */
      if (nr + newR > LANES || nq + (rowEnd - at) > LANES) {
        if (rows > 0) break;
        /* the first row alone does not fit: its retirements first (as many as there are lanes), then a piece of the row */
        if (newR > LANES) {
          for (int m = 0; m < LANES; m++) R[nr].frag = trail + m, R[nr].t = 0, R[nr].done = 0, nr++;
          trail += LANES;
          break; /* no queries in this batch */
        }
        for (int m = 0; m < newR; m++) R[nr].frag = trail + m, R[nr].t = 0, R[nr].done = 0, nr++;
        trail += newR;
        for (int l = 0; l < LANES && at + l < rowEnd; l++) Q[nq].frag = at + l, Q[nq].t = 0, Q[nq].done = 0, nq++;
        at += nq;
        rows = 1;
        break;
      }
      for (int m = 0; m < newR; m++) R[nr].frag = trail + m, R[nr].t = rows, R[nr].done = 0, nr++;
      trail += newR;
      for (int i = at; i < rowEnd; i++) Q[nq].frag = i, Q[nq].t = rows, Q[nq].done = 0, nq++;
      at = rowEnd;
      rows++;
    }
    g_batches++;
    g_batchEvents += nq + nr;
    g_batchRows += rows;
    /* ---- set phase: rounds of evaluate / find the first dirty event / apply */
    int left = nq + nr;
    while (left > 0) {
      g_batchRounds++;
      long longest = 0;
      /* evaluation of every pending lane against the set as it stands (reads only) */
      for (int m = 0; m < nr; m++) {
        if (R[m].done) continue;
        const frag *t = &F[R[m].frag];
        const int member = s.bit[t->y];
        R[m].store = member ? (s.eCost[t->y] < t->cost) : 1;
        R[m].reach = t->y;
        R[m].stop = t->y;
        if (R[m].store) {
          int sc = cs_succ_gt(&s, t->y);
          long it = 1;
          while (sc >= 0 && s.eCost[sc] > t->cost) R[m].reach = sc, sc = cs_succ_gt(&s, sc), it++;
          R[m].stop = sc >= 0 ? sc : INT_MAX;
          if (it > longest) longest = it;
        }
      }
      g_batchWalkRounds += longest;
      for (int l = 0; l < nq; l++) {
        if (Q[l].done) continue;
        const int p = cs_pred_le(&s, F[Q[l].frag].y);
        Q[l].p = p;
        if (p >= 0) Q[l].cpFrag = s.eFrag[p], Q[l].cpLen = s.eLen[p], Q[l].cpCost = s.eCost[p], Q[l].cpXY = s.eXY[p];
      }
      /* the first dirty event in sequence order: seq = (t, R before Q, lane) */
      int64_t firstDirty = INT64_MAX;
      for (int i = 0; i < nr; i++) {
        if (R[i].done || !R[i].store) continue;
        g_writerLoops++;
        const int64_t seqI = ((int64_t)R[i].t << 16) + i;
        const int wy = F[R[i].frag].y, wr = R[i].reach;
        for (int m = 0; m < nr; m++) {
          const int64_t seqM = ((int64_t)R[m].t << 16) + m;
          if (R[m].done || seqM <= seqI) continue;
          if (F[R[m].frag].y <= wr && R[m].stop >= wy && seqM < firstDirty) firstDirty = seqM;
        }
        for (int l = 0; l < nq; l++) {
          const int64_t seqL = ((int64_t)Q[l].t << 16) + 256 + l;
          if (Q[l].done || seqL <= seqI) continue;
          if (Q[l].p <= wr && F[Q[l].frag].y >= wy && seqL < firstDirty) firstDirty = seqL;
        }
      }
      /* everything before it is final: the retirements' writes (disjoint ranges), the queries' answers */
      for (int m = 0; m < nr; m++) {
        if (R[m].done || ((int64_t)R[m].t << 16) + m >= firstDirty) continue;
        R[m].done = 1;
        left--;
        if (!R[m].store) continue;
        const frag *t = &F[R[m].frag];
        for (int c = t->y + 1; c <= R[m].reach; c++) s.bit[c] = 0;
        s.bit[t->y] = 1;
        s.eFrag[t->y] = R[m].frag, s.eCost[t->y] = t->cost, s.eLen[t->y] = t->clen, s.eXY[t->y] = t->x + t->y;
      }
      for (int l = 0; l < nq; l++)
        if (!Q[l].done && ((int64_t)Q[l].t << 16) + 256 + l < firstDirty) Q[l].done = 1, left--;
    }
    /* ---- score phase: the rows of the batch in order (cp at hand; cl, ca from fragments already scored) */
    for (int l = 0; l < nq; l++) {
      const int i = Q[l].frag;
      frag *f = &F[i];
      int cp = INT_MAX, cl = INT_MAX, ca = INT_MAX, found = 0;
      if (Q[l].p >= 0) cp = Q[l].cpCost + isqrt_floor(abs((f->x + f->y) - Q[l].cpXY)) - f->len, found = 1;
      if (f->predCl >= 0) {
        const frag *p = &F[f->predCl];
        int ov = (int)((uint32_t)k - (uint32_t)(f->y - p->y)) * match;
        if (ov > 0) ov = 0;
        cl = p->cost + ov + indel_penalty(f->x, f->y, p->x, p->y, ins, del);
        found = 1;
      }
      if (f->above >= 0) {
        const frag *p = &F[f->above];
        ca = (int)((uint32_t)p->cost + ((uint32_t)k - (uint32_t)(f->y - p->y)) * (uint32_t)match +
                   (uint32_t)indel_penalty(f->x, f->y, p->x, p->y, ins, del));
        found = 1;
      }
      int mc = cl < ca ? cl : ca;
      if (cp < mc) mc = cp;
      if (found && (type == 1 || (type == 0 && mc < 0))) {
        f->cost = (int)((uint32_t)mc - (uint32_t)k);
        if (mc == cp) f->prev = Q[l].cpFrag, f->clen = Q[l].cpLen + 1;
        else if (mc == cl) f->prev = f->predCl, f->clen = F[f->predCl].clen + 1;
        else f->prev = f->above, f->clen = F[f->above].clen + 1;
      } else {
        f->prev = -1;
        f->clen = 1;
        if (type == 1) f->cost = (int)((uint32_t)(f->x + f->y) * (uint32_t)del + (uint32_t)k * (uint32_t)match - (uint32_t)k);
        else f->cost = (int)((uint32_t)k * (uint32_t)match - (uint32_t)k);
      }
      if (minCost > f->cost) minCost = f->cost, minIdx = i;
      if ((unsigned)f->clen > maxLen) maxLen = (unsigned)f->clen, maxLenIdx = i;
    }
    fScore += nq;
    fTrail = trail;
  }
  cs_free(&s);
  return type == 0 ? minIdx : maxLenIdx;
}

/* ---- fragment sets ---- */
static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd(void) {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 16);
}
static void random_seq(unsigned char *s, int n, int alphabet, int runLen) {
  for (int i = 0; i < n;) {
    const unsigned char c = (unsigned char)(rnd() % (unsigned)alphabet);
    int r = runLen > 1 ? 1 + (int)(rnd() % (unsigned)runLen) : 1;
    while (r-- > 0 && i < n) s[i++] = c;
  }
}
static int cmp_xy(const void *a, const void *b) {
  const frag *fa = (const frag *)a, *fb = (const frag *)b;
  if (fa->x != fb->x) return fa->x < fb->x ? -1 : 1;
  if (fa->y != fb->y) return fa->y < fb->y ? -1 : 1;
  return 0;
}
/* all exact k-mer matches between q and t (what StoreMatchingPositions yields without a per-position limit) */
static int kmer_matches(const unsigned char *q, int qn, const unsigned char *t, int tn, int k, frag **out, int cap) {
  frag *F = (frag *)malloc(sizeof(frag) * (size_t)cap);
  int n = 0;
  for (int x = 0; x + k <= qn; x++)
    for (int y = 0; y + k <= tn && n < cap; y++)
      if (memcmp(q + x, t + y, (size_t)k) == 0) {
        memset(&F[n], 0, sizeof(frag));
        F[n].x = x, F[n].y = y, F[n].len = k;
        n++;
      }
  qsort(F, (size_t)n, sizeof(frag), cmp_xy);
  *out = F;
  return n;
}

static int check_case(const char *what, frag *F, int n, int qLen, int tLen, int k, int type, int ins, int del, int match) {
  static_preds(F, n, k);
  frag *G = (frag *)malloc(sizeof(frag) * (size_t)(n > 0 ? n : 1));
  memcpy(G, F, sizeof(frag) * (size_t)n);
  const int endA = n > 0 ? sweep_serial(F, n, k, tLen + 1, ins, del, match, type) : -1;
  const int endB = n > 0 ? sweep_rows(G, n, k, tLen + 1, ins, del, match, type) : -1;
  int bad = endA != endB;
  for (int i = 0; i < n && !bad; i++)
    if (F[i].cost != G[i].cost || F[i].prev != G[i].prev || F[i].clen != G[i].clen) {
      fprintf(stderr, "%s: fragment %d of %d (x %d y %d): serial cost %d prev %d len %d, rows cost %d prev %d len %d\n", what, i, n,
              F[i].x, F[i].y, F[i].cost, F[i].prev, F[i].clen, G[i].cost, G[i].prev, G[i].clen);
      bad = 1;
    }
  memcpy(G, F, sizeof(frag) * (size_t)n);
  for (int i = 0; i < n; i++) G[i].cost = G[i].prev = G[i].clen = 0;
  const int endC = n > 0 ? sweep_batches(G, n, k, tLen + 1, ins, del, match, type) : -1;
  if (!bad && endA != endC) fprintf(stderr, "%s: batches end at %d, serial at %d\n", what, endC, endA), bad = 1;
  for (int i = 0; i < n && !bad; i++)
    if (F[i].cost != G[i].cost || F[i].prev != G[i].prev || F[i].clen != G[i].clen) {
      fprintf(stderr, "%s: fragment %d of %d (x %d y %d): serial cost %d prev %d len %d, batches cost %d prev %d len %d\n", what, i,
              n, F[i].x, F[i].y, F[i].cost, F[i].prev, F[i].clen, G[i].cost, G[i].prev, G[i].clen);
      bad = 1;
    }
  /* the plain restatement against the oracle's: the reported chain */
  if (!bad && n > 0) {
    int *xyl = (int *)malloc(sizeof(int) * 3 * (size_t)n), *chain = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; i++) xyl[3 * i] = F[i].x, xyl[3 * i + 1] = F[i].y, xyl[3 * i + 2] = F[i].len;
    const int cn = bo_sdp_chain(qLen, xyl, n, k, ins, del, match, type == 0 ? 0 : 1, chain, n);
    int at = endA, m = 0;
    for (int c = endA; c != -1; c = F[c].prev) m++;
    if (m != cn) bad = 1;
    for (int i = cn - 1; i >= 0 && !bad; i--) {
      if (chain[i] != at) bad = 1;
      at = F[at].prev;
    }
    if (bad) fprintf(stderr, "%s: the restatement's chain (%d fragments) differs from the oracle's (%d)\n", what, m, cn);
    free(xyl), free(chain);
  }
  free(G);
  return bad;
}

int main(int argc, char **argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 40;
  int failures = 0, cases = 0;
  long frags = 0;
  for (int r = 0; r < rounds; r++) {
    /* kinds: 0 scattered chance matches (k = 5 / 7), 1 near-diagonal (a noisy copy, k = 7 / 11), 2 low complexity (runs of a
     * base, two-letter alphabet: dense rows, ties, long successor runs), 3 tiny */
    const int kind = argc > 2 ? atoi(argv[2]) : r % 4; /* (argv: rounds [kind q t k]: one kind at a chosen size) */
    int qn, tn, k, alphabet = 4, runLen = 1;
    if (kind == 0) qn = 200 + (int)(rnd() % 1500), tn = 200 + (int)(rnd() % 3000), k = (rnd() & 1) ? 5 : 7;
    else if (kind == 1) qn = 300 + (int)(rnd() % 2000), tn = 0, k = (rnd() & 1) ? 7 : 11;
    else if (kind == 2) qn = 60 + (int)(rnd() % 300), tn = 60 + (int)(rnd() % 400), k = 3 + (int)(rnd() % 4), alphabet = 2 + (int)(rnd() % 2), runLen = 1 + (int)(rnd() % 6);
    else qn = 5 + (int)(rnd() % 40), tn = 5 + (int)(rnd() % 40), k = 2 + (int)(rnd() % 3);
    if (argc > 5) qn = atoi(argv[3]), tn = atoi(argv[4]), k = atoi(argv[5]);
    unsigned char *q = (unsigned char *)malloc((size_t)qn + 8), *t;
    random_seq(q, qn, alphabet, runLen);
    if (kind == 1) { /* t = q with 15 % indels / substitutions */
      t = (unsigned char *)malloc((size_t)qn * 2 + 8);
      tn = 0;
      for (int i = 0; i < qn; i++) {
        const unsigned u = rnd() % 100;
        if (u < 5) continue;
        if (u < 10) t[tn++] = (unsigned char)(rnd() % 4);
        t[tn++] = u < 13 ? (unsigned char)(rnd() % 4) : q[i];
      }
    } else {
      t = (unsigned char *)malloc((size_t)tn + 8);
      random_seq(t, tn, alphabet, runLen);
    }
    frag *F;
    const int n = kmer_matches(q, qn, t, tn, k, &F, 4000000);
    for (int type = 0; type < 2; type++) {
      frag *C = (frag *)malloc(sizeof(frag) * (size_t)(n > 0 ? n : 1));
      memcpy(C, F, sizeof(frag) * (size_t)n);
      char what[128];
      snprintf(what, sizeof(what), "round %d kind %d q %d t %d k %d type %d", r, kind, qn, tn, k, type);
      /* the mapper's parameters (sdpIns 5, sdpDel 5... MappingParameters.h) and a second set with different signs */
      const int ins = (r & 8) ? 3 : 5, del = (r & 16) ? 7 : 5, match = (r & 4) ? -3 : -5;
      failures += check_case(what, C, n, qn, tn, k, type, ins, del, match);
      cases++;
      frags += n;
      free(C);
    }
    free(F), free(q), free(t);
  }
  printf("{\"cases\": %d, \"failures\": %d, \"fragments\": %ld, \"rows\": %ld, \"fragments_per_row\": %.2f, \"widest_row\": %ld, "
         "\"retired_rows\": %ld, \"evaluation_rounds_per_retired_row\": %.3f, \"walk_rounds_per_retired_row\": %.2f, "
         "\"deleted_successors_per_retired_fragment\": %.3f, \"in_row_above\": %ld, \"above_rounds\": %ld, "
         "\"batches\": %ld, \"events_per_batch\": %.1f, \"rows_per_batch\": %.1f, \"rounds_per_batch\": %.2f, "
         "\"walk_rounds_per_batch\": %.2f, \"writers_per_round\": %.1f}\n",
         cases, failures, frags, g_rows, g_rows ? (double)g_rowFrags / g_rows : 0.0, g_maxRowLanes, g_retRows,
         g_retRows ? (double)g_retRounds / g_retRows : 0.0, g_retRows ? (double)g_walkRounds / g_retRows : 0.0,
         g_retFrags ? (double)g_walkIters / g_retFrags : 0.0, g_inRowAbove, g_aboveRounds, g_batches,
         g_batches ? (double)g_batchEvents / g_batches : 0.0, g_batches ? (double)g_batchRows / g_batches : 0.0,
         g_batches ? (double)g_batchRounds / g_batches : 0.0, g_batches ? (double)g_batchWalkRounds / g_batches : 0.0,
         g_batchRounds ? (double)g_writerLoops / g_batchRounds : 0.0);
  return failures != 0;
}
