/* TEST INFRASTRUCTURE -- see blasr_oracle.h.  Plain-C restatement of the
 * reference algorithms on BLASR's refine path.  Written for clarity, not speed.
 * Reference citations are file:line under the reference checkout.
 *
 * Third-party algorithm restated here: libstdc++'s std::sort (GCC 11.4.0,
 * bits/stl_algo.h introsort: median-of-3 quicksort to runs of <= 16, heapsort
 * after 2*floor(log2 n) levels, final insertion sort).  SDPAlign.h:256 sorts a
 * fragment list that can hold duplicate (x,y) keys of different `length`, and
 * which duplicate survives the following de-duplication is decided by that
 * schedule, so it is cloned exactly (bo_introsort_xy).
 */
#include "blasr_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define INF_INT INT_MAX /* defs.h:15-17 */
#define MAX_BAND_SIZE 250 /* GuidedAlign.h:29 */

/* ------------------------------------------------------------------------- */
/* Nucleotide code tables (NucConversion.h:7-41 TwoBit, :48-86 ThreeBit,     */
/* :88-121 IsACTG), rebuilt from their defining property.                     */
/* ------------------------------------------------------------------------- */
static int g_two[256], g_three[256], g_isactg[256], g_tables_ready;

static void init_tables(void) {
  if (g_tables_ready) return;
  for (int i = 0; i < 256; i++) {
    g_two[i] = 255;
    g_three[i] = 255;
    g_isactg[i] = 0;
  }
  for (int i = 0; i < 8; i++) g_two[i] = i & 3;
  for (int i = 0; i < 4; i++) {
    g_three[i] = i;
    g_isactg[i] = 1;
  }
  g_three[4] = 4;
  g_three['$'] = 5;
  const char *acgt = "ACGT";
  for (int i = 0; i < 4; i++) {
    int u = acgt[i], l = acgt[i] + 32;
    g_two[u] = g_two[l] = i;
    g_three[u] = g_three[l] = i;
    g_isactg[u] = g_isactg[l] = 1;
  }
  /* IUPAC ambiguity codes that ThreeBit maps to 4 ('N'). */
  const char *nUpper = "BDHKMNRSUVWY_";
  for (const char *p = nUpper; *p; p++) g_three[(int)*p] = 4;
  const char *nLower = "bdhkmnrsuvwxy";
  for (const char *p = nLower; *p; p++) g_three[(int)*p] = 4;
  g_tables_ready = 1;
}

typedef struct {
  int m[5][5];
  int ins, del, affineOpen, affineExtend;
} score_fn;

static void make_score_fn(score_fn *fn, const int *matrix25, int ins, int del, int affOpen,
                          int affExt) {
  init_tables();
  for (int i = 0; i < 5; i++)
    for (int j = 0; j < 5; j++) fn->m[i][j] = matrix25[i * 5 + j];
  fn->ins = ins;
  fn->del = del;
  fn->affineOpen = affOpen;
  fn->affineExtend = affExt;
}

/* DistanceMatrixScoreFunction<DNASequence,FASTQSequence>::Match
 * (DistanceMatrixScoreFunction.h:99-104): matrix[query][ref]. */
static int match_score(const score_fn *fn, unsigned char tc, unsigned char qc) {
  return fn->m[g_three[qc]][g_three[tc]];
}

/* ------------------------------------------------------------------------- */
/* Alignment container (datastructures/alignment/Alignment.h:17-41).         */
/* ------------------------------------------------------------------------- */
typedef struct {
  uint32_t q, t, len;
} block_t;
typedef struct {
  int seq, len;
} gap_t;
typedef struct {
  gap_t *g;
  int n, cap;
} gaplist_t;
typedef struct {
  uint32_t qPos, tPos;
  block_t *blocks;
  int nBlocks, blockCap;
  gaplist_t *gaps;
  int nGaps, gapCap;
  int score, nCells, nMatch, nMismatch, nIns, nDel;
  float pct;
} aln_t;

static void aln_init(aln_t *a) { memset(a, 0, sizeof(*a)); }
static void aln_free(aln_t *a) {
  free(a->blocks);
  for (int i = 0; i < a->nGaps; i++) free(a->gaps[i].g);
  free(a->gaps);
  aln_init(a);
}
static void aln_push_block(aln_t *a, block_t b) {
  if (a->nBlocks == a->blockCap) {
    a->blockCap = a->blockCap ? a->blockCap * 2 : 64;
    a->blocks = (block_t *)realloc(a->blocks, sizeof(block_t) * a->blockCap);
  }
  a->blocks[a->nBlocks++] = b;
}
static gaplist_t *aln_push_gaplist(aln_t *a) {
  if (a->nGaps == a->gapCap) {
    a->gapCap = a->gapCap ? a->gapCap * 2 : 64;
    a->gaps = (gaplist_t *)realloc(a->gaps, sizeof(gaplist_t) * a->gapCap);
  }
  gaplist_t *g = &a->gaps[a->nGaps++];
  g->g = NULL;
  g->n = g->cap = 0;
  return g;
}
static void gaplist_push(gaplist_t *l, int seq, int len) {
  if (l->n == l->cap) {
    l->cap = l->cap ? l->cap * 2 : 4;
    l->g = (gap_t *)realloc(l->g, sizeof(gap_t) * l->cap);
  }
  l->g[l->n].seq = seq;
  l->g[l->n].len = len;
  l->n++;
}

/* Path.h:4-21 (the arrows this path produces). */
enum { A_DIAG, A_UP, A_LEFT, A_INS_UP, A_INS_OPEN, A_INS_CLOSE, A_DEL_LEFT, A_DEL_OPEN, A_DEL_CLOSE, A_NONE = 12 };

/* Alignment::ArrowPathToAlignment (Alignment.h:190-254): RLE of a forward
 * arrow path into blocks (runs of Diagonal) and one gap list per inter-block
 * slot; gap list 0 is the leading one; a list that reaches the end of the
 * path is emptied. */
static void arrow_path_to_alignment(aln_t *a, const unsigned char *path, int n) {
  int q = 0, t = 0, i = 0, beforeFirst = 1;
  while (i < n) {
    if (!beforeFirst && path[i] == A_DIAG) {
      block_t b = {(uint32_t)q, (uint32_t)t, 0};
      while (i < n && path[i] == A_DIAG) {
        b.len++;
        i++;
        t++;
        q++;
      }
      aln_push_block(a, b);
    }
    gaplist_t *gl = aln_push_gaplist(a);
    while (i < n && (path[i] == A_LEFT || path[i] == A_UP)) {
      int start = i;
      if (path[i] == A_LEFT) {
        while (i < n && path[i] == A_LEFT) {
          t++;
          i++;
        }
        gaplist_push(gl, BO_GAP_QUERY, i - start);
      } else {
        while (i < n && path[i] == A_UP) {
          q++;
          i++;
        }
        gaplist_push(gl, BO_GAP_TARGET, i - start);
      }
    }
    if (i == n) gl->n = 0;
    beforeFirst = 0;
  }
}

/* RemoveAlignmentPrefixGaps (AlignmentUtils.h:620-644). */
static void remove_prefix_gaps(aln_t *a) {
  if (a->nGaps == 0) return;
  int tStart = 0, qStart = 0;
  for (int g = 0; g < a->gaps[0].n; g++) {
    if (a->gaps[0].g[g].seq == BO_GAP_TARGET)
      qStart += a->gaps[0].g[g].len;
    else
      tStart += a->gaps[0].g[g].len;
  }
  for (int b = 0; b < a->nBlocks; b++) {
    a->blocks[b].q -= qStart;
    a->blocks[b].t -= tStart;
  }
  a->gaps[0].n = 0;
  a->tPos += tStart;
  a->qPos += qStart;
}

/* ------------------------------------------------------------------------- */
/* ComputeAlignmentStats (AlignmentUtils.h:535-584) over the column strings   */
/* of CreateAlignmentStrings (:390-533) and ComputeAlignmentScore (:60-124).  */
/* The strings are materialised as two byte arrays ('-' = gap), as there.     */
/* ------------------------------------------------------------------------- */
typedef struct {
  unsigned char *q, *t;
  int n, cap;
} cols_t;
static void cols_push(cols_t *c, unsigned char qc, unsigned char tc) {
  if (c->n == c->cap) {
    c->cap = c->cap ? c->cap * 2 : 1024;
    c->q = (unsigned char *)realloc(c->q, c->cap);
    c->t = (unsigned char *)realloc(c->t, c->cap);
  }
  c->q[c->n] = qc;
  c->t[c->n] = tc;
  c->n++;
}

static void compute_alignment_stats(aln_t *a, const unsigned char *query, const unsigned char *text,
                                    const score_fn *fn, int useAffine) {
  cols_t c = {0, 0, 0, 0};
  uint32_t q = a->qPos, t = a->tPos;
  if (a->nBlocks > 0) {
    if (a->nGaps == 0) {
      /* AlignmentUtils.h:408-446: offset of the first block spelled out. */
      if (a->blocks[0].q > 0 || a->blocks[0].t > 0) {
        uint32_t qp = a->blocks[0].q, tp = a->blocks[0].t;
        int common = (int)qp;
        if ((uint32_t)common > tp) common = (int)tp;
        for (int g = 0; g < common; g++) cols_push(&c, query[q++], text[t++]);
        tp -= common;
        qp -= common;
        for (uint32_t p = 0; p < tp; p++) cols_push(&c, '-', text[t++]);
        for (uint32_t p = 0; p < qp; p++) cols_push(&c, query[q++], '-');
      }
    } else {
      for (int gi = 0; gi < a->gaps[0].n; gi++)
        for (int gp = 0; gp < a->gaps[0].g[gi].len; gp++) {
          if (a->gaps[0].g[gi].seq == BO_GAP_QUERY)
            cols_push(&c, '-', text[t++]);
          else
            cols_push(&c, query[q++], '-');
        }
    }
    for (int b = 0; b < a->nBlocks; b++) {
      for (uint32_t bl = 0; bl < a->blocks[b].len; bl++) cols_push(&c, query[q++], text[t++]);
      if (b == a->nBlocks - 1) continue;
      if (a->nGaps > 0) {
        gaplist_t *gl = &a->gaps[b + 1];
        for (int gi = 0; gi < gl->n; gi++)
          for (int gp = 0; gp < gl->g[gi].len; gp++) {
            if (gl->g[gi].seq == BO_GAP_QUERY)
              cols_push(&c, '-', text[t++]);
            else
              cols_push(&c, query[q++], '-');
          }
      } else {
        /* AlignmentUtils.h:494-529: no gap list -> query surplus, then text
         * surplus, then the common part as aligned columns. */
        int qGap = (int)(a->blocks[b + 1].q - a->blocks[b].q - a->blocks[b].len);
        int tGap = (int)(a->blocks[b + 1].t - a->blocks[b].t - a->blocks[b].len);
        if (qGap > 0 || tGap > 0) {
          int common = qGap;
          if (qGap > tGap) common = tGap;
          tGap -= common;
          qGap -= common;
          for (int g = 0; g < qGap; g++) cols_push(&c, query[q++], '-');
          for (int g = 0; g < tGap; g++) cols_push(&c, '-', text[t++]);
          for (int g = 0; g < common; g++) cols_push(&c, query[q++], text[t++]);
        }
      }
    }
  }
  int nMatch = 0, nMismatch = 0, nIns = 0, nDel = 0, tp = 0, qp = 0;
  for (int i = 0; i < c.n; i++) {
    if (c.t[i] != '-' && c.q[i] != '-') {
      if (g_three[c.t[i]] == g_three[c.q[i]])
        nMatch++;
      else
        nMismatch++;
      tp++;
      qp++;
    } else if (c.t[i] == '-' && c.q[i] != '-') {
      nIns++;
      qp++;
    } else if (c.q[i] == '-' && c.t[i] != '-') {
      nDel++;
      tp++;
    }
  }
  float pct = 0;
  if (tp + qp > 0 && c.n + c.n > 0) pct = (float)((nMatch * 2.0) / (double)(c.n + c.n) * 100);
  /* ComputeAlignmentScore(string) AlignmentUtils.h:60-124 */
  int score = 0;
  for (int i = 0; i < c.n; i++) {
    if (c.q[i] != '-' && c.t[i] != '-') {
      score += fn->m[g_three[c.q[i]]][g_three[c.t[i]]];
    } else if (useAffine) {
      int gapEnd = i;
      while (gapEnd < c.n && (c.q[gapEnd] == '-' || c.t[gapEnd] == '-')) ++gapEnd;
      int gapLength = gapEnd - i;
      int affineScore = gapLength * fn->affineExtend + fn->affineOpen;
      int gapScore = (c.q[gapEnd - 1] == '-') ? fn->del * gapLength : fn->ins * gapLength;
      score += (gapScore < affineScore) ? gapScore : affineScore;
      i = gapEnd - 1;
    } else {
      if (c.q[i] == '-' && c.t[i] != '-')
        score += fn->del;
      else if (c.q[i] != '-' && c.t[i] == '-')
        score += fn->ins;
      else
        score += fn->m[4][4];
    }
  }
  a->score = score;
  a->nMatch = nMatch;
  a->nMismatch = nMismatch;
  a->nIns = nIns;
  a->nDel = nDel;
  a->pct = pct;
  free(c.q);
  free(c.t);
}

/* ------------------------------------------------------------------------- */
/* SWAlign, Global mode (SWAlign.h:19-389): boundary del*c / ins*r (:49-64),  */
/* arrow priority Diagonal, Up, Left (:196-207), traceback (q,t)->(0,0)       */
/* (:231-238, :324-353).                                                      */
/* ------------------------------------------------------------------------- */
static int sw_align_global(const unsigned char *q, int qlen, const unsigned char *t, int tlen,
                           const score_fn *fn, aln_t *out) {
  int nRows = qlen + 1, nCols = tlen + 1;
  int *S = (int *)calloc((size_t)nRows * nCols, sizeof(int));
  unsigned char *P = (unsigned char *)malloc((size_t)nRows * nCols);
  memset(P, A_NONE, (size_t)nRows * nCols);
  for (int c = 0; c < nCols; c++) {
    S[c] = fn->del * c;
    P[c] = A_LEFT;
  }
  for (int r = 0; r < nRows; r++) {
    S[(size_t)r * nCols] = fn->ins * r;
    P[(size_t)r * nCols] = A_UP;
  }
  P[0] = A_DIAG;
  for (int r = 0; r < qlen; r++)
    for (int c = 0; c < tlen; c++) {
      int m = match_score(fn, t[c], q[r]) + S[(size_t)r * nCols + c];
      int qGap = S[(size_t)r * nCols + c + 1] + fn->ins;
      int tGap = S[(size_t)(r + 1) * nCols + c] + fn->del;
      int mn = m < qGap ? m : qGap;
      if (tGap < mn) mn = tGap;
      S[(size_t)(r + 1) * nCols + c + 1] = mn;
      P[(size_t)(r + 1) * nCols + c + 1] = (mn == m) ? A_DIAG : (mn == qGap) ? A_UP : A_LEFT;
    }
  int r = qlen, c = tlen, n = 0;
  unsigned char *path = (unsigned char *)malloc((size_t)qlen + tlen + 2);
  while (r > 0 || c > 0) {
    unsigned char a = P[(size_t)r * nCols + c];
    path[n++] = a;
    if (a == A_DIAG) {
      r--;
      c--;
    } else if (a == A_UP)
      r--;
    else if (a == A_LEFT)
      c--;
  }
  for (int i = 0; i < n / 2; i++) {
    unsigned char tmp = path[i];
    path[i] = path[n - 1 - i];
    path[n - 1 - i] = tmp;
  }
  if (n > 0) arrow_path_to_alignment(out, path, n);
  int score = S[(size_t)qlen * nCols + tlen];
  free(path);
  free(S);
  free(P);
  return score;
}

/* ------------------------------------------------------------------------- */
/* libstdc++ std::sort clone on fragments keyed by (x,y)                      */
/* ------------------------------------------------------------------------- */
typedef struct {
  uint32_t x, y, weight, length;
  int index, chainPrev, cost, above;
  uint32_t chainLength;
} frag_t; /* sdp/SDPFragment.h:5-17 */

static int frag_less_xy(const frag_t *a, const frag_t *b) { /* SDPFragment.h:60-67 */
  if (a->x < b->x) return 1;
  if (a->x == b->x) return a->y < b->y;
  return 0;
}

static void is_swap(frag_t *a, frag_t *b) {
  frag_t t = *a;
  *a = *b;
  *b = t;
}
static void is_adjust_heap(frag_t *first, long hole, long len, frag_t value) {
  long top = hole, child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (frag_less_xy(&first[child], &first[child - 1])) child--;
    first[hole] = first[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole = child - 1;
  }
  long parent = (hole - 1) / 2;
  while (hole > top && frag_less_xy(&first[parent], &value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}
static void is_heapsort(frag_t *first, frag_t *last) {
  long len = last - first;
  if (len >= 2) {
    long parent = (len - 2) / 2;
    for (;;) {
      frag_t v = first[parent];
      is_adjust_heap(first, parent, len, v);
      if (parent == 0) break;
      parent--;
    }
  }
  while (last - first > 1) {
    --last;
    frag_t v = *last;
    *last = *first;
    is_adjust_heap(first, 0, last - first, v);
  }
}
static void is_median_to_first(frag_t *result, frag_t *a, frag_t *b, frag_t *c) {
  if (frag_less_xy(a, b)) {
    if (frag_less_xy(b, c))
      is_swap(result, b);
    else if (frag_less_xy(a, c))
      is_swap(result, c);
    else
      is_swap(result, a);
  } else if (frag_less_xy(a, c))
    is_swap(result, a);
  else if (frag_less_xy(b, c))
    is_swap(result, c);
  else
    is_swap(result, b);
}
static frag_t *is_partition(frag_t *first, frag_t *last, frag_t *pivot) {
  for (;;) {
    while (frag_less_xy(first, pivot)) ++first;
    --last;
    while (frag_less_xy(pivot, last)) --last;
    if (!(first < last)) return first;
    is_swap(first, last);
    ++first;
  }
}
static void is_loop(frag_t *first, frag_t *last, long depth) {
  while (last - first > 16) {
    if (depth == 0) {
      is_heapsort(first, last);
      return;
    }
    --depth;
    frag_t *mid = first + (last - first) / 2;
    is_median_to_first(first, first + 1, mid, last - 1);
    frag_t *cut = is_partition(first + 1, last, first);
    is_loop(cut, last, depth);
    last = cut;
  }
}
static void is_unguarded_linear_insert(frag_t *last) {
  frag_t val = *last;
  frag_t *next = last - 1;
  while (frag_less_xy(&val, next)) {
    *last = *next;
    last = next;
    --next;
  }
  *last = val;
}
static void is_insertion_sort(frag_t *first, frag_t *last) {
  if (first == last) return;
  for (frag_t *i = first + 1; i != last; ++i) {
    if (frag_less_xy(i, first)) {
      frag_t val = *i;
      memmove(first + 1, first, (size_t)(i - first) * sizeof(frag_t));
      *first = val;
    } else
      is_unguarded_linear_insert(i);
  }
}
static void introsort_frags(frag_t *first, long n) {
  if (n <= 0) return;
  frag_t *last = first + n;
  long lg = 0;
  for (long m = n; m > 1; m >>= 1) lg++;
  is_loop(first, last, lg * 2);
  if (n > 16) {
    is_insertion_sort(first, first + 16);
    for (frag_t *i = first + 16; i != last; ++i) is_unguarded_linear_insert(i);
  } else
    is_insertion_sort(first, last);
}

void bo_introsort_xy(int *xyl, int n) {
  frag_t *f = (frag_t *)calloc(n > 0 ? n : 1, sizeof(frag_t));
  for (int i = 0; i < n; i++) {
    f[i].x = xyl[3 * i];
    f[i].y = xyl[3 * i + 1];
    f[i].length = xyl[3 * i + 2];
  }
  introsort_frags(f, n);
  for (int i = 0; i < n; i++) {
    xyl[3 * i] = (int)f[i].x;
    xyl[3 * i + 1] = (int)f[i].y;
    xyl[3 * i + 2] = (int)f[i].length;
  }
  free(f);
}

/* ------------------------------------------------------------------------- */
/* SDPLongestCommonSubsequence (sdp/SparseDynamicProgramming.h:70-322)        */
/* ------------------------------------------------------------------------- */

/* IndelPenalty (SparseDynamicProgramming.h:27-53) */
static int indel_penalty(int x1, int y1, int x2, int y2, int insertion, int deletion) {
  int drift = (x1 - y1) - (x2 - y2);
  if (drift > 0) return (int)((1.0 * drift) * insertion);
  if (drift < 0) return (int)((-1.0 * drift) * deletion);
  return 0;
}

static int cmp_frag_yx(const void *pa, const void *pb) { /* SDPFragment.h:69-76 */
  const frag_t *a = (const frag_t *)pa, *b = (const frag_t *)pb;
  if (a->y != b->y) return a->y < b->y ? -1 : 1;
  if (a->x != b->x) return a->x < b->x ? -1 : 1;
  return 0;
}
static int cmp_frag_xy(const void *pa, const void *pb) {
  const frag_t *a = (const frag_t *)pa, *b = (const frag_t *)pb;
  if (a->x != b->x) return a->x < b->x ? -1 : 1;
  if (a->y != b->y) return a->y < b->y ? -1 : 1;
  return 0;
}

/* The sweep's ordered set (SDPSet<Fragment>, sdp/SDPSet.h) keyed by
 * Fragment::operator< = (diag = y - x, then x) (SDPFragment.h:78-91); kept as
 * a sorted array of fragment indices. */
typedef struct {
  int *v;
  int n, cap;
} iset_t;
static void iset_insert_at(iset_t *s, int pos, int val) {
  if (s->n == s->cap) {
    s->cap = s->cap ? s->cap * 2 : 256;
    s->v = (int *)realloc(s->v, sizeof(int) * s->cap);
  }
  memmove(s->v + pos + 1, s->v + pos, sizeof(int) * (size_t)(s->n - pos));
  s->v[pos] = val;
  s->n++;
}
static void iset_erase_at(iset_t *s, int pos) {
  memmove(s->v + pos, s->v + pos + 1, sizeof(int) * (size_t)(s->n - pos - 1));
  s->n--;
}
static int frag_less_diag(const frag_t *a, const frag_t *b) {
  int da = (int)(a->y - a->x), db = (int)b->y - (int)b->x;
  if (da < db) return 1;
  if (da == db) return a->x < b->x;
  return 0;
}
/* first position whose element is not less than f */
static int sweep_lower_bound(const iset_t *s, const frag_t *F, const frag_t *f) {
  int lo = 0, hi = s->n;
  while (lo < hi) {
    int mid = (lo + hi) / 2;
    if (frag_less_diag(&F[s->v[mid]], f))
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}
/* column set (SDPSet<SDPColumn>, sdp/SDPColumn.h): sorted by col */
typedef struct {
  int *col, *opt;
  int n, cap;
} cset_t;
static int cset_lower_bound(const cset_t *s, int col) {
  int lo = 0, hi = s->n;
  while (lo < hi) {
    int mid = (lo + hi) / 2;
    if (s->col[mid] < col)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

/* probe (diagnostics for the device kernels' design, not part of any result): how large the column set of a sweep
 * gets and how far from its top the operations land.  out: [0] sweeps, [1] fragments, [2] largest set, [3] set
 * operations, [4] operations more than 16 entries below the top, [5..9] fragments in sweeps whose largest set was
 * <= 16, <= 64, <= 256, <= 1024, larger */
static int g_cs_probe = 0;
static long g_cs_stat[10];
void bo_probe_colset(int enable, long *out10) {
  if (out10) memcpy(out10, g_cs_stat, sizeof(g_cs_stat));
  g_cs_probe = enable;
  memset(g_cs_stat, 0, sizeof(g_cs_stat));
}
#define CS_PROBE_OP(pos, n)                              \
  do {                                                   \
    if (g_cs_probe) {                                    \
      g_cs_stat[3]++;                                    \
      if ((pos) < (n) - 16) g_cs_stat[4]++;              \
      if ((n) > csMax) csMax = (n);                      \
    }                                                    \
  } while (0)

static int sdp_lcs(uint32_t queryLength, frag_t *F, int nF, uint32_t fragmentLength, int insertion,
                   int deletion, int match, int alignType, int **chainOut) {
  *chainOut = NULL;
  if (nF < 1) return 0;
  int csMax = 0;
  qsort(F, nF, sizeof(frag_t), cmp_frag_xy); /* :78 (keys unique -> order unique) */
  for (int i = 0; i < nF; i++) F[i].index = i;
  /* StoreAbove :55-68 */
  qsort(F, nF, sizeof(frag_t), cmp_frag_yx);
  for (int i = 1; i < nF; i++)
    if (F[i - 1].x <= F[i].x && F[i - 1].x + F[i - 1].length > F[i].x && F[i - 1].y < F[i].y)
      F[i].above = F[i - 1].index;
  qsort(F, nF, sizeof(frag_t), cmp_frag_xy);

  iset_t sweep = {0, 0, 0};
  cset_t cols = {0, 0, 0, 0};
  uint32_t sweepRow = F[0].x;
  int fSweep = 0, fTrail = 0;
  uint32_t maxChainLength = 0;
  int maxChainFragment = -1, minFragmentCost = INF_INT, minFragmentIndex = -1;
  for (; sweepRow < queryLength + fragmentLength; sweepRow++) {
    int startF = fSweep;
    while (fSweep < nF && F[fSweep].x == sweepRow) {
      frag_t *f = &F[fSweep];
      int cp = INF_INT, cl = INF_INT, ca = INF_INT, foundPrev = 0;
      int cpFrag = -1, clFrag = -1, aboveIndex = -1;
      /* colSet.Predecessor: largest column <= f.y (SDPSet.h:94-119) */
      if (cols.n > 0) {
        int pos = cset_lower_bound(&cols, (int)f->y);
        CS_PROBE_OP(pos, cols.n);
        int have = 0;
        if (pos < cols.n && cols.col[pos] == (int)f->y)
          have = 1;
        else if (pos > 0) {
          pos--;
          have = 1;
        }
        if (have) {
          cpFrag = cols.opt[pos];
          const frag_t *p = &F[cpFrag];
          cp = p->cost + (int)sqrt(abs((int)(f->x + f->y) - (int)(p->x + p->y))) - (int)f->length;
          foundPrev = 1;
        }
      }
      /* sweepSet.Predecessor: greatest element below (diag,x) */
      if (sweep.n > 0) {
        int pos = sweep_lower_bound(&sweep, F, f);
        if (pos > 0) {
          clFrag = sweep.v[pos - 1];
          const frag_t *p = &F[clFrag];
          int ov = (int)(fragmentLength - (f->y - p->y)) * match;
          if (ov > 0) ov = 0;
          cl = p->cost + ov + indel_penalty((int)f->x, (int)f->y, (int)p->x, (int)p->y, insertion, deletion);
          foundPrev = 1;
        }
      }
      if (f->above >= 0) {
        aboveIndex = f->above;
        const frag_t *p = &F[aboveIndex];
        ca = (int)((uint32_t)p->cost +
                   (fragmentLength - (uint32_t)(int)(f->y - p->y)) * (uint32_t)match +
                   (uint32_t)indel_penalty((int)f->x, (int)f->y, (int)p->x, (int)p->y, insertion,
                                           deletion));
        foundPrev = 1;
      }
      int minCost = cl < ca ? cl : ca;
      if (cp < minCost) minCost = cp;
      if (foundPrev && (alignType == BO_GLOBAL || (alignType == BO_LOCAL && minCost < 0))) {
        f->cost = (int)((uint32_t)minCost - f->weight);
        if (minCost == cp)
          f->chainPrev = cpFrag;
        else if (minCost == cl)
          f->chainPrev = clFrag;
        else if (minCost == ca)
          f->chainPrev = aboveIndex;
        f->chainLength = F[f->chainPrev].chainLength + 1;
      } else {
        f->chainPrev = -1;
        f->chainLength = 1;
        if (alignType == BO_GLOBAL)
          f->cost = (int)((f->x + f->y) * (uint32_t)deletion + fragmentLength * (uint32_t)match - f->weight);
        else
          f->cost = (int)(fragmentLength * (uint32_t)match - f->weight);
      }
      if (minFragmentCost > f->cost) {
        minFragmentCost = f->cost;
        minFragmentIndex = fSweep;
      }
      if (f->chainLength > maxChainLength) {
        maxChainLength = f->chainLength;
        maxChainFragment = fSweep;
      }
      fSweep++;
    }
    for (int i = startF; i < fSweep; i++) {
      int pos = sweep_lower_bound(&sweep, F, &F[i]);
      iset_insert_at(&sweep, pos, i);
    }
    if (sweepRow >= fragmentLength + 1) {
      uint32_t trailRow = sweepRow - fragmentLength - 1;
      while (fTrail < nF && F[fTrail].x == trailRow) {
        frag_t *f = &F[fTrail];
        int pos = cset_lower_bound(&cols, (int)f->y);
        CS_PROBE_OP(pos, cols.n);
        int member = (pos < cols.n && cols.col[pos] == (int)f->y);
        int store = 0;
        if (member) {
          if (F[cols.opt[pos]].cost < f->cost) store = 1; /* :260-267, sic */
        } else
          store = 1;
        if (store) {
          if (member)
            cols.opt[pos] = fTrail;
          else {
            if (cols.n == cols.cap) {
              cols.cap = cols.cap ? cols.cap * 2 : 256;
              cols.col = (int *)realloc(cols.col, sizeof(int) * cols.cap);
              cols.opt = (int *)realloc(cols.opt, sizeof(int) * cols.cap);
            }
            memmove(cols.col + pos + 1, cols.col + pos, sizeof(int) * (size_t)(cols.n - pos));
            memmove(cols.opt + pos + 1, cols.opt + pos, sizeof(int) * (size_t)(cols.n - pos));
            cols.col[pos] = (int)f->y;
            cols.opt[pos] = fTrail;
            cols.n++;
          }
          /* Successor needs size >= 2 (SDPSet.h:72-87); delete costlier ones */
          int del = 0;
          while (cols.n - del >= 2 && pos + 1 + del < cols.n && F[cols.opt[pos + 1 + del]].cost > f->cost)
            del++;
          if (del) {
            memmove(cols.col + pos + 1, cols.col + pos + 1 + del, sizeof(int) * (size_t)(cols.n - pos - 1 - del));
            memmove(cols.opt + pos + 1, cols.opt + pos + 1 + del, sizeof(int) * (size_t)(cols.n - pos - 1 - del));
            cols.n -= del;
          }
        }
        int sp = sweep_lower_bound(&sweep, F, f);
        if (sp < sweep.n && sweep.v[sp] == fTrail) iset_erase_at(&sweep, sp);
        ++fTrail;
      }
    }
  }
  if (g_cs_probe) {
    g_cs_stat[0]++;
    g_cs_stat[1] += nF;
    if (csMax > g_cs_stat[2]) g_cs_stat[2] = csMax;
    g_cs_stat[csMax <= 16 ? 5 : csMax <= 64 ? 6 : csMax <= 256 ? 7 : csMax <= 1024 ? 8 : 9] += nF;
  }
  if (alignType == BO_LOCAL) maxChainFragment = minFragmentIndex;
  int n = 0, cap = 64;
  int *chain = (int *)malloc(sizeof(int) * cap);
  while (maxChainFragment != -1) {
    if (n == cap) {
      cap *= 2;
      chain = (int *)realloc(chain, sizeof(int) * cap);
    }
    chain[n++] = maxChainFragment;
    maxChainFragment = F[maxChainFragment].chainPrev;
  }
  for (int i = 0; i < n / 2; i++) {
    int tmp = chain[i];
    chain[i] = chain[n - 1 - i];
    chain[n - 1 - i] = tmp;
  }
  free(sweep.v);
  free(cols.col);
  free(cols.opt);
  *chainOut = chain;
  return n;
}

/* ------------------------------------------------------------------------- */
/* Tuples (tuples/DNATuple.h:308-359, TupleList.h:98-126, TupleMatching.h:37-61)*/
/* ------------------------------------------------------------------------- */
typedef struct {
  uint64_t tuple;
  uint32_t pos;
} ptuple_t;
static int cmp_ptuple(const void *pa, const void *pb) { /* DNATuple.h:147-154 */
  const ptuple_t *a = (const ptuple_t *)pa, *b = (const ptuple_t *)pb;
  if (a->tuple != b->tuple) return a->tuple < b->tuple ? -1 : 1;
  if (a->pos != b->pos) return a->pos < b->pos ? -1 : 1;
  return 0;
}
/* all k-mers whose k bases are ACGT; first base in the least-significant bits */
static int sequence_to_tuple_list(const unsigned char *s, uint32_t n, int k, ptuple_t **out) {
  ptuple_t *v = (ptuple_t *)malloc(sizeof(ptuple_t) * (n ? n : 1));
  int cnt = 0;
  if (k >= 1 && n >= (uint32_t)k) {
    uint32_t run = 0;
    uint64_t tup = 0;
    for (uint32_t i = 0; i < n; i++) {
      if (g_isactg[s[i]]) {
        run++;
        tup >>= 2;
        tup += ((uint64_t)g_two[s[i]]) << (2 * (k - 1));
        if (run >= (uint32_t)k) {
          v[cnt].tuple = tup;
          v[cnt].pos = i + 1 - k;
          cnt++;
        }
      } else {
        run = 0;
        tup = 0;
      }
    }
  }
  *out = v;
  return cnt;
}

typedef struct {
  frag_t *v;
  int n, cap;
} fvec_t;
static void fvec_push(fvec_t *fv, uint32_t x, uint32_t y, uint32_t weight, uint32_t length) {
  if (fv->n == fv->cap) {
    fv->cap = fv->cap ? fv->cap * 2 : 1024;
    fv->v = (frag_t *)realloc(fv->v, sizeof(frag_t) * fv->cap);
  }
  frag_t *f = &fv->v[fv->n++];
  memset(f, 0, sizeof(*f));
  f->x = x;
  f->y = y;
  f->weight = weight;
  f->length = length;
  f->above = -1;
}

/* StoreMatchingPositions (TupleMatching.h:37-61): note ThreeBit (not IsACTG)
 * decides k-mer validity on the query side; identical on ACGTN input.
 *
 * Words of 16 bases and more.  The query's tuples are NOT built the way the target's are: the first one of a clean stretch
 * comes from FromStringRL (DNATuple.h:55-83, 64-bit arithmetic, exact), every later one from ShiftAddRL (:85-94),
 *     tuple >>= 2;  tuple += (TwoBit[nuc] << ((tm.tupleSize-1)*2));
 * and TwoBit is `static int[]` (NucConversion.h:7): the shift is done in 32-bit int.  For tupleSize == 16 a G or T shifted by
 * 30 overflows int; from 17 on the count is >= 32.  Both are undefined in C++; what every build of the reference on x86-64
 * (and AArch64) computes is the shift with its count taken mod 32 and the int result sign-extended into the 64-bit sum --
 * `sall %cl, %eax; cltq; addq` -- so that from 16 bases on the query's tuples after the first of a stretch are not its words
 * and match the target's (which ARE its words, DNATuple.h:308-359) only by accident.  The recurrence is restated as
 * compiled; tests/test_oracle_vs_ref.py and tests/golden/long_words_golden.npz pin it to the reference built here. */
static void store_matching_positions(const unsigned char *qs, uint32_t qn, int k, const ptuple_t *tl,
                                     int tn, int maxMatches, uint32_t xOff, uint32_t yOff,
                                     uint32_t weight, uint32_t length, fvec_t *out) {
  if (k < 1 || qn < (uint32_t)k) return;
  int res = 0; /* the previous position had a tuple (TupleMatching.h:44-47) */
  uint64_t tup = 0;
  const int sh = (2 * (k - 1)) & 31;
  for (uint32_t s = 0; s + (uint32_t)k <= qn; s++) {
    if (res) { /* ShiftAddRL with the base that enters the window */
      const unsigned char nuc = qs[s + k - 1];
      if (g_three[nuc] > 3) {
        res = 0;
      } else {
        tup >>= 2;
        tup += (uint64_t)(int64_t)(int32_t)((uint32_t)g_two[nuc] << sh);
      }
    }
    if (!res) { /* FromStringRL: only a window of ACGT has a tuple */
      int clean = 1;
      for (int j = 0; j < k; j++) clean = clean && g_three[qs[s + j]] <= 3;
      if (clean) {
        tup = 0;
        for (int j = k - 1; j > 0; j--) {
          tup += (uint64_t)g_two[qs[s + j]];
          tup <<= 2;
        }
        tup += (uint64_t)g_two[qs[s]];
        res = 1;
      }
    }
    if (!res) continue;
    int lo = 0, hi = tn;
    while (lo < hi) {
      int mid = (lo + hi) / 2;
      if (tl[mid].tuple < tup)
        lo = mid + 1;
      else
        hi = mid;
    }
    int end = lo;
    while (end < tn && tl[end].tuple == tup) end++;
    if (maxMatches == 0 || end - lo <= maxMatches)
      for (int j = lo; j < end; j++) fvec_push(out, s + xOff, tl[j].pos + yOff, weight, length);
  }
}

/* ------------------------------------------------------------------------- */
/* SDPAlign (SDPAlign.h:95-626)                                               */
/* ------------------------------------------------------------------------- */
#define SDP_DETAILED_WORD_SIZE 5 /* SDPAlign.h:19 */

static uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

static int sdp_align(const unsigned char *query, uint32_t qLen, const unsigned char *target,
                     uint32_t tLen, const score_fn *fn, int wordSize, int sdpIns, int sdpDel,
                     float indelRate, aln_t *alignment, int alignType, int detailed,
                     int extendFront, int sdpPrefixLength, int recurse, int noRecurseUnder,
                     int maxMatchesPerPosition) {
  int smallWordSize = wordSize < SDP_DETAILED_WORD_SIZE ? wordSize : SDP_DETAILED_WORD_SIZE;
  /* :148-182 prefix / whole / suffix partition */
  uint32_t prefixLength = umin(tLen, (uint32_t)sdpPrefixLength);
  uint32_t suffixLength = umin(tLen - prefixLength, (uint32_t)sdpPrefixLength);
  uint32_t middleLength = tLen - prefixLength - suffixLength;
  uint32_t suffixPos = prefixLength + middleLength;
  uint32_t qPrefixLength = umin(qLen, (uint32_t)sdpPrefixLength);
  uint32_t qSuffixLength = umin(qLen - qPrefixLength, (uint32_t)sdpPrefixLength);
  uint32_t qMiddleLength = qLen - qPrefixLength - qSuffixLength;
  uint32_t qSuffixPos = qPrefixLength + qMiddleLength;

  ptuple_t *tp, *tsf, *tm;
  int ntp = sequence_to_tuple_list(target, prefixLength, smallWordSize, &tp);
  int nts = sequence_to_tuple_list(target + suffixPos, suffixLength, smallWordSize, &tsf);
  int ntm = sequence_to_tuple_list(target, tLen, wordSize, &tm);
  qsort(tp, ntp, sizeof(ptuple_t), cmp_ptuple);
  qsort(tsf, nts, sizeof(ptuple_t), cmp_ptuple);
  qsort(tm, ntm, sizeof(ptuple_t), cmp_ptuple);

  /* :200-242: prefix, middle, suffix fragments concatenated in that order */
  fvec_t fr = {0, 0, 0};
  store_matching_positions(query, qPrefixLength, smallWordSize, tp, ntp, maxMatchesPerPosition, 0, 0,
                           wordSize, smallWordSize, &fr);
  store_matching_positions(query, qLen, wordSize, tm, ntm, maxMatchesPerPosition, 0, 0, wordSize,
                           wordSize, &fr);
  store_matching_positions(query + qSuffixPos, qSuffixLength, smallWordSize, tsf, nts,
                           maxMatchesPerPosition, qSuffixPos, suffixPos, wordSize, smallWordSize, &fr);
  free(tp);
  free(tsf);
  free(tm);

  /* :256-269 std::sort + keep first of each (x,y) */
  introsort_frags(fr.v, fr.n);
  {
    int f = 0, cur = 0;
    while (f + 1 <= fr.n) {
      fr.v[cur] = fr.v[f];
      while (f < fr.n && fr.v[cur].x == fr.v[f].x && fr.v[cur].y == fr.v[f].y) f++;
      cur++;
    }
    fr.n = cur;
  }
  alignment->qPos = 0;
  alignment->tPos = 0;
  if (fr.n == 0) {
    free(fr.v);
    return 0;
  }
  int *chain;
  int nChain = sdp_lcs(qLen, fr.v, fr.n, (uint32_t)wordSize, sdpIns, sdpDel, fn->m[0][0], alignType, &chain);

  /* :308-335 merge +1/+1 runs into blocks */
  block_t *cb = (block_t *)malloc(sizeof(block_t) * (nChain ? nChain : 1));
  int ncb = 0;
  for (int f = 0; f < nChain; f++) {
    int startF = f;
    while (f < nChain - 1 && fr.v[chain[f]].x == fr.v[chain[f + 1]].x - 1 &&
           fr.v[chain[f]].y == fr.v[chain[f + 1]].y - 1)
      f++;
    cb[ncb].q = fr.v[chain[startF]].x;
    cb[ncb].t = fr.v[chain[startF]].y;
    cb[ncb].len = fr.v[chain[f]].x + fr.v[chain[f]].length - fr.v[chain[startF]].x;
    ncb++;
  }
  free(chain);
  free(fr.v);
  /* :349-357 trim overlaps */
  for (int b = 0; b + 1 < ncb; b++) {
    if (cb[b].q + cb[b].len > cb[b + 1].q) cb[b].len = cb[b + 1].q - cb[b].q;
    if (cb[b].t + cb[b].len > cb[b + 1].t) cb[b].len = cb[b + 1].t - cb[b].t;
  }
  /* :372-413 drop zero-length and off-diagonal blocks */
  unsigned char *good = (unsigned char *)malloc(ncb ? ncb : 1);
  for (int b = 0; b < ncb; b++) good[b] = cb[b].len != 0;
  for (int b = 1; b + 1 < ncb; b++) {
    int prevDiag = abs(((int)cb[b].t - (int)cb[b].q) - ((int)cb[b - 1].t - (int)cb[b - 1].q));
    int prevDist = (int)umin(cb[b].t - cb[b - 1].t, cb[b].q - cb[b - 1].q);
    int nextDiag = abs(((int)cb[b + 1].t - (int)cb[b + 1].q) - ((int)cb[b].t - (int)cb[b].q));
    int nextDist = (int)umin(cb[b + 1].t - cb[b].t, cb[b + 1].q - cb[b].q);
    if (prevDist * indelRate < prevDiag && nextDist * indelRate < nextDiag) good[b] = 0;
  }
  {
    int w = 0;
    for (int b = 0; b < ncb; b++)
      if (good[b]) cb[w++] = cb[b];
    ncb = w;
  }
  free(good);

  if (ncb > 0) {
    /* :423-474 front fill */
    if (cb[0].q > 0 && cb[0].t > 0 && (alignType == BO_GLOBAL || extendFront)) {
      uint32_t fq = cb[0].q, ft = cb[0].t;
      aln_t front;
      aln_init(&front);
      if (recurse == 0 && fq * ft < (uint32_t)noRecurseUnder)
        sw_align_global(query, (int)fq, target, (int)ft, fn, &front);
      else if (recurse != 0)
        /* :445-456: smithWatermanAlignType (EndAnchored = 6) lands in the
         * maxMatchesPerPosition slot */
        sdp_align(query, fq, target, ft, fn, wordSize - 4 > 5 ? wordSize - 4 : 5, sdpIns, sdpDel,
                  indelRate, &front, BO_GLOBAL, detailed, extendFront, sdpPrefixLength, recurse - 1,
                  noRecurseUnder, BO_END_ANCHORED);
      for (int i = 0; i < front.nBlocks; i++) {
        block_t b = front.blocks[i];
        b.t += front.tPos;
        b.q += front.qPos;
        aln_push_block(alignment, b);
      }
      aln_free(&front);
    }
    /* :481-536 between consecutive blocks */
    for (int b = 0; b + 1 < ncb; b++) {
      aln_push_block(alignment, cb[b]);
      uint32_t qOff = cb[b].q + cb[b].len, tOff = cb[b].t + cb[b].len;
      uint32_t fq = cb[b + 1].q - qOff, ft = cb[b + 1].t - tOff;
      if (fq > 0 && ft > 0 && detailed) {
        aln_t frag;
        aln_init(&frag);
        if (fq * ft < (uint32_t)noRecurseUnder)
          sw_align_global(query + qOff, (int)fq, target + tOff, (int)ft, fn, &frag);
        else if (recurse != 0)
          sdp_align(query + qOff, fq, target + tOff, ft, fn, wordSize - 4 > 5 ? wordSize - 4 : 5,
                    sdpIns, sdpDel, indelRate, &frag, BO_GLOBAL, detailed, 0, 0, recurse - 1,
                    noRecurseUnder, 0);
        for (int i = 0; i < frag.nBlocks; i++) {
          block_t nb = frag.blocks[i];
          nb.q += qOff;
          nb.t += tOff;
          aln_push_block(alignment, nb);
        }
        aln_free(&frag);
      }
    }
    /* :537-600 last block; tail fill only when extendFront */
    if (alignType == BO_GLOBAL || alignType == BO_LOCAL) {
      int last = ncb - 1;
      aln_push_block(alignment, cb[last]);
      if (alignType == BO_GLOBAL || extendFront) {
        uint32_t qOff = cb[last].q + cb[last].len, tOff = cb[last].t + cb[last].len;
        uint32_t fq = qLen - qOff, ft = tLen - tOff;
        if (fq > 0 && ft > 0 && extendFront) {
          aln_t frag;
          aln_init(&frag);
          if (recurse == 0 && fq * ft < (uint32_t)noRecurseUnder)
            sw_align_global(query + qOff, (int)fq, target + tOff, (int)ft, fn, &frag);
          else if (recurse != 0)
            sdp_align(query + qOff, fq, target + tOff, ft, fn, wordSize / 2 > 5 ? wordSize / 2 : 5,
                      sdpIns, sdpDel, indelRate, &frag, BO_GLOBAL, detailed, extendFront,
                      sdpPrefixLength, recurse - 1, noRecurseUnder, maxMatchesPerPosition);
          for (int i = 0; i < frag.nBlocks; i++) {
            block_t nb = frag.blocks[i];
            nb.q += qOff;
            nb.t += tOff;
            aln_push_block(alignment, nb);
          }
          aln_free(&frag);
        }
      }
    }
  }
  free(cb);
  /* :602-610 */
  if (alignType == BO_LOCAL && alignment->nBlocks > 0) {
    alignment->tPos = alignment->blocks[0].t;
    alignment->qPos = alignment->blocks[0].q;
    for (int b = 0; b < alignment->nBlocks; b++) {
      alignment->blocks[b].q -= alignment->qPos;
      alignment->blocks[b].t -= alignment->tPos;
    }
  }
  /* :624 ComputeAlignmentScore(alignment, ...) (AlignmentUtils.h:127-169), no gap lists */
  int score = 0;
  for (int b = 0; b < alignment->nBlocks; b++) {
    uint32_t q = alignment->qPos + alignment->blocks[b].q, t = alignment->tPos + alignment->blocks[b].t;
    for (uint32_t l = 0; l < alignment->blocks[b].len; l++, q++, t++)
      score += match_score(fn, target[t], query[q]);
  }
  return score;
}

/* ------------------------------------------------------------------------- */
/* AlignmentToGuide (GuidedAlign.h:104-259), StoreMatrixOffsets (:94-101)     */
/* ------------------------------------------------------------------------- */
typedef struct {
  int q, t, tPre, tPost;
  unsigned int matrixOffset;
} guide_row;

static int imin(int a, int b) { return a < b ? a : b; }

static int alignment_to_guide(const block_t *blocks, int nBlocks, int bandSize, guide_row **out) {
  *out = NULL;
  if (nBlocks == 0) return 0;
  int tStart = (int)blocks[0].t, qStart = (int)blocks[0].q;
  int qEnd = (int)(blocks[nBlocks - 1].q + blocks[nBlocks - 1].len);
  int n = qEnd - qStart + 1;
  guide_row *g = (guide_row *)calloc(n, sizeof(guide_row));
  g[0].t = tStart - 1;
  g[0].q = qStart - 1;
  int drift = abs(tStart - qStart);
  g[0].tPost = drift > bandSize ? drift : bandSize;
  g[0].tPre = 0;
  int gi = 1;
  for (int b = 0; b < nBlocks; b++) {
    for (uint32_t bp = 0; bp < blocks[b].len; bp++) {
      g[gi].t = (int)(blocks[b].t + bp);
      g[gi].q = (int)(blocks[b].q + bp);
      int reach = g[gi].t - (g[gi - 1].t - g[gi - 1].tPre);
      if (bp == 0) {
        g[gi].tPre = reach;
        g[gi].tPost = bandSize + abs(drift);
      } else {
        g[gi].tPre = imin(bandSize, reach);
        g[gi].tPost = imin(MAX_BAND_SIZE, bandSize);
      }
      gi++;
    }
    if (b < nBlocks - 1) {
      int qGap = (int)(blocks[b + 1].q - (blocks[b].q + blocks[b].len));
      int tGap = (int)(blocks[b + 1].t - (blocks[b].t + blocks[b].len));
      /* ComputeDrift (AlignmentUtils.h:586-600) */
      {
        int tg = tGap, qg = qGap, common = 0;
        if (tg > 0 && qg > 0) common = abs(tg - qg);
        tg -= common;
        qg -= common;
        drift = tg - qg;
      }
      int diagonalLength = imin(qGap, tGap);
      int qPos = (int)(blocks[b].q + blocks[b].len), tPos = (int)(blocks[b].t + blocks[b].len);
      int qE = (int)blocks[b + 1].q;
      for (int d = 0; d < diagonalLength; d++, tPos++, qPos++) {
        g[gi].t = tPos;
        g[gi].q = qPos;
        g[gi].tPre = imin(MAX_BAND_SIZE, g[gi].t - (g[gi - 1].t - g[gi - 1].tPre));
        g[gi].tPost = imin(MAX_BAND_SIZE, bandSize + abs(drift));
        gi++;
      }
      while (qPos < qE) {
        g[gi].t = tPos;
        g[gi].q = qPos;
        qPos++;
        g[gi].tPre = imin(MAX_BAND_SIZE, g[gi].t - (g[gi - 1].t - g[gi - 1].tPre));
        g[gi].tPost = imin(MAX_BAND_SIZE, bandSize + abs(drift));
        gi++;
      }
    }
  }
  unsigned int cur = 0;
  for (int r = 0; r < n; r++) {
    g[r].matrixOffset = g[r].tPre + cur;
    cur += (unsigned)(g[r].tPre + g[r].tPost + 1);
  }
  *out = g;
  return n;
}

/* ------------------------------------------------------------------------- */
/* AffineGuidedAlign (AffineGuidedAlign.h:31-488)                             */
/* ------------------------------------------------------------------------- */
static int cell_index(const guide_row *g, int nRows, int row, int col) {
  /* GetBufferIndexFunctor (GuidedAlign.h:45-79) without the "index++"
   * shortcut: a cell exists iff col lies in the row's [t-tPre, t+tPost]. */
  if (row < 0 || row >= nRows) return -1;
  if (col <= g[row].t && g[row].t - col <= g[row].tPre) return (int)g[row].matrixOffset - (g[row].t - col);
  if (col > g[row].t && col - g[row].t <= g[row].tPost) return (int)g[row].matrixOffset + (col - g[row].t);
  return -1;
}

/* A probe for the design of a 16-bit fill: over all rows filled since the last reset, the
 * largest spread of S within a row (max - min over its cells), the number of rows, and how many had a spread of
 * 512 / 1024 or more.  Test infrastructure like the rest of this file. */
static long g_probe[4];
static int g_probe_on;
void bo_probe_row_spread(int enable, long *out4) {
  if (out4 != NULL) memcpy(out4, g_probe, sizeof g_probe);
  g_probe_on = enable;
  memset(g_probe, 0, sizeof g_probe);
}

static int affine_guided_align(const unsigned char *qSeq, const unsigned char *tSeq,
                               const block_t *guideBlocks, int nGuideBlocks, const score_fn *fn,
                               int bandSize, aln_t *alignment) {
  guide_row *g;
  int nRows = alignment_to_guide(guideBlocks, nGuideBlocks, bandSize, &g);
  if (nRows == 0) return 0;
  long nElem = 0;
  for (int r = 0; r < nRows; r++) nElem += g[r].tPre + g[r].tPost + 1;
  int *S = (int *)calloc(nElem, sizeof(int));
  int *I = (int *)malloc(nElem * sizeof(int));
  int *D = (int *)malloc(nElem * sizeof(int));
  unsigned char *PS = (unsigned char *)malloc(nElem), *PI = (unsigned char *)malloc(nElem),
                *PD = (unsigned char *)malloc(nElem);
  for (long i = 0; i < nElem; i++) {
    I[i] = D[i] = fn->affineOpen; /* :107-116, sic: open, not infinity */
    PS[i] = PI[i] = PD[i] = A_NONE;
  }
  int qStart = g[1].q, tStart = g[1].t;
  int qEnd = g[nRows - 1].q + 1, tEnd = g[nRows - 1].t + 1;
  /* :166-208 corner + deletion row */
  {
    int c0 = cell_index(g, nRows, 0, tStart - 1);
    S[c0] = 0;
    PS[c0] = A_NONE;
    for (int t = tStart; t < tStart + g[0].tPost; t++) {
      int cur = cell_index(g, nRows, 0, t), del = cell_index(g, nRows, 0, t - 1);
      if (del >= 0) {
        S[cur] = S[del] + fn->del;
        D[cur] = fn->affineOpen;
        PD[cur] = A_DEL_OPEN;
        I[cur] = fn->affineOpen;
        PI[cur] = A_INS_OPEN;
        PS[cur] = A_LEFT;
      }
    }
  }
  /* :213-235 (stripe initialisation) is overwritten by the main loop below with
   * values computed from the same predecessors, so it is not restated. */
  for (int q = qStart; q < qEnd; q++) {
    int qi = q - qStart + 1;
    int rowMin = INF_INT, rowMax = -INF_INT;
    for (int t = g[qi].t - g[qi].tPre; t <= g[qi].t + g[qi].tPost + 1; t++) {
      if (t == g[qi].t + g[qi].tPost + 1) { /* the row is done */
        if (g_probe_on && rowMin != INF_INT) {
          long spread = (long)rowMax - rowMin;
          if (spread > g_probe[0]) g_probe[0] = spread;
          g_probe[1]++;
          if (spread >= 512) g_probe[2]++;
          if (spread >= 1024) g_probe[3]++;
        }
        break;
      }
      if (t < -1) continue;
      if (t >= tEnd) continue;
      int mi = cell_index(g, nRows, qi - 1, t - 1), ii = cell_index(g, nRows, qi - 1, t),
          di = cell_index(g, nRows, qi, t - 1), cur = cell_index(g, nRows, qi, t);
      int matchScore = INF_INT, insScore = INF_INT, insExt = INF_INT, delScore = INF_INT, delExt = INF_INT;
      if (mi >= 0) matchScore = S[mi] + match_score(fn, tSeq[t], qSeq[q]);
      if (ii >= 0) {
        insScore = S[ii] + fn->ins;
        insExt = I[ii] + fn->affineExtend;
      }
      if (di >= 0) {
        delScore = S[di] + fn->del;
        delExt = D[di] + fn->affineExtend;
      }
      int mn = matchScore;
      if (insScore < mn) mn = insScore;
      if (delScore < mn) mn = delScore;
      if (insExt < mn) mn = insExt;
      if (delExt < mn) mn = delExt;
      S[cur] = mn;
      if (mn != INF_INT) {
        if (mn < rowMin) rowMin = mn;
        if (mn > rowMax) rowMax = mn;
      }
      if (mn == INF_INT)
        PS[cur] = A_NONE;
      else if (mn == matchScore)
        PS[cur] = A_DIAG;
      else if (mn == delScore)
        PS[cur] = A_LEFT;
      else if (mn == insScore)
        PS[cur] = A_UP;
      else if (mn == insExt)
        PS[cur] = A_INS_CLOSE;
      else
        PS[cur] = A_DEL_CLOSE;
      int openScore = (int)((unsigned)mn + (unsigned)fn->affineOpen);
      if (openScore < insExt) {
        PI[cur] = A_INS_OPEN;
        I[cur] = openScore;
      } else {
        PI[cur] = A_INS_UP;
        I[cur] = insExt;
      }
      if (openScore < delExt) {
        PD[cur] = A_DEL_OPEN;
        D[cur] = openScore;
      } else {
        PD[cur] = A_DEL_LEFT;
        D[cur] = delExt;
      }
    }
  }
  /* :377-468 traceback */
  int q = qEnd - 1, t = tEnd - 1, n = 0, ok = 1;
  enum { M_MATCH, M_INS, M_DEL } cur = M_MATCH;
  unsigned char *path = (unsigned char *)malloc((size_t)(qEnd - qStart) + (tEnd - tStart) + 4);
  while (q >= qStart || t >= tStart) {
    int idx = cell_index(g, nRows, q - qStart + 1, t);
    if (idx < 0) {
      ok = 0;
      break;
    }
    if (cur == M_MATCH) {
      unsigned char a = PS[idx];
      if (a == A_DIAG) {
        path[n++] = A_DIAG;
        q--;
        t--;
      } else if (a == A_UP) {
        path[n++] = A_UP;
        q--;
      } else if (a == A_LEFT) {
        path[n++] = A_LEFT;
        t--;
      } else if (a == A_INS_CLOSE) {
        path[n++] = A_UP;
        cur = M_INS;
        q--;
      } else if (a == A_DEL_CLOSE) {
        t--;
        path[n++] = A_LEFT;
        cur = M_DEL;
      } else {
        ok = 0;
        break;
      }
    } else if (cur == M_INS) {
      unsigned char a = PI[idx];
      if (a == A_INS_OPEN)
        cur = M_MATCH;
      else if (a == A_INS_UP) {
        q--;
        path[n++] = A_UP;
      } else {
        ok = 0;
        break;
      }
    } else {
      unsigned char a = PD[idx];
      if (a == A_DEL_OPEN)
        cur = M_MATCH;
      else if (a == A_DEL_LEFT) {
        t--;
        path[n++] = A_LEFT;
      } else {
        ok = 0;
        break;
      }
    }
  }
  for (int i = 0; i < n / 2; i++) {
    unsigned char tmp = path[i];
    path[i] = path[n - 1 - i];
    path[n - 1 - i] = tmp;
  }
  alignment->nCells = (int)nElem;
  alignment->qPos = qStart;
  alignment->tPos = tStart;
  arrow_path_to_alignment(alignment, path, n);
  remove_prefix_gaps(alignment);
  int last = cell_index(g, nRows, qEnd - 1 - qStart + 1, tEnd - 1);
  int score = last >= 0 ? S[last] : 0;
  alignment->score = score;
  free(path);
  free(S);
  free(I);
  free(D);
  free(PS);
  free(PI);
  free(PD);
  free(g);
  return ok ? score : INT_MIN;
}

/* ------------------------------------------------------------------------- */
/* flat-array entry points                                                    */
/* ------------------------------------------------------------------------- */
static int flatten(const aln_t *a, int *blocks, int blockCap, int *nBlocks, int *gapCounts,
                   int gapListCap, int *nGapLists, int *gapItems, int gapItemCap, int *nGapItems) {
  *nBlocks = a->nBlocks;
  if (a->nBlocks > blockCap) return -1;
  for (int b = 0; b < a->nBlocks; b++) {
    blocks[3 * b] = (int)a->blocks[b].q;
    blocks[3 * b + 1] = (int)a->blocks[b].t;
    blocks[3 * b + 2] = (int)a->blocks[b].len;
  }
  if (nGapLists) {
    *nGapLists = a->nGaps;
    if (a->nGaps > gapListCap) return -2;
    int n = 0;
    for (int g = 0; g < a->nGaps; g++) {
      gapCounts[g] = a->gaps[g].n;
      for (int i = 0; i < a->gaps[g].n; i++) {
        if (n >= gapItemCap) return -3;
        gapItems[2 * n] = a->gaps[g].g[i].seq;
        gapItems[2 * n + 1] = a->gaps[g].g[i].len;
        n++;
      }
    }
    *nGapItems = n;
  }
  return 0;
}

static void unflatten(aln_t *a, int qPos, int tPos, const int *blocks, int nBlocks,
                      const int *gapCounts, int nGapLists, const int *gapItems) {
  aln_init(a);
  a->qPos = qPos;
  a->tPos = tPos;
  for (int b = 0; b < nBlocks; b++) {
    block_t bl = {(uint32_t)blocks[3 * b], (uint32_t)blocks[3 * b + 1], (uint32_t)blocks[3 * b + 2]};
    aln_push_block(a, bl);
  }
  int n = 0;
  for (int g = 0; g < nGapLists; g++) {
    gaplist_t *gl = aln_push_gaplist(a);
    for (int i = 0; i < gapCounts[g]; i++) {
      gaplist_push(gl, gapItems[2 * n], gapItems[2 * n + 1]);
      n++;
      gl = &a->gaps[g];
    }
  }
}

int bo_sdp_align(const unsigned char *q, int qlen, const unsigned char *t, int tlen,
                 const int *matrix25, int ins, int del, int affOpen, int affExt, int wordSize,
                 int sdpIns, int sdpDel, float indelRate, int alignType, int detailed,
                 int extendFront, int sdpPrefixLength, int recurse, int noRecurseUnder,
                 int maxPerPos, int *outQPos, int *outTPos, int *blocks, int blockCap,
                 int *nBlocks) {
  score_fn fn;
  make_score_fn(&fn, matrix25, ins, del, affOpen, affExt);
  aln_t a;
  aln_init(&a);
  int score = sdp_align(q, (uint32_t)qlen, t, (uint32_t)tlen, &fn, wordSize, sdpIns, sdpDel,
                        indelRate, &a, alignType, detailed, extendFront, sdpPrefixLength, recurse,
                        noRecurseUnder, maxPerPos);
  *outQPos = (int)a.qPos;
  *outTPos = (int)a.tPos;
  int rc = flatten(&a, blocks, blockCap, nBlocks, 0, 0, 0, 0, 0, 0);
  aln_free(&a);
  return rc ? INT_MIN : score;
}

int bo_sdp_chain(int queryLength, const int *xyl, int nFrag, int fragmentLength, int ins, int del,
                 int match, int alignType, int *chain, int cap) {
  init_tables();
  frag_t *F = (frag_t *)calloc(nFrag > 0 ? nFrag : 1, sizeof(frag_t));
  for (int i = 0; i < nFrag; i++) {
    F[i].x = xyl[3 * i];
    F[i].y = xyl[3 * i + 1];
    F[i].length = xyl[3 * i + 2];
    F[i].weight = fragmentLength;
    F[i].above = -1;
  }
  int *c;
  int n = sdp_lcs((uint32_t)queryLength, F, nFrag, (uint32_t)fragmentLength, ins, del, match, alignType, &c);
  for (int i = 0; i < n && i < cap; i++) chain[i] = c[i];
  free(c);
  free(F);
  return n;
}

int bo_sw_align(const unsigned char *q, int qlen, const unsigned char *t, int tlen,
                const int *matrix25, int ins, int del, int alignType, int *outQPos, int *outTPos,
                int *blocks, int blockCap, int *nBlocks) {
  if (alignType != BO_GLOBAL) return INT_MIN;
  score_fn fn;
  make_score_fn(&fn, matrix25, ins, del, 0, 0);
  aln_t a;
  aln_init(&a);
  int score = sw_align_global(q, qlen, t, tlen, &fn, &a);
  *outQPos = (int)a.qPos;
  *outTPos = (int)a.tPos;
  int rc = flatten(&a, blocks, blockCap, nBlocks, 0, 0, 0, 0, 0, 0);
  aln_free(&a);
  return rc ? INT_MIN : score;
}

int bo_alignment_to_guide(const int *blocks, int nBlocks, int bandSize, int *rows, int rowCap,
                          int *nCells) {
  block_t *b = (block_t *)malloc(sizeof(block_t) * (nBlocks ? nBlocks : 1));
  for (int i = 0; i < nBlocks; i++) {
    b[i].q = blocks[3 * i];
    b[i].t = blocks[3 * i + 1];
    b[i].len = blocks[3 * i + 2];
  }
  guide_row *g;
  int n = alignment_to_guide(b, nBlocks, bandSize, &g);
  int cells = 0;
  for (int r = 0; r < n; r++) {
    cells += g[r].tPre + g[r].tPost + 1;
    if (r < rowCap) {
      rows[5 * r] = g[r].q;
      rows[5 * r + 1] = g[r].t;
      rows[5 * r + 2] = g[r].tPre;
      rows[5 * r + 3] = g[r].tPost;
      rows[5 * r + 4] = (int)g[r].matrixOffset;
    }
  }
  *nCells = cells;
  free(g);
  free(b);
  return n;
}

int bo_affine_guided_align(const unsigned char *q, int qlen, const unsigned char *t, int tlen,
                           const int *guideBlocks, int nGuideBlocks, const int *matrix25, int ins,
                           int del, int affOpen, int affExt, int bandSize, int *outQPos,
                           int *outTPos, int *blocks, int blockCap, int *nBlocks, int *gapCounts,
                           int gapListCap, int *nGapLists, int *gapItems, int gapItemCap,
                           int *nGapItems, int *stats, float *pct) {
  (void)qlen;
  (void)tlen;
  score_fn fn;
  make_score_fn(&fn, matrix25, ins, del, affOpen, affExt);
  block_t *gb = (block_t *)malloc(sizeof(block_t) * (nGuideBlocks ? nGuideBlocks : 1));
  for (int i = 0; i < nGuideBlocks; i++) {
    gb[i].q = guideBlocks[3 * i];
    gb[i].t = guideBlocks[3 * i + 1];
    gb[i].len = guideBlocks[3 * i + 2];
  }
  aln_t a;
  aln_init(&a);
  int dp = affine_guided_align(q, t, gb, nGuideBlocks, &fn, bandSize, &a);
  free(gb);
  stats[0] = dp;
  stats[1] = a.nCells;
  compute_alignment_stats(&a, q, t, &fn, 1);
  stats[2] = a.score;
  stats[3] = a.nMatch;
  stats[4] = a.nMismatch;
  stats[5] = a.nIns;
  stats[6] = a.nDel;
  *pct = a.pct;
  *outQPos = (int)a.qPos;
  *outTPos = (int)a.tPos;
  int rc = flatten(&a, blocks, blockCap, nBlocks, gapCounts, gapListCap, nGapLists, gapItems, gapItemCap, nGapItems);
  aln_free(&a);
  return rc;
}

void bo_compute_alignment_stats(const unsigned char *q, const unsigned char *t, int qPos, int tPos,
                                const int *blocks, int nBlocks, const int *gapCounts,
                                int nGapLists, const int *gapItems, const int *matrix25, int ins,
                                int del, int affOpen, int affExt, int useAffine, int *stats,
                                float *pct) {
  score_fn fn;
  make_score_fn(&fn, matrix25, ins, del, affOpen, affExt);
  aln_t a;
  unflatten(&a, qPos, tPos, blocks, nBlocks, gapCounts, nGapLists, gapItems);
  compute_alignment_stats(&a, q, t, &fn, useAffine);
  stats[0] = a.score;
  stats[1] = a.nMatch;
  stats[2] = a.nMismatch;
  stats[3] = a.nIns;
  stats[4] = a.nDel;
  *pct = a.pct;
  aln_free(&a);
}

/* AlignIntervals' SDP step (Blasr.cpp:1716-1724) followed by RefineAlignment's
 * guided branch (Blasr.cpp:850-914) for one candidate window. */
int bo_refine_candidate(const unsigned char *q, int qlen, const unsigned char *t, int tlen,
                        const int *matrix25, int ins, int del, int affOpen, int affExt,
                        int wordSize, int sdpIns, int sdpDel, float indelRate, int sdpPrefixLength,
                        int recurse, int noRecurseUnder, int bandSize, int *out, float *pct,
                        int *outQPos, int *outTPos, int *blocks, int blockCap, int *nBlocks,
                        int *gapCounts, int gapListCap, int *nGapLists, int *gapItems,
                        int gapItemCap, int *nGapItems) {
  score_fn fn;
  make_score_fn(&fn, matrix25, ins, del, affOpen, affExt);
  aln_t sdp;
  aln_init(&sdp);
  out[0] = sdp_align(q, (uint32_t)qlen, t, (uint32_t)tlen, &fn, wordSize, sdpIns, sdpDel, indelRate,
                     &sdp, BO_LOCAL, 1, 0, sdpPrefixLength, recurse, noRecurseUnder, 0);
  compute_alignment_stats(&sdp, q, t, &fn, 1);
  out[1] = sdp.score;
  out[2] = (int)sdp.qPos;
  out[3] = (int)sdp.tPos;
  out[4] = sdp.nBlocks;
  *nBlocks = 0;
  if (nGapLists) *nGapLists = 0;
  if (nGapItems) *nGapItems = 0;
  for (int i = 5; i < 12; i++) out[i] = 0;
  *pct = 0;
  *outQPos = (int)sdp.qPos;
  *outTPos = (int)sdp.tPos;
  int rc = 0;
  if (sdp.nBlocks > 0) {
    /* Blasr.cpp:850-859: sequences trimmed to the SDP extent */
    const unsigned char *ts = t + sdp.tPos, *qs = q + sdp.qPos;
    aln_t ref;
    aln_init(&ref);
    out[5] = affine_guided_align(qs, ts, sdp.blocks, sdp.nBlocks, &fn, bandSize, &ref);
    out[6] = ref.nCells;
    compute_alignment_stats(&ref, qs, ts, &fn, 1);
    out[7] = ref.score;
    out[8] = ref.nMatch;
    out[9] = ref.nMismatch;
    out[10] = ref.nIns;
    out[11] = ref.nDel;
    *pct = ref.pct;
    /* Blasr.cpp:905-914: window-relative start of the refined alignment */
    *outQPos = (int)(sdp.qPos + ref.qPos);
    *outTPos = (int)(sdp.tPos + ref.tPos);
    rc = flatten(&ref, blocks, blockCap, nBlocks, gapCounts, gapListCap, nGapLists, gapItems, gapItemCap, nGapItems);
    aln_free(&ref);
  }
  aln_free(&sdp);
  return rc;
}

/* ------------------------------------------------------------------------- */
/* a-1: window / strand set-up of AlignIntervals (alignment/Blasr.cpp:1208-1336) */
/* ------------------------------------------------------------------------- */
/* out[8]: tAlignedSeqPos (on the strand of the interval), tAlignedSeqLength, tStrand, tIndex,
 * tLength, qAlignedSeqPos, qAlignedSeqLength, first base of the window on the FORWARD genome */
void bo_interval_window(unsigned genomeLen, unsigned readLen, unsigned subreadStart, unsigned subreadEnd,
                        unsigned ivStart, unsigned ivEnd, unsigned ivQStart, unsigned ivQEnd, int strand,
                        const unsigned *contigStarts, int nContigs, float approximateMaxInsertionRate,
                        int extendEnds, int doGlobalAlignment, int *out) {
  unsigned matchIntervalStart = ivStart, matchIntervalEnd = ivEnd; /* :1208-1210 */
  /* :1243-1248 the subread on the strand of the interval */
  unsigned sStart = subreadStart, sEnd = subreadEnd;
  if (strand == 1) {
    sEnd = (readLen - subreadStart - 1) + 1;
    sStart = readLen - (subreadEnd - 1) - 1;
  }
  if (extendEnds) { /* :1250-1266, float arithmetic exactly as written there */
    unsigned lengthBeforeFirstMatch = (unsigned)((ivQStart - sStart) * approximateMaxInsertionRate);
    unsigned lengthAfterLastMatch = (unsigned)((sEnd - ivQEnd) * approximateMaxInsertionRate);
    if (matchIntervalStart < lengthBeforeFirstMatch || doGlobalAlignment) matchIntervalStart = 0;
    else matchIntervalStart -= lengthBeforeFirstMatch;
    if (genomeLen < matchIntervalEnd + lengthAfterLastMatch || doGlobalAlignment) matchIntervalEnd = genomeLen;
    else matchIntervalEnd += lengthAfterLastMatch;
  }
  int tIndex = 0;
  unsigned tLength;
  if (contigStarts != NULL) { /* :1272-1298 */
    int nSeqPos = nContigs + 1, idx = 0;
    if (nSeqPos != 1) { /* SequenceIndexDatabase.h:108-118: upper_bound over seqStartPos[1..] */
      int lo = 1, hi = nSeqPos;
      while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (!(ivStart < contigStarts[mid])) lo = mid + 1;
        else hi = mid;
      }
      idx = lo - 1;
    }
    unsigned contigStartPos = contigStarts[idx];
    if (contigStartPos > matchIntervalStart) matchIntervalStart = contigStartPos;
    unsigned contigEndPos = contigStarts[idx + 1] - 1;
    if (contigEndPos < matchIntervalEnd) matchIntervalEnd = contigEndPos;
    tLength = contigEndPos - contigStartPos;
    tIndex = idx;
  } else
    tLength = genomeLen; /* :1300-1304 */
  unsigned tAlignedSeqPos = matchIntervalStart, tAlignedSeqLength = matchIntervalEnd - matchIntervalStart; /* :1309-1310 */
  out[7] = (int)tAlignedSeqPos;
  if (strand == 1) /* :1316-1324: the window is copied as its reverse complement and re-addressed on that strand */
    tAlignedSeqPos = genomeLen - (tAlignedSeqPos + tAlignedSeqLength - 1) - 1;
  out[0] = (int)tAlignedSeqPos;
  out[1] = (int)tAlignedSeqLength;
  out[2] = strand == 1 ? 1 : 0;
  out[3] = tIndex;
  out[4] = (int)tLength;
  out[5] = (int)subreadStart;          /* :1332 always the forward read */
  out[6] = (int)(sEnd - sStart);       /* :1333-1334 */
}

/* DNASequence::Copy / CopyAsRC (DNASequence.h:219-233) with ReverseComplementNuc (NucConversion.h:337-370) */
static unsigned char bo_rc_nuc(unsigned char c) {
  if (c < 4) return (unsigned char)(3 - c);
  switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    case 'N': return 'N';
    case 'a': return 't';
    case 'c': return 'g';
    case 'g': return 'c';
    case 't': return 'a';
    case 'n': return 'n';
    default: return 127;
  }
}
void bo_copy_window(const unsigned char *genome, unsigned fwdStart, unsigned len, int strand, unsigned char *out) {
  for (unsigned i = 0; i < len; i++)
    out[i] = strand == 1 ? bo_rc_nuc(genome[fwdStart + len - 1 - i]) : genome[fwdStart + i];
}

/* ------------------------------------------------------------------------- */
/* a-10: candidate ordering and pruning (alignment/Blasr.cpp:226-236, 2022-2161)  */
/* ------------------------------------------------------------------------- */
/* std::sort(alignmentPtrs, SortAlignmentPointersByScore): (score, tPos + tAlignedSeqPos); order[] in/out */
void bo_sort_by_score(int n, const int *score, const unsigned *tPos, const unsigned *tAlignedSeqPos, int *order) {
  frag_t *f = (frag_t *)calloc(n > 0 ? n : 1, sizeof(frag_t));
  for (int i = 0; i < n; i++) {
    const int c = order[i];
    f[i].x = (uint32_t)score[c] ^ 0x80000000u; /* order-preserving image of the signed score */
    f[i].y = tPos[c] + tAlignedSeqPos[c];
    f[i].index = c;
  }
  introsort_frags(f, n);
  for (int i = 0; i < n; i++) order[i] = f[i].index;
  free(f);
}

/* RemoveLowQualityAlignments (:2022-2079); score[] may be modified (maxGap flagging); returns the new count */
int bo_remove_low_quality(int n, int *order, int *score, const int *blocks, const int *blkOff, const int *nBlocks,
                          int nCandidates, int maxGap, int maxScore) {
  const int lim = nCandidates < n ? nCandidates : n;
  for (int i = 0; i < lim; i++) {
    const int c = order[i];
    if (maxGap > 0) {
      for (int j = 1; j < nBlocks[c]; j++) {
        const int *b = blocks + 3 * (blkOff[c] + j), *p = b - 3;
        const int gq = (int)((unsigned)b[0] - (unsigned)p[0] - (unsigned)b[2]);
        const int gt = (int)((unsigned)b[1] - (unsigned)p[1] - (unsigned)b[2]);
        if ((gq >= maxGap && gt < maxGap) || (gt > maxGap && gq < maxGap)) {
          score[c] = maxScore;
          break;
        }
      }
    }
    if (nBlocks[c] == 0 || score[c] > maxScore) return i; /* this and everything after it goes */
  }
  return n;
}

/* RemoveOverlappingAlignments (:2081-2161); tEnd = Alignment::TEnd() of every candidate */
int bo_remove_overlapping(int n, int *order, const int *score, const unsigned *tPos, const unsigned *tAlignedSeqPos,
                          const unsigned *tEnd, const int *tIndex, const int *tStrand, const float *pctSimilarity,
                          float minPctIdentity) {
  unsigned char *contained = (unsigned char *)calloc(n > 0 ? n : 1, 1);
  if (n > 0) {
    for (int i = 0; i < n - 1; i++) {
      const int a = order[i];
      if (pctSimilarity[a] < minPctIdentity) continue;
      const unsigned aB = tAlignedSeqPos[a] + tPos[a], aE = aB + tEnd[a];
      for (int j = i + 1; j < n; j++) {
        if (contained[j]) continue;
        const int b = order[j];
        if (tIndex[a] != tIndex[b]) continue;
        const unsigned bB = tAlignedSeqPos[b] + tPos[b], bE = bB + tEnd[b];
        if (aB <= bB && aE >= bE && tStrand[a] == tStrand[b]) {
          if (score[a] <= score[b]) contained[j] = 1;
        } else if (bB <= aB && bE >= aE && tStrand[a] == tStrand[b]) {
          if (score[b] <= score[a]) contained[i] = 1;
        }
      }
    }
  }
  int kept = 0;
  for (int i = 0; i < n; i++)
    if (!contained[i]) order[kept++] = order[i];
  free(contained);
  return kept;
}

/* ------------------------------------------------------------------------- */
/* a-11: AffineKBandAlign, Global (algorithms/alignment/AffineKBandAlign.h:12-401) */
/* ------------------------------------------------------------------------- */
/* S + affine insertion I + affine homopolymer insertion HP, linear deletion, fixed band of 2k+1
 * columns stored as column = k + t - q.  Cells outside the band / outside the sequences are never read. */
#define KB_INF (2147483647 - 1000) /* INF_INT - 1000, :30 */
static int three_bit(unsigned char c) { /* ThreeBit on the supported alphabet (ACGTN, both cases); else N */
  init_tables();
  const int v = g_three[c];
  return v > 4 ? 4 : v;
}
enum { KB_DIAG, KB_LEFT, KB_INS_CLOSE, KB_HP_CLOSE, KB_NONE };
static int affine_kband_align(const unsigned char *qs, uint32_t qLength, const unsigned char *ts, uint32_t tLength,
                              const int m[5][5], int hpInsOpen, int hpInsExtend, int insOpen, int insExtend,
                              int del, int k, aln_t *out) {
  uint32_t tLen, qLen; /* SetKBoundedLengths, KBandAlign.h:36-56 */
  if (tLength < qLength) {
    tLen = tLength;
    qLen = qLength < tLength + (uint32_t)k ? qLength : tLength + (uint32_t)k;
  } else if (qLength < tLength) {
    qLen = qLength;
    tLen = tLength < qLength + (uint32_t)k ? tLength : qLength + (uint32_t)k;
  } else {
    qLen = qLength;
    tLen = tLength;
  }
  const size_t nCols = 2 * (size_t)k + 1, total = ((size_t)qLen + 1) * nCols;
  int *S = (int *)calloc(total, sizeof(int)), *H = (int *)calloc(total, sizeof(int)), *I = (int *)calloc(total, sizeof(int));
  unsigned char *PS = (unsigned char *)malloc(total), *PH = (unsigned char *)calloc(total, 1), *PI = (unsigned char *)calloc(total, 1);
  memset(PS, KB_NONE, total);
#define IX(r, c) ((size_t)(r) * nCols + (size_t)(c))
  /* boundaries :96-150 (alignType Global); PH / PI: 1 = Open, 0 = Up */
  I[IX(0, k)] = 0;
  PI[IX(0, k)] = 1;
  for (int q = 1; q <= k && q < (int)qLen + 1; q++) I[IX(q, k - q)] = q * insExtend + insOpen;
  H[IX(0, k)] = 0;
  PH[IX(0, k)] = 1;
  for (int q = 1; q <= k && q < (int)qLen + 1; q++) H[IX(q, k - q)] = q * hpInsExtend + hpInsOpen;
  for (int t = k + 1; t < (int)nCols; t++) {
    H[IX(0, t)] = KB_INF;
    I[IX(0, t)] = KB_INF;
  }
  for (int q = 1; q <= k && q < (int)qLen + 1; q++) {
    S[IX(q, k - q)] = I[IX(q, k - q)];
    PS[IX(q, k - q)] = KB_INS_CLOSE;
  }
  for (int t = 1; t <= k; t++) {
    S[IX(0, t + k)] = t * del;
    PS[IX(0, t + k)] = KB_LEFT;
  }
  /* recurrence :161-279 */
  for (int q = 1; q <= (int)qLen; q++) {
    for (int t = q - k; t < q + k + 1; t++) {
      if (t < 1) continue;
      if ((uint32_t)t > tLen) break;
      const size_t upper = IX(q - 1, k + t - q + 1), cur = IX(q, k + t - q);
      const int hpOpenScore = t < q + k ? S[upper] + hpInsOpen : KB_INF;
      int hpExtendScore = KB_INF;
      if (q > 1 && qs[q - 1] == qs[q - 2] && t < q + k) hpExtendScore = H[upper] + hpInsExtend;
      int minHp;
      if (hpOpenScore < hpExtendScore) {
        PH[cur] = 1;
        minHp = hpOpenScore;
      } else {
        PH[cur] = 0;
        minHp = hpExtendScore;
      }
      H[cur] = minHp;
      int insOpenScore = KB_INF, insExtendScore = KB_INF;
      if (t < q + k) {
        insOpenScore = S[upper] + insOpen;
        insExtendScore = I[upper] + insExtend;
      }
      int minIns;
      if (insOpenScore < insExtendScore) {
        PI[cur] = 1;
        minIns = insOpenScore;
      } else {
        PI[cur] = 0;
        minIns = insExtendScore;
      }
      I[cur] = minIns;
      const int delScore = t == q - k ? KB_INF : S[IX(q, k + t - q - 1)] + del;
      const int matchScore = S[IX(q - 1, k + t - q)] + m[three_bit(qs[q - 1])][three_bit(ts[t - 1])];
      int minScore = minIns < minHp ? minIns : minHp;
      if (delScore < minScore) minScore = delScore;
      if (matchScore < minScore) minScore = matchScore;
      S[cur] = minScore;
      PS[cur] = minScore == matchScore ? KB_DIAG : minScore == delScore ? KB_LEFT : minScore == minIns ? KB_INS_CLOSE : KB_HP_CLOSE;
    }
  }
  /* traceback :299-391, Global */
  int q = (int)qLen, t = k - ((int)qLen - (int)tLen);
  const int optScore = S[IX(q, t)];
  unsigned char *path = (unsigned char *)malloc((size_t)qLen + tLen + 2 * (size_t)k + 16);
  int n = 0, mat = 0; /* 0 Match, 1 AffineIns, 2 AffineHPIns */
  while (q > 0 || (q == 0 && t > k)) {
    if (mat == 0) {
      const int a = PS[IX(q, t)];
      if (a == KB_DIAG) {
        path[n++] = A_DIAG;
        q--;
      } else if (a == KB_LEFT) {
        path[n++] = A_LEFT;
        t--;
      } else if (a == KB_INS_CLOSE) mat = 1;
      else if (a == KB_HP_CLOSE) mat = 2;
      else break; /* NoArrow: the reference would spin forever; cannot happen for a computed cell */
    } else {
      const int open = mat == 1 ? PI[IX(q, t)] : PH[IX(q, t)];
      if (open) mat = 0;
      path[n++] = A_UP;
      q--;
      t++;
    }
  }
  for (int i = 0; i < n / 2; i++) {
    unsigned char tmp = path[i];
    path[i] = path[n - 1 - i];
    path[n - 1 - i] = tmp;
  }
  arrow_path_to_alignment(out, path, n);
#undef IX
  free(path);
  free(S);
  free(H);
  free(I);
  free(PS);
  free(PH);
  free(PI);
  return optScore;
}

int bo_affine_kband_align(const unsigned char *q, int qlen, const unsigned char *t, int tlen, const int *matrix25,
                          int hpInsOpen, int hpInsExtend, int insOpen, int insExtend, int del, int k, int *score,
                          int *blocks, int blockCap, int *nBlocks, int *gapCounts, int gapListCap, int *nGapLists,
                          int *gapItems, int gapItemCap, int *nGapItems) {
  int m[5][5];
  for (int i = 0; i < 5; i++)
    for (int j = 0; j < 5; j++) m[i][j] = matrix25[i * 5 + j];
  aln_t a;
  aln_init(&a);
  *score = affine_kband_align(q, (uint32_t)qlen, t, (uint32_t)tlen, m, hpInsOpen, hpInsExtend, insOpen, insExtend, del,
                              k, &a);
  int rc = flatten(&a, blocks, blockCap, nBlocks, gapCounts, gapListCap, nGapLists, gapItems, gapItemCap, nGapItems);
  aln_free(&a);
  return rc;
}

/* ------------------------------------------------------------------------- */
/* a-11: RefineAlignment's branch under -noUseGuidedAlign (alignment/Blasr.cpp:917-1046)               */
/* ------------------------------------------------------------------------- */
/* ComputeDrift (AlignmentUtils.h:586-618): largest absolute running sum of (tGap - qGap) between
 * consecutive blocks (the "common gap" the two-block form subtracts from both cancels, :593-599). */
int bo_compute_drift(const int *blocks, int nBlocks) {
  int drift = 0, maxDrift = 0;
  for (int b = 0; b + 1 < nBlocks; b++) {
    const int *cur = blocks + 3 * b, *next = cur + 3;
    int tGap = (int)((uint32_t)next[1] - ((uint32_t)cur[1] + (uint32_t)cur[2]));
    int qGap = (int)((uint32_t)next[0] - ((uint32_t)cur[0] + (uint32_t)cur[2]));
    int common = 0;
    if (tGap > 0 && qGap > 0) common = abs(tGap - qGap);
    tGap -= common;
    qGap -= common;
    drift += tGap - qGap;
    if (abs(drift) > maxDrift) maxDrift = abs(drift);
  }
  return maxDrift;
}

/* The candidate of one window after the SDP step (q, t = the window's query / target, the SDP alignment's
 * qPos / tPos / blocks) through RefineAlignment with useGuidedAlign == false, qIsSubstring == true (:968-972):
 *   sequences trimmed to the SDP extent (:968-981), k = bandSize or |drift| * 1.5, at least 10 (:990-1003),
 *   PairwiseLocalAlign (:675-746): AffineKBandAlign(SMRTDistanceMatrix, indel + 2, indel - 3, indel + 2, indel - 1,
 *   indel, k * 1.2, Global), ComputeAlignmentStats(alignment, q, t, distScoreFn) in its non-affine form.
 * The reference's distScoreFn has no score matrix at that point (:689-690, SURVEY a-11): its printed score is
 * uninitialised memory.  stats[0] here is what the same formula gives with matrix25 in place; everything else the
 * reference leaves in the candidate is well defined and returned as it is there.
 * out[0..7] = drift, k (before the factor), k handed to AffineKBandAlign, its return value, nMatch, nMismatch, nIns, nDel;
 * the candidate's qPos / tPos are 0 afterwards (:1019, :1022), its sequences the trimmed ones. */
int bo_refine_nonguided(const unsigned char *q, const unsigned char *t, int sdpQPos, int sdpTPos, const int *sdpBlocks,
                        int nSdpBlocks, const int *matrix25, int ins, int del, int indel, int bandSize, int *out,
                        int *statsScore, float *pct, int *qTrimLen, int *tTrimLen, int *blocks, int blockCap,
                        int *nBlocks, int *gapCounts, int gapListCap, int *nGapLists, int *gapItems, int gapItemCap,
                        int *nGapItems) {
  for (int i = 0; i < 8; i++) out[i] = 0;
  *statsScore = 0;
  *pct = 0;
  *qTrimLen = *tTrimLen = 0;
  *nBlocks = 0;
  if (nGapLists) *nGapLists = 0;
  if (nGapItems) *nGapItems = 0;
  const int drift = bo_compute_drift(sdpBlocks, nSdpBlocks);
  out[0] = drift;
  if (nSdpBlocks == 0) return 0; /* :959-962: score = 0, nothing else changes */
  const int *last = sdpBlocks + 3 * (nSdpBlocks - 1);
  const int qLen = last[0] + last[2];                 /* :968-972 (qIsSubstring) */
  const int tLen = last[1] + last[2] - sdpBlocks[1];  /* :979-981 */
  const unsigned char *qs = q + sdpQPos, *ts = t + sdpTPos;
  int k;
  if (bandSize == 0) k = (int)(abs(drift) * 1.5); /* :991-996 */
  else k = bandSize;
  if (k < 10) k = 10; /* :1001-1003 */
  out[1] = k;
  const int kk = (int)(k * 1.2); /* :699: the double product converted to AffineKBandAlign's int parameter */
  out[2] = kk;
  int m[5][5];
  for (int i = 0; i < 5; i++)
    for (int j = 0; j < 5; j++) m[i][j] = matrix25[i * 5 + j];
  aln_t a;
  aln_init(&a);
  out[3] = affine_kband_align(qs, (uint32_t)qLen, ts, (uint32_t)tLen, m, indel + 2, indel - 3, indel + 2, indel - 1, indel,
                              kk, &a);
  score_fn fn;
  make_score_fn(&fn, matrix25, ins, del, 0, 0);
  compute_alignment_stats(&a, qs, ts, &fn, 0); /* :740 */
  *statsScore = a.score;
  out[4] = a.nMatch;
  out[5] = a.nMismatch;
  out[6] = a.nIns;
  out[7] = a.nDel;
  *pct = a.pct;
  *qTrimLen = qLen;
  *tTrimLen = tLen;
  int rc = flatten(&a, blocks, blockCap, nBlocks, gapCounts, gapListCap, nGapLists, gapItems, gapItemCap, nGapItems);
  aln_free(&a);
  return rc;
}

/* ------------------------------------------------------------------------- */
/* a-11: KBandAlign with a DistanceMatrixScoreFunction (algorithms/alignment/KBandAlign.h:74-403) */
/* ------------------------------------------------------------------------- */
/* AlignmentType (AlignmentUtils.h:14-58): Global 1, QueryFit 2, TargetFit 3, ..., Fit 7.
 * ins/del: boundary penalties; fnIns/fnDel: scoreFn.Insertion / Deletion inside the recurrence;
 * scoreFn.Match = matrix[ThreeBit[ref]][ThreeBit[query]] (DistanceMatrixScoreFunction.h:43-45).
 * Cells the loops skip keep the 0 / NoArrow the matrices are filled with, exactly as in the reference. */
enum { KT_GLOBAL = 1, KT_QUERYFIT = 2, KT_TARGETFIT = 3, KT_FIT = 7 };
static int kband_align(const unsigned char *qs, uint32_t qLength, const unsigned char *ts, uint32_t tLength,
                       const int m[5][5], int ins, int del, int fnIns, int fnDel, int k, int alignType, aln_t *out,
                       int *nCellsOut) {
  uint32_t tLen, qLen; /* SetKBoundedLengths, :36-56 */
  if (tLength < qLength) {
    tLen = tLength;
    qLen = qLength < tLength + (uint32_t)k ? qLength : tLength + (uint32_t)k;
  } else if (qLength < tLength) {
    qLen = qLength;
    tLen = tLength < qLength + (uint32_t)k ? tLength : qLength + (uint32_t)k;
  } else {
    qLen = qLength;
    tLen = tLength;
  }
  const size_t nCols = 2 * (size_t)k + 1, total = ((size_t)qLen + 1) * nCols;
  *nCellsOut = (int)total;
  int *S = (int *)calloc(total, sizeof(int));
  unsigned char *P = (unsigned char *)malloc(total);
  memset(P, A_NONE, total);
#define IX(r, c) ((size_t)(r) * nCols + (size_t)(c))
  int q, t;
  for (q = 1; q <= k && (uint32_t)q < qLen + 1; q++) { /* :113-116 */
    S[IX(q, k - q)] = q * ins;
    P[IX(q, k - q)] = A_UP;
  }
  if (alignType == KT_GLOBAL)
    for (t = 1; t <= k && (uint32_t)t < tLen; t++) { /* :117-122, note t < tLen */
      S[IX(0, t + k)] = t * del;
      P[IX(0, t + k)] = A_LEFT;
    }
  if (alignType == KT_QUERYFIT || alignType == KT_FIT)
    for (t = 1; (t <= k) & ((uint32_t)t < tLen); t++) {
      S[IX(0, t + k)] = 0;
      P[IX(0, t + k)] = A_LEFT;
    }
  if (alignType == KT_TARGETFIT || alignType == KT_FIT)
    for (q = 1; (q <= k) & ((uint32_t)q < qLen); q++) { /* :129-134: column 0, as written there */
      S[IX(q, 0)] = 0;
      P[IX(q, 0)] = A_UP;
    }
  S[IX(0, k)] = 0;
  P[IX(0, k)] = A_DIAG;
  for (q = 1; (uint32_t)q <= qLen; q++) { /* :144-222, samplePaths = false */
    for (t = q - k; t < q + k + 1; t++) {
      if (t < 1) continue;
      if ((uint32_t)t > tLen) continue;
      const int delScore = t == q - k ? 2147483647 : S[IX(q, k + t - q - 1)] + fnDel;
      const int matchScore = S[IX(q - 1, k + t - q)] + m[three_bit(ts[t - 1])][three_bit(qs[q - 1])];
      const int insScore = t == q + k ? 2147483647 : S[IX(q - 1, k + t - q + 1)] + fnIns;
      int minScore = insScore < delScore ? insScore : delScore;
      if (matchScore < minScore) minScore = matchScore;
      const size_t cur = IX(q, k + t - q);
      S[cur] = minScore;
      P[cur] = minScore == matchScore ? A_DIAG : (minScore == delScore ? A_LEFT : A_UP);
    }
  }
  /* end cell :228-296 */
  q = (int)qLen;
  t = k - (int)(qLen - tLen);
  const int globalMinScore = S[IX(q, t)];
  int minLastColScore = globalMinScore, minLastRowScore = globalMinScore;
  int minLastColScoreIndex = 0, minLastRowScoreIndex = 0;
  if (alignType == KT_QUERYFIT || alignType == KT_FIT) {
    int q2 = (int)qLen, t2, minScoreSet = 0;
    for (t2 = q - k; t2 < q2 + k + 1; t2++) {
      if (t2 < 1) continue;
      if ((uint32_t)t2 > tLen) continue;
      if (!minScoreSet || S[IX(q2, k + t2 - q)] < minLastRowScore) {
        minScoreSet = 1;
        minLastRowScore = S[IX(q2, k + t2 - q)];
        minLastRowScoreIndex = t2;
      }
    }
    if (minScoreSet) {
      t2 = k - (q - minLastRowScoreIndex);
      t = t2;
      q = q2;
    }
  }
  if (alignType == KT_TARGETFIT || alignType == KT_FIT) {
    int q2, t2 = k - (int)(qLen - tLen), minScoreSet = 0;
    for (q2 = (int)qLen; (uint32_t)q2 >= tLen - (uint32_t)k && q2 > 0; q2--) { /* unsigned compare, :279 */
      const size_t index = IX(q2, k + (int)tLen - q2);
      if (!minScoreSet || S[index] < minLastColScore) {
        minLastColScore = S[index];
        minScoreSet = 1;
        minLastColScoreIndex = q2;
      }
    }
    if (alignType == KT_FIT) {
      if (minLastColScore < minLastRowScore) {
        t = t2;
        q = minLastColScoreIndex;
      }
    } else {
      t = t2;
      q = minLastColScoreIndex;
    }
  }
  const int optScore = S[IX(q, t)];
  /* traceback :317-381 */
  unsigned char *path = (unsigned char *)malloc((size_t)qLen + tLen + 2 * (size_t)k + 16);
  int n = 0;
  for (;;) {
    int go;
    if (alignType == KT_GLOBAL || alignType == KT_QUERYFIT) go = q > 0 && (t < k ? (k - t != q) : 1);
    else if (alignType == KT_FIT) go = q > 0 && (t < k ? (k - t != q) : 1) && (q <= k ? k - q != t : 1);
    else go = q > 0 && (q < k ? k - q != t : 1);
    if (!go) break;
    const int a = P[IX(q, t)];
    if (a == A_NONE) break;
    path[n++] = (unsigned char)a;
    if (a == A_DIAG) q--;
    else if (a == A_UP) {
      q--;
      t++;
    } else
      t--;
  }
  out->qPos = (uint32_t)q;
  out->tPos = t < k ? (uint32_t)((k - t) - q) : (uint32_t)((t - k) - q);
  for (int i = 0; i < n / 2; i++) {
    unsigned char tmp = path[i];
    path[i] = path[n - 1 - i];
    path[n - 1 - i] = tmp;
  }
  arrow_path_to_alignment(out, path, n);
#undef IX
  free(path);
  free(S);
  free(P);
  return optScore;
}

int bo_kband_align(const unsigned char *q, int qlen, const unsigned char *t, int tlen, const int *matrix25, int ins,
                   int del, int fnIns, int fnDel, int k, int alignType, int *out, int *blocks, int blockCap,
                   int *nBlocks, int *gapCounts, int gapListCap, int *nGapLists, int *gapItems, int gapItemCap,
                   int *nGapItems) {
  int m[5][5];
  for (int i = 0; i < 5; i++)
    for (int j = 0; j < 5; j++) m[i][j] = matrix25[i * 5 + j];
  aln_t a;
  aln_init(&a);
  int nCells = 0;
  out[0] = kband_align(q, (uint32_t)qlen, t, (uint32_t)tlen, m, ins, del, fnIns, fnDel, k, alignType, &a, &nCells);
  out[1] = (int)a.qPos;
  out[2] = (int)a.tPos;
  int rc = flatten(&a, blocks, blockCap, nBlocks, gapCounts, gapListCap, nGapLists, gapItems, gapItemCap, nGapItems);
  aln_free(&a);
  return rc;
}

/* ---------------------------------------------------------------------------------------------------------
 * next-3 (first piece): ExtendAlignment (algorithms/alignment/ExtendAlign.h:115-363) behind its two entry points
 * ExtendAlignmentForward (:365-405) and ExtendAlignmentReverse (:407-445), the -extend step of AlignIntervals
 * (alignment/Blasr.cpp:1749-1940).  Band of 2k+1 diagonals around the main one, linear gaps, rows added until
 * maxNDrops rows have raised the row minimum; the alignment ends at the first cell holding the global minimum.
 * --------------------------------------------------------------------------------------------------------- */
/* idxQ / idxT: T_Index::queryPos / refPos; QuerySeqPos(q) = idxQ + q (ForwardIndex :63-69) or idxQ - q
 * (ReverseIndex :89-95).  Returns the function's value; *endQ / *endT = globalMinScoreQPos / TPos. */
static int extend_alignment(const unsigned char *qs, const unsigned char *ts, int idxQ, int idxT, int qAlignLen,
                            int tAlignLen, int reverse, const int m[5][5], int ins, int del, int k,
                            int minExtendNBases, int maxNDrops, aln_t *out, int *endQ, int *endT) {
  const int nCols = 2 * k + 2, middleCol = k + 1; /* :140-148 */
  *endQ = *endT = 0;
  if (qAlignLen < minExtendNBases || tAlignLen < minExtendNBases) return 0; /* :150-157 */
  const int maxRows = qAlignLen + 2;
  const size_t total = (size_t)maxRows * (size_t)nCols;
  int *S = (int *)calloc(total, sizeof(int));
  unsigned char *P = (unsigned char *)malloc(total);
  memset(P, A_NONE, total);
  /* RCToIndex (:13-41) with qStart = tStart = 0 */
#define EX_IN(r, c) ((r) >= 0 && (c) >= 0 && abs((r) - (c)) <= k && middleCol - ((r) - (c)) < nCols)
#define EX_IX(r, c) ((size_t)(r) * (size_t)nCols + (size_t)(middleCol - ((r) - (c))))
  for (int q = 1; q <= k && q - 1 != qAlignLen; q++) { /* :186-195 */
    S[EX_IX(q, 0)] = S[EX_IX(q - 1, 0)] + ins;
    P[EX_IX(q, 0)] = A_UP;
  }
  for (int t = 1; t <= k && t - 1 != tAlignLen; t++) { /* :201-212 */
    S[EX_IX(0, t)] = S[EX_IX(0, t - 1)] + del;
    P[EX_IX(0, t)] = A_LEFT;
  }
  int nDrops = 0, prevRowMin = INT_MAX, globalMin = INT_MAX, gq = 0, gt = 0;
  const int maxAlignLength = (qAlignLen < tAlignLen ? qAlignLen : tAlignLen) + maxNDrops; /* :228 */
  for (int q = 1; q - 1 != qAlignLen && nDrops < maxNDrops && q < maxAlignLength; q++) { /* :230-233 */
    int curRowMin = INT_MAX;
    const int tStart = q - k > 1 ? q - k : 1;
    int tEnd = q + k + 1 < tAlignLen + 1 ? q + k + 1 : tAlignLen + 1;
    if (maxAlignLength < tEnd) tEnd = maxAlignLength;
    for (int t = tStart; t < tEnd; t++) { /* :252-330 */
      const int qPos = reverse ? idxQ - (q - 1) : idxQ + (q - 1), tPos = reverse ? idxT - (t - 1) : idxT + (t - 1);
      int insScore = INT_MAX, delScore = INT_MAX, matchScore = INT_MAX;
      if (EX_IN(q - 1, t)) insScore = S[EX_IX(q - 1, t)] + ins;
      if (EX_IN(q, t - 1)) delScore = S[EX_IX(q, t - 1)] + del;
      if (EX_IN(q - 1, t - 1)) matchScore = S[EX_IX(q - 1, t - 1)] + m[three_bit(ts[tPos])][three_bit(qs[qPos])];
      int minScore = matchScore < delScore ? matchScore : delScore;
      if (insScore < minScore) minScore = insScore;
      S[EX_IX(q, t)] = minScore;
      if (minScore != INT_MAX) { /* :306-310: the last test that holds wins */
        if (minScore == insScore) P[EX_IX(q, t)] = A_UP;
        if (minScore == delScore) P[EX_IX(q, t)] = A_LEFT;
        if (minScore == matchScore) P[EX_IX(q, t)] = A_DIAG;
      } else {
        P[EX_IX(q, t)] = A_NONE;
      }
      if (minScore < curRowMin) curRowMin = minScore;
      if (minScore < globalMin) {
        globalMin = minScore;
        gq = q;
        gt = t;
      }
    }
    if (curRowMin > prevRowMin) nDrops++; /* :333-336 */
    prevRowMin = curRowMin;
  }
  /* traceback from the global minimum (:339-374) */
  int q = gq, t = gt, n = 0;
  unsigned char *path = (unsigned char *)malloc((size_t)gq + (size_t)gt + 1);
  if (P[EX_IX(q, t)] != A_NONE) {
    while (q > 0 || t > 0) {
      const unsigned char arrow = P[EX_IX(q, t)];
      path[n++] = arrow;
      if (arrow == A_DIAG) {
        q--;
        t--;
      } else if (arrow == A_LEFT) {
        t--;
      } else if (arrow == A_UP) {
        q--;
      } else {
        break; /* the reference asserts here; not reachable from a scored cell */
      }
    }
  }
  if (!reverse) /* ForwardIndex::OrderArrowVector :79-81; ReverseIndex keeps the order :105-106 */
    for (int i = 0; i < n / 2; i++) {
      const unsigned char sw = path[i];
      path[i] = path[n - 1 - i];
      path[n - 1 - i] = sw;
    }
  arrow_path_to_alignment(out, path, n);
  if (reverse) { /* :376-377 with ReverseIndex::GetQueryStartPos :97-103 */
    out->qPos = (uint32_t)(idxQ - (gq - 1));
    out->tPos = (uint32_t)(idxT - (gt - 1));
  }
  *endQ = gq;
  *endT = gt;
  free(path);
  free(S);
  free(P);
#undef EX_IN
#undef EX_IX
  return globalMin;
}

/* reverse 0: ExtendAlignmentForward (aligns query[queryPos..) to target[refPos..), qPos/tPos = queryPos/refPos),
 * reverse 1: ExtendAlignmentReverse (aligns the bases before queryPos / refPos, walking left).
 * out[5] = return value, qPos, tPos, globalMinScoreQPos, globalMinScoreTPos. */
int bo_extend_align(const unsigned char *q, int qlen, int queryPos, const unsigned char *t, int tlen, int refPos,
                    const int *matrix25, int ins, int del, int k, int minExtendNBases, int maxNDrops, int reverse,
                    int *out, int *blocks, int blockCap, int *nBlocks, int *gapCounts, int gapListCap, int *nGapLists,
                    int *gapItems, int gapItemCap, int *nGapItems) {
  int m[5][5];
  for (int i = 0; i < 5; i++)
    for (int j = 0; j < 5; j++) m[i][j] = matrix25[i * 5 + j];
  aln_t a;
  aln_init(&a);
  int endQ = 0, endT = 0;
  if (reverse) {
    out[0] = extend_alignment(q, t, queryPos - 1, refPos - 1, queryPos, refPos, 1, m, ins, del, k, minExtendNBases,
                              maxNDrops, &a, &endQ, &endT);
  } else {
    out[0] = extend_alignment(q, t, queryPos, refPos, qlen - queryPos, tlen - refPos, 0, m, ins, del, k,
                              minExtendNBases, maxNDrops, &a, &endQ, &endT);
    a.qPos = (uint32_t)queryPos; /* :401-402 */
    a.tPos = (uint32_t)refPos;
  }
  out[1] = (int)a.qPos;
  out[2] = (int)a.tPos;
  out[3] = endQ;
  out[4] = endT;
  int rc = flatten(&a, blocks, blockCap, nBlocks, gapCounts, gapListCap, nGapLists, gapItems, gapItemCap, nGapItems);
  aln_free(&a);
  return rc;
}

/* ---------------------------------------------------------------------------------------------------------
 * next-2: first pieces of the output epilogue
 * --------------------------------------------------------------------------------------------------------- */
/* SequenceIndexDatabase::SearchForIndex (SequenceIndexDatabase.h:108-118): upper_bound over seqStartPos[1..] */
static int search_for_index(const unsigned *seqStartPos, int nSeqPos, unsigned pos) {
  int lo = 1, hi = nSeqPos;
  if (nSeqPos == 1) return 0;
  while (lo < hi) {
    int mid = (lo + hi) / 2;
    if (seqStartPos[mid] <= pos) lo = mid + 1; else hi = mid;
  }
  return lo - 1;
}

/* AssignRefContigLocation (Blasr.cpp:2183-2220); out2 = {contig index, tAlignedSeqPos inside the contig} */
void bo_ref_contig_location(unsigned tAlignedSeqPos, unsigned tAlignedSeqLength, int tStrand, const unsigned *seqStartPos,
                            int nSeqPos, unsigned genomeLength, unsigned *out2) {
  if (tStrand == 0) {
    int idx = search_for_index(seqStartPos, nSeqPos, tAlignedSeqPos);
    out2[0] = (unsigned)idx;
    out2[1] = tAlignedSeqPos - seqStartPos[idx];
  } else {
    /* MakeRCCoordinate(p) = length - p - 1 (DNASequence.h:215-217) */
    unsigned forwardTPos = genomeLength - (tAlignedSeqPos + tAlignedSeqLength - 1) - 1;
    int idx = search_for_index(seqStartPos, nSeqPos, forwardTPos);
    unsigned reverseTOffset = genomeLength - (seqStartPos[idx + 1] - 2) - 1;
    out2[0] = (unsigned)idx;
    out2[1] = tAlignedSeqPos - reverseTOffset;
  }
}

typedef struct {
  int *size;
  char *op;
  long n, cap;
} cigar_ops;
static void cigar_push(cigar_ops *c, int size, char op) {
  if (c->n == c->cap) {
    c->cap = c->cap ? 2 * c->cap : 64;
    c->size = (int *)realloc(c->size, sizeof(int) * (size_t)c->cap);
    c->op = (char *)realloc(c->op, (size_t)c->cap);
  }
  c->size[c->n] = size;
  c->op[c->n] = op;
  c->n++;
}
/* AddUngappedOperations / AddUnmatchedOperations (SAMPrinter.h:138-201) */
static void cigar_match_runs(cigar_ops *c, const unsigned char *q, const unsigned char *t, int len) {
  int i = 0;
  while (i < len) {
    int start = i;
    while (i < len && q[i] != t[i]) i++;
    if (i > start) cigar_push(c, i - start, 'X');
    start = i;
    while (i < len && q[i] == t[i]) i++;
    if (i > start) cigar_push(c, i - start, '=');
  }
}

/* CreateCIGARString (SAMPrinter.h:346-416); returns the string's length, -1 if outCap is too small */
int bo_cigar(const unsigned char *q, const unsigned char *t, int qPos, int tPos, const unsigned *ops, long nOps,
             const int *blocks, long nBlocks, int clipping, int tStrand, unsigned readLength, unsigned lowQualityPrefix,
             unsigned lowQualitySuffix, unsigned subreadStart, unsigned subreadEnd, unsigned qAlignStart,
             unsigned qAlignEnd, char *out, int outCap, unsigned *clips4) {
  unsigned prefixSoft = 0, suffixSoft = 0, prefixHard = 0, suffixHard = 0;
  cigar_ops c = {NULL, NULL, 0, 0};
  long i, b;
  int at = 0, qp, tp;
  if (clipping == 0) { /* hard: SetHardClip :320-335 */
    prefixHard = qAlignStart;
    suffixHard = readLength - qAlignEnd;
    if (tStrand == 1) {
      unsigned sw = prefixHard;
      prefixHard = suffixHard;
      suffixHard = sw;
    }
  }
  if (clipping == 1 || clipping == 2) { /* soft, subread :366-382 */
    if (clipping == 1) {
      prefixHard = lowQualityPrefix;
      suffixHard = lowQualitySuffix;
    } else {
      prefixHard = subreadStart > lowQualityPrefix ? subreadStart : lowQualityPrefix;
      suffixHard = readLength - subreadEnd > lowQualitySuffix ? readLength - subreadEnd : lowQualitySuffix;
    }
    /* SetSoftClip :300-313 */
    prefixSoft = qAlignStart + (ops ? 0u : (unsigned)blocks[0]) - prefixHard;
    suffixSoft = readLength - suffixHard - qAlignEnd;
  }
  if (prefixHard > 0) cigar_push(&c, (int)prefixHard, 'H');
  if (prefixSoft > 0) cigar_push(&c, (int)prefixSoft, 'S');
  /* :388-389: qPos + blocks[0].qPos */
  qp = qPos + (ops ? 0 : blocks[0]);
  tp = tPos + (ops ? 0 : blocks[1]);
  if (ops) { /* CreateNoClippingCigarOps with gap lists: AddGaps(0) is empty, then block, AddGaps(b+1) */
    for (i = 0; i < nOps; i++) {
      int len = (int)(ops[i] >> 2), ty = (int)(ops[i] & 3u);
      if (ty == 0) {
        cigar_match_runs(&c, q + qp, t + tp, len);
        qp += len;
        tp += len;
      } else if (ty == 2) { /* Gap::Query -> D (:125-129) */
        cigar_push(&c, len, 'D');
        tp += len;
      } else { /* Gap::Target -> I */
        cigar_push(&c, len, 'I');
        qp += len;
      }
    }
  } else { /* nGaps == 0 (:228-287) */
    for (b = 0; b < nBlocks; b++) {
      int len = blocks[3 * b + 2], qGap = 0, tGap = 0, unmatched = 0;
      if (b + 1 < nBlocks) {
        qGap = blocks[3 * (b + 1)] - blocks[3 * b] - len;
        tGap = blocks[3 * (b + 1) + 1] - blocks[3 * b + 1] - len;
      }
      cigar_match_runs(&c, q + qp, t + tp, len);
      qp += len;
      tp += len;
      if (qGap > 0 && tGap > 0) {
        int common = qGap < tGap ? qGap : tGap;
        qGap -= common;
        tGap -= common;
        unmatched = common;
      }
      if (qGap > 0) {
        cigar_push(&c, qGap, 'I');
        qp += qGap;
      }
      if (tGap > 0) {
        cigar_push(&c, tGap, 'D');
        tp += tGap;
      }
      cigar_match_runs(&c, q + qp, t + tp, unmatched);
      qp += unmatched;
      tp += unmatched;
    }
  }
  if (suffixSoft > 0) cigar_push(&c, (int)suffixSoft, 'S');
  if (suffixHard > 0) cigar_push(&c, (int)suffixHard, 'H');
  clips4[0] = prefixSoft;
  clips4[1] = suffixSoft;
  clips4[2] = prefixHard;
  clips4[3] = suffixHard;
  for (i = 0; i < c.n; i++) {
    long k = tStrand == 1 ? c.n - 1 - i : i; /* :408-411 */
    int w = snprintf(out + at, at < outCap ? (size_t)(outCap - at) : 0, "%d%c", c.size[k], c.op[k]);
    at += w;
    if (at >= outCap) {
      at = -1;
      break;
    }
  }
  free(c.size);
  free(c.op);
  if (at == 0 && outCap > 0) out[0] = 0;
  return at;
}

/* IntervalAlignmentPrinter::Print (:9-38); ostream << float = %g with six significant digits */
int bo_m4_line(const char *qName, const char *tName, int score, float pctSimilarity, int qStrand, unsigned qAlignStart,
               unsigned qAlignEnd, unsigned qLength, int tStrand, unsigned tAlignStart, unsigned tAlignEnd,
               unsigned tLength, int mapQV, char *out, int outCap) {
  int w = snprintf(out, (size_t)outCap, "%s %s %d %g %d %u %u %u %d %u %u %u %d\n", qName, tName, score,
                   (double)pctSimilarity, qStrand, qAlignStart, qAlignEnd, qLength, tStrand, tAlignStart, tAlignEnd,
                   tLength, mapQV);
  return w < outCap ? w : -1;
}

/* ---- StoreMapQVs ------------------------------------------------------------------------------------------- */
/* LogSumOfTwo (utils/LogUtils.h:11-41); LOG10 there is the literal 2.3025850929 */
static double mq_log_sum_of_two(double value1, double value2) {
  double minValue = value1, maxValue = value2, difference;
  if (maxValue < minValue) {
    minValue = value2;
    maxValue = value1;
  }
  minValue *= 2.3025850929;
  maxValue *= 2.3025850929;
  difference = minValue - maxValue;
  if (difference < -30) return maxValue / 2.3025850929;
  if (difference < -8) return (maxValue + exp(difference)) / 2.3025850929;
  {
    float expv = exp(difference);
    float log1pv = log1p(expv);
    return (maxValue + log1pv) / 2.3025850929;
  }
}
/* AlignmentCandidate::GetQInterval (AlignmentCandidate.h:162-174) */
static void mq_interval(unsigned start, unsigned end, int qStrand, unsigned qLength, int forward, int *s, int *e) {
  *s = (int)start;
  *e = (int)end;
  if (forward && qStrand == 1) {
    int fs = (int)qLength - *e, fe = (int)qLength - *s;
    *s = fs;
    *e = fe;
  }
}
/* AlignmentsOverlap (Blasr.cpp:2679-2722) on forward-strand intervals */
static int mq_overlap(int aS, int aE, int bS, int bE, float minPercentOverlap) {
  int ovp = 0;
  float pct;
  if (aS <= bS && aE >= bE) return 1;
  if (bS <= aS && bE >= aE) return 1;
  if (aE >= bS && aE <= bE) ovp = aE - bS;
  else if (aS >= bS && aS <= bE) ovp = bE - aS;
  if (aE - aS > 0 && bE - bS > 0) {
    float x = (float)ovp / (float)(aE - aS), y = (float)ovp / (float)(bE - bS);
    pct = x > y ? x : y;
  } else {
    pct = 0;
  }
  return pct > minPercentOverlap;
}

void bo_store_map_qvs(int n, const unsigned char *const *qSeqs, const unsigned char *const *tSeqs, const int *qPos,
                      const int *tPos, const unsigned *ops, const long *opOff, const unsigned *qAlignStart,
                      const unsigned *qAlignEnd, const int *qStrand, const unsigned *qLength, const int *nMatch,
                      const unsigned char *qvs, const int *matrix25, int ins, int del, int affineOpen, int affineExtend,
                      int affineAlign, float minFraction, int *mapQV, float *probScore) {
  /* partitions as arrays: part[i] = partition of alignment i; members are visited in increasing index, which is
   * the iteration order of the reference's set<int> */
  int *part = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  int nPart = 0, a, p, i;
  int *pBegin, *pEnd, *pScore;
  for (a = 0; a < n; a++) { /* :2777-2785 with ComputeAlignmentScore (AlignmentUtils.h:127-170) */
    int score = 0, q = qPos[a], t = tPos[a];
    long k;
    for (k = opOff[a]; k < opOff[a + 1]; k++) {
      int len = (int)(ops[k] >> 2), ty = (int)(ops[k] & 3u), l;
      if (ty == 0) {
        for (l = 0; l < len; l++) score += matrix25[three_bit(tSeqs[a][t + l]) * 5 + three_bit(qSeqs[a][q + l])];
        q += len;
        t += len;
      } else {
        int unit = ty == 1 ? ins : del;
        if (affineAlign && affineOpen + len * affineExtend < len * unit) score += affineOpen + len * affineExtend;
        else score += len * unit;
        if (ty == 1) q += len; else t += len;
      }
    }
    probScore[a] = -score / 10.0;
  }
  for (i = 0; i < n; i++) { /* PartitionOverlappingAlignments :2726-2759 */
    int found = -1, iS, iE;
    mq_interval(qAlignStart[i], qAlignEnd[i], qStrand[i], qLength[i], 1, &iS, &iE);
    for (p = 0; p < nPart && found < 0; p++) {
      int j;
      for (j = 0; j < i && found < 0; j++) {
        int jS, jE;
        if (part[j] != p) continue;
        mq_interval(qAlignStart[j], qAlignEnd[j], qStrand[j], qLength[j], 1, &jS, &jE);
        if (mq_overlap(iS, iE, jS, jE, minFraction) ||
            (qAlignStart[i] <= qAlignStart[j] && qAlignEnd[i] > qAlignEnd[j]))
          found = p;
      }
    }
    part[i] = found >= 0 ? found : nPart++;
  }
  pBegin = (int *)malloc(sizeof(int) * (size_t)(nPart > 0 ? nPart : 1));
  pEnd = (int *)malloc(sizeof(int) * (size_t)(nPart > 0 ? nPart : 1));
  pScore = (int *)malloc(sizeof(int) * (size_t)(nPart > 0 ? nPart : 1));
  for (p = 0; p < nPart; p++) { /* :2797-2826 */
    int first = 1;
    for (i = 0; i < n; i++) {
      int s, e;
      if (part[i] != p) continue;
      mq_interval(qAlignStart[i], qAlignEnd[i], qStrand[i], qLength[i], 0, &s, &e);
      if (first || (e - s > pEnd[p] - pBegin[p] && nMatch[i] * 1.20 >= pScore[p])) {
        pBegin[p] = s;
        pEnd[p] = e;
        pScore[p] = nMatch[i];
      }
      first = 0;
    }
  }
  for (i = 0; i < n; i++) { /* :2851-2873, SumMismatches :2647-2671 */
    int s, e, mismatchSum = 0, x;
    p = part[i];
    mq_interval(qAlignStart[i], qAlignEnd[i], qStrand[i], qLength[i], 1, &s, &e);
    if (s - pBegin[p] > 5 || pEnd[p] - e > 5) {
      if (qvs != NULL) {
        for (x = pBegin[p]; x < s; x++) mismatchSum += qvs[x];
        for (x = e; x < pEnd[p]; x++) mismatchSum += qvs[x];
      } else {
        mismatchSum += 15 * ((s - pBegin[p]) + (pEnd[p] - e));
      }
    }
    probScore[i] += -(mismatchSum) * 0.5;
  }
  for (p = 0; p < nPart; p++) { /* :2878-2944 */
    float denominator = 0;
    int maxNMatch = 0, first = 1, size = 0;
    for (i = 0; i < n; i++) {
      if (part[i] != p) continue;
      size++;
      if (first) {
        denominator = probScore[i];
        maxNMatch = nMatch[i];
        first = 0;
      } else {
        if (nMatch[i] > maxNMatch) maxNMatch = nMatch[i];
        if (nMatch[i] * 1.2 >= maxNMatch) denominator = mq_log_sum_of_two(denominator, probScore[i]);
      }
    }
    for (i = 0; i < n; i++) {
      if (part[i] != p) continue;
      if (size == 1) {
        mapQV[i] = 254;
      } else if (probScore[i] - denominator < -20) {
        mapQV[i] = 0;
      } else {
        double log10v = log(10), sub = probScore[i] - denominator;
        double expo = exp(log10v * sub), diff = 1.0 - expo;
        int phred;
        if (expo == 0) phred = 0;
        else if (diff == 0) phred = 254;
        else { /* diff < 0 happens (exp rounds above 1): NaN to int is what cvttsd2si gives, INT_MIN -- spelled out */
          double v = -10 * log10(diff);
          phred = (v != v || v >= 2147483648.0 || v < -2147483648.0) ? INT_MIN : (int)v;
        }
        if (phred > 254) phred = 254;
        if (phred < 0) phred = 0;
        mapQV[i] = phred;
      }
    }
  }
  free(part);
  free(pBegin);
  free(pEnd);
  free(pScore);
}

/* ---- SAMOutput::PrintAlignment ----------------------------------------------------------------------------- */
static unsigned char sam_rc(unsigned char c) { /* ReverseComplementNuc (NucConversion.h:337-370) */
  switch (c) {
    case 0: return 3; case 1: return 2; case 2: return 1; case 3: return 0;
    case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; case 'N': return 'N';
    case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a'; case 'n': return 'n';
    default: return 127;
  }
}
int bo_sam_line(const char *qName, const char *tName, const char *readGroupId, const unsigned char *readSeq,
                const unsigned char *readQual, const char *cigar, const unsigned *clips4, const int *iv, const float *fv,
                char *out, int outCap) {
  const int clipping = iv[2], qStrand = iv[3], tStrand = iv[4], isPrimary = iv[5];
  const unsigned readLength = (unsigned)iv[6], lqp = (unsigned)iv[7], lqs = (unsigned)iv[8];
  const unsigned subreadStart = (unsigned)iv[9], subreadEnd = (unsigned)iv[10];
  const unsigned qAlignStart = (unsigned)iv[24], qAlignEnd = (unsigned)iv[25], tAlignStart = (unsigned)iv[26];
  const unsigned tEnd = (unsigned)iv[27], tLength = (unsigned)iv[14];
  unsigned start = 0, length = 0, flag = 0, pos, i;
  int at = 0;
#define SAM_PUT(...) do { int w_ = snprintf(out + at, at < outCap ? (size_t)(outCap - at) : 0, __VA_ARGS__); at += w_; \
                          if (at >= outCap) return -1; } while (0)
  /* SetAlignedSequence :58-105 */
  if (clipping == 3 || clipping == 0) {
    start = qAlignStart;
    length = qAlignEnd - qAlignStart;
  } else if (clipping == 1) {
    length = readLength - lqp - lqs;
    start = lqp;
  } else {
    length = subreadEnd - subreadStart;
    start = subreadStart;
  }
  if (tStrand != qStrand) flag |= 0x10; /* BuildFlag :29-50 */
  if (!isPrimary) flag |= 0x100;
  pos = tStrand == 0 ? tAlignStart + 1 : tLength - (tAlignStart + tEnd) + 1; /* :437-446 */
  SAM_PUT("%s\t%u\t%s\t%u\t%d\t%s\t*\t0\t%u\t", qName, flag, tName, pos, iv[16], cigar, tEnd);
  if (at + (int)length + 2 >= outCap) return -1;
  for (i = 0; i < length; i++) out[at++] = (char)(tStrand == 0 ? readSeq[start + i] : sam_rc(readSeq[start + length - 1 - i]));
  out[at++] = '\t';
  if (readQual != NULL) {
    if (at + (int)length + 2 >= outCap) return -1;
    for (i = 0; i < length; i++) out[at++] = (char)(readQual[tStrand == 0 ? start + i : start + length - 1 - i] + 33);
  } else {
    out[at++] = '*';
  }
  SAM_PUT("\tRG:Z:%s\tAS:i:%d\t", readGroupId, iv[17]);
  if (clipping == 3) {
    SAM_PUT("XS:i:%u\tXE:i:%u\t", qAlignStart + 1, qAlignEnd + 1);
  } else {
    unsigned xs = clips4[2], xe = readLength - clips4[3];
    if (tStrand == 1) {
      xs = clips4[3];
      xe = readLength - clips4[2];
    }
    SAM_PUT("XS:i:%u\tXE:i:%u\tqs:i:%d\tqe:i:%d\t", xs + 1, xe + 1, iv[20], iv[21]);
  }
  SAM_PUT("zm:i:%d\tXL:i:%u\tXT:i:1\tNM:i:%d\tFI:i:%u\tXQ:i:%u\t", iv[19], (unsigned)iv[12], iv[18], (unsigned)iv[11] + 1,
          (unsigned)iv[15]);
  SAM_PUT("rq:f:%.4g\tnp:i:%d\tcx:i:%d\tsn:B:f,%.4g,%.4g,%.4g,%.4g\n", (double)fv[0], iv[22], iv[23], (double)fv[1],
          (double)fv[2], (double)fv[3], (double)fv[4]);
#undef SAM_PUT
  return at;
}

/* Alignment::OrderGapsByType (Alignment.h:255-317): gap list g sits between block g-1 and block g; lists with more
 * than one gap are replaced by the net gap, and the bases that cancel lengthen block g-1. */
long bo_order_gaps_by_type(const unsigned *ops, long nOps, unsigned *out) {
  long i = 0, n = 0;
  while (i < nOps) {
    long j = i, k;
    unsigned targetLen = 0, queryLen = 0;
    if ((ops[i] & 3u) == 0) {
      out[n++] = ops[i++];
      continue;
    }
    while (j < nOps && (ops[j] & 3u) != 0) {
      if ((ops[j] & 3u) == 1) targetLen += ops[j] >> 2; else queryLen += ops[j] >> 2;
      j++;
    }
    if (j - i <= 1 || n == 0) {
      for (k = i; k < j; k++) out[n++] = ops[k];
    } else if (targetLen > queryLen) {
      out[n - 1] += queryLen << 2;
      out[n++] = ((targetLen - queryLen) << 2) | 1u;
    } else if (queryLen > targetLen) {
      out[n - 1] += targetLen << 2;
      out[n++] = ((queryLen - targetLen) << 2) | 2u;
    } else {
      out[n - 1] += targetLen << 2;
    }
    i = j;
  }
  return n;
}

/* ------------------------------------------------------------------------- */
/* next-1, device half: MapReadToGenome over the suffix array                 */
/* ------------------------------------------------------------------------- */
/* SearchLeftBound (datastructures/suffixarray/SuffixArray.h:733-775): first suffix in [l, r) whose character
 * `offset` bases in is >= queryChar; suffixes that end at or before `offset` sort low. */
static long sa_left_bound(const unsigned char *target, long targetLength, const unsigned *index, unsigned offset,
                          unsigned char queryChar, long l, long r) {
  long ll = l, lr = r;
  while (ll < lr) {
    const long m = (ll + lr) / 2;
    const long sufLen = targetLength - (long)index[m];
    if (sufLen == (long)offset) {
      ll = m + 1;
      continue;
    }
    int comp;
    if (sufLen < (long)offset)
      comp = -1;
    else
      comp = g_three[target[index[m] + offset]] - g_three[queryChar]; /* DefaultCompareStrings, CompareStrings.h:8-10 */
    if (comp < 0)
      ll = m + 1;
    else
      lr = m;
  }
  return ll;
}

/* SearchRightBound (SuffixArray.h:777-816): first suffix in [l, r) whose character is > queryChar; a suffix that ends
 * exactly at `offset` stops the search where it stands (the reference's early break, kept). */
static long sa_right_bound(const unsigned char *target, long targetLength, const unsigned *index, unsigned offset,
                           unsigned char queryChar, long l, long r) {
  long rl = l, rr = r;
  while (rl < rr) {
    const long m = (rl + rr) / 2;
    const long sufLen = targetLength - (long)index[m];
    if (sufLen == (long)offset) {
      rr = m;
      break;
    }
    if (sufLen < (long)offset) {
      rr = m;
    } else {
      const int comp = g_three[target[index[m] + offset]] - g_three[queryChar];
      if (comp <= 0)
        rl = m + 1;
      else
        rr = m;
    }
  }
  return rr;
}

/* StoreLCPBounds (SuffixArray.h:928-1070).  lows/highs receive one pair per matched length; returns lcpLength and
 * stores the number of pairs in *nBounds.  The pairs are kept in a growing array owned by the caller. */
typedef struct {
  long *low, *high;
  long n, cap;
} sa_bounds;
static void sa_bounds_push(sa_bounds *b, long l, long r) {
  if (b->n == b->cap) {
    b->cap = b->cap ? 2 * b->cap : 64;
    b->low = (long *)realloc(b->low, sizeof(long) * (size_t)b->cap);
    b->high = (long *)realloc(b->high, sizeof(long) * (size_t)b->cap);
  }
  b->low[b->n] = l;
  b->high[b->n] = r;
  b->n++;
}
static unsigned sa_store_lcp_bounds(const unsigned char *target, long targetLength, const unsigned *index,
                                    const unsigned *startPosTable, const unsigned *endPosTable,
                                    int lookupPrefixLength, const unsigned char *query, unsigned queryLength,
                                    unsigned queryAvail, int useLookupTable, int maxMatchLength, int stopOnceUnique,
                                    sa_bounds *b) {
  long l = 0, r = targetLength;
  unsigned lcpLength = 0;
  if (useLookupTable && startPosTable != NULL) {
    /* DNATuple::FromStringLR (tuples/DNATuple.h:24-52) */
    unsigned long tuple = 0;
    for (int i = 0; i < lookupPrefixLength; i++) {
      /* With a table prefix longer than minPrefixMatchLength the reference reads up to prefix - min bytes PAST the read's
       * buffer at the last positions (its own ASan report).  What it finds there is not defined; a byte that is
       * not there counts as "not a base" here (what a zero after the buffer gives the reference), and is not read. */
      if ((unsigned)i >= queryAvail || !g_isactg[query[i]]) return 0;
      tuple = (tuple << 2) + (unsigned long)g_two[query[i]];
    }
    l = startPosTable[tuple];
    r = endPosTable[tuple];
    lcpLength = (unsigned)lookupPrefixLength;
    if (l < r)
      sa_bounds_push(b, l, r);
    else
      return 0;
  }
  while (l < r && lcpLength < queryLength) {
    if (stopOnceUnique && l == r - 1) break;
    if (maxMatchLength && lcpLength >= (unsigned)maxMatchLength) break;
    if (g_three[target[index[l] + lcpLength]] >= 4) break;
    l = sa_left_bound(target, targetLength, index, lcpLength, query[lcpLength], l, r);
    r = sa_right_bound(target, targetLength, index, lcpLength, query[lcpLength], l, r);
    if (l == r || (long)index[l] + lcpLength >= targetLength || g_three[query[lcpLength]] >= 4 ||
        g_three[target[index[l] + lcpLength]] - g_three[query[lcpLength]] != 0)
      break;
    sa_bounds_push(b, l, r);
    lcpLength++;
  }
  return lcpLength;
}

/* LocateAnchorBoundsInSuffixArray (algorithms/anchoring/MapBySuffixArray.h:24-215) followed by MapReadToGenome's
 * emission loop (:217-311).  ap[7] = minMatchLength, expand, useLookupTable, maxAnchorsPerPosition,
 * advanceExactMatches, maxLCPLength, stopMappingOnceUnique (AnchorParameters.h:9-42; removeEncompassedMatches reads
 * past its arrays in the reference and is not restated).  The read is one strand, already reverse-complemented by
 * the caller for the other.  out: (t, q, l) triples in emission order.  Returns the count; -3 / -4 where the
 * reference's asserts (:283, :301-302) would fire. */
long bo_map_read_to_genome(const unsigned char *genome, unsigned genomeLength, const unsigned *saIndex,
                           const unsigned *startPosTable, const unsigned *endPosTable, int lookupPrefixLength,
                           const unsigned char *read, unsigned readLength, unsigned subreadStart, unsigned subreadEnd,
                           unsigned minPrefixMatchLength, const int *ap, unsigned *out, long cap) {
  init_tables();
  const unsigned minMatchLength = (unsigned)ap[0];
  const int expand = ap[1], useLookupTable = ap[2], maxAnchorsPerPosition = ap[3], advanceExactMatches = ap[4],
            maxLCPLength = ap[5], stopMappingOnceUnique = ap[6];
  if (subreadEnd - subreadStart < minMatchLength) return 0; /* :227-230 */
  unsigned *matchLow = NULL, *matchHigh = NULL, *matchLength = NULL;
  unsigned numSearched = 0;
  sa_bounds b = {NULL, NULL, 0, 0};
  if (!(minPrefixMatchLength > 0 && subreadEnd - subreadStart < minPrefixMatchLength)) { /* :39-42 */
    const unsigned matchEnd = subreadEnd - minPrefixMatchLength + 1;
    numSearched = matchEnd - subreadStart;
    matchLow = (unsigned *)calloc(numSearched ? numSearched : 1, sizeof(unsigned));
    matchHigh = (unsigned *)calloc(numSearched ? numSearched : 1, sizeof(unsigned));
    matchLength = (unsigned *)calloc(numSearched ? numSearched : 1, sizeof(unsigned));
    unsigned p, m;
    for (m = 0, p = subreadStart; p < matchEnd; p++, m++) {
      b.n = 0;
      unsigned lcpLength =
          sa_store_lcp_bounds(genome, (long)genomeLength, saIndex, startPosTable, endPosTable, lookupPrefixLength,
                              read + p, matchEnd - p, readLength - p, useLookupTable, maxLCPLength, stopMappingOnceUnique,
                              &b);
      matchLow[m] = matchHigh[m] = matchLength[m] = 0;
      if (b.n > 0) {
        long lcpSearchLength = b.n;
        while (lcpSearchLength > 0 && b.low[lcpSearchLength - 1] == b.high[lcpSearchLength - 1]) { /* :104-109 */
          lcpSearchLength--;
          lcpLength--;
        }
        matchLow[m] = (unsigned)b.low[lcpSearchLength - 1];
        matchHigh[m] = (unsigned)b.high[lcpSearchLength - 1];
        matchLength[m] = minPrefixMatchLength + (unsigned)lcpSearchLength - 1;
        if (matchLow[m] + 1 == matchHigh[m]) { /* unique: extend along the genome, :134-174 */
          lcpLength = minPrefixMatchLength + (unsigned)lcpSearchLength - 1;
          long refPos = (long)saIndex[matchLow[m]] + lcpLength - 1;
          long queryPos = (long)p + lcpLength - 1;
          int extended = 0;
          while (refPos + 1 < (long)genomeLength && queryPos + 1 < (long)readLength && genome[refPos + 1] != 'N' &&
                 genome[refPos + 1] == read[queryPos + 1] &&
                 (maxLCPLength == 0 || lcpLength < (unsigned)maxLCPLength)) {
            refPos++;
            queryPos++;
            lcpLength++;
            extended = 1;
          }
          if (extended) {
            matchLength[m] = lcpLength;
          } else {
            if (lcpSearchLength > 1) lcpSearchLength--;
            matchLow[m] = (unsigned)b.low[lcpSearchLength - 1];
            matchHigh[m] = (unsigned)b.high[lcpSearchLength - 1];
            matchLength[m] = minPrefixMatchLength + (unsigned)lcpSearchLength - 1;
          }
        } else { /* not unique: back off by expand, :176-193 */
          if (lcpSearchLength > expand)
            lcpSearchLength -= expand;
          else
            lcpSearchLength = 1;
          matchLow[m] = (unsigned)b.low[lcpSearchLength - 1];
          matchHigh[m] = (unsigned)b.high[lcpSearchLength - 1];
          matchLength[m] = minPrefixMatchLength + (unsigned)lcpSearchLength - 1;
        }
      }
      if (advanceExactMatches) { /* :206-212 */
        const int adv = (int)(lcpLength - (unsigned)advanceExactMatches);
        p += (unsigned)(adv > 1 ? adv : 1);
        m += (unsigned)(adv > 1 ? adv : 1);
      }
    }
  }
  free(b.low);
  free(b.high);
  /* MapReadToGenome's emission, :268-306 */
  unsigned trim = minMatchLength + 1;
  if ((unsigned)lookupPrefixLength + 1 > trim) trim = (unsigned)lookupPrefixLength + 1;
  const unsigned endOfMapping = subreadEnd < trim ? 0 : subreadEnd - trim;
  long n = 0;
  for (unsigned pos = subreadStart; pos < endOfMapping; pos++) {
    const unsigned mi = pos - subreadStart;
    if (mi >= numSearched) {
      n = -3;
      break;
    }
    if (matchHigh[mi] - matchLow[mi] <= (unsigned)maxAnchorsPerPosition) {
      for (unsigned mp = matchLow[mi]; mp < matchHigh[mi]; mp++) {
        if (matchLength[mi] < minMatchLength) continue;
        if (matchLength[mi] + pos > readLength) matchLength[mi] = readLength - pos;
        if ((unsigned long)saIndex[mp] + matchLength[mi] > genomeLength ||
            genome[saIndex[mp] + matchLength[mi] - 1] == 'N') {
          n = -4;
          break;
        }
        if (n < cap) {
          out[3 * n] = saIndex[mp];
          out[3 * n + 1] = pos;
          out[3 * n + 2] = matchLength[mi];
        }
        n++;
      }
      if (n < 0) break;
    }
  }
  free(matchLow);
  free(matchHigh);
  free(matchLength);
  return n;
}

/* SortMatchPosList (datastructures/anchoring/MatchPos.h:120-131,166-168): ascending t, then q.  (t, q) pairs are
 * distinct within one list, so the order is total. */
static int cmp_match_pos(const void *a, const void *b) {
  const unsigned *x = (const unsigned *)a, *y = (const unsigned *)b;
  if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
  if (x[1] != y[1]) return x[1] < y[1] ? -1 : 1;
  return 0;
}
void bo_sort_match_pos(unsigned *triples, long n) { qsort(triples, (size_t)n, 3 * sizeof(unsigned), cmp_match_pos); }
