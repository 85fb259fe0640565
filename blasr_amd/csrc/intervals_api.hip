// next-1, host half: from the sorted anchor lists of a read to its candidate intervals.
//
// Host C++, no device work: per BASELINE.md this is 1 % of the anchoring time (0.01 s per 200 reads against 0.97 s for
// the suffix-array search, which is the device half in anchor_api.hip).  Restates, for one read at a time, what MapRead
// does between SortMatchPosList and AlignIntervals (alignment/Blasr.cpp:2342-2500) with its default switches:
//   RemoveOverlappingAnchors      algorithms/anchoring/FindMaxInterval.h:282-315 (+ RemoveZeroLengthAnchors :267-280)
//   StoreLargestIntervals         :394-640, the non-overlapping branch, with AdvanceIndexToPastInterval :230-265
//   GlobalChain                   algorithms/anchoring/GlobalChain.h:101-207 over BasicEndpoint (BasicEndpoint.h:57-113)
//                                 and PrioritySearchTree (PrioritySearchTree.h:44-275)
//   ComputeLISPValue              algorithms/anchoring/LISPValue.h:84-172 with PMatch / POneOrMoreMatches /
//                                 SumRightShiftMarginalTupleCounts (ScoreAnchors.h:11-131) over the tuple count table
//   FindMaxIncreasingInterval     FindMaxInterval.h:661-843
//   WeightedIntervalSet::insert   datastructures/anchoring/WeightedInterval.h:189-268 (+ RemoveContained :133-181)
// The reference's behaviour is kept where it is surprising, because the intervals depend on it: the priority search
// tree's Activate pushes a displaced point down the NEW point's path; noOvpLisNBases is set by ComputeLISPValue and
// then added to again; StoreLargestIntervals looks at the element one past the end of the list (see `stale` in
// blasr_hip.h).  Sorting uses std::sort / std::multiset of the same libstdc++ the reference is built with, so ties fall
// the same way.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <set>
#include <thread>
#include <vector>

#include "../../include/blasr_hip.h"
#include "ctx.h"

#pragma STDC FP_CONTRACT OFF

namespace {

struct Match {  // ChainedMatchPos (MatchPos.h:11-40, :83-118)
  uint32_t t, q, l;
  int score;
  int prev;  // chainPrev as an index into the fragment array, -1 = NULL
};

// ---------------------------------------------------------------------------------------------------------------
// nucleotide tables (NucConversion.h:7-41 TwoBit; tuples/DNATuple.h:24-52 FromStringLR needs IsACTG :88-121)
// ---------------------------------------------------------------------------------------------------------------
inline int two_bit(uint8_t c) {
  if (c < 8) return c & 3;
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 255;
  }
}
inline bool is_actg(uint8_t c) {
  if (c < 4) return true;
  switch (c) {
    case 'A': case 'a': case 'C': case 'c': case 'G': case 'g': case 'T': case 't': return true;
    default: return false;
  }
}

struct Genome {
  const uint8_t *seq;
  int64_t length;
  const uint32_t *seqStartPos;  // SequenceIndexDatabase::seqStartPos, nSeqPos entries
  int nSeqPos;
  const int32_t *countTable;  // TupleCountTable::countTable, 4^tupleSize entries
  int tupleSize;
};

// DNATuple::FromStringLR on the genome at `at` (reads tupleSize bases; the genome ends in an 'N', so a tuple that would
// run past it fails on that N first)
inline bool tuple_at(const Genome &g, int64_t at, uint64_t &tuple) {
  tuple = 0;
  for (int i = 0; i < g.tupleSize; i++) {
    if (at + i >= g.length || !is_actg(g.seq[at + i])) return false;
    tuple = (tuple << 2) + (uint64_t)two_bit(g.seq[at + i]);
  }
  return true;
}

// SequenceIndexDatabase::SearchForIndex / SearchForStartBoundary (SequenceIndexDatabase.h:108-144)
inline uint32_t start_boundary(const Genome &g, uint32_t pos) {
  if (g.nSeqPos == 1) return g.seqStartPos[0];
  const uint32_t *it = std::upper_bound(g.seqStartPos + 1, g.seqStartPos + g.nSeqPos, pos);
  return g.seqStartPos[it - g.seqStartPos - 1];
}

// ---------------------------------------------------------------------------------------------------------------
// p-value of a chain (ScoreAnchors.h, LISPValue.h)
// ---------------------------------------------------------------------------------------------------------------
// SumRightShiftMarginalTupleCounts (ScoreAnchors.h:110-131)
inline int right_marginal(const Genome &g, uint64_t tuple, int nextNuc, int &nextSeqCount) {
  int total = 0;
  const uint64_t base = tuple & ~(uint64_t)3;
  for (int i = 0; i < 4; i++) {
    const int c = g.countTable[base + (uint64_t)i];
    total += c;
    if (i == nextNuc) nextSeqCount = c;
  }
  return total;
}

// PMatch (ScoreAnchors.h:23-88); pMatch is a log probability that starts at 1 in the reference and is kept so
int p_match(const Genome &g, uint32_t startPos, uint32_t length, float &pMatch) {
  pMatch = 1;
  uint64_t tuple;
  if (!tuple_at(g, startPos, tuple)) return 0;
  if (g.countTable[tuple] == 0) return 0;
  if (length < (uint32_t)g.tupleSize) {
    pMatch = 0;
    return 1;
  }
  for (uint32_t i = 1; i < length - (uint32_t)g.tupleSize; i++) {
    if (!tuple_at(g, (int64_t)startPos + i, tuple)) return 0;
    int nextTupleCount = 0;
    const int64_t last = (int64_t)startPos + i + g.tupleSize - 1;
    const int rightMarCount = right_marginal(g, tuple, two_bit(g.seq[last]), nextTupleCount);
    // tuple counts are not defined for N's
    if (last + 1 >= g.length || two_bit(g.seq[last + 1]) > 3) return 0;
    if (nextTupleCount == 0) return 0;
    pMatch += log((nextTupleCount / (1.0 * rightMarCount)));
  }
  return 1;
}

// ComputeLISPValue (LISPValue.h:84-172): POneOrMoreMatches (ScoreAnchors.h:90-107) always succeeds and hands back
// whatever PMatch left in pMatch
// (the chain enters only through its first anchor and its number of anchors)
float lis_p_value_core(const Genome &g, uint32_t firstT, uint32_t firstL, size_t count) {
  float p = 1;
  p_match(g, firstT, firstL, p);
  if (count == 1) return p;
  float pChain = p;
  for (size_t i = 1; i < count; i++) {
    const float qLambda = 1 / (1.0 * g.length);
    pChain = pChain + log(qLambda);
  }
  return pChain;
}
float lis_p_value(const Genome &g, const std::vector<Match> &lis, int &lisNBases, int &lisSize) {
  lisNBases = 0;
  for (size_t i = 0; i < lis.size(); i++) lisNBases += (int)lis[i].l;
  lisSize = (int)lis.size();
  return lis_p_value_core(g, lis[0].t, lis[0].l, lis.size());
}

// ---------------------------------------------------------------------------------------------------------------
// GlobalChain (GlobalChain.h:101-207)
// ---------------------------------------------------------------------------------------------------------------
struct Endpoint {  // BasicEndpoint<ChainedMatchPos>: p = (x, y) = (q, t) of the fragment's start or end
  uint32_t x, y;
  int side;      // 0 Start, 1 End
  int fragment;  // index into the fragment array
};
struct EndpointLess {  // Coordinate::operator< (BasicEndpoint.h:28-31)
  bool operator()(const Endpoint &a, const Endpoint &b) const {
    if (a.x == b.x) return a.y < b.y;
    return a.x < b.x;
  }
};

struct PstVertex {  // PSTVertex (PrioritySearchTree.h:23-42)
  unsigned leftChild = 0, rightChild = 0, isALeaf = 0;
  uint32_t medianKey = 0, maxKey = 0;
  unsigned pointIndex = 0;
  int maxScoreNode = -1;
};

struct Chainer {
  std::vector<Endpoint> points;
  std::vector<PstVertex> tree;
  Match *frags = nullptr;

  uint32_t key(int p) const { return points[(size_t)p].y; }
  int score(int p) const { return frags[points[(size_t)p].fragment].score; }

  // PrioritySearchTree::CreateTree (:53-110)
  uint32_t create(int start, int end, unsigned &iter) {
    const int median = (end + start) / 2;
    const unsigned cur = iter;
    tree[cur].medianKey = key(median);
    if (end == start) {
      tree[cur].pointIndex = (unsigned)start;
      return tree[cur].medianKey;
    }
    if (end - start == 1) {
      tree[cur].isALeaf = 1;
      tree[cur].medianKey = key(start);
      tree[cur].pointIndex = (unsigned)start;
      return tree[cur].medianKey;
    }
    tree[cur].isALeaf = 0;
    tree[cur].leftChild = ++iter;
    const uint32_t leftKey = create(start, median, iter);
    tree[cur].medianKey = leftKey;
    tree[cur].rightChild = ++iter;
    const uint32_t rightKey = create(median, end, iter);
    tree[cur].maxKey = rightKey;
    return rightKey;
  }

  // PrioritySearchTree::FindIndexOfMaxPoint (:112-175)
  int find_max(unsigned v, uint32_t maxKey, int &maxValue, int &maxIndex) const {
    const PstVertex &x = tree[v];
    if (x.maxScoreNode == -1) return 0;
    if (key(x.maxScoreNode) < maxKey) {
      if (score(x.maxScoreNode) >= maxValue) {
        maxValue = score(x.maxScoreNode);
        maxIndex = x.maxScoreNode;
        return 1;
      }
      return 0;
    }
    if (x.isALeaf) return 0;
    if (maxKey <= x.medianKey) return find_max(x.leftChild, maxKey, maxValue, maxIndex);
    const int l = find_max(x.leftChild, maxKey, maxValue, maxIndex);
    const int r = find_max(x.rightChild, maxKey, maxValue, maxIndex);
    return l || r;
  }

  // PrioritySearchTree::Activate (:225-253): the displaced point keeps travelling along the path of the point that
  // displaced it, and is compared with that point's score
  void activate(int pointIndex) {
    const int pointScore = score(pointIndex);
    const uint32_t pointKey = key(pointIndex);
    unsigned cur = 0;
    while (pointIndex != -1 && tree[cur].isALeaf == 0) {
      if (tree[cur].maxScoreNode == -1 || score(tree[cur].maxScoreNode) <= pointScore) {
        const int tmp = tree[cur].maxScoreNode;
        tree[cur].maxScoreNode = pointIndex;
        pointIndex = tmp;
      }
      cur = pointKey <= tree[cur].medianKey ? tree[cur].leftChild : tree[cur].rightChild;
    }
  }

  // Returns the chain as indices into frags, first fragment first.
  void chain(Match *fragments, uint32_t n, std::vector<uint32_t> &out) {
    out.clear();
    if (n == 0) return;
    frags = fragments;
    for (uint32_t f = 0; f < n; f++) frags[f].score = (int)frags[f].l;
    points.resize((size_t)n * 2);
    for (uint32_t f = 0; f < n; f++) {  // FragmentSetToEndpoints (:8-21)
      points[2 * f] = Endpoint{frags[f].q, frags[f].t, 0, (int)f};
      points[2 * f + 1] = Endpoint{frags[f].q + frags[f].l, frags[f].t + frags[f].l, 1, (int)f};
    }
    std::sort(points.begin(), points.end(), EndpointLess());
    tree.assign(points.size() * 2 - 1, PstVertex());
    unsigned iter = 0;
    create(0, (int)points.size(), iter);
    size_t best = 0;
    bool found = false;
    for (size_t p = 0; p < points.size(); p++) {
      Match &f = frags[points[p].fragment];
      if (points[p].side == 0) {
        int maxIndex = 0, maxValue = 0;
        if (tree[0].maxScoreNode != -1 && find_max(0, key((int)p), maxValue, maxIndex)) {
          f.prev = points[(size_t)maxIndex].fragment;
          f.score = score(maxIndex) + f.score;
        } else {
          f.prev = -1;
        }
      } else {
        activate((int)p);
        if (!found || score((int)best) < score((int)p)) {
          best = p;
          found = true;
        }
      }
    }
    if (!found) return;
    int f = points[best].fragment;
    while (f != -1) {
      out.push_back((uint32_t)f);
      f = frags[f].prev;
    }
    std::reverse(out.begin(), out.end());
  }
};

// ---------------------------------------------------------------------------------------------------------------
// clusters of anchors (FindMaxInterval.h)
// ---------------------------------------------------------------------------------------------------------------
// RemoveOverlappingAnchors + RemoveZeroLengthAnchors; the vector keeps its physical length, the logical size is returned
uint32_t remove_overlapping_anchors(std::vector<Match> &list, uint32_t size) {
  if (size == 0) return 0;
  for (int m = (int)size - 1; m > 0; m--) {
    int n = m - 1;
    while (n > 0 && list[n].t == list[m].t) n--;
    bool mergeFound = false;
    const int ni = n;
    while (!mergeFound && n > 0 && list[n].t == list[ni].t) {
      if (list[n].q < list[m].q && list[n].t < list[m].t && list[n].l + list[n].q >= list[m].l + list[m].q &&
          list[n].l + list[n].t >= list[m].l + list[m].t) {
        list[m].l = 0;
        mergeFound = true;
      }
      n--;
    }
  }
  uint32_t cur = 0;
  for (uint32_t m = 0; m < size; m++)
    if (list[m].l > 0) list[cur++] = list[m];
  return cur;
}

inline uint32_t num_remaining_bases(uint32_t curPos, uint32_t intervalLength) {
  return curPos > intervalLength ? 0 : intervalLength - curPos;
}

// AdvanceIndexToPastInterval (:230-265)
void advance_index(const Genome &g, const std::vector<Match> &pos, uint32_t nPos, uint32_t intervalLength,
                   uint32_t startIndex, uint32_t startBoundary, uint32_t &index, uint32_t &indexBoundary) {
  if (index >= nPos) return;
  indexBoundary = start_boundary(g, pos[index].t);
  while (index < nPos && pos[index].t - pos[startIndex].t < intervalLength && indexBoundary == startBoundary) {
    index++;
    if (index < nPos) indexBoundary = start_boundary(g, pos[index].t);
  }
}

inline int sum_anchors(const std::vector<Match> &pos, uint32_t start, uint32_t end) {
  int sum = 0;
  for (uint32_t i = start; i < end; i++) sum += (int)pos[i].l;
  return sum;
}

// StoreLargestIntervals (:394-640), params.overlap == false, params.noSelf == false.  pos has nPos + 1 elements: the
// reference reads pos[nPos] (:493, :581, :617).
void store_largest_intervals(const Genome &g, const std::vector<Match> &pos, uint32_t nPos, uint32_t intervalLength,
                             int minSize, std::vector<uint32_t> &start, std::vector<uint32_t> &end) {
  if (nPos == 0) return;
  if (nPos == 1) {
    start.push_back(0);
    end.push_back(0);
    return;
  }
  uint32_t cur = 0, endIndex = 1;
  uint32_t curBoundary = start_boundary(g, pos[cur].t);
  uint32_t endIndexBoundary = endIndex < nPos ? start_boundary(g, pos[endIndex].t) : start_boundary(g, pos[cur].t);
  (void)num_remaining_bases;
  advance_index(g, pos, nPos, intervalLength, cur, curBoundary, endIndex, endIndexBoundary);
  uint32_t maxStart = cur, maxEndIndex = endIndex;
  int maxSize = sum_anchors(pos, cur, endIndex);
  int curSize = maxSize;
  if (curSize > minSize) {
    start.push_back(cur);
    end.push_back(endIndex);
  }
  while (cur < nPos) {
    if (pos[cur].t >= pos[maxStart].t && maxEndIndex > 0 && pos[cur].t < pos[maxEndIndex].t &&
        curBoundary == endIndexBoundary) {
      if (curSize > maxSize) {
        maxSize = curSize;
        maxStart = cur;
        maxEndIndex = endIndex;
      }
    } else {
      if (maxSize > minSize) {
        start.push_back(maxStart);
        end.push_back(maxEndIndex);
      }
      maxStart = cur;
      maxEndIndex = endIndex;
      maxSize = curSize;
    }
    if (curBoundary != endIndexBoundary) {
      cur = endIndex;
      curBoundary = endIndexBoundary;
      if (endIndex < nPos) {
        endIndex++;
        advance_index(g, pos, nPos, intervalLength, cur, curBoundary, endIndex, endIndexBoundary);
        maxSize = 0;
        maxStart = cur;
        maxEndIndex = endIndex;
      }
      maxSize = curSize = sum_anchors(pos, cur, endIndex);
    } else {
      while (cur < endIndex && pos[cur].t + intervalLength < pos[endIndex].t) {
        curSize -= (int)pos[cur].l;
        cur++;
      }
      if (cur >= nPos) break;
      curSize += (int)pos[endIndex].l;
      endIndex++;
    }
    if (endIndex > nPos) break;
    curBoundary = start_boundary(g, pos[cur].t);
    endIndexBoundary = start_boundary(g, pos[endIndex].t);
  }
  if (curSize > minSize) {
    start.push_back(maxStart);
    end.push_back(maxEndIndex);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// the interval set (WeightedInterval.h)
// ---------------------------------------------------------------------------------------------------------------
struct Interval {
  uint32_t size, start, end, qStart, qEnd, qLength;
  int readIndex;
  float pValue;
  int nAnchors, totalAnchorSize;
  std::vector<uint32_t> chain;  // WeightedInterval::matches as (t, q, l) triplets -- only kept when the caller asks for them
  int64_t cluster = -1;         // (device route: the cluster of anchors the interval came from)
};
struct IntervalByPValue {  // CompareWeightedIntervalByPValue (:121-131)
  bool operator()(const Interval &a, const Interval &b) const {
    if (a.pValue != b.pValue) return a.pValue < b.pValue;
    return a.start < b.start;
  }
};
typedef std::multiset<Interval, IntervalByPValue> IntervalSet;

// WeightedIntervalSet::insert (:189-268)
bool insert_interval(IntervalSet &set, size_t maxSize, const Interval &intv) {
  bool isContained = false;
  IntervalSet::iterator it = set.begin();
  while (it != set.end() && !isContained) {
    if (intv.qStart >= it->qStart && intv.qEnd <= it->qEnd && intv.start >= it->start && intv.end <= it->end &&
        intv.readIndex == it->readIndex && intv.pValue >= it->pValue) {
      isContained = true;
    } else if (it->start >= intv.start && it->end <= intv.end && it->qStart >= intv.qStart && it->qEnd <= intv.qEnd &&
               it->readIndex == intv.readIndex && it->pValue >= intv.pValue) {
      it = set.erase(it);
    } else {
      ++it;
    }
  }
  if (set.size() >= maxSize && maxSize > 0) {
    IntervalSet::iterator last = set.end();
    --last;
    if (last->pValue < intv.pValue) return false;
  }
  if (isContained) return false;
  bool addInsert = false;
  if (set.empty()) {
    addInsert = true;
  } else {
    IntervalSet::iterator last = set.end();
    --last;
    if (set.size() < maxSize || last->pValue > intv.pValue) {
      addInsert = true;
      if (maxSize != 0 && set.size() >= maxSize && set.size() > 0) set.erase(last);
    }
  }
  if (addInsert) set.insert(intv);
  return true;
}

// WeightedIntervalSet::RemoveContained (:133-181), maxRatio 0.9
void remove_contained(IntervalSet &set) {
  const float maxRatio = 0.9;
  IntervalSet::iterator it = set.begin();
  while (it != set.end()) {
    uint32_t curStart = it->qStart, curEnd = it->qEnd;
    const int curAnchorSize = it->totalAnchorSize;
    if (it->readIndex == 1) {
      curStart = it->qLength - it->qEnd;
      curEnd = it->qLength - it->qStart;
    }
    IntervalSet::iterator nextIt = it;
    ++nextIt;
    while (nextIt != set.end()) {
      float nextStart = nextIt->qStart, nextEnd = nextIt->qEnd;
      if (nextIt->readIndex == 1) {
        nextEnd = nextIt->qLength - nextIt->qStart;
        nextStart = nextIt->qLength - nextIt->qEnd;
      }
      float overlapRatio = 0;
      if ((nextStart >= curStart && nextEnd < curEnd) || (nextStart <= curStart && nextEnd >= curEnd)) {
        overlapRatio = 1;
      } else if (nextEnd < curStart || nextStart > curEnd) {
        overlapRatio = 0;
      } else if (nextStart < curStart && nextEnd > curStart) {
        overlapRatio = (nextEnd - curStart) * 2 / (nextEnd - nextStart + curEnd - curStart);
      } else if (nextStart < curEnd && nextEnd > curEnd) {
        overlapRatio = (curEnd - nextStart) * 2 / (nextEnd - nextStart + curEnd - curStart);
      }
      const float anchorSizeRatio = ((float)nextIt->totalAnchorSize) / curAnchorSize;
      if (nextStart >= curStart && nextEnd <= curEnd && (overlapRatio > maxRatio || anchorSizeRatio < 0.25)) {
        nextIt = set.erase(nextIt);
      } else {
        ++nextIt;
      }
    }
    ++it;
  }
}

// FindMaxIncreasingInterval (:661-843) for one strand's list.  pos: nPos + 1 elements (the last one is what the
// reference finds past the end of its vector).
void find_max_increasing_interval(const Genome &g, int readDir, std::vector<Match> &pos, uint32_t nPos,
                                  uint32_t intervalLength, size_t nBest, const blasr_hip_interval_params &P,
                                  uint32_t queryLength, IntervalSet &queue, Chainer &chainer, bool keepChains) {
  if (nPos == 0) return;
  int noOvpLisSize = 0, noOvpLisNBases = 0;
  std::vector<uint32_t> start, end, lisIndices;
  std::vector<Match> lis;
  store_largest_intervals(g, pos, nPos, intervalLength, P.min_interval, start, end);
  for (size_t posi = 0; posi < start.size(); posi++) {
    lis.clear();
    lisIndices.clear();
    const uint32_t cur = start[posi], endIndex = end[posi];
    if (endIndex - cur == 1)
      lisIndices.push_back(0);
    else
      chainer.chain(&pos[cur], endIndex - cur, lisIndices);
    for (size_t i = 0; i < lisIndices.size(); i++) lis.push_back(pos[lisIndices[i] + cur]);  // StoreLIS (:650-657)
    int lisSize = 0;  // MatchListWeight (:171-179)
    for (size_t i = 0; i < lis.size(); i++) lisSize += (int)lis[i].l;
    float lisPValue = 0;
    if (!lis.empty()) lisPValue = lis_p_value(g, lis, noOvpLisNBases, noOvpLisSize);
    noOvpLisSize = (int)lis.size();
    for (size_t i = 0; i < lis.size(); i++) noOvpLisNBases += (int)lis[i].l;
    float lisWeight = 0;  // LISSizeWeightor (LISSizeWeightor.h:9-18): a float sum of the lengths
    for (size_t i = 0; i < lis.size(); i++) lisWeight += lis[i].l;
    if (lisPValue < P.max_p_value && lisSize > 0 && noOvpLisNBases > P.min_interval) {
      const Match &first = lis[0], &last = lis[lis.size() - 1];
      Interval iv;
      iv.size = (uint32_t)(int)lisWeight;
      iv.start = first.t;
      iv.end = last.t + last.l;
      iv.readIndex = readDir;
      iv.pValue = lisPValue;
      iv.qStart = first.q;
      iv.qEnd = last.q + last.l;
      iv.qLength = queryLength;
      iv.nAnchors = noOvpLisSize;
      iv.totalAnchorSize = noOvpLisNBases;
      if (keepChains) {
        iv.chain.reserve(3 * lis.size());
        for (size_t i = 0; i < lis.size(); i++) iv.chain.push_back(lis[i].t), iv.chain.push_back(lis[i].q), iv.chain.push_back(lis[i].l);
      }
      insert_interval(queue, nBest, iv);
    }
  }
}

void intervals_of_read(const Genome &g, const blasr_hip_read &rd, const blasr_hip_anchor *anchors, const int64_t *off,
                       const blasr_hip_anchor *stale, const blasr_hip_interval_params &P, int readIndex,
                       std::vector<blasr_hip_interval> &out, std::vector<float> &pValues,
                       std::vector<std::vector<uint32_t>> *chains) {
  // Blasr.cpp:2402-2406
  const uint32_t span = rd.subread_end - rd.subread_start;
  uint32_t intervalLength = (uint32_t)(span * (1 + P.indel_rate));
  const uint32_t maxGapLength = 50000;
  if (intervalLength - span > maxGapLength) intervalLength = span + maxGapLength;
  IntervalSet queue;
  Chainer chainer;
  std::vector<Match> pos;
  for (int strand = 0; strand < 2; strand++) {
    const int64_t n = off[strand + 1] - off[strand];
    pos.resize((size_t)n + 1);
    for (int64_t i = 0; i < n; i++) {
      const blasr_hip_anchor &a = anchors[off[strand] + i];
      pos[(size_t)i] = Match{a.t, a.q, a.l, 0, -1};
    }
    pos[(size_t)n] = stale ? Match{stale[strand].t, stale[strand].q, stale[strand].l, 0, -1} : Match{0, 0, 0, 0, -1};
    const uint32_t nPos = remove_overlapping_anchors(pos, (uint32_t)n);
    find_max_increasing_interval(g, strand, pos, nPos, intervalLength, (size_t)P.n_candidates, P, rd.length, queue,
                                 chainer, chains != nullptr);
  }
  if (P.remove_contained_intervals) remove_contained(queue);
  for (IntervalSet::iterator it = queue.begin(); it != queue.end(); ++it) {
    blasr_hip_interval iv;
    iv.start = it->start;
    iv.end = it->end;
    iv.q_start = it->qStart;
    iv.q_end = it->qEnd;
    iv.read = readIndex;
    iv.strand = it->readIndex;
    out.push_back(iv);
    pValues.push_back(it->pValue);
    if (chains) chains->push_back(it->chain);
  }
}

}  // namespace

// The END of FindMaxIncreasingInterval for a block of reads (FindMaxInterval.h:776-840) from the chains the device
// found (intervals_device.hip): per cluster of anchors, in StoreLargestIntervals' order, the chain's first and last anchor,
// the sum of its anchors' lengths and their number -> the chain's p-value (ComputeLISPValue: double-precision logarithms
// of the host's libm, which is why this part stays here), the tests of :805-825, and the read's WeightedIntervalSet.
int blasr_hip_intervals_from_clusters(const uint8_t *genome, int64_t genome_length, const uint32_t *seq_start_pos,
                                      int32_t n_seq_pos, const int32_t *count_table, int32_t count_tuple_size,
                                      int32_t n_reads, const blasr_hip_read *reads, const blasr_hip_iv_cluster *clusters,
                                      const int64_t *cluster_off, const blasr_hip_interval_params *params,
                                      blasr_hip_interval *intervals, float *p_values, int64_t capacity,
                                      int64_t *n_intervals, int64_t *read_interval_off, int64_t *cluster_of_interval) {
  Genome g{genome, genome_length, seq_start_pos, n_seq_pos, count_table, count_tuple_size};
  const blasr_hip_interval_params &P = *params;
  // a thread takes a run of consecutive reads and writes what it finds back to back into its own arrays (a vector or three
  // per READ were 300 k allocations a call at configs[2]: most of this function's 20 ms)
  struct Found {
    std::vector<blasr_hip_interval> ivs;
    std::vector<float> p;
    std::vector<int64_t> cluster;
    std::vector<int32_t> perRead;  // intervals of each read of the run
  };
  int threads = blasr_hip_host_threads(P.threads > 0 ? P.threads : (int)std::min(32u, std::max(1u, std::thread::hardware_concurrency())));
  threads = std::max(1, std::min(threads, n_reads));
  std::vector<Found> found((size_t)threads);
  const int32_t perThread = (n_reads + threads - 1) / threads;
  auto work = [&](int tid) {
    Found &f = found[(size_t)tid];
    const int32_t r0 = (int32_t)std::min<int64_t>((int64_t)tid * perThread, n_reads);
    const int32_t r1 = (int32_t)std::min<int64_t>((int64_t)r0 + perThread, n_reads);
    f.perRead.reserve((size_t)(r1 - r0));
    for (int32_t r = r0; r < r1; r++) {
      IntervalSet queue;
      for (int strand = 0; strand < 2; strand++) {
        int noOvpLisNBases = 0;  // (declared outside the loop over the clusters in the reference, :690)
        for (int64_t c = cluster_off[2 * r + strand]; c < cluster_off[2 * r + strand + 1]; c++) {
          const blasr_hip_iv_cluster &k = clusters[c];
          const int lisSize = k.sum_l;  // MatchListWeight
          float lisPValue = 0;
          if (k.count > 0) {
            lisPValue = lis_p_value_core(g, k.first_t, k.first_l, (size_t)k.count);
            noOvpLisNBases = k.sum_l;
          }
          noOvpLisNBases += k.sum_l;  // (:822-825: added to again)
          if (lisPValue < P.max_p_value && lisSize > 0 && noOvpLisNBases > P.min_interval) {
            Interval iv;
            iv.size = (uint32_t)k.sum_l;
            iv.start = k.first_t;
            iv.end = k.last_t + k.last_l;
            iv.readIndex = strand;
            iv.pValue = lisPValue;
            iv.qStart = k.first_q;
            iv.qEnd = k.last_q + k.last_l;
            iv.qLength = reads[r].length;
            iv.nAnchors = k.count;
            iv.totalAnchorSize = noOvpLisNBases;
            iv.cluster = c;
            insert_interval(queue, (size_t)P.n_candidates, iv);
          }
        }
      }
      if (P.remove_contained_intervals) remove_contained(queue);
      int32_t mine = 0;
      for (IntervalSet::iterator it = queue.begin(); it != queue.end(); ++it, mine++) {
        blasr_hip_interval iv;
        iv.start = it->start;
        iv.end = it->end;
        iv.q_start = it->qStart;
        iv.q_end = it->qEnd;
        iv.read = r;
        iv.strand = it->readIndex;
        f.ivs.push_back(iv);
        f.p.push_back(it->pValue);
        if (cluster_of_interval) f.cluster.push_back(it->cluster);
      }
      f.perRead.push_back(mine);
    }
  };
  if (const int trc = blasr_hip_run_threads(threads, work)) return trc;  // (a worker ran out of memory)
  int64_t total = 0;
  for (int tid = 0; tid < threads; tid++) {  // (the threads' runs of reads follow each other)
    const Found &f = found[(size_t)tid];
    const int32_t r0 = (int32_t)std::min<int64_t>((int64_t)tid * perThread, n_reads);
    size_t at = 0;
    for (size_t k = 0; k < f.perRead.size(); k++) {
      read_interval_off[r0 + (int32_t)k] = total;
      for (int32_t i = 0; i < f.perRead[k]; i++, at++, total++)
        if (total < capacity) {
          intervals[total] = f.ivs[at];
          if (p_values) p_values[total] = f.p[at];
          if (cluster_of_interval) cluster_of_interval[total] = f.cluster[at];
        }
    }
  }
  read_interval_off[n_reads] = total;
  *n_intervals = total;
  return total > capacity ? BLASR_HIP_ERR_CAPACITY : BLASR_HIP_OK;
}

extern "C" {

int blasr_hip_interval_default_params(blasr_hip_interval_params *p) try {
  if (!p) return BLASR_HIP_ERR_INVALID;
  p->n_candidates = 10;
  p->min_interval = 100;
  p->indel_rate = 0.30f;
  p->max_p_value = (float)log(0.5);
  p->remove_contained_intervals = 0;
  p->threads = 0;
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_count_tuples(const uint8_t *genome, int64_t genome_length, int32_t tuple_size, int32_t *count_table) try {
  if (!genome || genome_length < 0 || tuple_size < 1 || tuple_size > 14 || !count_table) return BLASR_HIP_ERR_INVALID;
  const int64_t entries = 1ll << (2 * tuple_size);
  memset(count_table, 0, sizeof(int32_t) * (size_t)entries);
  // TupleCountTable::AddSequenceTupleCountsLR (TupleCountTable.h:63-73): every position whose tuple is all ACGT
  const uint64_t mask = (uint64_t)entries - 1;
  uint64_t tuple = 0;
  int valid = 0;
  for (int64_t i = 0; i < genome_length; i++) {
    if (is_actg(genome[i])) {
      tuple = ((tuple << 2) + (uint64_t)two_bit(genome[i])) & mask;
      if (valid < tuple_size) valid++;  // (saturating: a run of ACGT can be longer than an int counts)
    } else {
      valid = 0;
      tuple = 0;
    }
    if (valid >= tuple_size) count_table[tuple]++;
  }
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_find_intervals_chains(const uint8_t *genome, int64_t genome_length, const uint32_t *seq_start_pos,
                                    int32_t n_seq_pos, const int32_t *count_table, int32_t count_tuple_size,
                                    int32_t n_reads, const blasr_hip_read *reads, const blasr_hip_anchor *anchors,
                                    const int64_t *list_off, const blasr_hip_anchor *stale,
                                    const blasr_hip_interval_params *params, blasr_hip_interval *intervals,
                                    float *p_values, int64_t capacity, int64_t *n_intervals, int64_t *read_interval_off,
                                    uint32_t *chains, int64_t chains_capacity, int64_t *chains_used, int64_t *chain_off) try {
  const bool keep = chains_used != nullptr;
  if (keep && (chains_capacity < 0 || (chains_capacity > 0 && !chains) || (capacity > 0 && !chain_off))) return BLASR_HIP_ERR_INVALID;
  if (!genome || genome_length <= 0 || genome_length >= (1ll << 32) || !seq_start_pos || n_seq_pos < 1 || !count_table ||
      count_tuple_size < 1 || count_tuple_size > 14 || n_reads < 0 || !params || !n_intervals || !read_interval_off ||
      capacity < 0 || (capacity > 0 && !intervals) || (n_reads > 0 && (!reads || !list_off)))
    return BLASR_HIP_ERR_INVALID;
  if (params->n_candidates < 0 || params->min_interval < 0) return BLASR_HIP_ERR_INVALID;
  for (int32_t r = 0; r < n_reads; r++)
    if (reads[r].subread_start > reads[r].subread_end || reads[r].subread_end > reads[r].length ||
        list_off[2 * r + 1] < list_off[2 * r] || list_off[2 * r + 2] < list_off[2 * r + 1] ||
        (list_off[2 * r + 2] > list_off[2 * r] && !anchors))
      return BLASR_HIP_ERR_INVALID;
  Genome g{genome, genome_length, seq_start_pos, n_seq_pos, count_table, count_tuple_size};
  std::vector<std::vector<blasr_hip_interval>> perRead((size_t)n_reads);
  std::vector<std::vector<float>> perReadP((size_t)n_reads);
  std::vector<std::vector<std::vector<uint32_t>>> perReadChains(keep ? (size_t)n_reads : 0);
  int threads = blasr_hip_host_threads(params->threads > 0 ? params->threads : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
  threads = std::max(1, std::min(threads, n_reads));
  auto work = [&](int tid) {
    for (int32_t r = tid; r < n_reads; r += threads)
      intervals_of_read(g, reads[r], anchors, list_off + 2 * r, stale ? stale + 2 * r : nullptr, *params, r,
                        perRead[(size_t)r], perReadP[(size_t)r], keep ? &perReadChains[(size_t)r] : nullptr);
  };
  if (const int trc = blasr_hip_run_threads(threads, work)) return trc;  // (a worker ran out of memory)
  int64_t total = 0, words = 0;
  for (int32_t r = 0; r < n_reads; r++) {
    read_interval_off[r] = total;
    for (size_t i = 0; i < perRead[(size_t)r].size(); i++, total++) {
      if (total < capacity) {
        intervals[total] = perRead[(size_t)r][i];
        if (p_values) p_values[total] = perReadP[(size_t)r][i];
      }
      if (keep) {
        const std::vector<uint32_t> &c = perReadChains[(size_t)r][i];
        if (total < capacity) chain_off[total] = words / 3;
        if (words + (int64_t)c.size() <= 3 * chains_capacity && !c.empty()) memcpy(chains + words, c.data(), sizeof(uint32_t) * c.size());
        words += (int64_t)c.size();
      }
    }
  }
  read_interval_off[n_reads] = total;
  *n_intervals = total;
  if (keep) {
    if (total <= capacity && capacity > 0) chain_off[total] = words / 3;
    *chains_used = words / 3;
    if (words > 3 * chains_capacity) return BLASR_HIP_ERR_CAPACITY;
  }
  return total > capacity ? BLASR_HIP_ERR_CAPACITY : BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_find_intervals(const uint8_t *genome, int64_t genome_length, const uint32_t *seq_start_pos,
                             int32_t n_seq_pos, const int32_t *count_table, int32_t count_tuple_size, int32_t n_reads,
                             const blasr_hip_read *reads, const blasr_hip_anchor *anchors, const int64_t *list_off,
                             const blasr_hip_anchor *stale, const blasr_hip_interval_params *params,
                             blasr_hip_interval *intervals, float *p_values, int64_t capacity, int64_t *n_intervals,
                             int64_t *read_interval_off) try {
  return blasr_hip_find_intervals_chains(genome, genome_length, seq_start_pos, n_seq_pos, count_table, count_tuple_size,
                                         n_reads, reads, anchors, list_off, stale, params, intervals, p_values, capacity,
                                         n_intervals, read_interval_off, nullptr, 0, nullptr, nullptr);
} BLASR_HIP_CATCH_CODE

}  // extern "C"
