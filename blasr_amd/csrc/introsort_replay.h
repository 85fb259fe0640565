// introsort_replay.h -- libstdc++'s std::sort (GCC 11.4 bits/stl_algo.h: introsort with median-of-three pivots, heapsort
// at the depth limit, a final insertion sort over ranges of up to sixteen) replayed step by step by ONE lane on an
// array in LDS or global memory.  The reference sorts sequences whose keys can tie with std::sort (SDPAlign.h:256, the
// fragments; GlobalChain.h:137, the endpoints of the chained anchors), and which of two tied elements comes first --
// and what follows from it -- is decided by exactly this schedule.
#ifndef BLASR_AMD_INTROSORT_REPLAY_H_
#define BLASR_AMD_INTROSORT_REPLAY_H_
#include <hip/hip_runtime.h>

namespace blasr_hip {

template <class T>
__device__ __forceinline__ void is_swap(T *a, T *b) {
  T t = *a;
  *a = *b;
  *b = t;
}
template <class T, class Less>
__device__ __forceinline__ void is_adjust_heap(T *first, long hole, long len, T value, Less less) {
  const long top = hole;
  long child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (less(first[child], first[child - 1])) child--;
    first[hole] = first[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole = child - 1;
  }
  long parent = (hole - 1) / 2;
  while (hole > top && less(first[parent], value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}
template <class T, class Less>
__device__ __forceinline__ void is_heapsort(T *first, T *last, Less less) {
  const long len = last - first;
  if (len >= 2) {
    long parent = (len - 2) / 2;
    for (;;) {
      T v = first[parent];
      is_adjust_heap(first, parent, len, v, less);
      if (parent == 0) break;
      parent--;
    }
  }
  while (last - first > 1) {
    --last;
    T v = *last;
    *last = *first;
    is_adjust_heap(first, 0, last - first, v, less);
  }
}
template <class T, class Less>
__device__ __forceinline__ void is_unguarded_linear_insert(T *last, Less less) {
  T val = *last;
  T *next = last - 1;
  while (less(val, *next)) {
    *last = *next;
    last = next;
    --next;
  }
  *last = val;
}
template <class T, class Less>
__device__ __forceinline__ void is_insertion_sort(T *first, T *last, Less less) {
  if (first == last) return;
  for (T *i = first + 1; i != last; ++i) {
    if (less(*i, *first)) {
      T val = *i;
      for (T *p = i; p != first; --p) *p = *(p - 1);
      *first = val;
    } else
      is_unguarded_linear_insert(i, less);
  }
}
template <class T, class Less>
__device__ __forceinline__ void introsort_replay(T *first, long n, Less less) {
  if (n <= 0) return;
  long lg = 0;
  for (long m = n; m > 1; m >>= 1) lg++;
  // explicit stack for the recursion on the right partition
  T *stFirst[64], *stLast[64];
  long stDepth[64];
  int sp = 0;
  stFirst[0] = first;
  stLast[0] = first + n;
  stDepth[0] = lg * 2;
  sp = 1;
  while (sp > 0) {
    --sp;
    T *f = stFirst[sp], *l = stLast[sp];
    long depth = stDepth[sp];
    // __introsort_loop(f, l, depth): right parts are pushed and processed first, as the recursion does
    // (the order of processing disjoint ranges does not change the result)
    while (l - f > 16) {
      if (depth == 0) {
        is_heapsort(f, l, less);
        break;
      }
      --depth;
      T *mid = f + (l - f) / 2;
      // __move_median_to_first(f, f+1, mid, l-1)
      T *a = f + 1, *b = mid, *c = l - 1;
      if (less(*a, *b)) {
        if (less(*b, *c)) is_swap(f, b);
        else if (less(*a, *c)) is_swap(f, c);
        else is_swap(f, a);
      } else if (less(*a, *c)) is_swap(f, a);
      else if (less(*b, *c)) is_swap(f, c);
      else is_swap(f, b);
      // __unguarded_partition(f+1, l, f)
      T *lo = f + 1, *hi = l;
      for (;;) {
        while (less(*lo, *f)) ++lo;
        --hi;
        while (less(*f, *hi)) --hi;
        if (!(lo < hi)) break;
        is_swap(lo, hi);
        ++lo;
      }
      stFirst[sp] = lo;
      stLast[sp] = l;
      stDepth[sp] = depth;
      sp++;
      l = lo;
    }
  }
  if (n > 16) {
    is_insertion_sort(first, first + 16, less);
    for (T *i = first + 16; i != first + n; ++i) is_unguarded_linear_insert(i, less);
  } else
    is_insertion_sort(first, first + n, less);
}

}  // namespace blasr_hip
#endif
