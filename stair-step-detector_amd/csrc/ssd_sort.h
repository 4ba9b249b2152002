/*
 * ssd_sort.h — std::sort as GNU libstdc++ 11 performs it (bits/stl_algo.h __sort: median-of-three introsort down to
 * runs of 16, heapsort when the depth limit 2*lg(n) is exhausted, then one insertion sort), restated for keys that
 * are (double distance, int index) compared by distance only.
 *
 * Why: the reference picks the vertical-edge point at rank 2n/3 of the distance-sorted probe points with
 * std::ranges::sort (segmentation.cpp:724), which is not stable — among points at EXACTLY equal distance the one that
 * lands on that rank is whatever this algorithm leaves there.  The kernels select by rank (no sort) and fall back to
 * this restatement only when the selected distance is duplicated and the duplicates would give different lines.
 * Compiled for host and device from this one definition; tests compare it with std::sort itself (oracle, host).
 *
 * Provenance: written from the algorithm's published description (Musser's introsort as libstdc++ configures it:
 * threshold 16, depth limit 2 * floor(lg n), median of first + 1 / middle / last - 1 moved to the front, unguarded
 * partition, heap-select + sort-heap fallback, final guarded / unguarded insertion sort).  It restates the ALGORITHM and
 * its constants so that ties land where std::sort leaves them; it contains no text of libstdc++.
 */
#ifndef SSD_SORT_H_
#define SSD_SORT_H_

#include <hip/hip_runtime.h>

namespace ssd
{

struct SortKeys
{
  double *d;     /* distance */
  int *i;        /* payload */
};

__host__ __device__ inline void gs_swap(const SortKeys &k, int a, int b)
{
  const double td = k.d[a]; k.d[a] = k.d[b]; k.d[b] = td;
  const int ti = k.i[a]; k.i[a] = k.i[b]; k.i[b] = ti;
}

/* __move_median_to_first(result, a, b, c) */
__host__ __device__ inline void gs_median_to_first(const SortKeys &k, int result, int a, int b, int c)
{
  if(k.d[a] < k.d[b])
  {
    if(k.d[b] < k.d[c]) gs_swap(k, result, b);
    else if(k.d[a] < k.d[c]) gs_swap(k, result, c);
    else gs_swap(k, result, a);
  }
  else if(k.d[a] < k.d[c]) gs_swap(k, result, a);
  else if(k.d[b] < k.d[c]) gs_swap(k, result, c);
  else gs_swap(k, result, b);
}

/* __unguarded_partition(first, last, pivot) */
__host__ __device__ inline int gs_partition(const SortKeys &k, int first, int last, int pivot)
{
  while(true)
  {
    while(k.d[first] < k.d[pivot]) ++first;
    --last;
    while(k.d[pivot] < k.d[last]) --last;
    if(!(first < last)) return first;
    gs_swap(k, first, last);
    ++first;
  }
}

/* __push_heap(first, hole, top, value) with comp(parent, value) */
__host__ __device__ inline void gs_push_heap(const SortKeys &k, int first, int hole, int top, double vd, int vi)
{
  int parent = (hole - 1) / 2;
  while(hole > top && k.d[first + parent] < vd)
  {
    k.d[first + hole] = k.d[first + parent]; k.i[first + hole] = k.i[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  k.d[first + hole] = vd; k.i[first + hole] = vi;
}

/* __adjust_heap(first, hole, len, value) */
__host__ __device__ inline void gs_adjust_heap(const SortKeys &k, int first, int hole, int len, double vd, int vi)
{
  const int top = hole;
  int second = hole;
  while(second < (len - 1) / 2)
  {
    second = 2 * (second + 1);
    if(k.d[first + second] < k.d[first + (second - 1)])
      second--;
    k.d[first + hole] = k.d[first + second]; k.i[first + hole] = k.i[first + second];
    hole = second;
  }
  if((len & 1) == 0 && second == (len - 2) / 2)
  {
    second = 2 * (second + 1);
    k.d[first + hole] = k.d[first + (second - 1)]; k.i[first + hole] = k.i[first + (second - 1)];
    hole = second - 1;
  }
  gs_push_heap(k, first, hole, top, vd, vi);
}

/* __partial_sort(first, last, last): __heap_select = __make_heap (nothing beyond middle), then __sort_heap */
__host__ __device__ inline void gs_heapsort(const SortKeys &k, int first, int last)
{
  const int len = last - first;
  if(len >= 2)
  {
    int parent = (len - 2) / 2;
    while(true)
    {
      const double vd = k.d[first + parent];
      const int vi = k.i[first + parent];
      gs_adjust_heap(k, first, parent, len, vd, vi);
      if(parent == 0) break;
      parent--;
    }
  }
  while(last - first > 1)
  {
    --last;
    /* __pop_heap(first, last, last): value = *result; *result = *first; adjust(first, 0, last - first, value) */
    const double vd = k.d[last];
    const int vi = k.i[last];
    k.d[last] = k.d[first]; k.i[last] = k.i[first];
    gs_adjust_heap(k, first, 0, last - first, vd, vi);
  }
}

/* __unguarded_linear_insert(last) */
__host__ __device__ inline void gs_linear_insert(const SortKeys &k, int last)
{
  const double vd = k.d[last];
  const int vi = k.i[last];
  int next = last - 1;
  while(vd < k.d[next])
  {
    k.d[last] = k.d[next]; k.i[last] = k.i[next];
    last = next;
    --next;
  }
  k.d[last] = vd; k.i[last] = vi;
}

/* __insertion_sort(first, last) */
__host__ __device__ inline void gs_insertion_sort(const SortKeys &k, int first, int last)
{
  if(first == last) return;
  for(int i = first + 1; i != last; ++i)
  {
    if(k.d[i] < k.d[first])
    {
      const double vd = k.d[i];
      const int vi = k.i[i];
      for(int j = i; j > first; --j) { k.d[j] = k.d[j - 1]; k.i[j] = k.i[j - 1]; }     /* move_backward(first, i, i + 1) */
      k.d[first] = vd; k.i[first] = vi;
    }
    else
      gs_linear_insert(k, i);
  }
}

/* std::sort(first = 0, last = n) on the keys; n <= 2^15.  The recursion is an explicit stack of at most kSortStack ranges (the depth
 * limit bounds it) in storage the caller provides - three arrays of kSortStack ints: k_outline hands over LDS (round 6: as local
 * arrays they were the kernel's only scratch memory, 496 bytes per thread for a path one lane in a thousand blocks takes). */
constexpr int kSortStack = 40;
__host__ __device__ inline void gnu_sort_on(const SortKeys &k, int n, int *stFirst, int *stLast, int *stDepth)
{
  if(n <= 0) return;
  int lg = 0;
  for(int v = n; v > 1; v >>= 1) lg++;                    /* std::__lg */
  /* __introsort_loop(0, n, 2 * lg): the recursive call takes [cut, last), the loop continues with [first, cut) */
  int sp = 0;
  stFirst[0] = 0; stLast[0] = n; stDepth[0] = 2 * lg; sp = 1;
  while(sp > 0)
  {
    sp--;
    int first = stFirst[sp], last = stLast[sp], depth = stDepth[sp];
    /* ranges still to be looped over after the recursive calls are pushed in reverse so that execution order is the
     * library's: (cut,last) fully first, then (first,cut).  The order does not matter for the result (disjoint
     * ranges), only which ranges get which depth. */
    while(last - first > 16)
    {
      if(depth == 0)
      {
#ifdef SSD_SORT_TRACE
        SSD_SORT_TRACE(first, last);
#endif
        gs_heapsort(k, first, last);
        break;
      }
      --depth;
      const int mid = first + (last - first) / 2;
      gs_median_to_first(k, first, first + 1, mid, last - 1);
      const int cut = gs_partition(k, first + 1, last, first);
      stFirst[sp] = cut; stLast[sp] = last; stDepth[sp] = depth; sp++;        /* the recursive call */
      last = cut;
    }
  }
  /* __final_insertion_sort */
  if(n > 16)
  {
    gs_insertion_sort(k, 0, 16);
    for(int i = 16; i != n; ++i)
      gs_linear_insert(k, i);
  }
  else
    gs_insertion_sort(k, 0, n);
}
/* the same with the stack in local arrays (host, test hooks) */
__host__ __device__ inline void gnu_sort(const SortKeys &k, int n)
{
  int stFirst[kSortStack], stLast[kSortStack], stDepth[kSortStack];
  gnu_sort_on(k, n, stFirst, stLast, stDepth);
}

} // namespace ssd

#endif /* SSD_SORT_H_ */
