// stdsort_replica.h — the exact permutation libstdc++'s std::sort produces, as plain
// host/device code.
//
// connectLimbs*/rtpose.cpp:953-954 orders the candidate limb connections with
//     std::sort(temp.begin(), temp.end(), ColumnCompare());      // lhs[2] > rhs[2]
// std::sort is not stable, so for equal scores (common on synthetic / saturated PAFs) the greedy
// assignment that follows depends on the implementation's data movements.  The reference links
// libstdc++; this file restates libstdc++'s algorithm (bits/stl_algo.h, bits/stl_heap.h:
// introsort with median-of-3 to first, unguarded Hoare partition, threshold 16, heap-sort
// fallback at depth 2*floor(log2 n), final guarded/unguarded insertion sort) so that a single
// GPU lane reproduces the same order.  tests/test_stdsort_replica.py checks it element-for-
// element against the real std::sort (g++ 11) on tie-heavy inputs.
#pragma once

#if defined(__HIPCC__)
#define RTP_HD __host__ __device__
#else
#define RTP_HD
#endif

namespace rtp {

struct Cand {
  float score;  // ColumnCompare key: temp[row][2]
  int ij;       // (i << 16) | j   (1-based peak ordinals, rtpose.cpp:942-943)
};

RTP_HD inline bool cand_less(const Cand& a, const Cand& b) { return a.score > b.score; }  // "comp(a,b)"

RTP_HD inline void cand_swap(Cand& a, Cand& b) {
  Cand t = a;
  a = b;
  b = t;
}

// ---- heap helpers (stl_heap.h) -----------------------------------------------------------
RTP_HD inline void ss_push_heap(Cand* first, int holeIndex, int topIndex, Cand value) {
  int parent = (holeIndex - 1) / 2;
  while (holeIndex > topIndex && cand_less(first[parent], value)) {
    first[holeIndex] = first[parent];
    holeIndex = parent;
    parent = (holeIndex - 1) / 2;
  }
  first[holeIndex] = value;
}

RTP_HD inline void ss_adjust_heap(Cand* first, int holeIndex, int len, Cand value) {
  const int topIndex = holeIndex;
  int secondChild = holeIndex;
  while (secondChild < (len - 1) / 2) {
    secondChild = 2 * (secondChild + 1);
    if (cand_less(first[secondChild], first[secondChild - 1])) secondChild--;
    first[holeIndex] = first[secondChild];
    holeIndex = secondChild;
  }
  if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
    secondChild = 2 * (secondChild + 1);
    first[holeIndex] = first[secondChild - 1];
    holeIndex = secondChild - 1;
  }
  ss_push_heap(first, holeIndex, topIndex, value);
}

RTP_HD inline void ss_heap_sort(Cand* first, int len) {  // __partial_sort(first, last, last)
  if (len >= 2) {                                          // __make_heap
    int parent = (len - 2) / 2;
    while (true) {
      Cand value = first[parent];
      ss_adjust_heap(first, parent, len, value);
      if (parent == 0) break;
      parent--;
    }
  }
  int last = len;  // __sort_heap
  while (last > 1) {
    --last;
    Cand value = first[last];  // __pop_heap(first, last, last)
    first[last] = first[0];
    ss_adjust_heap(first, 0, last, value);
  }
}

// ---- introsort pieces (stl_algo.h) -------------------------------------------------------
RTP_HD inline void ss_move_median_to_first(Cand* v, int result, int a, int b, int c) {
  if (cand_less(v[a], v[b])) {
    if (cand_less(v[b], v[c])) cand_swap(v[result], v[b]);
    else if (cand_less(v[a], v[c])) cand_swap(v[result], v[c]);
    else cand_swap(v[result], v[a]);
  } else if (cand_less(v[a], v[c])) cand_swap(v[result], v[a]);
  else if (cand_less(v[b], v[c])) cand_swap(v[result], v[c]);
  else cand_swap(v[result], v[b]);
}

RTP_HD inline int ss_unguarded_partition(Cand* v, int first, int last, int pivot) {
  while (true) {
    while (cand_less(v[first], v[pivot])) ++first;
    --last;
    while (cand_less(v[pivot], v[last])) --last;
    if (!(first < last)) return first;
    cand_swap(v[first], v[last]);
    ++first;
  }
}

RTP_HD inline void ss_unguarded_linear_insert(Cand* v, int last) {
  Cand val = v[last];
  int next = last - 1;
  while (cand_less(val, v[next])) {
    v[last] = v[next];
    last = next;
    --next;
  }
  v[last] = val;
}

RTP_HD inline void ss_insertion_sort(Cand* v, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (cand_less(v[i], v[first])) {
      Cand val = v[i];
      for (int k = i; k > first; --k) v[k] = v[k - 1];  // move_backward(first, i, i+1)
      v[first] = val;
    } else {
      ss_unguarded_linear_insert(v, i);
    }
  }
}

// std::sort(v, v+n, ColumnCompare()) — in place.
RTP_HD inline void std_sort_replica(Cand* v, int n) {
  if (n <= 0) return;
  // __introsort_loop(first, last, 2*__lg(n)) with the tail recursion on [cut,last) turned into
  // an explicit stack (depth <= 2*lg n <= 64).
  int lg = 0;
  for (int t = n; t > 1; t >>= 1) ++lg;
  // ONE thread of a workgroup runs this on the device (the connect kernel's rare exact path): the explicit stack lives in LDS there.  As
  // private arrays it gave every wave of every launch of the kernel a 784-byte scratch segment (61 MB for the single-launch chain).
#ifdef __HIP_DEVICE_COMPILE__
  __shared__ int stack_first[64], stack_last[64], stack_depth[64];
#else
  int stack_first[64], stack_last[64], stack_depth[64];
#endif
  int sp = 0;
  stack_first[0] = 0; stack_last[0] = n; stack_depth[0] = 2 * lg;
  sp = 1;
  while (sp > 0) {
    --sp;
    int first = stack_first[sp], last = stack_last[sp], depth = stack_depth[sp];
    // The recursive call handles [cut,last) FIRST and then the loop continues on [first,cut).
    // Both sub-ranges are disjoint, so the order of processing does not change the result.
    while (last - first > 16) {
      if (depth == 0) {
        ss_heap_sort(v + first, last - first);
        break;
      }
      --depth;
      const int mid = first + (last - first) / 2;
      ss_move_median_to_first(v, first, first + 1, mid, last - 1);
      const int cut = ss_unguarded_partition(v, first + 1, last, first);
      // defer [cut,last) with the decremented depth; continue on [first,cut)
      stack_first[sp] = cut; stack_last[sp] = last; stack_depth[sp] = depth;
      ++sp;
      last = cut;
    }
  }
  // __final_insertion_sort
  if (n > 16) {
    ss_insertion_sort(v, 0, 16);
    for (int i = 16; i != n; ++i) ss_unguarded_linear_insert(v, i);
  } else {
    ss_insertion_sort(v, 0, n);
  }
}

}  // namespace rtp
