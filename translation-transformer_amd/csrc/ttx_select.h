// Which of several EQUAL maxima `tensor.topk(1)` returns — restated so that the HIP path picks the same one.
//
// The reference chooses the best draft of a candidate with `n_accepted_in_drafts.topk(1, dim=-1)`
// (src/decoding/speculative_decoding.py:553) and, in smart-drafts mode, with `topk_in_each_group(..., k=1, pad=-1)`
// (:779-784 -> :225 `score_2d.topk(k=k)`).  The scores are small integers (accepted draft tokens), so ties are the
// norm, and two drafts with the same count need not hold the same tokens (a token is "accepted" when it is among the
// n_best nucleus tokens of its position) — the choice changes which hypotheses come out.  torch's CPU top-k
// (aten/src/ATen/native/cpu/TopKImpl.h, torch 2.x) fills a vector of (value, index) pairs per slice and calls
//     k * 64 <= n :  std::partial_sort(begin, begin + k, end, greater-by-value)
//     otherwise   :  std::nth_element(begin, begin + k - 1, end, greater-by-value)
// whose treatment of equal keys is whatever libstdc++'s heap-select / introselect happen to do.  Both are
// deterministic, and below they are restated for k = 1 step for step (median-of-three to the front, unguarded
// partition, final insertion sort on <= 3 elements, heap-select when the depth limit 2*floor(log2 n) runs out).
// tests/test_select_host.py checks the restatement against torch.topk itself on the CPU for every n in 1..130.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define TTX_HD __host__ __device__ __forceinline__
#else
#define TTX_HD inline
#endif

namespace ttxsel {

TTX_HD void swap_pair(long long* v, int* ix, int a, int b) {
  const long long tv = v[a]; v[a] = v[b]; v[b] = tv;
  const int ti = ix[a]; ix[a] = ix[b]; ix[b] = ti;
}

// std::__heap_select(first, first + 1, last) on [lo, hi): the front element is replaced by every later element that
// compares strictly greater, the displaced value taking that element's place (std::__pop_heap on a heap of one).
TTX_HD void heap_select_one(long long* v, int* ix, int lo, int hi) {
  for (int i = lo + 1; i < hi; ++i)
    if (v[i] > v[lo]) swap_pair(v, ix, i, lo);
}

// Index (into the ORIGINAL slice) of the element torch.topk(k=1, largest=True) reports for the n values in `v`.
// `v` and `ix` are scratch of n entries each; `v` holds the values on entry and both are permuted.
TTX_HD int topk1_index(long long* v, int* ix, int n) {
  for (int i = 0; i < n; ++i) ix[i] = i;
  if (n <= 1) return 0;
  if (64 <= n) {                               // use_partial_sort: k * 64 <= n
    heap_select_one(v, ix, 0, n);
    return ix[0];
  }
  int lo = 0, hi = n;
  int depth = 0;
  for (int t = n; t > 1; t >>= 1) ++depth;     // std::__lg(n)
  depth *= 2;
  while (hi - lo > 3) {
    if (depth == 0) {
      heap_select_one(v, ix, lo, hi);
      return ix[0];                            // nth == begin, and lo never moves (see below)
    }
    --depth;
    // std::__unguarded_partition_pivot: median of (lo+1, mid, hi-1) to lo, partition (lo+1, hi) around it
    const int mid = lo + (hi - lo) / 2;
    const int a = lo + 1, b = mid, c = hi - 1;
    int med;
    if (v[a] > v[b]) {
      if (v[b] > v[c]) med = b;
      else if (v[a] > v[c]) med = c;
      else med = a;
    } else if (v[a] > v[c]) med = a;
    else if (v[b] > v[c]) med = c;
    else med = b;
    swap_pair(v, ix, lo, med);
    int first = lo + 1, last = hi;
    for (;;) {
      while (v[first] > v[lo]) ++first;
      --last;
      while (v[lo] > v[last]) --last;
      if (!(first < last)) break;
      swap_pair(v, ix, first, last);
      ++first;
    }
    // cut = first; nth (= begin = lo) < cut always, so the search continues in [lo, cut)
    hi = first;
  }
  // std::__insertion_sort on the remaining <= 3 elements (stable for equal keys)
  for (int i = lo + 1; i < hi; ++i) {
    const long long tv = v[i];
    const int ti = ix[i];
    if (tv > v[lo]) {
      for (int j = i; j > lo; --j) { v[j] = v[j - 1]; ix[j] = ix[j - 1]; }
      v[lo] = tv; ix[lo] = ti;
    } else {
      int j = i;
      while (tv > v[j - 1]) { v[j] = v[j - 1]; ix[j] = ix[j - 1]; --j; }
      v[j] = tv; ix[j] = ti;
    }
  }
  return ix[0];
}

}  // namespace ttxsel
