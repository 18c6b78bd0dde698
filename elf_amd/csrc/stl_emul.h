// Exact emulation of two libstdc++ behaviours that are semantically visible in the reference's MCTS.
//
// (1) Iteration order of std::unordered_map<Coord, EdgeInfo> (NodeT::stateActions_,
//     elf/ai/tree_search/tree_search_node.h:310) after the insertions of NodeT::setEvaluation (:186-189).
//     PUCT tie-breaks (:338-341,:370-385), the assignment of Dirichlet draws to edges (:149-154) and
//     MCTSResultT::addActions (tree_search_base.h:248-292) all follow that order.
//     libstdc++ (GCC 11) facts used, all checked against the real container by tests/test_stl_emul.py:
//       * std::hash<unsigned short> is the identity, bucket = key % bucket_count;
//       * a default-constructed map grows 1 -> 13 -> 29 -> 59 -> 127 -> 257 -> 541 buckets, rehashing
//         BEFORE the insert that would exceed load factor 1;
//       * an insert into an empty bucket puts the node at the head of the global list, an insert into a
//         non-empty bucket puts it at the front of that bucket's run; a rehash re-inserts the nodes in
//         their current iteration order under the same two rules.
//     Hence one "epoch" (a sequence inserted into an empty table of nb buckets) iterates as the REVERSE
//     of: buckets in order of first appearance, each bucket's keys in insertion order.
// (2) std::sort with comparator a.second > b.second (go/mcts/mcts.h:292-297): introsort + final insertion
//     sort; not stable, so the order of equal priors is whatever this exact algorithm produces.
//
// Host and device (gfx950) compile the same code; the device uses the serial forms only on rare paths.
#pragma once
#include <stdint.h>
#if !defined(__HIP_DEVICE_COMPILE__)
#include <vector>
#endif

#if defined(__HIPCC__)
#define STL_HD __host__ __device__ __forceinline__
#define STL_HD_MEMBER __host__ __device__ __forceinline__
#else
#define STL_HD static inline
#define STL_HD_MEMBER inline
#endif

namespace stl_emul {

// bucket counts of successive rehashes and the element count each can hold before the next one
constexpr int kNumEpochs = 6;
constexpr STL_HD int epoch_buckets(int e) {
  return e == 0 ? 13 : e == 1 ? 29 : e == 2 ? 59 : e == 3 ? 127 : e == 4 ? 257 : 541;
}

// Serial reference form. keys[0..n) = insertion order (distinct, < 65536), n <= 541.
// order[i] = index into keys of the i-th element in iteration order. tmp: 2*n ints of scratch.
template <typename K>
STL_HD void umap_iteration_order(const K* keys, int n, int* order, int* tmp) {
  int* cur = tmp;        // current iteration order (indices into keys)
  int* seq = tmp + n;    // insertion sequence of this epoch
  int have = 0, done = 0;
  for (int e = 0; e < kNumEpochs && done < n; ++e) {
    const int nb = epoch_buckets(e);
    const int take = (n < nb ? n : nb) - done;
    int m = 0;
    for (int i = 0; i < have; ++i) seq[m++] = cur[i];
    for (int i = 0; i < take; ++i) seq[m++] = done + i;
    done += take;
    // groups in order of first appearance, members in insertion order; then reverse
    int w = 0;
    for (int i = 0; i < m; ++i) {
      const int b = (int)keys[seq[i]] % nb;
      bool first = true;
      for (int j = 0; j < i; ++j)
        if ((int)keys[seq[j]] % nb == b) { first = false; break; }
      if (!first) continue;
      for (int j = i; j < m; ++j)
        if ((int)keys[seq[j]] % nb == b) cur[m - 1 - w++] = seq[j];
    }
    have = m;
  }
  for (int i = 0; i < n; ++i) order[i] = cur[i];
}

// ---- std::sort (libstdc++ bits/stl_algo.h: __introsort_loop, __final_insertion_sort) -------------
// Elements are (key, val) pairs held in two parallel arrays; comp(a, b) = val[a] > val[b].
template <typename K>
struct PairRef {
  using key_t = K;
  K* k;
  float* v;
};
// the same view over interleaved 8-byte pairs {float value, u32 key}: what the expand kernel keeps in LDS
struct PairRefInterleaved {
  using key_t = uint32_t;
  struct KeyView { uint32_t* w; STL_HD_MEMBER uint32_t& operator[](int i) const { return w[2 * i + 1]; } };
  struct ValView { uint32_t* w; STL_HD_MEMBER float& operator[](int i) const { return reinterpret_cast<float*>(w)[2 * i]; } };
  KeyView k;
  ValView v;
};

template <typename K>
STL_HD void pr_swap(PairRef<K> p, int a, int b) {
  K tk = p.k[a]; p.k[a] = p.k[b]; p.k[b] = tk;
  float tv = p.v[a]; p.v[a] = p.v[b]; p.v[b] = tv;
}

template <typename K>
STL_HD void unguarded_linear_insert(PairRef<K> p, int last) {
  const K vk = p.k[last];
  const float vv = p.v[last];
  int next = last - 1;
  while (vv > p.v[next]) {
    p.k[last] = p.k[next]; p.v[last] = p.v[next];
    last = next;
    --next;
  }
  p.k[last] = vk; p.v[last] = vv;
}

template <typename K>
STL_HD void insertion_sort(PairRef<K> p, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (p.v[i] > p.v[first]) {
      const K vk = p.k[i];
      const float vv = p.v[i];
      for (int j = i; j > first; --j) { p.k[j] = p.k[j - 1]; p.v[j] = p.v[j - 1]; }
      p.k[first] = vk; p.v[first] = vv;
    } else {
      unguarded_linear_insert(p, i);
    }
  }
}

template <typename P>
STL_HD void adjust_heap(P p, int first, int hole, int len, typename P::key_t vk, float vv) {
  const int top = hole;
  int second = hole;
  while (second < (len - 1) / 2) {
    second = 2 * (second + 1);
    if (p.v[first + second] > p.v[first + (second - 1)]) second--;
    p.k[first + hole] = p.k[first + second]; p.v[first + hole] = p.v[first + second];
    hole = second;
  }
  if ((len & 1) == 0 && second == (len - 2) / 2) {
    second = 2 * (second + 1);
    p.k[first + hole] = p.k[first + (second - 1)]; p.v[first + hole] = p.v[first + (second - 1)];
    hole = second - 1;
  }
  int parent = (hole - 1) / 2;   // __push_heap
  while (hole > top && p.v[first + parent] > vv) {
    p.k[first + hole] = p.k[first + parent]; p.v[first + hole] = p.v[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  p.k[first + hole] = vk; p.v[first + hole] = vv;
}

template <typename P>
STL_HD void heap_sort(P p, int first, int last) {   // __partial_sort(first, last, last)
  const int len = last - first;
  if (len >= 2) {                                             // __make_heap
    int parent = (len - 2) / 2;
    for (;;) {
      adjust_heap(p, first, parent, len, p.k[first + parent], p.v[first + parent]);
      if (parent == 0) break;
      parent--;
    }
  }
  while (last - first > 1) {                                  // __sort_heap / __pop_heap
    --last;
    const typename P::key_t vk = p.k[last];
    const float vv = p.v[last];
    p.k[last] = p.k[first]; p.v[last] = p.v[first];
    adjust_heap(p, first, 0, last - first, vk, vv);
  }
}

template <typename K>
STL_HD int partition_pivot(PairRef<K> p, int first, int last) {   // __unguarded_partition_pivot
  const int mid = first + (last - first) / 2;
  const int a = first + 1, b = mid, c = last - 1;                 // __move_median_to_first(first, a, b, c)
  if (p.v[a] > p.v[b]) {
    if (p.v[b] > p.v[c]) pr_swap(p, first, b);
    else if (p.v[a] > p.v[c]) pr_swap(p, first, c);
    else pr_swap(p, first, a);
  } else if (p.v[a] > p.v[c]) pr_swap(p, first, a);
  else if (p.v[b] > p.v[c]) pr_swap(p, first, c);
  else pr_swap(p, first, b);
  int lo = first + 1, hi = last;                                  // __unguarded_partition(first+1, last, first)
  for (;;) {
    while (p.v[lo] > p.v[first]) ++lo;
    --hi;
    while (p.v[first] > p.v[hi]) --hi;
    if (!(lo < hi)) return lo;
    pr_swap(p, lo, hi);
    ++lo;
  }
}

// std::sort(v.begin(), v.end(), [](a, b){ return a.second > b.second; }) on n <= 1024 pairs
// `stk` = 3 * kSortStack ints of scratch for the explicit introsort stack.  The device passes LDS: a private array would give the
// whole kernel a scratch segment (and with it a fraction of the occupancy) for a path that runs on prior ties only.
constexpr int kSortStack = 64;
template <typename K>
STL_HD void sort_desc(K* keys, float* vals, int n, int* stk) {
  if (n <= 0) return;
  PairRef<K> p{keys, vals};
  int lg = 0;
  for (int t = n; t > 1; t >>= 1) ++lg;
  // __introsort_loop with an explicit stack (recursion on the right part, loop on the left)
  int* stk_first = stk; int* stk_last = stk + kSortStack; int* stk_depth = stk + 2 * kSortStack;
  int sp = 0;
  stk_first[0] = 0; stk_last[0] = n; stk_depth[0] = 2 * lg; sp = 1;
  while (sp > 0) {
    --sp;
    int first = stk_first[sp], last = stk_last[sp], depth = stk_depth[sp];
    while (last - first > 16) {
      if (depth == 0) { heap_sort(p, first, last); break; }
      --depth;
      const int cut = partition_pivot(p, first, last);
      // the recursive call on [cut, last) runs to completion BEFORE the loop continues on [first, cut);
      // the two ranges are disjoint, so deferring the left part (stack) and continuing right is equivalent
      stk_first[sp] = first; stk_last[sp] = cut; stk_depth[sp] = depth; ++sp;
      first = cut;
    }
  }
  if (n > 16) {                                                   // __final_insertion_sort
    insertion_sort(p, 0, 16);
    for (int i = 16; i < n; ++i) unguarded_linear_insert(p, i);
  } else {
    insertion_sort(p, 0, n);
  }
}

// ---- the same std::sort in a form that maps onto a wavefront ------------------------------------------------------------------
// (a) __unguarded_partition is a pairing problem.  With pivot P = v[first] let U = positions in (first, last) holding v <= P in
//     ASCENDING order (where the up-scan `while (v[lo] > P) ++lo` can stop) and D = positions holding v >= P in DESCENDING order
//     (where the down-scan `while (P > v[hi]) --hi` can stop).  The loop swaps (U[t], D[t]) for t = 0, 1, ... while U[t] < D[t];
//     with T swaps done it returns U[T] if that still lies below D[T-1] (or below `last` when T = 0), else D[T-1], the position
//     that received the last swapped-in value <= P.  Flags, ranks and swaps are all data-parallel.
// (b) __final_insertion_sort only ever moves an element left past strictly smaller ones: its result is the STABLE descending sort
//     of whatever the introsort loop left behind, i.e. a sort by (value desc, position asc) -- a bitonic network on the device.
// sort_desc_pairing is this formulation run serially; tests/native/stl_emul_check.cc checks it against std::sort.  The expand
// kernel's form is (e) below; (c) and (d) are what it ran in rounds 5-6a, kept as independent cross-checks of the pairing idea.
template <typename K>
STL_HD void median_to_first(PairRef<K> p, int first, int last) {   // __move_median_to_first(first, first+1, mid, last-1)
  const int mid = first + (last - first) / 2;
  const int a = first + 1, b = mid, c = last - 1;
  if (p.v[a] > p.v[b]) {
    if (p.v[b] > p.v[c]) pr_swap(p, first, b);
    else if (p.v[a] > p.v[c]) pr_swap(p, first, c);
    else pr_swap(p, first, a);
  } else if (p.v[a] > p.v[c]) pr_swap(p, first, a);
  else if (p.v[b] > p.v[c]) pr_swap(p, first, c);
  else pr_swap(p, first, b);
}

#if !defined(__HIP_DEVICE_COMPILE__)
template <typename K>
static inline int partition_pairing(PairRef<K> p, int first, int last, int* upos, int* dpos) {
  median_to_first(p, first, last);
  const float P = p.v[first];
  int nu = 0, nd = 0;
  for (int i = first + 1; i < last; ++i) if (p.v[i] <= P) upos[nu++] = i;
  for (int i = last - 1; i > first; --i) if (p.v[i] >= P) dpos[nd++] = i;
  int T = 0;
  while (T < nu && T < nd && upos[T] < dpos[T]) ++T;
  for (int t = 0; t < T; ++t) pr_swap(p, upos[t], dpos[t]);
  const int hi_prev = T > 0 ? dpos[T - 1] : last;
  return (T < nu && upos[T] < hi_prev) ? upos[T] : hi_prev;
}

template <typename K>
static inline void sort_desc_pairing(K* keys, float* vals, int n) {
  if (n <= 0) return;
  PairRef<K> p{keys, vals};
  int lg = 0;
  for (int t = n; t > 1; t >>= 1) ++lg;
  std::vector<int> upos(n + 1), dpos(n + 1);
  int stk_first[kSortStack], stk_last[kSortStack], stk_depth[kSortStack], sp = 0;
  stk_first[0] = 0; stk_last[0] = n; stk_depth[0] = 2 * lg; sp = 1;
  while (sp > 0) {
    --sp;
    int first = stk_first[sp], last = stk_last[sp], depth = stk_depth[sp];
    while (last - first > 16) {
      if (depth == 0) { heap_sort(p, first, last); break; }
      --depth;
      const int cut = partition_pairing(p, first, last, upos.data(), dpos.data());
      stk_first[sp] = first; stk_last[sp] = cut; stk_depth[sp] = depth; ++sp;
      first = cut;
    }
  }
  // stable descending sort = __final_insertion_sort
  std::vector<int> idx(n);
  for (int i = 0; i < n; ++i) idx[i] = i;
  std::vector<K> k2(keys, keys + n);
  std::vector<float> v2(vals, vals + n);
  for (int i = 0; i < n; ++i) {
    int r = 0;
    for (int j = 0; j < n; ++j) r += (v2[j] > v2[i]) || (v2[j] == v2[i] && j < i);
    keys[r] = k2[i]; vals[r] = v2[i];
  }
}

// (c) The introsort loop is a tree of INDEPENDENT partitions: after a segment has been cut, its two parts never see each other again
//     (the recursion on [cut, last) and the loop on [first, cut) touch disjoint ranges), and every segment of recursion depth g has
//     the same remaining depth limit 2 lg n - g.  So all segments of one depth can be partitioned AT ONCE ("generation" g): per element
//     its segment (first, last), the segment's pivot, the two stop flags; ranks inside the segment from two prefix counts over the
//     whole array (rank = prefix at the element - prefix at the segment's start); the t-th up-stop and the t-th down-stop of a segment
//     meet in slot first + t of two position arrays; the pairs that swap are a prefix of the slots (upos ascending, dpos descending),
//     so the slot where "swap" turns into "no swap" knows T and with it the cut.  ~10 generations for 362 elements instead of ~55
//     partitions one after the other.  sort_desc_generations is this formulation run serially, array for array what the expand kernel
//     kept in LDS in rounds 5-6a; tests/native/stl_emul_check.cc checks it against std::sort.
template <typename K>
static inline void sort_desc_generations(K* keys, float* vals, int n) {
  if (n <= 0) return;
  PairRef<K> p{keys, vals};
  int lg = 0;
  for (int t = n; t > 1; t >>= 1) ++lg;
  std::vector<int> sf(n, 0), sl(n, n), PU(n + 1), PD(n + 1), UP(n + 1, 0), DP(n + 1, 0), CUT(n + 1, 0);
  std::vector<char> u(n), d(n);
  std::vector<float> P(n);
  int depth = 2 * lg;
  for (;;) {
    bool any = false;
    for (int e = 0; e < n; ++e) any = any || (sl[e] - sf[e] > 16);
    if (!any) break;
    if (depth == 0) {                                   // __partial_sort fallback for every segment that is still long
      for (int e = 0; e < n; ++e)
        if (e == sf[e] && sl[e] - sf[e] > 16) heap_sort(p, sf[e], sl[e]);
      break;
    }
    --depth;
    for (int e = 0; e < n; ++e)                         // 1. pivots
      if (e == sf[e] && sl[e] - sf[e] > 16) median_to_first(p, sf[e], sl[e]);
    for (int e = 0; e < n; ++e) {                       // 2./3. stop flags
      const bool act = sl[e] - sf[e] > 16, in = act && e > sf[e];
      P[e] = p.v[sf[e]];
      u[e] = in && p.v[e] <= P[e];
      d[e] = in && p.v[e] >= P[e];
    }
    PU[0] = PD[0] = 0;                                  // 4. prefix counts over the whole array
    for (int e = 0; e < n; ++e) { PU[e + 1] = PU[e] + u[e]; PD[e + 1] = PD[e] + d[e]; }
    for (int e = 0; e < n; ++e) {                       // 5. t-th up-stop / down-stop of the segment -> slot first + t
      if (u[e]) UP[sf[e] + (PU[e] - PU[sf[e] + 1])] = e;
      if (d[e]) DP[sf[e] + (PD[sl[e]] - PD[e] - 1)] = e;
    }
    std::vector<float> v0(vals, vals + n);              // (the wave reads every pair before it writes any)
    std::vector<K> k0(keys, keys + n);
    for (int q = 0; q < n; ++q) {                       // 6.-8. slot q of its segment: swap?  the slot where that ends knows the cut
      if (!(sl[q] - sf[q] > 16)) continue;
      const int first = sf[q], last = sl[q], t = q - first;
      const int nu = PU[last] - PU[first + 1], nd = PD[last] - PD[first + 1];
      const int mn = nu < nd ? nu : nd;
      const bool ok = t < mn && UP[q] < DP[q];
      const bool ok_next = t + 1 < mn && q + 1 < last && UP[q + 1] < DP[q + 1];
      if (ok) { keys[UP[q]] = k0[DP[q]]; vals[UP[q]] = v0[DP[q]]; keys[DP[q]] = k0[UP[q]]; vals[DP[q]] = v0[UP[q]]; }
      if (ok && !ok_next) {                             // T = t + 1
        const int hi_prev = DP[q];
        const int ut = t + 1 < nu ? UP[q + 1] : 0x7FFFFFFF;
        CUT[first] = ut < hi_prev ? ut : hi_prev;
      } else if (t == 0 && !ok) {                       // T = 0
        const int ut = nu > 0 ? UP[q] : 0x7FFFFFFF;
        CUT[first] = ut < last ? ut : last;
      }
    }
    for (int e = 0; e < n; ++e) {                       // 9. the two parts
      if (!(sl[e] - sf[e] > 16)) continue;
      const int cut = CUT[sf[e]];
      if (e < cut) sl[e] = cut; else sf[e] = cut;
    }
  }
  // stable descending sort = __final_insertion_sort
  std::vector<K> k2(keys, keys + n);
  std::vector<float> v2(vals, vals + n);
  for (int i = 0; i < n; ++i) {
    int r = 0;
    for (int j = 0; j < n; ++j) r += (v2[j] > v2[i]) || (v2[j] == v2[i] && j < i);
    keys[r] = k2[i]; vals[r] = v2[i];
  }
}

// (d) Rounds 5-6a, measured and not used: generations while some segment is longer than `serial_below` elements (few, long segments: the
//     wave-wide passes pay), then every remaining long segment is finished by ONE lane with the serial __introsort_loop and the depth
//     limit that is left (many short segments: a generation would still cost its full wave-wide passes).  Same partitions, other order.
template <typename K>
static inline void sort_desc_hybrid(K* keys, float* vals, int n, int serial_below) {
  if (n <= 0) return;
  PairRef<K> p{keys, vals};
  int lg = 0;
  for (int t = n; t > 1; t >>= 1) ++lg;
  std::vector<int> sf(n, 0), sl(n, n), upos(n + 1), dpos(n + 1);
  int depth = 2 * lg;
  for (;;) {
    bool longer = false;
    for (int e = 0; e < n; ++e) longer = longer || (sl[e] - sf[e] > serial_below && sl[e] - sf[e] > 16);
    if (!longer || depth == 0) break;
    --depth;
    std::vector<int> cuts(n, -1);
    for (int e = 0; e < n; ++e)
      if (e == sf[e] && sl[e] - sf[e] > 16) cuts[e] = partition_pairing(p, sf[e], sl[e], upos.data(), dpos.data());
    for (int e = 0; e < n; ++e) {
      if (!(sl[e] - sf[e] > 16)) continue;
      const int cut = cuts[sf[e]];
      if (e < cut) sl[e] = cut; else sf[e] = cut;
    }
  }
  for (int e = 0; e < n; ++e) {                          // one lane per remaining long segment
    if (!(e == sf[e] && sl[e] - sf[e] > 16)) continue;
    int stk_first[kSortStack], stk_last[kSortStack], stk_depth[kSortStack], sp = 0;
    stk_first[0] = sf[e]; stk_last[0] = sl[e]; stk_depth[0] = depth; sp = 1;
    while (sp > 0) {
      --sp;
      int first = stk_first[sp], last = stk_last[sp], d = stk_depth[sp];
      while (last - first > 16) {
        if (d == 0) { heap_sort(p, first, last); break; }
        --d;
        const int cut = partition_pivot(p, first, last);
        stk_first[sp] = first; stk_last[sp] = cut; stk_depth[sp] = d; ++sp;
        first = cut;
      }
    }
  }
  std::vector<K> k2(keys, keys + n);                     // stable descending sort = __final_insertion_sort
  std::vector<float> v2(vals, vals + n);
  for (int i = 0; i < n; ++i) {
    int r = 0;
    for (int j = 0; j < n; ++j) r += (v2[j] > v2[i]) || (v2[j] == v2[i] && j < i);
    keys[r] = k2[i]; vals[r] = v2[i];
  }
}

// (e) What the expand kernel runs since round 6b (mcts.cuh, introsort_segments_wave): the partition tree walked ONE SEGMENT AT A TIME by the whole wave.  A row of 362
//     priors has only ~40 partitions (at most ~8 long segments exist at any time, most of them <= 65 pairs = one round of 64 lanes), and
//     with the segment's bounds and pivot wave-uniform every per-element table of form (c) goes away:
//       * whether an element swaps is a LOCAL question.  With nub(e) = up-stops strictly before e and nda(e) = down-stops strictly
//         after e (both inside the segment): the up-stop at e is U[nub] and swaps iff D[nub] > e iff nda(e) >= nub(e) + 1; the
//         down-stop at e is D[nda] and swaps iff U[nda] < e iff nub(e) > nda(e).  (Never both: the two conditions contradict.)
//       * partners find each other through two rank-indexed position arrays; each swapper then WRITES ITS OWN OLD PAIR to its
//         partner's position (no read of a pair that another lane may already have replaced).
//       * cut = min(U[T], D[T-1], last) is the lowest position holding an up-stop that does not swap or a down-stop that does.
//     __final_insertion_sort: the segments the loop leaves (<= 16 pairs each) are ordered among themselves -- every pair of a segment
//     compares >= every pair of the next -- so the stable sort of the whole array is the stable sort of each segment: the final place
//     of the pair at e in segment [sf, sl) is sf + #{j in [sf, sl): v[j] > v[e]} + #{j in [sf, e): v[j] == v[e]}, and because of that
//     same order a 16-wide window from sf may run past sl without counting anything.  Segments finished by the heap-sort fallback
//     are sorted already: every one of their positions is marked as a boundary.
template <typename K>
static inline void sort_desc_segments(K* keys, float* vals, int n) {
  if (n <= 0) return;
  PairRef<K> p{keys, vals};
  int lg = 0;
  for (int t = n; t > 1; t >>= 1) ++lg;
  std::vector<char> bnd(n + 1, 0);
  bnd[0] = 1;
  std::vector<int> UP(n + 1), DP(n + 1), nub(n), nda(n);
  std::vector<char> su(n), sd(n), u(n), d(n);
  int stk_first[kSortStack], stk_last[kSortStack], stk_depth[kSortStack], sp = 0;
  stk_first[0] = 0; stk_last[0] = n; stk_depth[0] = 2 * lg; sp = 1;
  while (sp > 0) {
    --sp;
    int first = stk_first[sp], last = stk_last[sp], depth = stk_depth[sp];
    while (last - first > 16) {
      if (depth == 0) {
        heap_sort(p, first, last);
        for (int e = first; e < last; ++e) bnd[e] = 1;
        break;
      }
      --depth;
      median_to_first(p, first, last);
      const float P = p.v[first];
      int cu = 0, cd = 0;
      for (int e = first + 1; e < last; ++e) { u[e] = p.v[e] <= P; d[e] = p.v[e] >= P; nub[e] = cu; cu += u[e]; cd += d[e]; }
      const int ndt = cd;
      cd = 0;
      for (int e = first + 1; e < last; ++e) { cd += d[e]; nda[e] = ndt - cd; }   // down-stops strictly after e
      int cut = last;
      for (int e = last - 1; e > first; --e) {
        su[e] = u[e] && nda[e] >= nub[e] + 1;
        sd[e] = d[e] && nub[e] > nda[e];
        if (su[e]) UP[nub[e]] = e;
        if (sd[e]) DP[nda[e]] = e;
        if ((u[e] && !su[e]) || sd[e]) cut = e;
      }
      std::vector<float> v0(vals + first, vals + last);
      std::vector<K> k0(keys + first, keys + last);
      for (int e = first + 1; e < last; ++e) {
        if (su[e]) { const int q = DP[nub[e]]; keys[q] = k0[e - first]; vals[q] = v0[e - first]; }
        if (sd[e]) { const int q = UP[nda[e]]; keys[q] = k0[e - first]; vals[q] = v0[e - first]; }
      }
      bnd[cut] = 1;
      stk_first[sp] = first; stk_last[sp] = cut; stk_depth[sp] = depth; ++sp;
      first = cut;
    }
  }
  std::vector<K> k2(keys, keys + n);
  std::vector<float> v2(vals, vals + n);
  for (int e = 0; e < n; ++e) {
    int sf = e;
    while (!bnd[sf]) --sf;
    int r = 0;
    for (int t = 0; t < 16; ++t) {
      const int j = sf + t;
      if (j >= n) break;
      r += (v2[j] > v2[e]) || (j < e && v2[j] == v2[e]);
    }
    keys[sf + r] = k2[e]; vals[sf + r] = v2[e];
  }
}

template <typename K>
static inline void sort_desc(K* keys, float* vals, int n) {   // host convenience (tests)
  int stk[3 * kSortStack];
  sort_desc<K>(keys, vals, n, stk);
}
#endif

}  // namespace stl_emul
