// Host check of elf_amd/csrc/stl_emul.h against the real libstdc++ containers/algorithms.
// Built and run by tests/test_stl_emul.py (CPU only).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../elf_amd/csrc/stl_emul.h"

typedef unsigned short Coord;

static int check_umap(std::mt19937& rng, int n, int universe) {
  std::vector<Coord> all(universe);
  for (int i = 0; i < universe; ++i) all[i] = (Coord)i;
  std::shuffle(all.begin(), all.end(), rng);
  std::vector<Coord> keys(all.begin(), all.begin() + n);
  std::unordered_map<Coord, int> m;
  for (int i = 0; i < n; ++i) m.insert(std::make_pair(keys[i], i));
  std::vector<int> order(n), tmp(2 * n + 2);
  stl_emul::umap_iteration_order(keys.data(), n, order.data(), tmp.data());
  int i = 0;
  for (const auto& p : m) {
    if (p.second != order[i]) return 1;
    ++i;
  }
  return 0;
}

static int check_sort(std::vector<float> vals) {
  const int n = (int)vals.size();
  std::vector<std::pair<Coord, float>> ref(n);
  std::vector<Coord> k(n);
  std::vector<float> v(n);
  for (int i = 0; i < n; ++i) { ref[i] = std::make_pair((Coord)i, vals[i]); k[i] = (Coord)i; v[i] = vals[i]; }
  using T = std::pair<Coord, float>;
  std::sort(ref.begin(), ref.end(), [](const T& a, const T& b) { return a.second > b.second; });
  stl_emul::sort_desc(k.data(), v.data(), n);
  for (int i = 0; i < n; ++i)
    if (ref[i].first != k[i] || ref[i].second != v[i]) return 1;
  // the wavefront formulation (pairing partition + stable final sort) must give the same permutation
  std::vector<Coord> k2(n);
  std::vector<float> v2(n);
  for (int i = 0; i < n; ++i) { k2[i] = (Coord)i; v2[i] = vals[i]; }
  stl_emul::sort_desc_pairing(k2.data(), v2.data(), n);
  for (int i = 0; i < n; ++i)
    if (ref[i].first != k2[i] || ref[i].second != v2[i]) return 1;
  // ... and so must the generation formulation (all segments of one recursion depth partitioned at once: the expand kernel's form)
  std::vector<Coord> k3(n);
  std::vector<float> v3(n);
  for (int i = 0; i < n; ++i) { k3[i] = (Coord)i; v3[i] = vals[i]; }
  stl_emul::sort_desc_generations(k3.data(), v3.data(), n);
  for (int i = 0; i < n; ++i)
    if (ref[i].first != k3[i] || ref[i].second != v3[i]) return 1;
  {   // one segment at a time, local swap rule, window ranks (the expand kernel's form since round 6b)
    std::vector<Coord> k5(n);
    std::vector<float> v5(n);
    for (int i = 0; i < n; ++i) { k5[i] = (Coord)i; v5[i] = vals[i]; }
    stl_emul::sort_desc_segments(k5.data(), v5.data(), n);
    for (int i = 0; i < n; ++i)
      if (ref[i].first != k5[i] || ref[i].second != v5[i]) return 1;
  }
  for (int below : {16, 40, 64, 100000}) {   // generations, then one lane per remaining long segment (the kernel uses 64)
    std::vector<Coord> k4(n);
    std::vector<float> v4(n);
    for (int i = 0; i < n; ++i) { k4[i] = (Coord)i; v4[i] = vals[i]; }
    stl_emul::sort_desc_hybrid(k4.data(), v4.data(), n, below);
    for (int i = 0; i < n; ++i)
      if (ref[i].first != k4[i] || ref[i].second != v4[i]) return 1;
  }
  return 0;
}

// heap_sort over the interleaved {value, key} pairs of the expand kernel's LDS array against the same over two parallel arrays
static int check_heap_interleaved(std::mt19937& rng, int n, int levels) {
  std::vector<unsigned> k(n), w(2 * n);
  std::vector<float> v(n);
  for (int i = 0; i < n; ++i) {
    v[i] = (float)(rng() % (unsigned)levels) / 3.0f; k[i] = (unsigned)i | 0x10000u;
    w[2 * i + 1] = k[i];
    reinterpret_cast<float*>(w.data())[2 * i] = v[i];
  }
  const int first = n / 5, last = n - n / 7;
  stl_emul::heap_sort(stl_emul::PairRef<unsigned>{k.data(), v.data()}, first, last);
  stl_emul::heap_sort(stl_emul::PairRefInterleaved{{w.data()}, {w.data()}}, first, last);
  for (int i = 0; i < n; ++i)
    if (w[2 * i + 1] != k[i] || reinterpret_cast<float*>(w.data())[2 * i] != v[i]) return 1;
  return 0;
}

int main() {
  std::mt19937 rng(12345);
  int bad = 0, cases = 0;
  for (int n = 0; n <= 362; ++n)
    for (int rep = 0; rep < 6; ++rep) { bad += check_umap(rng, n, 441); ++cases; }
  for (int n = 0; n <= 82; ++n)
    for (int rep = 0; rep < 6; ++rep) { bad += check_umap(rng, n, 121); ++cases; }
  for (int n = 363; n <= 441; n += 7) { bad += check_umap(rng, n, 441); ++cases; }
  printf("umap cases %d bad %d\n", cases, bad);
  int sbad = 0, scases = 0;
  for (int n : {0, 1, 2, 3, 15, 16, 17, 18, 31, 33, 64, 82, 100, 200, 361, 362, 500, 1000}) {
    for (int levels : {1, 2, 3, 5, 17, 1000, 1 << 20}) {
      for (int rep = 0; rep < 8; ++rep) {
        std::vector<float> v(n);
        for (auto& x : v) x = (float)(rng() % (unsigned)levels) / (float)levels;
        sbad += check_sort(v); ++scases;
      }
    }
    std::vector<float> v(n);
    for (int i = 0; i < n; ++i) v[i] = (float)i;                 sbad += check_sort(v); ++scases;   // ascending
    for (int i = 0; i < n; ++i) v[i] = (float)(n - i);           sbad += check_sort(v); ++scases;   // descending
    for (int i = 0; i < n; ++i) v[i] = (float)(i < n / 2 ? i : n - i);  sbad += check_sort(v); ++scases;  // organ pipe
    for (int i = 0; i < n; ++i) v[i] = (float)(i % 2 ? i : -i);  sbad += check_sort(v); ++scases;
  }
  // median-of-3 killer (forces the heapsort fallback of introsort) for the descending comparator
  for (int n : {362, 512, 1000}) {
    std::vector<float> v(n);
    int k = n / 2;
    for (int i = 1; i <= k; ++i) {
      if (i % 2 == 1) { v[i - 1] = (float)-i; v[i] = (float)-(k + i); }
      v[k + i - 1] = (float)-(2 * i);
    }
    sbad += check_sort(v); ++scases;
  }
  for (int rep = 0; rep < 4000; ++rep) {   // the expand kernel's shape: 362 / 82 priors, few distinct values
    const int n = (rep & 1) ? 362 : 82;
    const int levels = 1 + (int)(rng() % (rep % 5 == 0 ? 400u : 12u));
    std::vector<float> v(n);
    for (auto& x : v) x = (float)(rng() % (unsigned)levels) / 7.0f;
    sbad += check_sort(v); ++scases;
  }
  for (int n : {17, 40, 100, 362})
    for (int levels : {2, 7, 1000}) { sbad += check_heap_interleaved(rng, n, levels); ++scases; }
  printf("sort cases %d bad %d\n", scases, sbad);
  return (bad || sbad) ? 1 : 0;
}
