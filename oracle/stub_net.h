/* TEST INFRASTRUCTURE ONLY -- deterministic stand-in for the policy/value net.
 *
 * The reference calls its net through the batch interface (src_py/elf/utils_elf.py:368-414): input
 * "s" f32 [B,18,N,N], reply "pi" f32 [B,N*N+1], "V" f32 [B].  For MCTS parity runs both the real
 * reference stack (oracle/ref_selfplay.cc) and the HIP search are fed by THIS function, evaluated
 * on the host, so every difference in visit counts is a search difference.
 *
 * Properties the parity tests rely on:
 *  - pure function of the 0/1 pattern of s (and `salt`), plain C float arithmetic, no libm;
 *  - V is a multiple of 1/256 in [-1,1]: sums of <= 2^15 such values are exact in fp32, so the
 *    reference's heap-address-dependent backup order (tree_search.h:216,245)
 *    cannot change any edge statistic;
 *    (salt bit 31 switches this off for the order-sensitivity test, see below);
 *  - tie_levels > 0 quantises the priors to that many distinct values, forcing equal priors
 *    (exercises std::sort tie order, go/mcts/mcts.h:292-297).
 */
#ifndef ORACLE_STUB_NET_H_
#define ORACLE_STUB_NET_H_
#include <stddef.h>
#include <stdint.h>

static inline uint64_t stubnet_mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

/* one position: s = 18*N*N floats, pi = N*N+1 floats */
static inline void stubnet_eval1(const float* s, int n, uint32_t salt, int tie_levels, float* pi, float* v) {
  const int np = n * n, na = np + 1, nf = 18 * np;
  uint64_t seed = 0xCBF29CE484222325ULL ^ salt;
  for (int i = 0; i < nf; ++i) {
    seed ^= (uint64_t)(s[i] != 0.0f ? (uint32_t)(i * 2 + 1) : 0u);
    seed *= 0x100000001B3ULL;
  }
  float sum = 0.0f;
  for (int a = 0; a < na; ++a) {
    uint64_t h = stubnet_mix(seed ^ ((uint64_t)(a + 1) * 0xD6E8FEB86659FD93ULL));
    uint32_t u = (uint32_t)(h >> 40) & 0xFFFFu;
    if (tie_levels > 0) u = (u % (uint32_t)tie_levels) * (65536u / (uint32_t)tie_levels);
    float w = (float)(u + 1) * (1.0f / 65536.0f);
    w = w * w;
    w = w * w;
    w = w * w;   /* peaky: a handful of moves carry most of the mass */
    pi[a] = w;
    sum += w;
  }
  for (int a = 0; a < na; ++a) pi[a] = pi[a] / sum;
  uint64_t hv = stubnet_mix(seed ^ 0xA5A5A5A5DEADBEEFULL);
  if (salt & 0x80000000u) {
    /* un-quantised value head (salt bit 31): an arbitrary fp32 in (-1, 1).  fp32 sums of such values depend on the backup
     * order, so this mode is only comparable between engines that share one order (the HIP search and oracle/mcts_oracle.cc
     * both use first occurrence; the reference's own order follows heap addresses) */
    *v = (float)((double)(hv >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0);
    return;
  }
  *v = (float)((int)((hv >> 20) % 513u) - 256) * (1.0f / 256.0f);
}

static inline void stubnet_eval(const float* s, int batch, int n, uint32_t salt, int tie_levels, float* pi, float* v) {
  const int np = n * n;
  for (int b = 0; b < batch; ++b) stubnet_eval1(s + (size_t)b * 18 * np, n, salt, tie_levels, pi + (size_t)b * (np + 1), v + b);
}
#endif
