// Host-side engine object shared by the translation units of libelf_amd.so (board engine, MCTS, self-play).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/elf_amd.h"
#include "go_board.cuh"

using namespace elfgo;

template <int N>
struct Pool {
  Slot<N>* slots;
  u64* sk_rec;    // [capacity][MAXMOVE+2][SKW]  superko records {hash, black words, white words}
  const u64* zob; // internal index order
  __device__ __forceinline__ u64* skr(int b) const { return sk_rec + (size_t)b * (Geo<N>::MAXMOVE + 2) * Geo<N>::SKW; }
};

struct ElfGoEngine {
  int n = 0, capacity = 0, device = 0;
  void* slots = nullptr;
  u64* sk_rec = nullptr;
  u64* zob = nullptr;
  size_t slot_bytes = 0;
};

// Every ABI entry that touches the device runs on the engine's device, whatever the calling thread's current device is, and
// leaves the caller's current device as it found it (several engines on different GPUs may live in one process).
struct DevGuard {
  int prev = -1;
  explicit DevGuard(int dev) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != dev && hipSetDevice(dev) == hipSuccess) prev = cur;
  }
  ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  DevGuard(const DevGuard&) = delete;
  DevGuard& operator=(const DevGuard&) = delete;
};

#define HIPCHK(x)                         \
  do {                                    \
    hipError_t _e = (x);                  \
    if (_e != hipSuccess) return (int)_e; \
  } while (0)

template <int N>
static Pool<N> pool_of(const ElfGoEngine* e) {
  Pool<N> p;
  p.slots = reinterpret_cast<Slot<N>*>(e->slots);
  p.sk_rec = e->sk_rec;
  p.zob = e->zob;
  return p;
}

#define DISPATCH(e, CALL)                 \
  do {                                    \
    if ((e)->n == 19) { constexpr int N = 19; CALL; } \
    else { constexpr int N = 9; CALL; }   \
  } while (0)

